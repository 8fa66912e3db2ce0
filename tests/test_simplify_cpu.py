"""ls_simplify_mesh_f64_host (csrc/simplify.cpp) against tests/golden/simplify.npz, recorded from the REFERENCE's own libsimplify
(occnet_utils/utils/libsimplify: simplify_mesh.pyx + Simplify.h, built out-of-tree by tests/golden/build_ref_native.py;
generator tests/golden/make_golden_simplify.py).  Host code on both sides (the reference decimates on the CPU too), so these run
without a GPU.  Bar: vertices (float64), faces and their ORDER bit-identical."""
import re

import numpy as np
import pytest

from livingscenes_amd import mesh_extractor2


def _cases(g):
    for k in g:
        m = re.match(r"(\w+?)_t(\d+)_a(\d)_v$", k)
        if m:
            yield k[:-2], m.group(1), int(m.group(2)), float(m.group(3))


def test_simplify_bit_identical_to_reference_fixture(golden):
    g = golden("simplify")
    n = 0
    for key, name, target, agg in _cases(g):
        v, f = mesh_extractor2.simplify_mesh_arrays(g[name + "_v"], g[name + "_f"], target, agg)
        assert v.dtype == np.float64 and f.dtype == np.int64
        assert np.array_equal(v, g[key + "_v"]), key
        assert np.array_equal(f, g[key + "_f"]), key
        n += 1
    assert n == 16


def test_simplify_properties_and_mesh_wrapper(golden):
    g = golden("simplify")
    v, f = g["torus_v"], g["torus_f"]
    for target in (100, 400, 1000):
        vo, fo = mesh_extractor2.simplify_mesh_arrays(v, f, target, 5.0)
        # stops at the target (a collapse removes 2 faces), or earlier when the 100 passes / the flip tests run out of legal collapses
        assert target - 1 <= fo.shape[0] < f.shape[0] and (target < 200 or fo.shape[0] <= target + 1)
        assert fo.min() == 0 and fo.max() == vo.shape[0] - 1                      # compacted: every vertex is referenced
        assert len(np.unique(fo)) == vo.shape[0]
        # closed input stays closed: every edge belongs to exactly two faces
        e = np.sort(np.concatenate([fo[:, [0, 1]], fo[:, [1, 2]], fo[:, [2, 0]]]), 1)
        _, cnt = np.unique(e, axis=0, return_counts=True)
        assert (cnt == 2).all()
    # target above the face count: nothing collapses, unreferenced vertices are dropped, order kept
    vo, fo = mesh_extractor2.simplify_mesh_arrays(v, f, f.shape[0] + 1, 5.0)
    assert np.array_equal(vo, v) and np.array_equal(fo, f)
    # the published algorithm (border flags zero during the initial cost pass) is a different, also valid, decimation
    va, fa = mesh_extractor2.simplify_mesh_arrays(v, f, 400, 5.0, initial_border=0)
    assert fa.shape[0] <= 401 and not np.array_equal(va, mesh_extractor2.simplify_mesh_arrays(v, f, 400, 5.0)[0])
    mesh = mesh_extractor2.simplify_mesh(mesh_extractor2.SimpleMesh(v, f), 400, 5.0)   # libsimplify/__init__.py:7-17 surface
    assert np.asarray(mesh.faces).shape[0] <= 401
    with pytest.raises(Exception):
        mesh_extractor2.simplify_mesh_arrays(v, f + 10 ** 6, 10, 5.0)                   # out-of-range vertex index
