"""tests/golden/simplify.npz from the REFERENCE's libsimplify (occnet_utils/utils/libsimplify: simplify_mesh.pyx + Simplify.h), built
out-of-tree by tests/golden/build_ref_native.py.  Inputs: marching-cubes meshes of analytic fields (closed sphere / torus, an
OPEN sheet whose border vertices exercise the border rules, a flat grid where every quadric is singular) produced by the pinned
numpy marching-cubes restatement (oracle/mcubes.py); outputs: the reference's simplified vertices (float64) and faces, in its
order, for several targets and aggressiveness values (mesh_extractor2.py:207 calls it with 5.0; the module default is 7.0)."""
import sys
import numpy as np

sys.path.insert(0, "/tmp/simplify_build")
import simplify_mesh as ref  # noqa: E402  (the reference's Cython module)

ROOT = __file__.rsplit("/tests/", 1)[0]
sys.path.insert(0, ROOT)
from oracle import mcubes  # noqa: E402


def field_mesh(fn, n, iso=0.0, pad=True):
    lin = np.linspace(-0.55, 0.55, n)
    g = np.stack(np.meshgrid(lin, lin, lin, indexing="ij"), -1)
    vol = fn(g).astype(np.float64)
    if pad:
        vol = np.pad(vol, 1, "constant", constant_values=-1e6)
    v, f = mcubes.marching_cubes(vol, iso)
    return np.ascontiguousarray(v, np.float64), np.ascontiguousarray(f, np.int64)


cases = {
    "sphere": field_mesh(lambda g: 0.35 - np.linalg.norm(g, axis=-1), 21),
    "torus": field_mesh(lambda g: 0.12 - np.sqrt((np.sqrt(g[..., 0] ** 2 + g[..., 1] ** 2) - 0.3) ** 2 + g[..., 2] ** 2), 25),
    "open_sheet": field_mesh(lambda g: 0.1 * np.sin(5 * g[..., 0]) * np.cos(4 * g[..., 1]) - g[..., 2], 15, pad=False),
    "flat": field_mesh(lambda g: -g[..., 2] + 0.013, 9, pad=False),
}
out = {}
for name, (v, f) in cases.items():
    out[f"{name}_v"], out[f"{name}_f"] = v, f
    for target, agg in ((max(8, f.shape[0] // 4), 5.0), (max(8, f.shape[0] // 10), 5.0), (max(8, f.shape[0] // 3), 7.0), (f.shape[0] + 10, 5.0)):
        vo, fo = ref.mesh_simplify(v.copy(), f.copy(), int(target), float(agg))
        key = f"{name}_t{target}_a{int(agg)}"
        out[key + "_v"], out[key + "_f"] = np.asarray(vo), np.asarray(fo)
        print(key, v.shape[0], f.shape[0], "->", vo.shape[0], fo.shape[0])
np.savez_compressed(ROOT + "/tests/golden/simplify.npz", **out)
