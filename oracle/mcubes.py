"""CPU restatement of the reference's marching cubes (PyMCubes as vendored in libmcubes) -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/lib_shape_prior/core/models/utils/occnet_utils/utils/libmcubes/marchingcubes.h:23-196
(mc::marching_cubes: cube order, `<=` inside test, which cube creates which edge vertex and in which order, the 0.5 offset)
and marchingcubes.cpp:290-326 (vertex interpolation), as called by pywrapper.cpp:90-108 on a 3-D array.
Pinned by tests/golden/mcubes.npz (tests/golden/make_golden_mcubes.py: the reference's library built out-of-tree).
The triangulation table is the one that script recovers by probing the library (also in the fixture).

The reference walks the cubes sequentially and numbers vertices as it creates them; here the same numbering comes from an
exclusive prefix sum over per-cube creation counts, which is also how the HIP kernels (csrc/mcubes.hip) do it.
"""
import os

import numpy as np

CORNER = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)])
EDGE_A = np.array([0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3])
EDGE_B = np.array([1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7])
ORDER = [6, 5, 10, 0, 1, 2, 3, 4, 7, 8, 9, 11]            # creation order inside a cube (marchingcubes.h:70-186)
# edge -> (di, dj, dk, slot) of the neighbour cube that owns it (slot 0/1/2 = that cube's edge 6/5/10), marchingcubes.h:96-186
OWNER = {0: (0, -1, -1, 0), 1: (0, 0, -1, 1), 2: (0, 0, -1, 0), 3: (-1, 0, -1, 1), 4: (0, -1, 0, 0), 7: (-1, 0, 0, 1),
         8: (-1, -1, 0, 2), 9: (0, -1, 0, 2), 11: (-1, 0, 0, 2)}
SLOT_EDGE = {0: 6, 1: 5, 2: 10}


def _tables():
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "mcubes.npz"))
    return g["tri_table"].astype(np.int64), g["edge_table"].astype(np.int64)


def marching_cubes(volume, isovalue):
    """volume [nx,ny,nz] -> (vertices float64 [nv,3], faces int64 [nf,3]) exactly as libmcubes.marching_cubes."""
    tri_table, edge_table = _tables()
    isovalue = float(np.float32(isovalue))   # mcubes.pyx:22 declares `float isovalue`: the Python float is narrowed to 32 bit
    vol = np.asarray(volume, np.float64)
    nx, ny, nz = (s - 1 for s in vol.shape)
    if min(nx, ny, nz) < 1:
        return np.zeros((0, 3)), np.zeros((0, 3), np.int64)
    I, J, K = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    v = np.stack([vol[I + dx, J + dy, K + dz] for dx, dy, dz in CORNER], -1)           # [nx,ny,nz,8]
    cfg = ((v <= isovalue) << np.arange(8)).sum(-1)
    edges = edge_table[cfg]
    crossed = (edges[..., None] >> np.arange(12) & 1).astype(bool)                       # [nx,ny,nz,12]
    i0, j0, k0 = I == 0, J == 0, K == 0
    creates = np.zeros_like(crossed)
    for e in (6, 5, 10):
        creates[..., e] = crossed[..., e]
    cond = {0: j0 | k0, 1: k0, 2: k0, 3: i0 | k0, 4: j0, 7: i0, 8: i0 | j0, 9: j0, 11: i0}
    for e, c in cond.items():
        creates[..., e] = crossed[..., e] & c
    nnew = creates.sum(-1)
    base = np.concatenate([[0], np.cumsum(nnew.reshape(-1))[:-1]]).reshape(nnew.shape)  # cube order = C order over (i,j,k)
    rank = np.zeros(creates.shape, np.int64)
    run = np.zeros(nnew.shape, np.int64)
    for e in ORDER:
        rank[..., e] = run
        run = run + creates[..., e]
    own_idx = np.where(creates, base[..., None] + rank, -1)                             # index of the vertex this cube creates
    # vertices
    nv = int(nnew.sum())
    verts = np.zeros((nv, 3))
    pos = np.stack([I, J, K], -1).astype(np.float64) + 0.5                               # marchingcubes.h:41-53 (dx/2 offset)
    for e in range(12):
        m = creates[..., e]
        if not m.any():
            continue
        a, b = EDGE_A[e], EDGE_B[e]
        f1, f2 = v[..., a][m], v[..., b][m]
        p1 = pos[m] + CORNER[a]
        p2 = pos[m] + CORNER[b]
        axis = int(np.nonzero(CORNER[a] != CORNER[b])[0][0])
        x1, x2 = p1[:, axis], p2[:, axis]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(f2 == f1, (x2 + x1) / 2, (x2 - x1) * (isovalue - f1) / (f2 - f1) + x1)   # marchingcubes.cpp:290-297
        p = p1.copy()
        p[:, axis] = t
        verts[own_idx[..., e][m]] = p
    # per-cube vertex index of every crossed edge
    idx12 = own_idx.copy()
    for e, (di, dj, dk, slot) in OWNER.items():
        need = crossed[..., e] & ~creates[..., e]
        if not need.any():
            continue
        ii, jj, kk = I[need] + di, J[need] + dj, K[need] + dk
        idx12[..., e][need] = own_idx[ii, jj, kk, SLOT_EDGE[slot]]
    # faces
    tri = tri_table[cfg.reshape(-1)]                                                     # [ncubes,16]
    valid = tri >= 0
    flat = np.take_along_axis(idx12.reshape(-1, 12), np.where(valid, tri, 0), 1)[valid]
    assert (flat >= 0).all()
    return verts, flat.reshape(-1, 3).astype(np.int64)
