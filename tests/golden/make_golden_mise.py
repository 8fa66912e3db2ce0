"""Generate tests/golden/mise.npz from the REFERENCE's MISE (libmise/mise.pyx), built out-of-tree in the dev container:

    mkdir -p /tmp/mise_build && cd /tmp/mise_build      # scratch dir outside the repo
    (setup.py: cythonize the reference's mise.pyx where it lies -> mise.cpython-*.so)
    python tests/golden/make_golden_mise.py

Analytic fields stand in for the decoder (the driver only sees values): per round the set of queried lattice points and
the final dense grid are recorded.  The fields are evaluated in float32 exactly as committed here, so the oracle and the HIP
path can reproduce the same values on the GPU box without the reference.
"""
import sys
import numpy as np

sys.path.insert(0, "/tmp/mise_build")
import mise as ref_mise  # noqa: E402  (the reference's Cython module)

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
sys.path.insert(0, __file__.rsplit("/tests/", 1)[0] + "/tests")
from mise_fields import FIELDS  # noqa: E402


def drive(field, res0, depth, thr, box=1.1):
    m = ref_mise.MISE(res0, depth, thr)
    pts = m.query()
    rounds = []
    while pts.shape[0] != 0:
        pf = pts.astype(np.float32) / np.float32(m.resolution)
        pf = np.float32(box) * (pf - np.float32(0.5))
        vals = field(pf).astype(np.float64)
        rounds.append(pts.copy())
        m.update(pts, vals)
        pts = m.query()
    return rounds, m.to_dense()


out = {}
cases = [("sphere", 8, 2, 0.0), ("torus", 8, 2, 0.0), ("two_blobs", 4, 3, 0.0), ("sphere", 16, 1, 0.05), ("plane_tie", 4, 2, 0.0),
         ("empty", 4, 2, 0.0)]
for name, res0, depth, thr in cases:
    rounds, dense = drive(FIELDS[name], res0, depth, thr)
    key = f"{name}_{res0}_{depth}"
    out[key + "_cfg"] = np.array([res0, depth, thr], np.float64)
    out[key + "_nrounds"] = np.array(len(rounds))
    for i, r in enumerate(rounds):
        lin = (r[:, 0] * (dense.shape[0]) + r[:, 1]) * dense.shape[0] + r[:, 2]
        out[key + f"_round{i}"] = np.sort(lin).astype(np.int32)
    out[key + "_dense"] = dense.astype(np.float32)     # values are float32-representable by construction
    print(key, [len(r) for r in rounds], dense.shape, float(np.nanmin(dense)), float(np.nanmax(dense)))
np.savez_compressed(__file__.rsplit("/", 1)[0] + "/mise.npz", **out)
