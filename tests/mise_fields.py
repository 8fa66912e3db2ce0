"""Analytic float32 test fields for the MISE driver (stand-ins for the decoder's logits: positive inside).

Used by tests/golden/make_golden_mise.py (against the reference's MISE), the oracle test and the GPU parity test, which must
all see bit-identical values: plain float32 numpy / torch arithmetic only, no transcendental functions.
"""
import numpy as np


def _f32(p):
    return np.asarray(p, np.float32)


def sphere(p):
    p = _f32(p)
    return (np.float32(0.09) - (p * p).sum(-1)).astype(np.float32)


def torus(p):
    p = _f32(p)
    q = (p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1] + p[:, 2] * p[:, 2] + np.float32(0.0675))
    return (np.float32(4.0) * np.float32(0.09) * (p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) - q * q).astype(np.float32)


def two_blobs(p):
    p = _f32(p)
    a = p - np.array([0.2, 0.1, -0.15], np.float32)
    b = p - np.array([-0.25, -0.2, 0.2], np.float32)
    return np.maximum(np.float32(0.02) - (a * a).sum(-1), np.float32(0.035) - (b * b).sum(-1)).astype(np.float32)


def plane_tie(p):
    """Exactly 0 (== threshold) on the lattice plane x = 0.5 R: exercises the >= / <= tie handling of mise.pyx:216-219."""
    p = _f32(p)
    return (-p[:, 0]).astype(np.float32)


def empty(p):
    p = _f32(p)
    return np.full(p.shape[0], -1.0, np.float32)


FIELDS = {"sphere": sphere, "torus": torus, "two_blobs": two_blobs, "plane_tie": plane_tie, "empty": empty}
