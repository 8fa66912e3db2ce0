"""3x4 / 4x4 rigid-transform helpers with the reference's names and conventions
(/root/reference/lib_math/torch_se3.py:6-92).  Twelve-float algebra: plain torch on whatever device the inputs live."""
import torch


def identity(batch_size):
    return torch.eye(3, 4)[None].repeat(batch_size, 1, 1)


def inverse(g):
    """(B,3/4,4) -> (B,3,4) inverse transform (torch_se3.py:10-25)."""
    Rt = g[..., :3, :3].transpose(-1, -2)
    return torch.cat([Rt, -(Rt @ g[..., :3, 3:4])], dim=-1)


def concatenate(a, b):
    """a @ b for 3x4 transforms (torch_se3.py:28-49)."""
    Ra, Rb = a[..., :3, :3], b[..., :3, :3]
    return torch.cat([Ra @ Rb, Ra @ b[..., :3, 3:4] + a[..., :3, 3:4]], dim=-1)


def transform(g, a, normals=None):
    """Apply g to points a [(B,)N,3] (torch_se3.py:52-78)."""
    R, p = g[..., :3, :3], g[..., :3, 3]
    if g.dim() != a.dim():
        raise NotImplementedError
    b = a @ R.transpose(-1, -2) + p[..., None, :]
    if normals is not None:
        return b, normals @ R.transpose(-1, -2)
    return b


def Rt_to_SE3(R, t):
    """(B,3,3),(B,3,1) -> (B,4,4) (torch_se3.py:81-92)."""
    T = torch.zeros(R.shape[0], 4, 4, device=R.device, dtype=R.dtype)
    T[:, :3, :3], T[:, :3, 3:4], T[:, 3, 3] = R, t, 1
    return T
