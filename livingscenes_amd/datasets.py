"""On-disk format of the FlyingShape benchmark as the reference's evaluation reads it (FlyingShape part):

    <root>/<group dir, name ending in _<object count>>/<scene dir>/*.npz    one npz per scan of the scene, sorted by name;
    npz['pc'] [n_obj, N, 3] float, npz['transform'] [n_obj, 3 or 4, 4]       (+ optional 'class_id', 'obj_id')

/root/reference/eval_flyingshape.py:33-60 (the dataset class: directory walk, sorted listing, np.load per scan) and :76,
:119-129 (which fields the loops use: data[0] = reference scan, data[1:] = rescans).  The real files are not in the container
(.MISSING_LARGE_BLOBS); ``write_scene`` produces the same layout from synthetic scenes so that the harness can be driven from
disk exactly as the reference drives it.  The 3RScan formats (PLY + json + npz) are in rscan.py.
"""
import glob
import os
import os.path as osp

import numpy as np
import torch


class FlyingShape:
    """``dataset[i]`` = the scans of scene i as np.load handles, reference scan first (same listing order as the reference's
    dataset class: object-count directories sorted by name, scenes sorted inside each, scans sorted by file name)."""

    def __init__(self, path):
        self.path = path
        self.n_shape_lst = sorted(os.listdir(path))
        self.scene_lst = []
        for group in self.n_shape_lst:
            if not group.rsplit("_", 1)[-1].isdigit():
                raise ValueError(f"FlyingShape: directory name {group!r} does not end in _<object count>")
            self.scene_lst.extend(osp.join(path, group, scene) for scene in sorted(os.listdir(osp.join(path, group))))

    def __len__(self):
        return len(self.scene_lst)

    def __getitem__(self, idx):
        scans = sorted(glob.glob(osp.join(self.scene_lst[idx], "*.npz")))   # IndexError past the end ends iteration
        return [np.load(f) for f in scans]


def write_scene(root, group_dir, scene_name, scans):
    """scans: list of {'pc': [n_obj,N,3], 'transform': [n_obj,4,4], ...}; the first one is the reference scan."""
    d = osp.join(root, group_dir, scene_name)
    os.makedirs(d, exist_ok=True)
    for i, sc in enumerate(scans):
        np.savez(osp.join(d, f"scan_{i:02d}.npz"), **{k: np.asarray(v) for k, v in sc.items()})
    return d


def scene_from_scans(data):
    """list of scans (as FlyingShape yields them) -> the in-memory scene dict of synth.make_scene_pair / harness.py
    (reference scan + FIRST rescan, as the relocalisation loop uses them, eval_flyingshape.py:120-124)."""
    def t(a):
        return torch.from_numpy(np.asarray(a)).float()
    return {"ref": t(data[0]["pc"]), "rescan": t(data[1]["pc"]), "ref_T": t(data[0]["transform"]), "rescan_T": t(data[1]["transform"])}
