"""On-disk formats of the 3RScan evaluation as the reference reads them (SURVEY.md 8 f-4, 3RScan part).

    <root>/3RScan.json                      list of scenes {'reference': id, 'scans': [{'reference': id, 'transform': [16 floats,
                                            column-major 4x4 rescan->reference], 'rigid': [{'instance_reference': objectId,
                                            'transform': [16, column-major, reference->rescan]}, ...]}, ...]}
    <root>/../splits/<split>.txt            reference scan ids of the split, one per line
    <root>/<split>_set/<scan id>/semseg.v2.json                   {'segGroups': [{'objectId': int, 'label': str, ...}]}
    <root>/<split>_set/<scan id>/pointcloud.instances.align.ply   vertices x y z (+ whatever else)
    <root>/<split>_set/<scan id>/pointcloud.labels.npz            npz['objectId'] [n_vertices] (or a predicted-mask file)

Behaviour follows /root/reference/eval_3rscan.py:50-72 (constructor: split file, scene list filtered by reference id),
:78-95 (zero padding + boolean mask for clouds of different sizes), :97-155 (per-scan instance extraction: category filter,
instances with fewer than 1024 points dropped, background = every 5th point of the other instances below the tallest kept
instance) and :160-187 (scene = reference scan + rescans with their rescan->reference transform and the moving / static
split of the annotated rigid instances: rotation difference > 1 degree or translation difference > 0.05).  The reference reads
the PLY through point_cloud_utils and hard-codes .cuda(); here the vertex reader is a small PLY parser (ascii and
binary_little_endian) and the device is a constructor argument.  None of the dataset is in the container
(.MISSING_LARGE_BLOBS): ``write_scan`` / ``write_index`` produce the same layout from synthetic scenes for the tests.
"""
import json
import os
import os.path as osp

import numpy as np
import torch

from .lib_more.pose_estimation import inverse_3d_transform, rotation_error, translation_error

# category mapping between ShapeNet and RIO labels (eval_3rscan.py:25-39)
SHAPENET_CATE = ["chair", "table", "bench", "sofa", "pillow", "bed", "trash_bin"]
RIO_CATE = [
    ["dinning chair", "rocking chair", "armchair", "chair"],
    ["couching table", "dining table", "computer desk", "round table", "side table", "stand", "desk", "coffee table"],
    ["bench"],
    ["sofa", "sofa chair", "couch", "ottoman", "footstool"],
    ["cushion", "pillow"],
    ["bed"],
    ["trash can"],
]


def get_shapenet_category(rio_cate):
    for shapenet_cate, rio_list in zip(SHAPENET_CATE, RIO_CATE):
        if rio_cate in rio_list:
            return shapenet_cate
    return "others"


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def load_ply_vertices(path):
    """[n, 3] vertex positions of a PLY file (ascii or binary little/big endian); other vertex properties are skipped."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n_vert, props, in_vertex, before = None, None, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: PLY header not terminated")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if tok[1] == "vertex":
                    n_vert, in_vertex = int(tok[2]), True
                else:
                    if n_vert is None:
                        before += 1          # an element in front of the vertices: not produced by the dataset's tools
                    in_vertex = False
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property on vertices is not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if n_vert is None or fmt is None or before:
            raise ValueError(f"{path}: unsupported PLY layout")
        names = [p[0] for p in props]
        if not all(a in names for a in "xyz"):
            raise ValueError(f"{path}: vertices have no x/y/z")
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=n_vert, ndmin=2)
            return np.stack([rows[:, names.index(a)] for a in "xyz"], axis=1).astype(np.float32)
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, end + t) for n, t in props])
        v = np.frombuffer(f.read(n_vert * dt.itemsize), dtype=dt, count=n_vert)
        return np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float32)


def _mat4(flat, device):
    """16 numbers, column-major -> [1,4,4] (eval_3rscan.py:171: reshape(1,4,4).transpose(-1,-2))"""
    return torch.tensor(flat, dtype=torch.float32, device=device).reshape(1, 4, 4).transpose(-1, -2).contiguous()


class Dataset_3RScan(torch.utils.data.Dataset):
    def __init__(self, cfg, device="cuda"):
        self.device = torch.device(device)
        self.root_path = cfg["root_path"]
        self.split = cfg["split"]
        self.data_path = osp.join(self.root_path, f"{self.split}_set")
        cl = cfg["category_list"]
        if isinstance(cl, str):
            with open(cl) as f:
                cl = f.read().splitlines()
        self.category_list = list(cl)
        self.n_point_per_instance = cfg["n_point_per_instance"]
        self.scan_list = os.listdir(self.data_path)
        with open(osp.join(self.root_path, "..", f"splits/{self.split}.txt")) as f:
            self.split_indices = f.read().splitlines()
        with open(osp.join(self.root_path, "3RScan.json")) as f:
            scene_json = json.load(f)
        self.scene_list = [scene for scene in scene_json if scene["reference"] in self.split_indices]
        self.use_gt_mask = cfg["use_gt_mask"]
        if not self.use_gt_mask:
            self.mask_name = cfg["mask_name"]

    def __len__(self):
        return len(self.scene_list)

    def __getitem__(self, idx):
        return self._get_scene(idx)

    def _heterogeneous_batching(self, pc_list):
        """[1,3,n_i] clouds -> zero-padded [B,3,n_max] + mask [B,1,n_max] (what Shape_Prior.encode_fps takes)"""
        n_max = max(pc.shape[-1] for pc in pc_list)
        B = len(pc_list)
        batch = torch.zeros(B, 3, n_max, device=self.device)
        mask = torch.zeros(B, 1, n_max, dtype=torch.bool, device=self.device)
        for i, pc in enumerate(pc_list):
            n = pc.shape[-1]
            batch[i, :, :n] = pc[0]
            mask[i, :, :n] = True
        return batch, mask

    def _load_scan(self, scan_id):
        scan_path = osp.join(self.data_path, scan_id)
        with open(osp.join(scan_path, "semseg.v2.json")) as f:
            semseg_list = json.load(f)["segGroups"]
        scan_pc = load_ply_vertices(osp.join(scan_path, "pointcloud.instances.align.ply"))
        labels = np.load(osp.join(scan_path, "pointcloud.labels.npz" if self.use_gt_mask else self.mask_name), allow_pickle=True)
        obj_of_point = labels["objectId"]
        pc_list, id_list, label_list, full_gt_id_list, bg_pc = [], [], [], [], []
        z_max = -100.0   # the background is trimmed at the top of the tallest kept instance
        for inst in semseg_list:
            if inst["label"] not in self.category_list:
                continue
            label_list.append((inst["objectId"], inst["label"], get_shapenet_category(inst["label"])))
            inst_id = torch.tensor([int(inst["objectId"])], device=self.device)
            full_gt_id_list.append(inst_id)
            pts = scan_pc[obj_of_point == inst["objectId"]]
            if len(pts) == 0:
                continue
            z_max = max(z_max, float(pts[:, -1].max()))
            if pts.shape[0] < 1024:
                continue
            pc_list.append(torch.from_numpy(pts).float().to(self.device).unsqueeze(0).permute(0, 2, 1))
            id_list.append(inst_id)
        for inst in semseg_list:
            if inst["label"] not in self.category_list:
                pts = scan_pc[obj_of_point == inst["objectId"]]
                bg_pc.append(pts[pts[:, 2] < z_max])
        if len(pc_list) == 0:
            return None
        batch_pc, batch_mask = self._heterogeneous_batching(pc_list)
        bg = np.concatenate(bg_pc, axis=0)[::5] if bg_pc else np.zeros((0, 3), np.float32)
        return {"pc": batch_pc, "pc_mask": batch_mask, "objectId": torch.cat(id_list, dim=0), "bg_pc": bg, "id_label": label_list,
                "full_objectId": torch.cat(full_gt_id_list)}

    def _get_scene(self, idx):
        if not 0 <= idx < len(self.scene_list):
            raise IndexError("scene index out of range!")
        scene = self.scene_list[idx]
        reference = self._load_scan(scene["reference"])
        rescans = []
        for scan in scene["scans"]:
            rescan = self._load_scan(scan["reference"])
            if rescan is None:
                continue
            scene_tsfm = _mat4(scan["transform"], self.device)
            moving, static = [], []
            for rigid in scan["rigid"]:
                obj_tsfm = inverse_3d_transform(_mat4(rigid["transform"], self.device))      # object: rescan -> reference
                rot_diff = rotation_error(obj_tsfm[:, :3, :3], scene_tsfm[:, :3, :3])
                t_diff = translation_error(obj_tsfm[:, :3, 3], scene_tsfm[:, :3, 3])
                (moving if (float(rot_diff) > 1 or float(t_diff) > 0.05) else static).append(rigid["instance_reference"])
            rescan["moving_ids"] = torch.tensor(moving, dtype=torch.float32, device=self.device)
            rescan["static_ids"] = torch.tensor(static, dtype=torch.float32, device=self.device)
            rescan["rescan2ref_tsfm"] = scene_tsfm
            rescans.append(rescan)
        return reference, rescans


# ------------------------------------------------------------------------------------------------ writers (synthetic fixtures)
def write_scan(data_path, scan_id, points, object_ids, seg_groups, binary=True, extra_uchar=True):
    """One scan directory: ``points`` [n,3], ``object_ids`` [n] int, ``seg_groups`` [{'objectId', 'label'}]."""
    d = osp.join(data_path, scan_id)
    os.makedirs(d, exist_ok=True)
    pts = np.asarray(points, np.float32)
    n = len(pts)
    with open(osp.join(d, "pointcloud.instances.align.ply"), "wb") as f:
        hdr = ["ply", f"format {'binary_little_endian' if binary else 'ascii'} 1.0", "comment synthetic", f"element vertex {n}",
               "property float x", "property float y", "property float z"]
        if extra_uchar:
            hdr += ["property uchar red", "property uchar green", "property uchar blue"]
        hdr += ["element face 0", "property list uchar int vertex_indices", "end_header"]
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        if binary:
            dt = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")] + ([("r", "u1"), ("g", "u1"), ("b", "u1")] if extra_uchar else [])
            v = np.zeros(n, dtype=dt)
            v["x"], v["y"], v["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
            f.write(v.tobytes())
        else:
            for p in pts:
                f.write((" ".join(repr(float(c)) for c in p) + (" 10 20 30" if extra_uchar else "") + "\n").encode("ascii"))
    np.savez(osp.join(d, "pointcloud.labels.npz"), objectId=np.asarray(object_ids))
    with open(osp.join(d, "semseg.v2.json"), "w") as f:
        json.dump({"scan_id": scan_id, "segGroups": list(seg_groups)}, f)
    return d


def write_index(root_path, split, scenes):
    """3RScan.json + ../splits/<split>.txt for ``scenes`` (the json structure documented at the top of this file)."""
    os.makedirs(osp.join(root_path, "..", "splits"), exist_ok=True)
    with open(osp.join(root_path, "3RScan.json"), "w") as f:
        json.dump(list(scenes), f)
    with open(osp.join(root_path, "..", f"splits/{split}.txt"), "w") as f:
        f.write("\n".join(s["reference"] for s in scenes) + "\n")
