"""The build's counterpart of the reference's FlyingShape evaluation loops:
/root/reference/eval_flyingshape.py:62-107 (eval_matching) and :110-173 (eval_relocalization), on synthetic scenes with
the same npz semantics ({'pc': [n_obj,N,3], 'transform': [n_obj,4,4]}; GT = rescan_T @ inv(ref_T), :129; GT match =
identity permutation, :87; symmetry-folded RRE min(r, |180-r|, |90-r|), :140; recalls at 5/10 degrees, :160-168).
All instances of a scene are encoded in ONE batch and all pairs registered in ONE batched call (the reference loops
pair by pair with B=1 encoder calls, :130).  No dataset I/O: the real FlyingShape npz files are not in the tree."""
import numpy as np
import torch

from .lib_math.torch_se3 import concatenate, inverse
from .lib_more.pose_estimation import compute_transformation_error, rotation_error, translation_error


def scene_recall(ratios):
    r = np.asarray(ratios) * 100
    return {f"scene_recall@{t}": float((r >= t).mean() * 100) for t in (25, 50, 75, 100)}


@torch.no_grad()
def eval_matching(scenes, solver, method="sequential"):
    """scenes: list of dicts from synth.make_scene_pair.  -> metrics dict (eval_flyingshape.py:62-107)."""
    n_correct = n_total = 0
    ratios = []
    for sc in scenes:
        dev = next(solver.model.parameters()).device
        ref = sc["ref"].to(dev).transpose(1, 2).contiguous()
        res = sc["rescan"].to(dev).transpose(1, 2).contiguous()
        n = ref.shape[0]
        code = solver.model.encode(torch.cat([ref, res], 0))
        cr = {k: v[:n] for k, v in code.items()}
        cs = {k: v[n:] for k, v in code.items()}
        m = solver._solve_object_matching(cr, cs, method)["matches0"]
        ok = int((m == torch.arange(n, device=m.device)).sum())
        n_correct += ok
        n_total += n
        ratios.append(ok / n)
    out = {"object_recall": 100.0 * n_correct / max(n_total, 1)}
    out.update(scene_recall(ratios))
    return out


@torch.no_grad()
def eval_relocalization(scenes, solver, icp=True):
    """Pairwise registration of every (ref_i, rescan_i) instance pair (eval_flyingshape.py:110-173)."""
    rre, rte, te, poses = [], [], [], []
    for sc in scenes:
        dev = next(solver.model.parameters()).device
        ref, res = sc["ref"].to(dev), sc["rescan"].to(dev)
        n = ref.shape[0]
        R, t = solver._solve_pairwise_registration_batch([ref[i] for i in range(n)], [res[i] for i in range(n)], icp=icp)
        gt = concatenate(sc["rescan_T"].to(dev)[:, :3], inverse(sc["ref_T"].to(dev)[:, :3]))
        r = rotation_error(R, gt[:, :, :3]).reshape(-1)
        r = torch.minimum(torch.minimum(r, (180 - r).abs()), (90 - r).abs())  # symmetry fold (:140)
        rre.append(r.cpu())
        rte.append(translation_error(t, gt[:, :, 3:4]).reshape(-1).cpu())
        pred = torch.cat([R, t], 2)
        poses.append(pred.cpu())
        te.append(torch.stack([compute_transformation_error(ref[i:i + 1], res[i:i + 1], pred[i:i + 1], gt[i:i + 1]) for i in range(n)]).cpu())
    rre, rte, te = torch.cat(rre).numpy(), torch.cat(rte).numpy(), torch.cat(te).numpy()

    def med(x):
        return float(np.median(x)) if len(x) else float("nan")
    return {"recall_rre5": float((rre < 5).mean() * 100), "recall_rre10": float((rre < 10).mean() * 100),
            "median_rre_5": med(rre[rre < 5]), "median_rte_5": med(rte[rre < 5]), "te_cm_5": med(te[rre < 5]) * 100,
            "median_rre_all": med(rre), "rre": rre, "rte": rte, "te": te,
            "poses": torch.cat(poses).numpy()}   # [n,3,4] predicted (R | t), for parity checks at matrix level


# ------------------------------------------------------------------------------------------------ 3RScan matching evaluation
def disambiguate(pred_ids, gt_ids, ambiguity, max_hops=200):
    """/root/reference/eval_3rscan.py:189-228: 3RScan annotates symmetric / interchangeable instances of a scene as groups of
    {'instance_source', 'instance_target', 'transform'} links.  A prediction counts as the ground truth when the ground-truth id
    is reachable from the predicted id by following those links (source -> target, first matching link each hop, until the walk
    returns to its start, dead-ends, or ``max_hops``).  Returns a corrected copy of ``pred_ids``."""
    links = [(p["instance_source"], p["instance_target"]) for group in ambiguity for p in group]
    out = pred_ids.clone()
    for i in range(gt_ids.shape[0]):
        start = int(pred_ids[i])
        chain = [t for s, t in links if s == start]
        hops = 0
        while chain and hops < max_hops:
            nxt = next((t for s, t in links if s == chain[-1]), None)
            if nxt is None or nxt == start:
                break
            chain.append(nxt)
            hops += 1
        if int(gt_ids[i]) in chain:
            out[i] = gt_ids[i]
    return out


@torch.no_grad()
def eval_3rscan_matching(dataset, solver, method_list=("sequential",)):
    """Object matching between the reference scan and every rescan of each scene of a ``rscan.Dataset_3RScan``
    (eval_3rscan.py:232-335): object-level recall over the instances present in both scans (all / static / dynamic, the split
    coming from the rescan's rigid annotations), and scene-level recall = share of (scene, rescan) pairs whose hit ratio reaches
    75 / 50 / 25 %.  Codes come from ``model.encode_fps`` on the padded clouds, matches from ``solver._solve_object_matching``."""
    model = solver.model
    n_methods = len(method_list)
    n_total = 0
    n_correct = np.zeros(n_methods)
    scene_total = np.zeros(3)
    scene_count = np.zeros(3)      # hits @75, @50, @25
    tot_dyn = cor_dyn = tot_sta = cor_sta = 0
    for i_s, scene in enumerate(dataset.scene_list):
        ref, rescans = dataset._get_scene(i_s)
        if ref is None or len(rescans) == 0:
            continue
        ref_codes = model.encode_fps(ref["pc"], ref["pc_mask"])
        ref_ids = ref["objectId"]
        for rescan in rescans:
            codes = model.encode_fps(rescan["pc"], rescan["pc_mask"])
            res_ids = rescan["objectId"]
            moving = set(int(v) for v in rescan["moving_ids"].tolist())
            valid = torch.tensor([int(i) in set(res_ids.tolist()) for i in ref_ids.tolist()], device=ref_ids.device)
            moving_mask = torch.tensor([int(i) in moving for i in ref_ids.tolist()], device=ref_ids.device)
            for mi, method in enumerate(method_list):
                m0 = solver._solve_object_matching(ref_codes, codes, method)["matches0"].to(ref_ids.device)
                matched = res_ids[m0.clamp(min=0)]
                if len(scene.get("ambiguity", [])) != 0:
                    matched = disambiguate(matched.view(-1), ref_ids, scene["ambiguity"])
                matched = torch.where(m0 != -1, matched, torch.full_like(matched, -1))
                hit = matched == ref_ids
                n_match = int(valid.sum())
                ok = int(hit[valid].sum())
                n_correct[mi] += ok
                n_total += n_match
                scene_total += 1
                ratio = ok / n_match if n_match else 0.0
                if ratio >= 0.75:
                    scene_count[:] += 1
                elif ratio >= 0.5:
                    scene_count[1:] += 1
                elif ratio >= 0.25:
                    scene_count[2:] += 1
                tot_dyn += int((valid & moving_mask).sum()); cor_dyn += int(hit[valid & moving_mask].sum())
                tot_sta += int((valid & ~moving_mask).sum()); cor_sta += int(hit[valid & ~moving_mask].sum())
    per_method_total = n_total / max(n_methods, 1)
    pct = lambda a, b: 100.0 * a / b if b else float("nan")
    out = {f"object_recall[{m}]": pct(n_correct[i], per_method_total) for i, m in enumerate(method_list)}
    out.update({"static_recall": pct(cor_sta, tot_sta), "dynamic_recall": pct(cor_dyn, tot_dyn)})
    out.update({f"scene_recall@{t}": pct(scene_count[i], scene_total[i]) for i, t in enumerate((75, 50, 25))})
    return out


@torch.no_grad()
def eval_3rscan_relocalization(dataset, solver, optim=True):
    """Instance re-localisation over a ``rscan.Dataset_3RScan`` (eval_3rscan.py:337-456): for every annotated rigid instance that
    is present in both the reference scan and the rescan, register its reference cloud to its rescan cloud
    (``solver._solve_pairwise_registration``; the rescan is first moved back into its own frame with the inverse scene
    transform), then relative rotation error (folded by the annotation's symmetry class: 1 -> min(r, |180-r|), 2 -> also
    |90-r|), relative translation error, end-point RMSE and the chamfer distance of every tenth point.  Reported as the
    reference does: recall at RMSE < 0.1 m with the medians over RMSE < 0.2 m, recall at RRE < 10 deg with the medians over it,
    median chamfer distance."""
    from .evaluate import chamfer_distance_torch
    from .lib_math import torch_se3
    rre_l, rte_l, err_l, cd_l, shape_l = [], [], [], [], []
    for i_s, scene in enumerate(dataset.scene_list):
        ref, rescans = dataset._get_scene(i_s)
        if ref is None:
            continue
        dev = ref["pc"].device
        for rescan, sg in zip(rescans, [s for s in scene["scans"]]):
            scene_tsfm = rescan["rescan2ref_tsfm"]
            pc = torch_se3.transform(torch_se3.inverse(scene_tsfm), rescan["pc"].transpose(-1, -2)).transpose(-1, -2)
            ref_ids, res_ids = ref["objectId"].tolist(), rescan["objectId"].tolist()
            for rigid in sg["rigid"]:
                if rigid["instance_reference"] not in ref_ids or rigid.get("instance_rescan", rigid["instance_reference"]) not in res_ids:
                    continue
                gt = torch.tensor(rigid["transform"], dtype=torch.float32, device=dev).reshape(1, 4, 4).transpose(-1, -2).contiguous()
                a = ref_ids.index(rigid["instance_reference"])
                b = res_ids.index(rigid.get("instance_rescan", rigid["instance_reference"]))
                inst_ref = ref["pc"][a].T[ref["pc_mask"][a, 0]].unsqueeze(0).contiguous()
                inst_res = pc[b].T[rescan["pc_mask"][b, 0]].unsqueeze(0).contiguous()
                with torch.enable_grad():
                    R, t = solver._solve_pairwise_registration(inst_ref, inst_res, optim=optim)
                rre = float(rotation_error(R, gt[:, :3, :3]))
                sym = rigid.get("symmetry", 0)
                if sym == 1:
                    rre = min(rre, abs(180 - rre))
                elif sym == 2:
                    rre = min(rre, abs(180 - rre), abs(90 - rre))
                pred = torch_se3.Rt_to_SE3(R, t)
                rre_l.append(rre)
                rte_l.append(float(translation_error(t, gt[:, :3, 3:4])))
                err_l.append(float(compute_transformation_error(inst_ref, inst_res, pred, gt)))
                cd_l.append(float(chamfer_distance_torch(inst_ref[:, ::10].contiguous(), inst_res[:, ::10].contiguous(), pred, gt)))
                shape_l.append(ref["id_label"][[l[0] for l in ref["id_label"]].index(rigid["instance_reference"])][-1])
    rre, rte, err, cd = (np.asarray(v, dtype=np.float64) for v in (rre_l, rte_l, err_l, cd_l))
    med = lambda v, m: float(np.median(v[m])) if m.any() else float("nan")
    return {"n_pairs": int(len(rre)),
            "recall[T<0.1m]": float(100 * (err < 0.1).mean()) if len(err) else float("nan"),
            "rre_median[T<0.2m]": med(rre, err < 0.2), "rte_median[T<0.2m]": med(rte, err < 0.2),
            "recall[RRE<10deg]": float(100 * (rre < 10).mean()) if len(rre) else float("nan"),
            "rre_median[RRE<10deg]": med(rre, rre < 10), "rte_median[RRE<10deg]": med(rte, rre < 10),
            "chamfer_median": float(np.median(cd)) if len(cd) else float("nan"), "shape": shape_l}
