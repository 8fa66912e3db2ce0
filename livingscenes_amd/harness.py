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
    rre, rte, te = [], [], []
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
        te.append(torch.stack([compute_transformation_error(ref[i:i + 1], res[i:i + 1], pred[i:i + 1], gt[i:i + 1]) for i in range(n)]).cpu())
    rre, rte, te = torch.cat(rre).numpy(), torch.cat(rte).numpy(), torch.cat(te).numpy()

    def med(x):
        return float(np.median(x)) if len(x) else float("nan")
    return {"recall_rre5": float((rre < 5).mean() * 100), "recall_rre10": float((rre < 10).mean() * 100),
            "median_rre_5": med(rre[rre < 5]), "median_rte_5": med(rte[rre < 5]), "te_cm_5": med(te[rre < 5]) * 100,
            "median_rre_all": med(rre), "rre": rre, "rte": rte, "te": te}
