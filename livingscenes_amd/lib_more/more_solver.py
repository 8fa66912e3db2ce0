"""Mirror of /root/reference/lib_more/more_solver.py (class More_Solver) for the accelerated path:
_solve_object_matching, _solve_pairwise_registration(optim=False), _transform_latent and the encode / match / register
part of _solve_end2end, plus batched variants the reference lacks (it registers one pair at a time with B=1 encoder
calls, eval_flyingshape.py:130).  _optimize_code (code half) runs with the decoder forward / backward in the
HIP library; the SE(3) / Sinkhorn registration branch (optim=True: torchlie + geomloss + roma, none installed) raises
NotImplementedError; mesh extraction (_mesh_from_latent / _mesh_from_pc, 8 f-2) runs MISE and marching cubes on the device
(livingscenes_amd/mesh_extractor2.py)."""
import logging

import torch

from .. import ops
from ..lib_math.torch_se3 import Rt_to_SE3, inverse, transform
from ..mesh_extractor2 import Generator3D as Generator3D_MC
from ..model_utils import fps, load_ckpt_from_log, mesh_from_latent
from .matcher_new import eq_seq_matcher, nn_matcher, sequential_matcher, sim3_seq_matcher, sinkhorn_matcher
from .pose_estimation import kabsch_transformation_estimation


class More_Solver:
    def __init__(self, cfg, model=None):
        """cfg as configs/more_3rscan.yaml; ``model`` (a Shape_Prior) may be injected instead of loading
        cfg['shape_priors']['ckpt_dir'] (more_solver.py:26-34)."""
        logging.info("Configuring MoRE solver")
        self.cfg = cfg
        # Generator3D (device MISE + marching cubes): more_solver.py:30
        self.mesh_extractor = Generator3D_MC(**cfg["mesh_extractor"]) if "mesh_extractor" in cfg else None
        if model is None:
            model = load_ckpt_from_log(cfg["shape_priors"]["ckpt_dir"])[cfg["shape_priors"]["prior_name"]]
        self.model = model

    # -------------------------------------------------------------------------------------------- matching
    def _solve_object_matching(self, src_codes, tgt_codes, method):
        """more_solver.py:71-93."""
        inv_src = src_codes["z_inv"].detach().clone()
        inv_tgt = tgt_codes["z_inv"].detach().clone()
        if method == "nn":
            return nn_matcher(inv_src.T[None], inv_tgt.T[None])
        if method == "sinkhorn":
            return sinkhorn_matcher(inv_src.T[None], inv_tgt.T[None])
        if method == "sequential":
            return sequential_matcher(inv_src, inv_tgt)
        if method == "sim3_seq":
            return sim3_seq_matcher(src_codes, tgt_codes)
        if method == "eq_seq":
            return eq_seq_matcher(src_codes, tgt_codes)

    # -------------------------------------------------------------------------------------------- registration
    def _register_from_codes(self, code1, code2):
        """more_solver.py:114-116: Kabsch on the 256 equivariant pseudo-points z_so3 + t."""
        R, t, _, _ = kabsch_transformation_estimation(code1["z_so3"] + code1["t"], code2["z_so3"] + code2["t"])
        return R, t

    def _icp(self, pc1, pc2, R, t):
        """more_solver.py:181-187: ICP refinement from the Kabsch initialisation (row-vector convention inside)."""
        Ri, Ti, _, _ = ops.icp(pc1, pc2, R.transpose(-1, -2).contiguous(), t.squeeze(2).contiguous())
        return Ri.transpose(-1, -2), Ti.unsqueeze(2)

    def _solve_pairwise_registration(self, pc1_full, pc2_full, optim=False):
        """more_solver.py:95-189.  pc1 [1,N,3], pc2 [1,M,3] -> R [1,3,3], t [1,3,1] mapping pc1 -> pc2."""
        if optim:
            raise NotImplementedError("optimisation-based registration (torchlie/geomloss/decoder backward) is a "
                                      "SURVEY.md 8(f-1) 'next' row; use optim=False")
        return self._solve_pairwise_registration_batch([pc1_full[0]], [pc2_full[0]])

    def _solve_pairwise_registration_batch(self, pcs1, pcs2, icp=True):
        """Batched form: lists of clouds [Ni,3] / [Mi,3] -> R [P,3,3], t [P,3,1].  One ragged FPS launch per side,
        ONE encoder batch of 2P instances, one Kabsch launch, one ICP launch."""
        n_in = self.cfg["shape_priors"]["n_input_point"]
        assert self.cfg.get("fps", {}).get("n_init", 1) == 1, "fps.n_init > 1 is not used by the released configs"
        P = len(pcs1)
        dev = pcs1[0].device

        def sample(pcs):
            lens = torch.tensor([p.shape[0] for p in pcs], device=dev)
            buf = torch.zeros(len(pcs), int(lens.max()), 3, device=dev)
            for i, p in enumerate(pcs):
                buf[i, : p.shape[0]] = p
            idx = ops.fps(buf, n_in, lengths=lens)
            return torch.gather(buf, 1, idx.long()[..., None].expand(-1, -1, 3))
        pc1, pc2 = sample(pcs1), sample(pcs2)
        with torch.no_grad():
            code = self.model.encode(torch.cat([pc1, pc2], 0).transpose(-1, -2).contiguous())
        c1 = {k: v[:P] for k, v in code.items()}
        c2 = {k: v[P:] for k, v in code.items()}
        R, t = self._register_from_codes(c1, c2)
        if icp:
            R, t = self._icp(pc1, pc2, R, t)
        return R, t

    def _optimize_code(self, code, pc, mask, n_steps=200):
        """more_solver.py:191-228: Adam on (z_inv, t, z_so3) against MSE(sdf(pc; code), 0), lr 1e-5 / 1e-4 / 5e-4, x0.1 at step
        160, best-loss snapshot.  The decoder forward / backward run in the HIP library (ls_sdf_decode_train / ls_sdf_backward)."""
        valid_pc = pc.T[mask.squeeze()].squeeze()[None]
        pc, _ = fps(valid_pc, K=self.cfg["shape_priors"]["n_input_point"])
        params = [{"params": code["z_inv"], "lr": 1e-5}, {"params": code["t"], "lr": 1e-4}, {"params": code["z_so3"], "lr": 5e-4}]
        for p in params:
            p["params"].requires_grad_(True)
        optimizer = torch.optim.Adam(params)
        scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[160], gamma=0.1)
        loss_fn = torch.nn.MSELoss()
        min_loss, best_code = 100.0, None
        for _ in range(n_steps):
            optimizer.zero_grad()
            sdf_output = self.model.decoder(pc, None, code, return_sdf=True)
            loss = loss_fn(sdf_output, torch.zeros_like(sdf_output))
            loss.backward()
            optimizer.step()
            scheduler.step()
            if loss < min_loss:   # (the snapshot is taken AFTER the step, as in the reference)
                min_loss = loss.item()
                best_code = {k: code[k].detach() for k in ("z_inv", "z_so3", "s", "t")}
            optimizer.zero_grad()
        return best_code

    def _mesh_from_latent(self, latent_code):
        """more_solver.py:37-58: mesh of the canonical shape (t = 0, s = 1), then scaled and moved to the instance pose."""
        if self.mesh_extractor is None:
            raise ValueError("More_Solver: cfg has no 'mesh_extractor' section")
        return mesh_from_latent(self.mesh_extractor, latent_code, self.model.decoder)

    def _mesh_from_pc(self, pc):
        """more_solver.py:60-69."""
        pc_down, _ = fps(pc, K=self.cfg["shape_priors"]["n_input_point"])
        return self._mesh_from_latent(self.model.encode(pc_down.transpose(-1, -2)))

    def _transform_latent(self, code, tsfm):
        """more_solver.py:230-244: rotate z_so3, move t."""
        R = tsfm[:, :, :3]
        return {"z_so3": (code["z_so3"] @ R.transpose(-1, -2)).detach().clone(), "z_inv": code["z_inv"].detach().clone(),
                "t": transform(tsfm, code["t"]).detach().clone(), "s": code["s"].detach().clone()}

    def _solve_end2end(self, ref, rescan, optim=False, mesh=False):
        """more_solver.py:246-299 without the mesh step: encode both scenes (one batch each), sequential matching,
        batched registration of the matched pairs, transformed latent codes.  ref / rescan: {'pc' [n,3,Nmax], 'pc_mask'}."""
        if ref is None:
            return None
        if optim or mesh:
            raise NotImplementedError("optim / mesh branches are SURVEY.md 8(f) 'next' rows")

        def valid_clouds(scene):
            return [pc.T[mask.reshape(-1).bool()] for pc, mask in zip(scene["pc"], scene["pc_mask"])]
        ref_full, res_full = valid_clouds(ref), valid_clouds(rescan)
        ref_codes = self.model.encode_fps(ref["pc"], ref["pc_mask"])
        res_codes = self.model.encode_fps(rescan["pc"], rescan["pc_mask"])
        matches = self._solve_object_matching(ref_codes, res_codes, "sequential")
        m0 = matches["matches0"]
        out = {"ref_pc_lst": ref_full, "rescan_pc_lst": res_full, "matches": m0, "registration": [None] * len(ref_full),
               "codes": [None] * len(ref_full)}
        pairs = [(i, int(j)) for i, j in enumerate(m0.tolist()) if j >= 0]
        if pairs:
            R, t = self._solve_pairwise_registration_batch([ref_full[i] for i, _ in pairs], [res_full[j] for _, j in pairs])
            T = Rt_to_SE3(R, t)
            for k, (i, j) in enumerate(pairs):
                out["registration"][i] = T[k:k + 1]
                cur = {key: res_codes[key][j][None] for key in ("z_so3", "z_inv", "s", "t")}
                out["codes"][i] = self._transform_latent(cur, inverse(T[k:k + 1]))
        return out
