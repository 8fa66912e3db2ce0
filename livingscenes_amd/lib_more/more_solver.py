"""Mirror of /root/reference/lib_more/more_solver.py (class More_Solver) for the accelerated path:
_solve_object_matching, _solve_pairwise_registration(optim=False), _transform_latent and the encode / match / register
part of _solve_end2end, plus batched variants the reference lacks (it registers one pair at a time with B=1 encoder
calls, eval_flyingshape.py:130).  _optimize_code and the optim=True registration branch run with the decoder
forward / backward and the Sinkhorn softmins in the HIP library; torchlie / geomloss / roma are not installed, so the SE(3)
update and the Sinkhorn divergence of that branch follow this build's documented definitions (parity unpinned); mesh extraction (_mesh_from_latent / _mesh_from_pc, 8 f-2) runs MISE and marching cubes on the device
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
            return self._solve_pairwise_registration_optim(pc1_full, pc2_full)
        return self._solve_pairwise_registration_batch([pc1_full[0]], [pc2_full[0]])

    def _solve_pairwise_registration_optim(self, pc1_full, pc2_full):
        """more_solver.py:118-189 (optim=True): refine the Kabsch pose by Adam on SE(3) against SmoothL1(sdf(g.src; shared code))
        + Sinkhorn(g.src, tgt), best-loss snapshot, then ICP.  The decoder forward / backward and the Sinkhorn softmins run in the
        HIP library.  torchlie / geomloss / roma are not installed: the manifold update and the Sinkhorn divergence follow this
        build's documented definitions (PARITY UNPINNED): left-multiplicative retraction g <- exp(-step) g with Adam
        moments kept on the 6-vector (v, omega) of the left tangent space; livingscenes_amd/sinkhorn.py."""
        from ..sinkhorn import sinkhorn_divergence
        n_in = self.cfg["shape_priors"]["n_input_point"]
        assert self.cfg.get("fps", {}).get("n_init", 1) == 1, "fps.n_init > 1 is not used by the released configs"
        pc1, _ = fps(pc1_full, K=n_in)
        pc2, _ = fps(pc2_full, K=n_in)
        with torch.no_grad():
            code1 = self.model.encode(pc1.transpose(-1, -2).contiguous())
            code2 = self.model.encode(pc2.transpose(-1, -2).contiguous())
            code1_se3, code2_se3 = code1["z_so3"] + code1["t"], code2["z_so3"] + code2["t"]
            R, t, _, _ = kabsch_transformation_estimation(code1_se3, code2_se3)
            sdf_error1 = self.model.decoder(pc1, None, code1, return_sdf=True).abs().mean()
            sdf_error2 = self.model.decoder(pc2, None, code2, return_sdf=True).abs().mean()
        reverse = bool(sdf_error1 < sdf_error2)            # keep the code that explains its own points better (:124-135)
        if reverse:
            shared_code, src_pc, tgt_pc = code1, pc2, pc1
            with torch.no_grad():
                R, t, _, _ = kabsch_transformation_estimation(code2_se3, code1_se3)
        else:
            shared_code, src_pc, tgt_pc = code2, pc1, pc2
        shared_code = {k: v.detach() for k, v in shared_code.items()}
        reg = self.cfg["registration"]
        lr0, n_steps, stop = reg["step_size"]["so3"], reg["n_steps"], reg["early_stop_threshold"]
        g = torch.cat([R, t], 2)[0].detach().clone()        # [3,4]
        init_R = g[:, :3].clone()
        m1 = torch.zeros(6, device=g.device)
        m2 = torch.zeros(6, device=g.device)
        b1, b2, eps_adam = 0.9, 0.999, 1e-8
        min_loss, best_g = 100.0, g.clone()
        src = src_pc[0]
        for i in range(n_steps):
            lr = lr0 * (0.1 ** sum(i >= ms for ms in (300, 340, 380)))          # MultiStepLR([300,340,380], 0.1), :143
            query = (src @ g[:, :3].T + g[:, 3]).detach().requires_grad_(True)
            sdf = self.model.decoder(query[None], None, shared_code, return_sdf=True)
            loss = torch.nn.functional.smooth_l1_loss(sdf, torch.zeros_like(sdf)) + sinkhorn_divergence(query[None], tgt_pc)
            loss.backward()
            G = query.grad
            with torch.no_grad():
                grad = torch.cat([G.sum(0), torch.cross(query.detach(), G, dim=1).sum(0)])   # d loss / d (v, omega), left tangent
                m1 = b1 * m1 + (1 - b1) * grad
                m2 = b2 * m2 + (1 - b2) * grad * grad
                step = lr * (m1 / (1 - b1 ** (i + 1))) / ((m2 / (1 - b2 ** (i + 1))).sqrt() + eps_adam)
                g = _se3_exp(-step) @ torch.cat([g, g.new_tensor([[0.0, 0.0, 0.0, 1.0]])], 0)
                g = g[:3]
                if float(loss) < min_loss:                   # snapshot AFTER the step, as the reference (:166-168)
                    min_loss, best_g = float(loss), g.clone()
                cosang = ((g[:, :3] @ init_R.T).diagonal().sum() - 1) / 2
                if float(torch.acos(cosang.clamp(-1, 1))) > stop:   # radians against the configured number, as the reference (:172-173)
                    break
        if reverse:
            Rb = best_g[:, :3].T
            best_g = torch.cat([Rb, -(Rb @ best_g[:, 3:4])], 1)
        R, t = best_g[None, :, :3].contiguous(), best_g[None, :, 3:4].contiguous()
        return self._icp(pc1, pc2, R, t)

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

    def _solve_end2end(self, ref, rescan, optim=False, mesh=None):
        """more_solver.py:246-299: encode both scenes (one batch each), sequential matching, registration of the matched pairs,
        transformed latent codes and (``mesh``, default = whether cfg has a mesh_extractor, as the reference always meshes) their
        meshes.  ref / rescan: {'pc' [n,3,Nmax], 'pc_mask'}.  optim=False registers all pairs in one batched call; optim=True
        runs the per-pair optimisation loop like the reference."""
        if ref is None:
            return None
        if mesh is None:
            mesh = self.mesh_extractor is not None

        def valid_clouds(scene):
            return [pc.T[mask.reshape(-1).bool()] for pc, mask in zip(scene["pc"], scene["pc_mask"])]
        ref_full, res_full = valid_clouds(ref), valid_clouds(rescan)
        ref_codes = self.model.encode_fps(ref["pc"], ref["pc_mask"])
        res_codes = self.model.encode_fps(rescan["pc"], rescan["pc_mask"])
        matches = self._solve_object_matching(ref_codes, res_codes, "sequential")
        m0 = matches["matches0"]
        out = {"ref_pc_lst": ref_full, "rescan_pc_lst": res_full, "matches": m0, "registration": [None] * len(ref_full),
               "codes": [None] * len(ref_full), "mesh_lst": [None] * len(ref_full)}
        pairs = [(i, int(j)) for i, j in enumerate(m0.tolist()) if j >= 0]
        if pairs:
            if optim:
                Rt = [self._solve_pairwise_registration(ref_full[i][None], res_full[j][None], optim=True) for i, j in pairs]
                R, t = torch.cat([r for r, _ in Rt], 0), torch.cat([tt for _, tt in Rt], 0)
            else:
                R, t = self._solve_pairwise_registration_batch([ref_full[i] for i, _ in pairs], [res_full[j] for _, j in pairs])
            T = Rt_to_SE3(R, t)
            for k, (i, j) in enumerate(pairs):
                out["registration"][i] = T[k:k + 1]
                cur = {key: res_codes[key][j][None] for key in ("z_so3", "z_inv", "s", "t")}
                out["codes"][i] = self._transform_latent(cur, inverse(T[k:k + 1]))
                if mesh:
                    out["mesh_lst"][i] = self._mesh_from_latent(out["codes"][i])
        return out


def _scene_clouds(scene):
    return [pc.T[mask.reshape(-1).bool()] for pc, mask in zip(scene["pc"], scene["pc_mask"])]


def solve_end2end_batch(solver, pairs, mesh=False):
    """Batched form of More_Solver._solve_end2end over MANY (reference scan, rescan) pairs -- an extension the reference lacks
    (eval_3rscan.py walks the scenes one pair at a time): every scan of every pair goes through ONE ragged FPS launch and ONE
    encoder batch, every matched pair of every scene through ONE registration batch (ragged FPS, encode, Kabsch, ICP).  The
    ragged FPS runs one workgroup per raw cloud (18 ms for a 60 000-point cloud), so a single scene pair leaves the GPU idle;
    hundreds of clouds per launch fill it.  Returns one dict per pair, as _solve_end2end (optim=False)."""
    scans, where = [], []
    for ref, res in pairs:
        where.append((len(scans), len(scans) + 1))
        scans += [ref, res]
    clouds = [_scene_clouds(s) for s in scans]
    flat = [c for cl in clouds for c in cl]
    dev = flat[0].device
    lens = torch.tensor([c.shape[0] for c in flat], device=dev)
    buf = torch.zeros(len(flat), 3, int(lens.max()), device=dev)
    mask = torch.zeros(len(flat), 1, int(lens.max()), dtype=torch.bool, device=dev)
    for i, c in enumerate(flat):
        buf[i, :, : c.shape[0]] = c.T
        mask[i, :, : c.shape[0]] = True
    codes = solver.model.encode_fps(buf, mask)
    starts = [0]
    for cl in clouds:
        starts.append(starts[-1] + len(cl))
    outs, reg1, reg2, slots = [], [], [], []
    for p, (a, b) in enumerate(where):
        cr = {k: v[starts[a]:starts[a + 1]] for k, v in codes.items()}
        cs = {k: v[starts[b]:starts[b + 1]] for k, v in codes.items()}
        m0 = solver._solve_object_matching(cr, cs, "sequential")["matches0"]
        n = len(clouds[a])
        out = {"ref_pc_lst": clouds[a], "rescan_pc_lst": clouds[b], "matches": m0, "registration": [None] * n, "codes": [None] * n,
               "mesh_lst": [None] * n, "_res_codes": cs}
        for i, j in enumerate(m0.tolist()):
            if j >= 0:
                reg1.append(clouds[a][i]); reg2.append(clouds[b][j]); slots.append((p, i, j))
        outs.append(out)
    if slots:
        R, t = solver._solve_pairwise_registration_batch(reg1, reg2)
        T = Rt_to_SE3(R, t)
        for k, (p, i, j) in enumerate(slots):
            out = outs[p]
            out["registration"][i] = T[k:k + 1]
            cur = {key: out["_res_codes"][key][j][None] for key in ("z_so3", "z_inv", "s", "t")}
            out["codes"][i] = solver._transform_latent(cur, inverse(T[k:k + 1]))
        if mesh:
            import numpy as np
            group = 16                                     # instances whose MISE rounds advance in lock-step
            for g0 in range(0, len(slots), group):
                part = slots[g0:g0 + group]
                cl = [outs[p]["codes"][i] for p, i, _ in part]
                canon = {k: torch.cat([c[k] for c in cl], 0) for k in ("z_so3", "z_inv")}
                canon["t"] = torch.zeros_like(torch.cat([c["t"] for c in cl], 0))   # canonical pose, as model_utils.py:296-298
                canon["s"] = torch.ones_like(torch.cat([c["s"] for c in cl], 0))
                meshes = solver.mesh_extractor.generate_from_latent_batch(canon, solver.model.decoder)
                for (p, i, _), c, msh in zip(part, cl, meshes):
                    tsfm = np.eye(4) * c["s"].squeeze().item()
                    tsfm[-1, -1] = 1
                    tsfm[:3, 3] = c["t"].squeeze().view(-1).detach().cpu().numpy()
                    if hasattr(msh, "apply_transform"):
                        msh.apply_transform(tsfm)
                    else:
                        msh.vertices = msh.vertices @ tsfm[:3, :3].T + tsfm[:3, 3]
                    outs[p]["mesh_lst"][i] = msh
    for out in outs:
        del out["_res_codes"]
    return outs


def _se3_exp(xi):
    """exp of the twist (v, omega) in R^6 -> [4,4] (Rodrigues + the left Jacobian for the translation)."""
    v, w = xi[:3], xi[3:]
    th = w.norm()
    K = xi.new_zeros(3, 3)
    K[0, 1], K[0, 2], K[1, 0], K[1, 2], K[2, 0], K[2, 1] = -w[2], w[1], w[2], -w[0], -w[1], w[0]
    eye = torch.eye(3, device=xi.device, dtype=xi.dtype)
    if float(th) < 1e-6:
        R, V = eye + K, eye + 0.5 * K
    else:
        a, b, c = torch.sin(th) / th, (1 - torch.cos(th)) / th ** 2, (th - torch.sin(th)) / th ** 3
        R, V = eye + a * K + b * (K @ K), eye + b * K + c * (K @ K)
    out = torch.eye(4, device=xi.device, dtype=xi.dtype)
    out[:3, :3], out[:3, 3] = R, V @ v
    return out

