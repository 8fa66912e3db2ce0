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
        return ops.kabsch_codes(code1["z_so3"], code1["t"], code2["z_so3"], code2["t"])   # z_so3 + t formed inside the launch

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
        """more_solver.py:118-189 (optim=True) for ONE pair: the batched path with P = 1 (same arithmetic: a pair's trajectory does
        not depend on the batch it rides in)."""
        return self._solve_pairwise_registration_optim_batch([pc1_full[0]], [pc2_full[0]])

    def _solve_pairwise_registration_optim_batch(self, pcs1, pcs2, icp=True, return_info=False):
        """more_solver.py:118-189 (optim=True) for P pairs IN LOCK-STEP -- what eval_3rscan.py:381 runs per matched instance: keep the
        code that explains its own points better (:124-135), refine the Kabsch pose by Adam on SE(3) against
        SmoothL1(sdf(g.src; shared code)) + Sinkhorn(g.src, tgt) (:137-173: MultiStepLR [300, 340, 380], best-loss snapshot after the
        step, geodesic early stop), invert for the reversed direction (:175-179), then ICP (:181-187).
        Per step the host only sequences launches: decoder forward + backward on P x N queries (ls_sdf_decode_train /
        ls_sdf_backward), per-pair SmoothL1 + its gradient, the batched Sinkhorn softmins, and ONE kernel for the tangent gradient,
        the Adam moments, the retraction, the snapshot, the early-stop test and the next transformed cloud (csrc/optim.hip).
        torchlie / geomloss / roma are not installed: the manifold update (left-multiplicative retraction g <- exp(-step) g, Adam
        moments on the 6-vector (v, omega) of the left tangent space) and the Sinkhorn divergence follow this build's documented
        definitions -- PARITY UNPINNED.  lists of clouds [Ni,3] / [Mi,3] -> R [P,3,3], t [P,3,1] mapping pc1 -> pc2."""
        from .. import _lib
        from ..sinkhorn import divergence_batch
        n_in = self.cfg["shape_priors"]["n_input_point"]
        assert self.cfg.get("fps", {}).get("n_init", 1) == 1, "fps.n_init > 1 is not used by the released configs"
        reg = self.cfg["registration"]
        lr0, n_steps, stop = reg["step_size"]["so3"], reg["n_steps"], reg["early_stop_threshold"]
        P = len(pcs1)
        pc1, pc2 = self._sample(pcs1, n_in), self._sample(pcs2, n_in)
        hip = self.model.hip_model()
        with torch.no_grad():
            code = self.model.encode(torch.cat([pc1, pc2], 0).transpose(-1, -2).contiguous())
            c1 = {k: v[:P] for k, v in code.items()}
            c2 = {k: v[P:] for k, v in code.items()}
            se1, se2 = c1["z_so3"] + c1["t"], c2["z_so3"] + c2["t"]
            R12, t12, _, _ = kabsch_transformation_estimation(se1, se2)
            R21, t21, _, _ = kabsch_transformation_estimation(se2, se1)
            err1 = self.model.decoder(pc1, None, c1, return_sdf=True).abs().mean(1)
            err2 = self.model.decoder(pc2, None, c2, return_sdf=True).abs().mean(1)
            reverse = err1 < err2                                 # keep the code that explains its own points better (:124-135)

            def pick(a, b):                                       # a where reversed, b otherwise
                return torch.where(reverse.view(-1, *([1] * (a.dim() - 1))), a, b)
            shared = {k: pick(c1[k], c2[k]).contiguous() for k in ("z_so3", "z_inv", "s", "t")}
            src, tgt = pick(pc2, pc1).contiguous(), pick(pc1, pc2).contiguous()
            g0 = torch.cat([pick(R21, R12), pick(t21, t12)], 2).contiguous()
            two_piece = int(reg.get("decoder_bf16_pieces", 3)) == 2
            opt, steps_run = self._refine_se3(shared, src, tgt, g0, n_steps, lr0, stop, two_piece=two_piece)
            best = opt.best_g
            Rb = best[:, :, :3].transpose(1, 2)
            inv = torch.cat([Rb, -(Rb @ best[:, :, 3:4])], 2)
            best = pick(inv, best)
            R, t = best[:, :, :3].contiguous(), best[:, :, 3:4].contiguous()
        if icp:
            R, t = self._icp(pc1, pc2, R, t)
        if return_info:
            return R, t, {"reverse": reverse, "min_loss": opt.min_loss, "active": opt.active, "steps": steps_run, "pre_icp": best}
        return R, t

    def _refine_se3(self, shared, src, tgt, g0, n_steps, lr0, stop, two_piece=False, trace=None):
        """The refinement loop of more_solver.py:137-173 for P pairs in lock-step (csrc/optim.hip; oracle twin: oracle/optim.py
        registration_loop): shared = the code dict the decoder is conditioned on (P rows), src / tgt [P,N,3] / [P,M,3], g0 [P,3,4].
        -> (ops.Se3Adam state, steps run).  ``trace`` (list) receives (g [P,3,4], loss [P]) after every step (tests)."""
        from .. import _lib
        from ..sinkhorn import divergence_batch
        hip = self.model.hip_model()
        # batch-invariant decoder arithmetic (P pairs == each pair alone); opt-in (not in the reference's yaml):
        # registration.decoder_bf16_pieces = 2 runs the refinement's decoder GEMMs with two-piece bf16 products (2^-16 per product).
        # The handle's previous settings are restored afterwards (a caller-chosen option is not clobbered).
        prev_split = hip.set_option(_lib.OPT_SDF_TRAIN_SPLITK, 0)
        prev_x2 = hip.set_option(_lib.OPT_SDF_BF16X2, 1) if two_piece else None
        try:
            opt = ops.Se3Adam(g0, src, stop)
            steps_run = 0
            # length of the Sinkhorn epsilon-schedule loop: read back once, then guessed as (largest seen + 1) and VERIFIED at the
            # host read every 16 steps (a schedule grows by one entry when a pair's bounding-box diameter doubles; 16 steps of
            # at most lr radians / units each cannot do that) -- no device -> host round trip inside a step
            sched_len, needs = None, []
            for i in range(n_steps):
                lr = lr0 * (0.1 ** sum(i >= ms for ms in (300, 340, 380)))          # MultiStepLR([300, 340, 380], 0.1), :143
                sdf, saved = hip.sdf_decode_train(opt.query, shared["z_so3"], shared["z_inv"], shared["s"], shared["t"])
                loss, gsdf = ops.smooth_l1(sdf)
                gq = hip.sdf_backward(saved, gsdf, need_code_grad=False)[0]              # the code is fixed here (:137-141)
                sl, sg, need = divergence_batch(opt.query, tgt, lmax=sched_len, return_need=True)
                if sched_len is None:
                    sched_len = int(need) + 1
                needs.append(need)
                opt.step(gq + sg, loss + sl, lr)
                steps_run = i + 1
                if trace is not None:
                    trace.append((opt.g.clone(), (loss + sl).clone()))
                if i % 16 == 15:                                            # one host read per 16 steps
                    worst = int(torch.stack(needs).max())
                    if worst > sched_len:
                        # a pose ran away far enough to double a bounding box within 16 steps: those steps used a schedule one
                        # entry short (a coarser final epsilon for that pair); widen and carry on rather than abort the whole batch
                        import warnings
                        warnings.warn(f"Sinkhorn schedule grew from {sched_len} to {worst} entries within 16 steps (diverging pose?)")
                    sched_len, needs = worst + 1, []
                    if not bool(opt.active.any()):                          # every pair stopped early
                        break
        finally:
            hip.set_option(_lib.OPT_SDF_TRAIN_SPLITK, prev_split)
            if two_piece:
                hip.set_option(_lib.OPT_SDF_BF16X2, prev_x2)
        return opt, steps_run

    def _sample(self, pcs, n_in):
        """Ragged FPS of a list of clouds [Ni,3] to n_in points each: ONE launch."""
        dev = pcs[0].device
        lens = torch.tensor([p.shape[0] for p in pcs], device=dev)
        buf = torch.zeros(len(pcs), int(lens.max()), 3, device=dev)
        for i, p in enumerate(pcs):
            buf[i, : p.shape[0]] = p
        idx = ops.fps(buf, n_in, lengths=lens)
        return torch.gather(buf, 1, idx.long()[..., None].expand(-1, -1, 3))

    def _solve_pairwise_registration_batch(self, pcs1, pcs2, icp=True):
        """Batched form: lists of clouds [Ni,3] / [Mi,3] -> R [P,3,3], t [P,3,1].  One ragged FPS launch per side,
        ONE encoder batch of 2P instances, one Kabsch launch, one ICP launch."""
        n_in = self.cfg["shape_priors"]["n_input_point"]
        assert self.cfg.get("fps", {}).get("n_init", 1) == 1, "fps.n_init > 1 is not used by the released configs"
        P = len(pcs1)
        pc1, pc2 = self._sample(pcs1, n_in), self._sample(pcs2, n_in)
        with torch.no_grad():
            code = self.model.encode(torch.cat([pc1, pc2], 0).transpose(-1, -2).contiguous())
        c1 = {k: v[:P] for k, v in code.items()}
        c2 = {k: v[P:] for k, v in code.items()}
        R, t = self._register_from_codes(c1, c2)
        if icp:
            R, t = self._icp(pc1, pc2, R, t)
        return R, t

    def _optimize_code(self, code, pc, mask, n_steps=200):
        """more_solver.py:191-228 for one instance (the batched path with P = 1)."""
        valid_pc = pc.T[mask.squeeze()].squeeze()
        best, improved = self._optimize_code_batch(code, [valid_pc], n_steps=n_steps)
        return best if bool(improved[0]) else None

    def _optimize_code_batch(self, code, pcs, n_steps=200):
        """more_solver.py:191-228 for P instances at once (eval_3rscan.py:489 calls it per instance): Adam on (z_inv, t, z_so3) against
        MSE(sdf(pc; code), 0), lr 1e-5 / 1e-4 / 5e-4, x0.1 at step 160.  `code` holds P rows; pcs = list of P clouds [Ni,3].  The
        loss is the SUM over instances of their mean-squared SDF, so every instance sees exactly the gradients (and, Adam being
        element-wise, exactly the updates) of its own run; the decoder forward / backward run in the HIP library for all P x N
        queries per step (ls_sdf_decode_train / ls_sdf_backward, split-K off: a row's value does not depend on the batch).
        The reference snapshots `best_code` with .detach() WITHOUT .clone(), i.e. the snapshot aliases the live parameters and ends
        up holding the FINAL values whenever the loss improved at least once (min_loss starts at 100): returned here as (code dict
        of the final values, improved [P] bool) -- an instance that never improved has best_code = None in the reference."""
        from .. import _lib
        n_in = self.cfg["shape_priors"]["n_input_point"]
        pc = self._sample(pcs, n_in)
        P = pc.shape[0]
        dev = pc.device
        # the live parameters (updated in place by the device Adam; written back into the caller's tensors at the end: in the reference
        # the caller's code tensors ARE the optimizer's parameters, more_solver.py:195-200)
        z_inv = code["z_inv"].detach().float().contiguous().clone()
        t = code["t"].detach().float().reshape(P, 3).contiguous().clone()
        z_so3 = code["z_so3"].detach().float().contiguous().clone()
        s_ = code["s"].detach().float().contiguous()
        opt = ops.Adam([(z_inv, 1e-5), (t, 1e-4), (z_so3, 5e-4)])            # more_solver.py:199-203
        min_loss = torch.full((P,), 100.0, device=dev)                        # :205
        improved = torch.zeros(P, dtype=torch.int32, device=dev)
        hip = self.model.hip_model()
        prev_split = hip.set_option(_lib.OPT_SDF_TRAIN_SPLITK, 0)
        try:
            # per step: decoder forward (activations kept), MSE + its gradient + the best-loss bookkeeping, decoder backward w.r.t. the
            # code, ONE Adam launch for the three tensors -- no autograd graph, no ATen kernel, no host read inside the loop
            for i in range(n_steps):
                sdf, saved = hip.sdf_decode_train(pc, z_so3, z_inv, s_, t)
                _, gsdf = ops.mse(sdf, min_loss, improved)                       # MSELoss(sdf, 0) of every instance (:213), :219-221
                _, gso3, ginv, _, gt = hip.sdf_backward(saved, gsdf, need_query_grad=False)
                opt.step([ginv, gt, gso3], lr_scale=0.1 if i >= 160 else 1.0)   # MultiStepLR([160], 0.1) (:204)
        finally:
            hip.set_option(_lib.OPT_SDF_TRAIN_SPLITK, prev_split)
        with torch.no_grad():
            code["z_inv"].copy_(z_inv.reshape(code["z_inv"].shape))
            code["z_so3"].copy_(z_so3.reshape(code["z_so3"].shape))
            code["t"].copy_(t.reshape(code["t"].shape))
        improved = improved.bool()
        return {k: code[k].detach() for k in ("z_inv", "z_so3", "s", "t")}, improved

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
        meshes.  ref / rescan: {'pc' [n,3,Nmax], 'pc_mask'}.  Both modes register all matched pairs of the
        scene in one batched call (optim=True: the 400-step refinement in lock-step)."""
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
            if optim:   # all matched pairs of the scene advance in lock-step (the reference loops over them, more_solver.py:272-282)
                R, t = self._solve_pairwise_registration_optim_batch([ref_full[i] for i, _ in pairs], [res_full[j] for _, j in pairs])
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


def solve_end2end_batch(solver, pairs, mesh=False, optim=False, sharded=False, optim_chunk=128):
    """Batched form of More_Solver._solve_end2end over MANY (reference scan, rescan) pairs -- an extension the reference lacks
    (eval_3rscan.py walks the scenes one pair at a time): every scan of every pair goes through ONE ragged FPS launch and ONE
    encoder batch, every matched pair of every scene through ONE registration batch (ragged FPS, encode, Kabsch, ICP; optim=True:
    the 400-step refinement in lock-step, optim_chunk pairs per call).  The ragged FPS runs one workgroup per raw cloud (18 ms for a
    60 000-point cloud), so a single scene pair leaves the GPU idle; hundreds of clouds per launch fill it.  Returns one dict per
    pair, as _solve_end2end.

    sharded=True (one process per GPU, torch.distributed initialised; SURVEY.md 8e, eval_3rscan.py:337-463 sharded over the node):
    the flat (scene, instance) list is block-partitioned over the ranks for FPS + encode and the codes all-gathered (4.1 KB each);
    the per-scene matchers run replicated (deterministic, 32 x 32); the flat list of matched pairs is block-partitioned again for
    the registration and the (R | t) rows all-gathered (48 B per pair); every rank then holds every pose and every transformed code,
    and meshes the pairs of ITS block (the SDF grids and meshes stay on the rank that made them: mesh_lst is None elsewhere)."""
    from .. import sharding
    scans, where = [], []
    for ref, res in pairs:
        where.append((len(scans), len(scans) + 1))
        scans += [ref, res]
    clouds = [_scene_clouds(s) for s in scans]
    flat = [c for cl in clouds for c in cl]
    dev = flat[0].device
    if sharded:
        codes = sharding.sharded_encode_fps(solver.model, flat)
    else:
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
        def register(lo, hi):
            if not optim:
                return solver._solve_pairwise_registration_batch(reg1[lo:hi], reg2[lo:hi])
            Rs, ts = [], []
            for c0 in range(lo, hi, optim_chunk):
                Rc, tc = solver._solve_pairwise_registration_optim_batch(reg1[c0:min(hi, c0 + optim_chunk)], reg2[c0:min(hi, c0 + optim_chunk)])
                Rs.append(Rc), ts.append(tc)
            return torch.cat(Rs, 0), torch.cat(ts, 0)
        R, t = sharding.sharded_pairs(len(slots), register) if sharded else register(0, len(slots))
        mine = range(*sharding.shard_range(len(slots))) if sharded else range(len(slots))     # the pairs this rank meshes
        T = Rt_to_SE3(R, t)
        for k, (p, i, j) in enumerate(slots):
            out = outs[p]
            out["registration"][i] = T[k:k + 1]
            cur = {key: out["_res_codes"][key][j][None] for key in ("z_so3", "z_inv", "s", "t")}
            out["codes"][i] = solver._transform_latent(cur, inverse(T[k:k + 1]))
        if mesh:
            import numpy as np
            group = 16                                     # instances whose MISE rounds advance in lock-step
            own = [slots[k] for k in mine]
            for g0 in range(0, len(own), group):
                part = own[g0:g0 + group]
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


def _se3_exp(xi):  # host twin of the retraction in csrc/optim.hip (tests compare both with torch.matrix_exp)
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

