"""VolSDF error-bounded importance sampler with the reference's class names and signatures
(spurfies/model/ray_sampler.py:17-59 UniformSampler, :337-588 ErrorBoundSampler_pn).

Host structure mirrors the reference so that its random draws come from the CPU generator in
the same order (ray_sampler.py:55,514,550,562 — SURVEY.md F8): a host that issues the same
calls reproduces the reference's samples bit for bit.  The SDF evaluations go through
`model.sdf_importance`, i.e. the HIP kNN + fused geometry kernels; the per-ray arithmetic runs
on the GPU without host synchronisation in training mode (`fast=1`).
"""
from __future__ import annotations

import abc

import torch


class RaySampler(metaclass=abc.ABCMeta):
    def __init__(self, near, far):
        self.near = near
        self.far = far

    @abc.abstractmethod
    def get_z_vals(self, ray_dirs, cam_loc, model):
        ...


class UniformSampler(RaySampler):
    def __init__(self, scene_bounding_sphere, near, N_samples, take_sphere_intersection=False, far=-1):
        super().__init__(near, 2.0 * scene_bounding_sphere if far == -1 else far)
        self.N_samples = N_samples
        self.scene_bounding_sphere = scene_bounding_sphere
        if take_sphere_intersection:
            raise NotImplementedError("inverse_sphere_bg is off in every shipped config (vol/*.yaml)")

    def get_z_vals(self, ray_dirs, cam_loc, model, iter_step=None):
        dev = ray_dirs.device
        n = ray_dirs.shape[0]
        near = self.near * torch.ones(n, 1, device=dev)
        far = self.far * torch.ones(n, 1, device=dev)
        t = torch.linspace(0.0, 1.0, steps=self.N_samples).to(dev)
        z = near * (1.0 - t) + far * t
        if model.training:
            mids = 0.5 * (z[..., 1:] + z[..., :-1])
            upper = torch.cat([mids, z[..., -1:]], -1)
            lower = torch.cat([z[..., :1], mids], -1)
            t_rand = torch.rand(z.shape).to(dev)  # CPU generator, then moved (reference: torch.rand(...).cuda())
            z = lower + (upper - lower) * t_rand
        return z


class ErrorBoundSampler_pn(RaySampler):
    def __init__(self, scene_bounding_sphere, near, far, N_samples, N_samples_eval, N_samples_extra, eps, beta_iters,
                 max_total_iters, inverse_sphere_bg=False, N_samples_inverse_sphere=0, add_tiny=0.0):
        # the config's `far` is accepted and ignored, as in the reference (ray_sampler.py:342,353)
        super().__init__(near, 2.0 * scene_bounding_sphere)
        if inverse_sphere_bg:
            raise NotImplementedError("inverse_sphere_bg is off in every shipped config (vol/*.yaml)")
        self.N_samples = N_samples
        self.N_samples_eval = N_samples_eval
        self.uniform_sampler = UniformSampler(scene_bounding_sphere, near, N_samples_eval)
        self.N_samples_extra = N_samples_extra
        self.eps = eps
        self.beta_iters = beta_iters
        self.max_total_iters = max_total_iters
        self.scene_bounding_sphere = scene_bounding_sphere
        self.add_tiny = add_tiny
        self._last_iters, self._flags = 0, None
        self.last_points = None
        # evaluation mode: enqueue all iterations with device-side loop control instead of one host synchronisation per iteration
        # (spf_sampler_iter: flags / it); used when the model offers `sdf_importance_gated`.  False = the reference's host-side loop.
        self.device_loop = True
        # ... and each iteration of that loop as TWO launches behind the SDF kernel (spf_sampler_eval: test, then merge | final + finish + the
        # main pass's slot assignment) instead of ~14; False = the separate launches (tests compare both bit for bit)
        self.fused_eval = True
        self.last_slots = None      # (slot_sample int32 [R,SR], ray_valid uint8 [R]) of the main pass when the fused loop assigned them
        self.shard = None   # (rank, world) for ray-sharded batches: CPU-generator draws are made for the whole batch, this rank's rows kept
        self.draws = None   # sync-free / graph mode: {'t_rand' [R,128], 'u' [R,N_samples], 'sel' int32 [N_extra]} device tensors the
        #                     caller fills from the CPU generator (same calls, same order) before every step
        # Lemma-2 constant exactly as the reference forms it in float32 (ray_sampler.py:389)
        self._bound_coef = float(1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0))))
        self._lin = {}

    @property
    def last_iters(self):
        """Sampler iterations realised by the last get_z_vals call (reading it after a device-controlled loop synchronises once)."""
        if self._flags is not None:
            self._last_iters, self._flags = int(self._flags[0][: self._flags[1]].sum().item()), None
        return self._last_iters

    @last_iters.setter
    def last_iters(self, v):
        self._last_iters, self._flags = int(v), None

    def _linspace(self, n, dev):
        key = (n, str(dev))
        if key not in self._lin:
            self._lin[key] = torch.linspace(0.0, 1.0, steps=n).to(dev)
        return self._lin[key]

    def _rand_rows(self, R, n):
        """torch.rand([R, n]) from the CPU generator; for a ray-sharded batch the draw covers all ranks' rays and this rank keeps
        rows rank::world (the order dist.shard_rays deals rays out), so the generator advances as in a single-GPU run."""
        if self.shard is None:
            return torch.rand((R, n))
        rank, world = self.shard
        return torch.rand((R * world, n))[rank::world].contiguous()

    def get_z_vals(self, ray_dirs, cam_loc, model, fast=-1, iter_step=None):
        """Same contract as the reference (:377-574): returns (z_vals [R, N_samples + 2 + N_samples_extra], z_samples_eik).
        Per-ray arithmetic = three HIP kernels per call in training (uniform, iterate, finish); the main-pass points
        o + z d are left in `self.last_points`."""
        from .. import ops

        dev = ray_dirs.device
        R = ray_dirs.shape[0]
        ray_dirs, cam_loc = ray_dirs.detach().contiguous(), cam_loc.detach().contiguous()
        max_total_iters = fast if fast >= 0 else self.max_total_iters
        self.last_slots = None
        beta0 = (model.density.get_beta_value() if hasattr(model.density, "get_beta_value") else model.density.get_beta().detach()).reshape(1).contiguous()
        n0 = self.N_samples_eval
        if (not model.training and self.device_loop and max_total_iters > 1 and hasattr(model, "sdf_importance_gated") and ray_dirs.is_cuda):
            if (self.fused_eval and hasattr(model, "sdf_pairs_gated") and self.N_samples_eval <= 128 and self.N_samples_eval * max_total_iters <= 640
                    and self.N_samples + 2 + self.N_samples_extra <= 128):
                return self._z_vals_device_loop_fused(ray_dirs, cam_loc, model, max_total_iters, beta0)
            return self._z_vals_device_loop(ray_dirs, cam_loc, model, max_total_iters, beta0)
        ext = self.draws if (self.draws is not None and model.training) else None
        if ext is not None:
            t_rand = ext["t_rand"]
        else:
            t_rand = self._rand_rows(R, n0).to(dev) if model.training else None  # CPU generator, as the reference (:55)
        z_vals, points = ops.sampler_uniform(self._linspace(n0, dev), t_rand, cam_loc, ray_dirs, self.near, self.far)
        samples, samples_idx, sdf, beta = z_vals, None, None, None
        total_iters, not_converge = 0, True
        while not_converge and total_iters < max_total_iters:
            with torch.no_grad():
                s_sdf = model.sdf_importance(points.view(-1, 3)).view(R, -1)
            if samples_idx is not None:
                sdf = torch.gather(torch.cat([sdf, s_sdf], -1), 1, samples_idx)
            else:
                sdf = s_sdf
            total_iters += 1
            more = False
            if total_iters < max_total_iters:
                # convergence test needs beta first (a host sync, as in the reference :468): beta-only pass
                _, beta, _, _ = ops.sampler_iter(z_vals, sdf, beta, beta0, self.eps, self._bound_coef, self.beta_iters, False, 0.0, None, 0)
                more = bool(beta.max() > beta0)
                iters_left = 0                                   # beta is final: second pass only samples
            else:
                iters_left = self.beta_iters
            not_converge = more
            if more:
                N, u = self.N_samples_eval, self._linspace(self.N_samples_eval, dev)
            else:
                N = self.N_samples
                if not model.training:
                    u = self._linspace(N, dev)
                else:
                    u = ext["u"] if ext is not None else self._rand_rows(R, N).to(dev).contiguous()
            samples, beta, zm, mi = ops.sampler_iter(z_vals, sdf, beta, beta0, self.eps, self._bound_coef, iters_left, more, self.add_tiny, u, N)
            if more:
                z_vals, samples_idx = zm, mi.long()
                points = cam_loc.unsqueeze(1) + samples.unsqueeze(2) * ray_dirs.unsqueeze(1)
        self.last_iters = total_iters
        if total_iters == 0:                                     # fast=0: the reference then takes `samples = z_vals`
            samples = z_vals
        if self.N_samples_extra > 0:
            if ext is not None:
                sel = ext["sel"]
            elif model.training:
                sel = torch.randperm(z_vals.shape[1])[: self.N_samples_extra].to(torch.int32).to(dev)
            else:
                sel = torch.linspace(0, z_vals.shape[1] - 1, self.N_samples_extra).long().to(torch.int32).to(dev)
        else:
            sel = None
        z_out, self.last_points = ops.sampler_finish(samples.contiguous(), z_vals, sel, self.near, self.far, cam_loc, ray_dirs)
        if ext is not None:
            return z_out, None                                            # the caller drew (and dropped) the eikonal index
        rank, world = self.shard if (self.shard is not None and model.training) else (0, 1)
        idx = torch.randint(z_out.shape[-1], (z_out.shape[0] * world,))[rank::world].to(dev)   # consumes the generator like :562
        z_samples_eik = torch.gather(z_out, 1, idx.unsqueeze(-1))
        return z_out, z_samples_eik

    # ------------------------------------------------------------------ evaluation: the device-controlled loop, two launches per iteration
    def _z_vals_device_loop_fused(self, ray_dirs, cam_loc, model, max_iters, beta0):
        """_z_vals_device_loop with every iteration's sampler work in two launches (spf_sampler_eval) behind the neighbour search, the
        compaction and the SDF-only geometry launch of the iteration's NEW samples: "test" (per-sample SDF from the pair scratch and the previous
        row, beta bisection, the batch-wide convergence flag) and "step" (the flag decides: merge + next query points, or final samples +
        finish + the main pass's slot assignment); the last allowed iteration is one launch.  Same values as the separate launches."""
        from .. import ops

        dev, R = ray_dirs.device, ray_dirs.shape[0]
        n0, Nf, Ne = self.N_samples_eval, self.N_samples, self.N_samples_extra
        key = ("flags", max_iters, str(dev))
        if key not in self._lin:
            t = torch.zeros((max_iters + 2,), dtype=torch.int32)
            t[0] = 1
            self._lin[key] = t.to(dev)
        flags = self._lin[key].clone()
        z_vals, points = ops.sampler_uniform(self._linspace(n0, dev), None, cam_loc, ray_dirs, self.near, self.far)
        M, SR = Nf + 2 + Ne, int(model.conf.max_shading_pts)
        z_out = torch.empty((R, M), dtype=torch.float32, device=dev)
        pts_out = torch.empty((R, M, 3), dtype=torch.float32, device=dev)
        slot_sample = torch.empty((R, SR), dtype=torch.int32, device=dev)
        ray_valid = torch.empty((R,), dtype=torch.uint8, device=dev)
        beta = torch.empty((R,), dtype=torch.float32, device=dev)
        grid_h = model._grid()._h
        common = dict(beta=beta, beta0=beta0, eps=self.eps, bound_coef=self._bound_coef, add_tiny=self.add_tiny, beta_iters=self.beta_iters,
                      u_more=self._linspace(n0, dev), N_more=n0, u_fin=self._linspace(Nf, dev), N_fin=Nf, Ne=Ne, near=self.near, far=self.far,
                      cam_loc=cam_loc, ray_dirs=ray_dirs, z_out=z_out, points_out=pts_out, SR=SR, slot_sample=slot_sample, ray_valid=ray_valid, flags=flags)
        sdf_prev, merged, n_prev = None, None, 0
        for it in range(max_iters):
            n = z_vals.shape[1]
            with torch.no_grad():
                tmp, pl = model.sdf_pairs_gated(points.view(-1, 3), flags[it: it + 1])
            skey = ("sel", n, Ne, str(dev))
            if Ne > 0 and skey not in self._lin:
                self._lin[skey] = torch.linspace(0, n - 1, Ne).long().to(torch.int32).to(dev)
            sdf_cur = torch.empty((R, n), dtype=torch.float32, device=dev)
            args = dict(common, z=z_vals, n=n, n_prev=n_prev, sdf_prev=sdf_prev, merged_idx=merged, pair_tmp=tmp, pair_off=pl.pair_off,
                        slot_point=pl.slot_point, sdf_cur=sdf_cur, sel=self._lin[skey] if Ne > 0 else None, it=it)
            if it < max_iters - 1:
                zm = torch.empty((R, n + n0), dtype=torch.float32, device=dev)
                mi = torch.empty((R, n + n0), dtype=torch.int32, device=dev)
                pn = torch.empty((R, n0, 3), dtype=torch.float32, device=dev)
                args.update(z_merged=zm, merged_out=mi, points_new=pn)
                ops.sampler_eval(0, grid_h, R, **args)
                ops.sampler_eval(1, grid_h, R, **args)
                sdf_prev, merged, n_prev, z_vals, points = sdf_cur, mi, n, zm, pn
            else:
                ops.sampler_eval(2, grid_h, R, **args)
        self._flags = (flags, max_iters)
        self.last_points = pts_out
        self.last_slots = (slot_sample, ray_valid)
        if torch.cuda.is_current_stream_capturing():         # hipGraph capture (eval_graph.py): the CPU-generator draw below is the replayer's job
            return z_out, None
        idx = torch.randint(z_out.shape[-1], (z_out.shape[0],)).to(dev)       # the eikonal sample (:562-563) consumes the generator as the reference
        return z_out, torch.gather(z_out, 1, idx.unsqueeze(-1))

    # ------------------------------------------------------------------ evaluation: the same loop as separate launches
    def _z_vals_device_loop(self, ray_dirs, cam_loc, model, max_iters, beta0):
        """get_z_vals for evaluation (:377-574, no random draws) WITHOUT the host-side convergence test of :468.  The iteration count is
        data-dependent but every iteration's shapes are static (128 (it + 1) samples per ray), so all `max_iters` iterations are enqueued and
        a device flag per iteration decides which of their launches do work (include/spurfies_hip.h: spf_sampler_iter) — exactly the passes the
        reference's loop would have run, bit for bit the same kernels on the same inputs; an iteration that is not reached costs its (empty)
        launches, a reached one no pipeline drain."""
        from .. import ops

        dev, R = ray_dirs.device, ray_dirs.shape[0]
        n0, Nf, Ne = self.N_samples_eval, self.N_samples, self.N_samples_extra
        key = ("flags", max_iters, str(dev))
        if key not in self._lin:
            t = torch.zeros((max_iters + 2,), dtype=torch.int32)
            t[0] = 1
            self._lin[key] = t.to(dev)
        flags = self._lin[key].clone()                                   # flags[i] != 0 <=> the loop reaches iteration i
        z_vals, points = ops.sampler_uniform(self._linspace(n0, dev), None, cam_loc, ray_dirs, self.near, self.far)
        z_out = torch.empty((R, Nf + 2 + Ne), dtype=torch.float32, device=dev)
        pts_out = torch.empty((R, Nf + 2 + Ne, 3), dtype=torch.float32, device=dev)
        beta = torch.zeros((R,), dtype=torch.float32, device=dev)      # one buffer, updated in place by the passes that run
        sdf, samples_idx = None, None
        u_more, u_fin = self._linspace(n0, dev), self._linspace(Nf, dev)
        for it in range(max_iters):
            with torch.no_grad():
                s_sdf = model.sdf_importance_gated(points.view(-1, 3), flags[it: it + 1]).view(R, -1)
            sdf = s_sdf if samples_idx is None else torch.gather(torch.cat([sdf, s_sdf], -1), 1, samples_idx)
            n = z_vals.shape[1]
            skey = ("sel", n, Ne, str(dev))
            if Ne > 0 and skey not in self._lin:
                self._lin[skey] = torch.linspace(0, n - 1, Ne).long().to(torch.int32).to(dev)
            sel = self._lin[skey] if Ne > 0 else None
            beta_in = None if it == 0 else beta
            if it < max_iters - 1:
                # the convergence test (:468) = a beta-only pass that sets flags[it + 1]; then BOTH continuations are enqueued: the merging
                # pass runs iff the flag is set, the final pass + finish iff it is clear
                ops.sampler_iter(z_vals, sdf, beta_in, beta0, self.eps, self._bound_coef, self.beta_iters, False, 0.0, None, 0, flags=flags, it=it, beta_out=beta)
                samples, _, zm, mi = ops.sampler_iter(z_vals, sdf, beta, beta0, self.eps, self._bound_coef, 0, True, self.add_tiny, u_more, n0,
                                                      flags=flags, it=it, beta_out=beta)
                fin, _, _, _ = ops.sampler_iter(z_vals, sdf, beta, beta0, self.eps, self._bound_coef, 0, False, self.add_tiny, u_fin, Nf,
                                                flags=flags, it=it, beta_out=beta)
                ops.sampler_finish(fin, z_vals, sel, self.near, self.far, cam_loc, ray_dirs, flags=flags, it=it, out=(z_out, pts_out))
                z_vals, samples_idx = zm, mi.long()
                points = cam_loc.unsqueeze(1) + samples.unsqueeze(2) * ray_dirs.unsqueeze(1)
            else:                                                       # the last iteration samples with a full bisection (:434-445), no test
                fin, _, _, _ = ops.sampler_iter(z_vals, sdf, beta_in, beta0, self.eps, self._bound_coef, self.beta_iters, False, self.add_tiny, u_fin, Nf,
                                                flags=flags, it=it, beta_out=beta)
                ops.sampler_finish(fin, z_vals, sel, self.near, self.far, cam_loc, ray_dirs, flags=flags, it=it, out=(z_out, pts_out))
        self._flags = (flags, max_iters)
        self.last_points = pts_out
        if torch.cuda.is_current_stream_capturing():         # hipGraph capture (eval_graph.py): the CPU-generator draw below is the replayer's job
            return z_out, None
        # the eikonal sample (:562-563, unused by the caller) consumes the CPU generator as in the reference
        idx = torch.randint(z_out.shape[-1], (z_out.shape[0],)).to(dev)
        return z_out, torch.gather(z_out, 1, idx.unsqueeze(-1))
