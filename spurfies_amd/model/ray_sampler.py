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
        self.last_iters = 0
        self.last_points = None
        self.shard = None   # (rank, world) for ray-sharded batches: CPU-generator draws are made for the whole batch, this rank's rows kept
        self.draws = None   # sync-free / graph mode: {'t_rand' [R,128], 'u' [R,N_samples], 'sel' int32 [N_extra]} device tensors the
        #                     caller fills from the CPU generator (same calls, same order) before every step
        # Lemma-2 constant exactly as the reference forms it in float32 (ray_sampler.py:389)
        self._bound_coef = float(1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0))))
        self._lin = {}

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
        beta0 = (model.density.get_beta_value() if hasattr(model.density, "get_beta_value") else model.density.get_beta().detach()).reshape(1).contiguous()
        n0 = self.N_samples_eval
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
