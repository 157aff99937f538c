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

    def get_z_vals(self, ray_dirs, cam_loc, model, fast=-1, iter_step=None):
        dev = ray_dirs.device
        max_total_iters = fast if fast >= 0 else self.max_total_iters
        beta0 = model.density.get_beta().detach()
        z_vals = self.uniform_sampler.get_z_vals(ray_dirs, cam_loc, model, iter_step=iter_step)
        samples, samples_idx = z_vals, None
        R = z_vals.shape[0]
        dists = z_vals[:, 1:] - z_vals[:, :-1]
        bound = (1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0)))).to(dev) * (dists ** 2.0).sum(-1)
        beta = torch.sqrt(bound)
        total_iters, not_converge = 0, True
        sdf = None
        zero_col = torch.zeros(R, 1, device=dev)
        while not_converge and total_iters < max_total_iters:
            points = cam_loc.unsqueeze(1) + samples.unsqueeze(2) * ray_dirs.unsqueeze(1)
            with torch.no_grad():
                samples_sdf = model.sdf_importance(points.reshape(-1, 3))
            if samples_idx is not None:
                merged = torch.cat([sdf.reshape(-1, z_vals.shape[1] - samples.shape[1]),
                                    samples_sdf.reshape(-1, samples.shape[1])], -1)
                sdf = torch.gather(merged, 1, samples_idx).reshape(-1, 1)
            else:
                sdf = samples_sdf
            d = sdf.reshape(z_vals.shape)
            dists = z_vals[:, 1:] - z_vals[:, :-1]
            a, b, c = dists, d[:, :-1].abs(), d[:, 1:].abs()
            first = a.pow(2) + b.pow(2) <= c.pow(2)
            second = a.pow(2) + c.pow(2) <= b.pow(2)
            s = (a + b + c) / 2.0
            area = s * (s - a) * (s - b) * (s - c)
            tri = ~first & ~second & (b + c - a > 0)
            d_star = torch.where(tri, (2.0 * torch.sqrt(area)) / a, torch.zeros_like(a))
            d_star = torch.where(first, b, d_star)
            d_star = torch.where(second, c, d_star)   # the reference assigns `second` last (ray_sampler.py:424-425)
            d_star = (d[:, 1:].sign() * d[:, :-1].sign() == 1) * d_star

            curr_error = self.get_error_bound(beta0, model, sdf, z_vals, dists, d_star)
            beta = torch.where(curr_error <= self.eps, beta0, beta)
            beta_min, beta_max = beta0.unsqueeze(0).repeat(R), beta
            for _ in range(self.beta_iters):
                beta_mid = (beta_min + beta_max) / 2.0
                curr_error = self.get_error_bound(beta_mid.unsqueeze(-1), model, sdf, z_vals, dists, d_star)
                ok = curr_error <= self.eps
                beta_max = torch.where(ok, beta_mid, beta_max)
                beta_min = torch.where(ok, beta_min, beta_mid)
            beta = beta_max

            density = model.density(d, beta=beta.unsqueeze(-1))
            dists_inf = torch.cat([dists, torch.full((R, 1), 1e10, device=dev)], -1)
            free_energy = dists_inf * density
            shifted = torch.cat([zero_col, free_energy[:, :-1]], dim=-1)
            alpha = 1 - torch.exp(-free_energy)
            transmittance = torch.exp(-torch.cumsum(shifted, dim=-1))
            weights = alpha * transmittance

            total_iters += 1
            # evaluate the convergence test (a host sync) only if another iteration is allowed
            more = total_iters < max_total_iters and bool(beta.max() > beta0)
            not_converge = more
            if more:
                N = self.N_samples_eval
                err_sec = torch.exp(-d_star / beta.unsqueeze(-1)) * (dists_inf[:, :-1] ** 2.0) / (4 * beta.unsqueeze(-1) ** 2)
                err_int = torch.cumsum(err_sec, dim=-1)
                pdf = (torch.clamp(torch.exp(err_int), max=1.0e6) - 1.0) * transmittance[:, :-1] + self.add_tiny
            else:
                N = self.N_samples
                pdf = weights[..., :-1] + 1e-5
            pdf = pdf / torch.sum(pdf, -1, keepdim=True)
            cdf = torch.cumsum(pdf, -1)
            cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
            if more or not model.training:
                u = torch.linspace(0.0, 1.0, steps=N).to(dev).unsqueeze(0).repeat(R, 1)
            else:
                u = torch.rand(list(cdf.shape[:-1]) + [N]).to(dev)
            u = u.contiguous()
            inds = torch.searchsorted(cdf, u, right=True)
            below = torch.clamp(inds - 1, min=0)
            above = torch.clamp(inds, max=cdf.shape[-1] - 1)
            cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
            bin_b, bin_a = torch.gather(z_vals, 1, below), torch.gather(z_vals, 1, above)
            denom = cdf_a - cdf_b
            denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
            samples = bin_b + (u - cdf_b) / denom * (bin_a - bin_b)
            if more:
                z_vals, samples_idx = torch.sort(torch.cat([z_vals, samples], -1), -1)
        self.last_iters = total_iters

        z_samples = samples
        near = self.near * torch.ones(R, 1, device=dev)
        far = self.far * torch.ones(R, 1, device=dev)
        if self.N_samples_extra > 0:
            if model.training:
                sampling_idx = torch.randperm(z_vals.shape[1])[: self.N_samples_extra]
            else:
                sampling_idx = torch.linspace(0, z_vals.shape[1] - 1, self.N_samples_extra).long()
            z_extra = torch.cat([near, far, z_vals[:, sampling_idx.to(dev)]], -1)
        else:
            z_extra = torch.cat([near, far], -1)
        z_vals, _ = torch.sort(torch.cat([z_samples, z_extra], -1), -1)
        idx = torch.randint(z_vals.shape[-1], (z_vals.shape[0],)).to(dev)
        z_samples_eik = torch.gather(z_vals, 1, idx.unsqueeze(-1))
        return z_vals, z_samples_eik

    def get_error_bound(self, beta, model, sdf, z_vals, dists, d_star):
        density = model.density(sdf.reshape(z_vals.shape), beta=beta)
        shifted = torch.cat([torch.zeros(dists.shape[0], 1, device=dists.device), dists * density[:, :-1]], dim=-1)
        integral = torch.cumsum(shifted, dim=-1)
        err = torch.exp(-d_star / beta) * (dists ** 2.0) / (4 * beta ** 2)
        err_int = torch.cumsum(err, dim=-1)
        bound = (torch.clamp(torch.exp(err_int), max=1.0e6) - 1.0) * torch.exp(-integral[:, :-1])
        return bound.max(-1)[0]
