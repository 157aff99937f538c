"""PointVolSDF with the reference's public interface (spurfies/model/pointneus_disent.py:24-908):
constructor `(conf, scan_id, dataset)`, `forward(input, fast=-1) -> dict`, `get_sdf_eval`,
`sdf_importance`, `pseudo_sdf`, the same parameter / buffer names (state_dict keys), `.density`,
`.ray_sampler`.

What differs is HOW it runs:
  * the voxel grid is built once per cloud (the reference rebuilds it 3-4x per step),
  * kNN, gather, RBF weights, F_geometry + T and the input Jacobian are HIP kernels working on
    worst-case-sized dense [R, SR] buffers with device-side point lists (no masked_select syncs),
  * d sdf/d x comes out of the fused kernel's Jacobian sweep instead of an autograd
    double-backward graph (its gradient w.r.t. every trainable tensor is exactly zero —
    tests/test_oracle_golden.py::test_eikonal_term_has_zero_gradient_for_trainables),
  * invalid rays are carried as all-masked rows instead of being compacted away.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import feat_utils, ops
from ..torch_knnquery import VoxelGrid
from ..utils import rend_util
from .density import LaplaceDensity
from .embedder import get_embedder
from .ray_sampler import ErrorBoundSampler_pn
from .utils import TVGraph, load_neural_points

SDF_FILL = ops.SDF_FILL


class DistPoints(torch.autograd.Function):
    """o + t d for the rendered depth t (pointneus_disent.py:765-767); gradient to t only (rays carry none)."""

    @staticmethod
    def forward(ctx, cam_loc, ray_dirs, dist):
        ctx.save_for_backward(ray_dirs)
        return torch.addcmul(cam_loc, ray_dirs, dist.unsqueeze(-1))

    @staticmethod
    def backward(ctx, g):
        (ray_dirs,) = ctx.saved_tensors
        return None, None, (g * ray_dirs).sum(-1)


class PointVolSDF(nn.Module):
    def __init__(self, conf, scan_id, dataset, neural_points=None, device="cuda"):
        """`neural_points` ({'pts': [N,3], 'colors': [N,3] 0..255}) bypasses the .ply on disk
        (the reference always reads ./data/<dataset>/scan<id>/<id>.ply, :131-144)."""
        super().__init__()
        self.conf = conf
        self.scan_id = scan_id
        self.dataset = dataset
        self.feature_vector_size = conf.get_int("feature_vector_size")
        self.scene_bounding_sphere = conf.get_float("scene_bounding_sphere", default=1.0)
        self.white_bkgd = conf.get_bool("white_bkgd", default=False)
        self.conf.rbf = 45  # hard-set by the reference (:42)
        wide = self.scan_id in ["garden", "stump"] and self.dataset == "mipnerf"
        self.grid_ranges = (-2, -2, -2, 2, 2, 2) if wide else (-1, -1, -1, 1, 1, 1)
        if conf.get_list("grid_ranges", default=None) is not None:
            self.grid_ranges = tuple(conf.get_list("grid_ranges"))
        self._voxel_grid_neural = VoxelGrid((0.025, 0.025, 0.025), (3, 3, 3), (3, 3, 3), 26, 20000, self.grid_ranges)
        self._init_neural_info(neural_points, device)

        self.position_encoding, pos_in_dim = get_embedder(multires=6, input_dims=3)
        self.view_encoding, dir_in_dim = get_embedder(multires=3, input_dims=3)
        fvs = self.feature_vector_size

        def lrelu():
            return nn.LeakyReLU(inplace=True)

        self.F_color = nn.Sequential(nn.Linear(fvs + pos_in_dim, 256), lrelu(), nn.Linear(256, 256), lrelu(),
                                     nn.Linear(256, 256), lrelu(), nn.Linear(256, 256))
        self.F_geometry = nn.Sequential(nn.Linear(fvs // 2 + 3, 256), lrelu(), nn.Linear(256, 256), lrelu(),
                                        nn.Linear(256, 256), lrelu(), nn.Linear(256, 256), lrelu(), nn.Linear(256, 256))
        self.T = nn.Sequential(nn.Linear(256, 1))
        self.R = nn.Sequential(nn.Linear(256 + dir_in_dim, 256), lrelu(), nn.Linear(256, 256), lrelu(),
                               nn.Linear(256, 3), nn.Sigmoid())
        self.density = LaplaceDensity(**conf.get_config("density"))
        self.ray_sampler = ErrorBoundSampler_pn(self.scene_bounding_sphere, **conf.get_config("ray_sampler"))
        self.to(self.neural_pts.device)
        self._packed_geo = None
        self._packed_key = None
        self._tv_graph = None
        self._tv_key = None
        self.stats = {}
        self.keep_stages = False
        self.stages = None
        self.sync_free = False   # training only: static shapes, no host synchronisation (spurfies_amd/train.py sets it)
        # evaluation only: None = every output of the reference's forward (:851-892); a tuple of keys out of ("rgb_values", "depth_values",
        # "normal_map", "weights") = the render outputs the reference's evaluation loops read (train.py:419-424, eval_spurfies.py:282-287) and
        # nothing else — no pseudo-point pass, TV term or per-slot plot maps, composites + normals + depth fill in ONE launch (eval_graph.py)
        self.eval_keys = None
        # evaluation only, OPT-IN (default False): the sampler's SDF-only passes with reduced products (SPF_ARITH_LITE: ~16 mantissa bits) — their
        # values only steer where samples go; the main pass evaluates every rendered point with the full products.  Tolerance study in DESIGN.md.
        self.sampler_lite = False
        self._cp_sync = ops.CompactSync()     # this model's own word buffers of the one-launch compaction, one per kNN pass

    # ------------------------------------------------------------------ initialisation (:116-205)
    @staticmethod
    def _init_neural_feats(neural_feats):
        neural_feats.uniform_(-1e-4, 1e-4)

    @staticmethod
    def init_latent_codes(neural_feats):
        torch.nn.init.normal_(neural_feats, mean=0.0, std=0.01)
        with torch.no_grad():
            norms = neural_feats.norm(dim=-1, keepdim=True)
            neural_feats *= torch.clamp(norms, max=1) / (norms + 1e-7)

    def _init_neural_info(self, neural_points, device):
        if neural_points is None:
            if self.dataset == "dtu":
                path = f"./data/{self.dataset}/scan{self.scan_id}/{self.scan_id}.ply"
            elif self.dataset in ["mipnerf", "own_data"]:
                path = f"./data/{self.dataset}/{self.scan_id}/{self.scan_id}.ply"
            else:
                raise NotImplementedError
            self.conf.pointcloud_path = path
            neural_points = load_neural_points(path, vox_res=self.conf.vox_res, device=device)
        pts = torch.as_tensor(neural_points["pts"]).float().to(device)
        n = len(pts)
        self.register_buffer("neural_pts", pts.clone().contiguous())
        self.neural_feats_color = nn.Parameter(torch.empty((n, self.feature_vector_size), dtype=torch.float32, device=device))
        self.neural_feats_geometry = nn.Parameter(torch.empty((n, self.feature_vector_size // 2), dtype=torch.float32, device=device))
        self._init_neural_feats(self.neural_feats_color.data)
        self.init_latent_codes(self.neural_feats_geometry.data)
        if self.conf.get_bool("initialize_colors", default=False) and "colors" in neural_points:
            col = torch.as_tensor(neural_points["colors"]).float().to(device)
            assert col.shape == (n, 3)
            self.neural_feats_color.data[:, :3] = col * 2.0 / 255.0 - 1.0

    # ------------------------------------------------------------------ cached device state
    def _grid(self):
        self._voxel_grid_neural.set_pointset(self.neural_pts.unsqueeze(0))  # no-op while the buffer is unchanged
        return self._voxel_grid_neural

    def _packed(self):
        """Packed F_geometry/T image; re-packed only when those (frozen, train.py:151-154) weights change."""
        ps = [p for m in (self.F_geometry, self.T) for p in m.parameters()]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if key != self._packed_key:
            if any(p.requires_grad for p in ps) and torch.is_grad_enabled() and self.training:
                raise RuntimeError(
                    "F_geometry / T must be frozen (requires_grad_(False)) as spurfies/train.py:151-154 does: the fused "
                    "geometry kernel does not produce their weight gradients")
            sd = {f"F_geometry.{i}.{n}": getattr(self.F_geometry[i], n) for i in (0, 2, 4, 6, 8) for n in ("weight", "bias")}
            sd.update({"T.0.weight": self.T[0].weight, "T.0.bias": self.T[0].bias})
            self._packed_geo = ops.pack_geometry_weights(sd)
            self._packed_key = key
        return self._packed_geo

    def cache_key(self):
        """Identity of everything the cached device state (cell table, TV graph, packed prior image) was derived from; a
        captured hipGraph holds pointers into that state and must be dropped when the key changes (train.py:TrainStep)."""
        self._packed()
        # ... and the storage of every parameter the forward reads: a captured graph bakes those addresses in, and FlatAdam._flatten (after
        # optimizer.load_state_dict) or module.to() re-point p.data — a replay would then render from the old, possibly freed, storage
        return (self.neural_pts.data_ptr(), self.neural_pts._version, self.neural_pts.shape[0], self._packed_key,
                tuple(p.data_ptr() for p in self.parameters()))

    def tv_graph(self):
        """Static neighbour graph of tv_regul (utils.py:221-282), rebuilt only when the cloud (or k / r) changes."""
        key = (self.neural_pts.data_ptr(), self.neural_pts._version, self.neural_pts.shape[0], self.conf.k, self.conf.r)
        if self._tv_graph is None or self._tv_key != key:
            self._tv_graph = TVGraph(self._grid(), self.neural_pts, self.conf.k, self.conf.r)
            self._tv_key = key
        return self._tv_graph

    def freeze_prior(self):
        """train.py:151-154: the local geometry prior (F_geometry, T) is not optimised."""
        for m in (self.F_geometry, self.T):
            for p in m.parameters():
                p.requires_grad_(False)
        return self

    def _zero_scalar(self, dev):
        """A cached 0-dim zero (never written to): the `local_loss` value of steps without local_data, without a fill launch per step."""
        z = getattr(self, "_zero", None)
        if z is None or z.device != torch.device(dev):
            self._zero = z = torch.zeros((), device=dev)
        return z

    # ------------------------------------------------------------------ geometry at free points
    def _sdf_points(self, x, with_grad, gate=None, role="points"):
        """x [M,3] -> dict(sdf [M] (1000 where no neighbour), grad, valid u8 [M], pairs); no host sync.
        gate: device int32 [1] — 0 makes the pass a no-op on the MLP side (every row gets the 1000 filler): a sampler iteration the
        device-controlled loop did not reach (ray_sampler.py).  role: which of the model's compaction word buffers the pass uses (passes
        that may overlap in time — forked graph branches — must not share one)."""
        grid = self._grid()
        x = x.contiguous()
        q = grid.query_dense(x.detach().unsqueeze(1), self.conf.k, self.conf.r, 1)
        M = x.shape[0]          # the compaction pass also lays down the 1000 filler / zero gradient of the rows without a neighbour
        sdf_buf = torch.empty((M,), dtype=torch.float32, device=x.device)
        grad_buf = torch.empty((M, 3), dtype=torch.float32, device=x.device) if with_grad else None
        pl = ops.PairList.from_slots(q["slot_valid"], q["pidx"].view(-1, self.conf.k), fill_sdf=sdf_buf, fill_grad=grad_buf, gate=gate,
                                     sync=self._cp_sync.get(role, x.device, M))
        if with_grad:
            sdf, grad, _ = ops.GeoSDF.apply(x, self.neural_feats_geometry, pl, self.neural_pts, self._packed(), float(self.conf.rbf), sdf_buf, grad_buf)
        else:
            res = ops.geo_forward(x.detach(), pl, self.neural_pts, self.neural_feats_geometry.detach(), self._packed(),
                                  float(self.conf.rbf), with_grad=False, sdf_out=sdf_buf)
            sdf, grad = res["sdf"], None
        return {"sdf": sdf, "grad": grad, "valid": q["slot_valid"].view(-1), "pairs": pl}

    def sdf_importance(self, inputs):
        """:348-421 — SDF at sampler points, 1000 where a point has no neighbour (callers wrap in no_grad)."""
        return self._sdf_points(inputs, with_grad=False, role="sampler")["sdf"]

    def sdf_importance_gated(self, inputs, gate):
        """sdf_importance behind a device-side gate (int32 [1]; 0 = skip the MLP work): the sampler's sync-free evaluation loop."""
        return self._sdf_points(inputs, with_grad=False, gate=gate, role="sampler")["sdf"]

    def sdf_pairs_gated(self, inputs, gate):
        """The SDF pass of one evaluation-sampler iteration WITHOUT its per-point reduction: neighbour search, compaction and the geometry
        kernel's per-pair scratch (ops.geo_forward(reduce=False)) -> (pair_tmp, PairList); the consumer (spf_sampler_eval) reduces."""
        grid = self._grid()
        x = inputs.detach().contiguous()
        q = grid.query_dense(x.unsqueeze(1), self.conf.k, self.conf.r, 1)
        pl = ops.PairList.from_slots(q["slot_valid"], q["pidx"].view(-1, self.conf.k), gate=gate, sync=self._cp_sync.get("sampler", x.device, x.shape[0]))
        tmp = ops.geo_forward(x, pl, self.neural_pts, self.neural_feats_geometry.detach(), self._packed(), float(self.conf.rbf), with_grad=False,
                              reduce=False, lite=self.sampler_lite)["pair_tmp"]
        return tmp, pl

    def get_sdf_eval(self, inputs):
        """:249-298 — mesh-extraction entry."""
        return self._sdf_points(inputs, with_grad=False)["sdf"]

    def sdf_eval_grid(self, xs, ys, zs, chunk=1 << 22, span=1 << 26):
        """get_sdf_eval over the regular grid np.meshgrid(xs, ys, zs) (plots.py:249-253 evaluates it in 100 000-point chunks of a host
        array; chunking is not observable) -> device float32 [len(ys), len(xs), len(zs)] (the 'xy' layout), 1000 where a point has no
        neighbour.  The points are generated on the device from the axes (cast to float32 exactly as `torch.tensor(grid, dtype=torch.float)`
        casts them, plots.py:327-330), ~88 % of them fail the dilated-occupancy test every point query starts with and are filled at once
        (spf_grid_sweep_hits); the rest — compacted — go through the neighbour search and the SDF kernels `chunk` points at a time and
        are scattered back.  One host read-back (the hit count) per `span` grid points."""
        dev = self.neural_pts.device
        grid = self._grid()
        ax = [torch.as_tensor(np.asarray(a)).to(torch.float32).to(dev).contiguous() for a in (xs, ys, zs)]
        nx, ny, nz = (int(a.numel()) for a in ax)
        M = nx * ny * nz
        out = torch.empty((M,), dtype=torch.float32, device=dev)
        cap = min(M, int(span))
        pts = torch.empty((cap, 3), dtype=torch.float32, device=dev)
        idx = torch.empty((cap,), dtype=torch.int64, device=dev)
        counter = torch.zeros((1,), dtype=torch.int64, device=dev)
        with torch.no_grad():
            for first in range(0, M, cap):
                cnt = min(cap, M - first)
                counter.zero_()
                grid.sweep_hits(ax[0], ax[1], ax[2], out[first: first + cnt], first, cnt, pts, idx, counter, fill_value=SDF_FILL)
                n = int(counter.item())
                for a in range(0, n, int(chunk)):
                    b = min(n, a + int(chunk))
                    out.index_copy_(0, idx[a:b], self._sdf_points(pts[a:b], with_grad=False)["sdf"])
        return out.view(ny, nx, nz)

    def pseudo_sdf(self, inputs):
        """:423-495 — differentiable SDF of the valid rows only ([Nv,1]); a constant 1000-vector over
        all inputs when none is valid.  (Data-dependent shape: synchronises, like the reference.)"""
        r = self._sdf_points(inputs, with_grad=torch.is_grad_enabled())
        valid = r["valid"].bool()
        if not bool(valid.any()):
            return torch.ones((inputs.shape[0]), device=inputs.device) * SDF_FILL
        return r["sdf"][valid].unsqueeze(-1)

    # ------------------------------------------------------------------ SDF zero crossings (:586-612)
    @staticmethod
    def find_surface_points(sdf, d_all, device=None):
        """Per ray, the first interval where the SDF goes from + to - between ADJACENT slots (slots without a point hold 1000
        and never form a crossing), and the linearly interpolated depth of the zero.  sdf, d_all [..., SR] ->
        (d_surface [...] (0 where there is none), network_mask bool [...]).  Differentiable w.r.t. sdf.  Unlike the
        reference (which overwrites the 1000s of its input with NaN in place, :587) the input is left untouched."""
        ok = sdf != SDF_FILL
        s0, s1 = sdf[..., :-1], sdf[..., 1:]
        cross = ok[..., :-1] & ok[..., 1:] & (s1 * s0 < 0) & (s1 < s0)
        hit = cross.any(-1)
        first = (torch.cumsum(cross.to(torch.int32), -1) == 0).sum(-1, keepdim=True).clamp(max=cross.shape[-1] - 1)
        # rays without a crossing get a harmless (1, -1) pair: no NaN / Inf enters the backward of the division
        a = torch.where(hit, torch.gather(sdf, -1, first).squeeze(-1), torch.ones_like(sdf[..., 0]))
        b = torch.where(hit, torch.gather(sdf, -1, first + 1).squeeze(-1), -torch.ones_like(sdf[..., 0]))
        d0, d1 = torch.gather(d_all, -1, first).squeeze(-1), torch.gather(d_all, -1, first + 1).squeeze(-1)
        d_surface = torch.where(hit, (a * d1 - b * d0) / (a - b), torch.zeros_like(a))
        return d_surface, hit

    # ------------------------------------------------------------------ rays
    def get_importance_rays(self, cam_loc, ray_dirs, model, fast=-1, iter_step=None):
        ray_dirs = ray_dirs.reshape(-1, 3)
        cam_loc = cam_loc.unsqueeze(1).repeat(1, ray_dirs.shape[0], 1).reshape(-1, 3)
        z_vals, _ = self.ray_sampler.get_z_vals(ray_dirs, cam_loc, model, fast, iter_step)
        points = self.ray_sampler.last_points                     # o + z d, written by the sampler's finish kernel
        return points, z_vals, cam_loc, ray_dirs

    def volume_rendering(self, deltas, density):
        """:894-908."""
        free_energy = deltas * density
        shifted = torch.cat([torch.zeros(deltas.shape[0], 1, device=deltas.device), free_energy[:, :-1]], dim=-1)
        alpha = 1 - torch.exp(-free_energy)
        transmittance = torch.exp(-torch.cumsum(shifted, dim=-1))
        return alpha * transmittance

    def _colors(self, n_valid, x, wn, pl, n_pairs, ray_dirs, SR, pre=(None, None)):
        """n_valid / n_pairs = None selects the sync-free (worst-case buffers, device-side counts) mode."""
        return self._colors_impl(n_valid, x, wn, pl, n_pairs, ray_dirs, SR, pre)

    def _colors_impl(self, n_valid, x, wn, pl, n_pairs, ray_dirs, SR, pre=(None, None)):
        """:325-346 — colours of the P valid points, written at their slot rows of a dense [R*SR,3] array (0 elsewhere).
        F_color's activated layers + RBF-weighted mean (per pair), then F_color's linear last layer + the R head (per
        point) are fused HIP kernels (spf_color_*, spf_rhead_*).  pre: weight images packed ahead of time (ops.prepack_*)."""
        fc, rh = self.F_color, self.R
        agg3 = ops.ColorAgg.apply(self.neural_feats_color, fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias, fc[4].weight,
                                  fc[4].bias, x, wn, pl, self.neural_pts, n_valid, n_pairs, pre[0])
        return ops.RHead.apply(agg3, fc[6].weight, fc[6].bias, rh[0].weight, rh[0].bias, rh[2].weight, rh[2].bias, rh[4].weight,
                               rh[4].bias, ray_dirs.detach().contiguous(), pl.point_slot, pl.n_points, SR, x.shape[0], n_valid is None, pre[1])

    # ------------------------------------------------------------------ forward (:614-892)
    def forward(self, input, fast=-1):
        intrinsics, uv, pose = input["intrinsics"], input["uv"], input["pose"]
        iter_step = input.get("iter_step", 1)
        dev = self.neural_pts.device
        conf = self.conf
        SR, k = conf.max_shading_pts, conf.k
        grid = self._grid()

        # one launch for the ray set-up (None for multi-view batches); in sync-free training it also forms this forward's effective
        # Laplace scale |beta| + beta_min, which the sampler and the compositing kernels read (density.get_beta_value)
        fuse_beta = ((self.sync_free and self.training) or (not self.training and self.eval_keys is not None)) and self.density.beta.is_cuda
        beta_fwd = torch.empty((), dtype=torch.float32, device=dev) if fuse_beta else None
        smp = self.ray_sampler
        if (fuse_beta and fast == 1 and ops._FUSED_SAMPLER[0] and smp.draws is not None and smp.N_samples_extra > 0
                and smp.N_samples_eval <= 128 and smp.N_samples + 2 + smp.N_samples_extra <= 128):
            out = self._forward_train_fused(input, beta_fwd)
            if out is not None:
                return out
        rays = ops.camera_rays(uv, pose, intrinsics, self.density.beta, self.density.beta_min_value, beta_fwd)
        self.density._beta_forward = beta_fwd if rays is not None else None
        slots = None
        try:
            if rays is not None:
                ray_dirs, cam_loc, depth_scale = rays
                self.ray_sampler.get_z_vals(ray_dirs, cam_loc, self, fast, iter_step)
                points = self.ray_sampler.last_points
                slots = self.ray_sampler.last_slots          # the evaluation loop's last launch assigned the main pass's slots on its way
            else:
                ray_dirs, cam_loc = rend_util.get_camera_params(uv, pose, intrinsics)
                dirs_cam, _ = rend_util.get_camera_params(uv, torch.eye(4, device=dev)[None], intrinsics)
                depth_scale = dirs_cam[0, :, 2:]
                points, _, cam_loc, ray_dirs = self.get_importance_rays(cam_loc, ray_dirs, self, fast, iter_step)
            return self.render_points(points, ray_dirs, cam_loc, depth_scale, input.get("local_data"), slots=slots)
        finally:
            self.density._beta_forward = None

    def _forward_train_fused(self, input, beta_fwd):
        """The optimisation step's forward (sync-free, fast = 1) with the sampler chain as fused launches (round 5): camera rays + uniform
        samples in one launch; the SDF kernel without its per-point reduction; then ONE launch that reduces, finds beta, draws the samples,
        sorts them, forms the main-pass points and assigns the main pass's kNN slots (spf_sampler_train) — 12 launches -> 7 between the start of
        the step and the main pass's geometry kernel, same values (tests/test_gpu_sampler.py).  None: shapes the fused form does not cover."""
        dev = self.neural_pts.device
        conf, smp = self.conf, self.ray_sampler
        ext = smp.draws
        n0 = smp.N_samples_eval
        # ... and the per-point TV terms of the geometry latents in the same launch (their backward rides in the loss backward launch)
        tv_in = ops.scatter_mode() != "fixed" and ops._sink(self.neural_feats_geometry) is not None
        res = ops.camera_uniform(input["uv"], input["pose"], input["intrinsics"], self.density.beta, self.density.beta_min_value, beta_fwd,
                                 smp._linspace(n0, dev), ext["t_rand"], smp.near, smp.far, tv_graph=self.tv_graph() if tv_in else None,
                                 tv_feat=self.neural_feats_geometry if tv_in else None,
                                 pack=(self.F_color, self.R, input["uv"].shape[1] * conf.max_shading_pts))
        if res is None:
            return None
        ray_dirs, cam_loc, depth_scale, z_vals, pts0, tv_pre, pre = res
        R = ray_dirs.shape[0]
        grid = self._grid()
        self.density._beta_forward = beta_fwd
        try:
            with torch.no_grad():
                x0 = pts0.view(-1, 3)
                q0 = grid.query_dense(x0.unsqueeze(1), conf.k, conf.r, 1)
                pl0 = ops.PairList.from_slots(q0["slot_valid"], q0["pidx"].view(-1, conf.k), sync=self._cp_sync.get("sampler", dev, x0.shape[0]))
                tmp = ops.geo_forward(x0, pl0, self.neural_pts, self.neural_feats_geometry.detach(), self._packed(), float(conf.rbf), with_grad=False,
                                      reduce=False)["pair_tmp"]
                _, z_out, points, slot_sample, ray_valid = ops.sampler_train(z_vals, tmp, pl0, beta_fwd.reshape(1), smp.eps, smp._bound_coef, smp.beta_iters,
                                                                              ext["u"], ext["sel"], smp.near, smp.far, cam_loc, ray_dirs, grid._h,
                                                                              conf.max_shading_pts)
            smp.last_iters = 1
            smp.last_points = points
            return self.render_points(points, ray_dirs, cam_loc, depth_scale, input.get("local_data"), slots=(slot_sample, ray_valid), tv_pre=tv_pre, pre=pre)
        finally:
            self.density._beta_forward = None

    def render_points(self, points, ray_dirs, cam_loc, depth_scale, local_data=None, slots=None, tv_pre=None, pre=None):
        """Everything of forward() behind the sampler (:654-892): main-pass kNN of the sample positions `points` [R,D,3] (the
        sampler's o + z d), filter_points, SDF + normals + colours at the hits, compositing, the pseudo-point / local / TV terms
        and the output dict.  Split out so that a stage test can feed recorded sample positions."""
        dev = self.neural_pts.device
        conf = self.conf
        SR, k = conf.max_shading_pts, conf.k
        grid = self._grid()
        R = ray_dirs.shape[0]

        static = self.sync_free and self.training                 # fused-loss mode of the optimisation step
        # forked step (ops.branch; TrainStep(fork=True) — experimental, OFF by default: measured +-1 % on ROCm 7.2): passes that do not depend on each other are
        # issued on side streams = parallel branches of the step's hipGraph.  Branch "aux": the two weight-packing launches (they only read
        # the parameters) and the TV term, beside the main pass's kNN / geometry kernel.
        if not self.training:
            local_data = None                                     # the feature-consistency term is a training loss (:731)
        fork = static and ops.fork_enabled() and local_data is None and ops._sink(self.density.beta) is not None
        ev_pack, tv_early = None, None
        if fork and (pre is None or tv_pre is None):
            with ops.branch("aux", dev) as aux:
                if pre is None:
                    pre = (ops.prepack_color(self.F_color, R * SR, dev), ops.prepack_rhead(self.F_color[6], self.R, R * SR, dev))
                    ev_pack = aux.record()
                if tv_pre is None:
                    tv_early = self.tv_graph().loss(self.neural_feats_geometry, reduce=False)
        if pre is None:
            pre = (None, None)

        # ---- kNN of the main pass, dense [R,SR] ------------------------------------------------
        q = grid.query_dense(points.detach(), k, conf.r, SR, slots=slots)
        # evaluation needs no exact-size training buffers either: worst-case colour buffers + device-side counts, no host read-back
        dense = static or not self.training
        # the bool forms of the two masks are read by the reference-shaped outputs only (two conversion launches)
        light = (not self.training) and self.eval_keys is not None
        valid = None if (static or light) else q["slot_valid"].bool()        # [R,SR]  == reference `mask`
        ray_mask = None if (static or light) else q["ray_valid"].bool()      # [R]
        sdf_buf = torch.empty((R * SR,), dtype=torch.float32, device=dev)
        grad_buf = torch.empty((R * SR, 3), dtype=torch.float32, device=dev)
        fuse_filter = (static or not self.training) and ops._FUSED_SAMPLER[0]     # filter_points (:207-239) rides in the compaction launch
        pl = ops.PairList.from_slots(q["slot_valid"], q["pidx"].view(R * SR, k), fill_sdf=sdf_buf, fill_grad=grad_buf,
                                     sync=self._cp_sync.get("main", dev, R * SR), filt=(q["loc"], cam_loc, ray_dirs) if fuse_filter else None)
        point_slot = pl.point_slot

        # ---- filter_points (:207-239) on dense rows (HIP) ---------------------------------------
        z_slots, deltas, x = pl.filtered if pl.filtered is not None else ops.filter_points(q["loc"], q["slot_valid"], cam_loc.detach(), ray_dirs.detach())

        # ---- geometry: sdf, d sdf/d x, normalised RBF weights (HIP) -----------------------------
        sdf_flat, gradients, wn = ops.GeoSDF.apply(x, self.neural_feats_geometry, pl, self.neural_pts, self._packed(), float(conf.rbf), sdf_buf, grad_buf)
        sdf = sdf_flat.view(R, SR)                                # gradients: [R*SR,3], zero rows where not a point

        # ---- forked step, branch "pseudo": everything that needs the compositing WEIGHTS only — depth, dist_map, the rendered surface
        #      points and the whole pseudo-point pass behind them (:765-780: kNN, compaction, geometry kernel with the Jacobian sweep on R
        #      points) — beside the colour stage instead of behind it; the colour composite follows on its own (ops.RenderRGB)
        pr, ev_w = None, None
        ev_geo = ops.mark(dev) if fork else None                  # the pseudo branch starts HERE, but is issued behind this stream's own
        ops.wait(ev_pack)                                         # continuation (the colour stage): see ops.branch(after=...)

        # ---- colours (PyTorch ops on the P valid points; one host sync for P) -------------------
        if dense:        # no host round trip: worst-case buffers, counts stay on the device
            P, n_pairs, rows = 1, None, None
            self.stats = {"rays": R, "counts": pl.counts}
            colors = self._colors(None, x, wn, pl, None, ray_dirs, SR, pre)
        else:
            P, n_pairs = pl.host_counts()
            rows = point_slot[:P].long()
            self.stats = {"valid_points": P, "pairs": n_pairs, "rays": R}
            colors = self._colors(P, x, wn, pl, n_pairs, ray_dirs, SR) if P > 0 else torch.zeros((R * SR, 3), device=dev)
        if fork:
            with ops.branch("pseudo", dev, after=ev_geo) as br:
                weights, depth, dist_map, acc, pts_rendered = ops.RenderW.apply(sdf, self.density.get_beta_value(), q["slot_valid"], z_slots, deltas,
                                                                                self.density.beta, cam_loc, ray_dirs)
                ev_w = br.record()
                pr = self._sdf_points(pts_rendered, with_grad=True, role="pseudo")
        colors = colors.view(R, SR, 3)
        if light:        # the evaluation loops' outputs and nothing else: composites, normals and the depth fill in one launch
            weights, rgb, depth_values, normal_map = ops.render_eval(sdf, colors, self.density.get_beta_value(), q["slot_valid"], z_slots, deltas,
                                                                     gradients, q["ray_valid"], depth_fill=1.0)
            full = {"rgb_values": rgb, "depth_values": depth_values, "normal_map": normal_map, "weights": weights}
            return {k_: full[k_] for k_ in self.eval_keys}

        # ---- density + compositing (:714-723, 765-795, 894-908), one HIP kernel each way --------------
        local_terms = None
        if fork:
            ops.wait(ev_w)
            rgb = ops.RenderRGB.apply(weights, colors)
            ops.join(dev)                                         # the pseudo-point pass and the TV term: read by the loss kernels next
        elif static and ops._sink(self.density.beta) is not None:    # beta's gradient goes straight into its .grad buffer
            # ... and the rendered surface points o + d * dist_map of the pseudo-point loss (:765-767) come out of the same launch
            if local_data is not None:
                # multi-view feature consistency at the SDF zero crossings (:727-763; DTU recipe, local_weight 0.5): ONE launch (crossing search,
                # projection into the 1 + m views, bilinear taps, cosine term, tangents); its sum joins the loss kernels' partial sums and its
                # gradient joins g_sdf inside the compositing backward — no other launch, no host synchronisation (ops.LocalTerms)
                local_terms = ops.local_forward(feat_utils.local_desc(local_data, dev), sdf, z_slots, cam_loc, ray_dirs)
            weights, rgb, depth, dist_map, acc, pts_rendered = ops.Render.apply(sdf, colors, self.density.get_beta_value(), q["slot_valid"], z_slots,
                                                                                deltas, self.density.beta, cam_loc, ray_dirs, local_terms)
        else:
            pts_rendered = None
            weights, rgb, depth, dist_map, acc = ops.Render.apply(sdf, colors, self.density.get_beta(), q["slot_valid"], z_slots, deltas)
        output = {"rgb_values": rgb, "weights": weights, "local_loss": self._zero_scalar(dev)}
        if self.keep_stages:        # stage tests: dense-row intermediates next to the reference's per-stage tensors
            self.stages = {"pidx": q["pidx"], "loc": q["loc"], "slot_valid": q["slot_valid"], "ray_valid": q["ray_valid"], "x": x, "z_slots": z_slots,
                           "deltas": deltas, "sdf": sdf, "gradients": gradients, "colors": colors, "dist_map": dist_map}

        # ---- multi-view feature consistency at the SDF zero crossings (:727-763), DTU training only ---------
        if local_terms is not None:
            if self.keep_stages:
                self.stages.update({"d_surface": local_terms.d_surface, "network_mask": local_terms.lfirst >= 0})
        elif local_data is not None and self.training:
            # the same launch behind an autograd.Function of its own (ops.LocalLoss): the default training mode
            lsum, lcnt, d_surface, hit = ops.LocalLoss.apply(sdf, z_slots, cam_loc, ray_dirs, feat_utils.local_desc(local_data, dev))
            output["local_loss"] = lsum / lcnt.clamp(min=1.0)          # 0 when no ray crosses the surface, as feat_utils.py:390-391
            output["local_sum"], output["local_count"] = lsum, lcnt    # ray-sharded steps normalise by the global count
            if self.keep_stages:
                self.stages.update({"d_surface": d_surface, "network_mask": hit})
        if not static:          # per-slot maps the trainer never reads in an optimisation step (plots only)
            far_fill = float(conf.ray_sampler.far)
            output["depth_values"] = torch.where(ray_mask[:, None], depth, torch.ones_like(depth))
            output["depth_vals"] = torch.where(ray_mask[:, None], z_slots * depth_scale, torch.full_like(z_slots, far_fill))
            output["xyz"] = torch.where(valid.unsqueeze(-1), x.view(R, SR, 3), torch.zeros(1, device=dev))

        # ---- pseudo-point loss (:765-780) --------------------------------------------------------
        if static:
            # SDF at the rendered points, dense [R] (1000 where no neighbour); the masked mean is formed by the loss kernels
            if pr is None:
                pr = self._sdf_points(pts_rendered if pts_rendered is not None else DistPoints.apply(cam_loc, ray_dirs, dist_map), with_grad=True, role="pseudo")
            output["_fused"] = {"acc": acc, "grad": gradients.detach(), "slot_valid": q["slot_valid"].view(-1), "n_points": pl.n_points,
                                "psdf": pr["sdf"], "pvalid": pr["valid"], "ray_valid": q["ray_valid"], "local": local_terms}
        else:
            pseudo_pts_loss = torch.zeros((), device=dev)
            pseudo_sum, pseudo_cnt = pseudo_pts_loss, pseudo_pts_loss
            if P > 0:
                pts_rendered = cam_loc + ray_dirs * dist_map[:, None]
                pr = self._sdf_points(pts_rendered, with_grad=True, role="pseudo")
                use = pr["valid"].bool() & ray_mask
                cnt = use.sum()
                pseudo_sum, pseudo_cnt = torch.where(use, pr["sdf"].abs(), torch.zeros_like(pr["sdf"])).sum(), cnt
                l1 = pseudo_sum / cnt.clamp(min=1)
                # no rendered point has a neighbour -> the reference's constant 1000 (no gradient)
                pseudo_pts_loss = torch.where(cnt > 0, l1, torch.full_like(l1, SDF_FILL))
            output.update({"pseudo_pts_loss": pseudo_pts_loss, "pseudo_sum": pseudo_sum, "pseudo_count": pseudo_cnt})
        # sync-free training: the per-point TV terms — their mean is formed inside the fused loss kernels (one reduction launch less)
        if tv_pre is not None:         # per-point values from the step's first launch; FusedLoss runs their backward (tv_ctx)
            output["tv_loss"] = tv_pre
            output["_fused"]["tv_ctx"] = (self.neural_feats_geometry, self.tv_graph())
        else:
            output["tv_loss"] = tv_early if tv_early is not None else self.tv_graph().loss(self.neural_feats_geometry, reduce=not static)
        if not self.training:
            g = gradients.view(R, SR, 3)
            nrm = torch.where(valid.unsqueeze(-1), g / g.norm(2, -1, keepdim=True), torch.zeros(1, device=dev))
            output["normal_map"] = torch.sum(weights.unsqueeze(-1) * nrm, 1).detach()
        elif static:
            output["grad_theta"] = None      # eikonal: a value only (zero gradient), formed by the loss kernels
        else:
            output["grad_theta"] = gradients[rows] if P > 0 else None
        return output
