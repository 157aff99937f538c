#!/usr/bin/env python
"""Contract benchmark: ray-samples/sec of one optimisation step of the Spurfies hot path
(kNN + SDF + render, forward + backward + Adam) on synthetic DTU-scan24-shaped batches.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  A "step" = `TrainStep` (spurfies_amd/train.py) on 1024 rays per GPU:
sampler pass (128 samples/ray through kNN + SDF), main pass (98 samples/ray: kNN, SDF + Jacobian,
colour MLPs, compositing), pseudo-point loss, TV, backward, grad all-reduce (N > 1), clip, Adam.
value = 226 * rays_total * K / seconds (SURVEY.md §8(d) counting convention).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLES_PER_RAY = 128 + 98          # sampler + main pass
F_FWD = 2.0 * (35 * 256 + 3 * 256 * 256 + 256)      # F_geometry (4 layers; the 5th folds into T) + T, per pair
F_JAC = 2.0 * (3 * 256 * 256 + 256 * 35)            # input-Jacobian sweep, per pair
PEAK_F32_MFMA_TFLOPS = 157.3                        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2516.6                      # dense bf16 MFMA: 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz
# The dominant kernel forms every fp32 product exactly from 6 bf16 piece products: its matrix-pipe ceiling in
# ALGORITHMIC (fp32) FLOP/s is the bf16 peak / 6.
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=1024, help="rays per GPU per step (config/ours.yaml:14 num_pixels)")
    ap.add_argument("--points", type=int, default=10000, help="neural points (DTU-like cloud)")
    ap.add_argument("--spacing", type=float, default=0.025, help="nearest-neighbour spacing of the synthetic cloud (0.0125 = dense stress cloud)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync", action="store_true", help="default (reference-shaped) step with one host read-back per step")
    ap.add_argument("--graph", action="store_true", help="replay forward + loss + backward as one hipGraph (measured no faster than the eager sync-free "
                    "step on MI355X: ~3 us of dependency handling per graph node; the eager launches run ahead of the GPU)")
    ap.add_argument("--no-graph", action="store_true", help="(default since the sync-free step became GPU-bound; kept for old command lines)")
    ap.add_argument("--cpu-rays", type=int, default=1024)
    return ap.parse_args()


def make_batches(scene, n_steps, rays_total, rank, world, device):
    """Seeded synthetic batches, resident on the device before the timed region."""
    from spurfies_amd import synthetic as syn

    g = torch.Generator().manual_seed(12345)
    K = torch.from_numpy(scene["intrinsics"])[None].to(device)
    out = []
    for s in range(n_steps):
        uv = torch.from_numpy(syn.make_pixels(rays_total, g))
        rgb = torch.rand((rays_total, 3), generator=g)
        mask = (torch.rand((rays_total,), generator=g) > 0.1).float()
        sel = torch.arange(rank, rays_total, world)
        pose = torch.from_numpy(scene["poses"][s % len(scene["poses"])])[None].to(device)
        out.append(({"intrinsics": K, "uv": uv[sel][None].to(device), "pose": pose, "local_data": None},
                    {"rgb": rgb[sel][None].to(device), "mask": mask[sel][None, :, None].repeat(1, 1, 3).to(device)}))
    return out


def cpu_baseline(scene, n_rays):
    """The oracle (CPU restatement of the reference's PyTorch path, validated against the reference
    through tests/golden) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import path as P
    from spurfies_amd import synthetic as syn

    cores = min(os.cpu_count() or 1, 8)   # more intra-op threads only slow these small CPU ops down
    torch.set_num_threads(cores)
    st = P.load_state(scene["state"])
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    grid = P.make_grid(cfg, st["neural_pts"])
    g = torch.Generator().manual_seed(999)
    uv = torch.from_numpy(syn.make_pixels(n_rays, g))[None]
    inp = {"intrinsics": torch.from_numpy(scene["intrinsics"])[None], "uv": uv, "pose": torch.from_numpy(scene["poses"][0])[None]}
    rgb, mask = torch.rand((n_rays, 3), generator=g), torch.ones(n_rays)
    n_steps = 2
    t0 = time.time()
    for _ in range(n_steps):
        P.train_step_grads(inp, rgb, mask, st, cfg, grid=grid)
    dt = time.time() - t0
    return {"value": SAMPLES_PER_RAY * n_rays * n_steps / dt, "unit": "ray-samples/s", "cores": cores, "kind": "port",
            "sample": f"{n_steps} train steps (fwd+bwd, no optimiser) of {n_rays} rays on the same cloud, {dt:.1f} s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SPF_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo only for single-GPU smoke tests
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)

    from spurfies_amd import ops, synthetic as syn
    from spurfies_amd.conf import default_model_conf
    from spurfies_amd.model.pointneus_disent import PointVolSDF
    from spurfies_amd.train import TrainStep

    torch.manual_seed(0)
    scene = syn.make_scene(args.points, seed=0, spacing=args.spacing)
    st = scene["state"]
    conf = default_model_conf(near=0.5, grid_ranges=list(scene["ranges"]))
    model = PointVolSDF(conf, 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]}, device=device)
    model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}, strict=False)
    use_graph = args.graph and not args.sync and world == 1
    step = TrainStep(model, sync_free=not args.sync, use_graph=use_graph)
    rays_total = args.rays * world
    batches = make_batches(scene, args.warmup + args.steps, rays_total, rank, world, device)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    torch.manual_seed(1 + rank * 0)   # identical CPU draws on every rank (stratified jitter is per ray anyway)
    for i in range(args.warmup):
        step(*batches[i])
    if not use_graph:
        ops.profile_start()          # HIP events around every spf_geo_forward launch of the timed region
    sync()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        losses, out = step(*batches[i])
    sync()
    dt = time.perf_counter() - t0
    loss_last = float(losses["loss"].item())
    if use_graph:
        # events cannot be placed inside a hipGraph replay: time the dominant kernel over eager forward+backward passes of
        # the SAME batches right after the timed region (same kernels, same inputs; no optimiser step)
        ops.profile_start()
        for i in range(args.warmup, args.warmup + args.steps):
            step._forward_backward(dict(batches[i][0]), batches[i][1])
        sync()
    prof = ops.profile_stop()
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())

    # dominant kernel: geo_pairs_kernel<true> of the main pass (the largest with-Jacobian launch per step)
    main = [p for p in prof if p["with_grad"] and p["rows"] >= args.rays * 2]
    roof = None
    traffic = None   # HBM bytes per launch from rocprofv3 PMC passes (cannot be collected from inside this process)
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc):
        rec = json.load(open(pmc))
        if rec["config"] == {"points": args.points, "rays": args.rays}:
            hit = [v for k, v in rec["kernels"].items() if "geo_pairs_x3_kernel<true>" in k] or \
                  [v for k, v in rec["kernels"].items() if "geo_pairs_kernel<true>" in k]
            traffic = hit[0]["hbm_bytes_max_corrected"] if hit else None
    if main:
        ms = sum(p["ms"] for p in main)
        pairs = sum(p["pairs"] for p in main)
        ach = pairs * (F_FWD + F_JAC) / (ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_SPLIT_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_SPLIT_TFLOPS,
                "peak_basis": "algorithmic fp32 FLOP/s; each fp32 product = 6 exact bf16 piece products on v_mfma_f32_32x32x16_bf16, so the "
                              "ceiling is the dense bf16 MFMA peak (2516.6 TFLOP/s) / 6; for scale, the fp32-MFMA peak is 157.3 TFLOP/s",
                "achieved_over_fp32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
                "power_envelope_note": "peak is the 2.4 GHz data-sheet figure; the library's GEMM loop alone (tools/micro/x3_loop_rate.hip) holds 1.53 GHz "
                                       "at 90 % matrix-pipe duty and 2.2 GHz at 66-76 %, i.e. 240-290 algorithmic TFLOP/s is what this instruction mix "
                                       "can draw (DESIGN.md section 6)",
                "traffic": traffic, "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
                if traffic else None, "kernel": "geo_pairs_x3_kernel<true> (+ geo_point_reduce_kernel, < 1 % of the launch)",
                "timing": ("HIP events over eager passes of the timed batches, right after the timed region (events cannot sit inside a "
                           "hipGraph replay)") if use_graph else "HIP events over the timed region", "launches": len(main), "avg_ms": ms / len(main),
                "pairs_per_launch": pairs / len(main)}
    res = {
        "metric": "ray-samples/sec (kNN+SDF+render, train step)", "value": SAMPLES_PER_RAY * rays_total * args.steps / dt,
        "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[1] shape (DTU scan24 3-view optimisation, 1024-ray batches) on a synthetic scene: {args.points} neural points, {args.rays} rays/GPU/step x "
                               f"(128 sampler + 98 main) samples, fast=1 optimisation step (fwd+bwd+clip+Adam)",
                   "rays_per_gpu": args.rays, "neural_points": args.points, "k": 8, "max_shading_pts": 80,
                   "parallelism": f"ray-sharded dp{world}", "valid_points_last_step": model.stats.get("valid_points", int(model.stats["counts"][0].item()) if "counts" in model.stats else None),
                   "host_syncs_per_step": 1 if args.sync else 0,
                   "arithmetic": "fp32 throughout; every MLP kernel (geometry, colour trunk, per-point head) and the weight-gradient GEMMs form each fp32 product exactly from three bf16 pieces per operand (6 bf16 MFMAs, fp32 accumulate)",
                   "launch": "hipGraph replay (fwd+loss+bwd) + 3 eager launches (clip + non-finite guard + Adam)" if use_graph else ("eager, reference-shaped (one host read-back)" if args.sync else "eager launches, no host synchronisation (~91 per step)")},
        "roofline": roof,
        "loss_last": loss_last,
    }
    if rank == 0:
        res["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(scene, args.cpu_rays)     # N = 1 only
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
