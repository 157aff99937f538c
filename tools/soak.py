"""Soak run of the optimisation step on the bench scene (round-4 verdict item 7).

Leg 1 (the soak): --steps sync-free graph-mode steps at the bench shape (150 k points, 1024 rays), cosine learning-rate schedule as the
reference (train.py:170: CosineAnnealingLR(T_max=100000, eta_min=3e-4)), cycling 64 batches.  Half way the whole-step state
(TrainStep.state_dict) is written to disk, model and step are thrown away, a NEW model + step is built, the state loaded, and the run
continues from there — the second half is a resumed run.  Every --block steps: wall ms/step of the block, loss terms, beta, learning rate,
skipped (non-finite) updates, allocator high-water marks.

Leg 2 (determinism of the resume): under ops.set_scatter_mode("fixed") an uninterrupted run of --det-steps steps and a run that is
checkpointed at the middle, rebuilt and resumed must end in bit-identical parameters and moments.

    python3 tools/soak.py --steps 20000 --out profiles/r05_soak.json
"""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, ".")
_own = argparse.ArgumentParser()
_own.add_argument("--steps", type=int, default=20000)
_own.add_argument("--block", type=int, default=1000)
_own.add_argument("--det-steps", type=int, default=400)
_own.add_argument("--out", default="gpurun_out/r05_soak.json")
_own.add_argument("--local", action="store_true", help="the DTU recipe's real step: every batch carries its view's synthetic local_data (feature-consistency "
                  "term inside the replayed graph, the view descriptor overwritten per step)")
own, rest = _own.parse_known_args()
sys.argv = ["bench.py", "--no-cpu-baseline", "--sustained", "0"] + rest
import bench  # noqa: E402
from spurfies_amd import ops  # noqa: E402

args = bench.parse()
dev = torch.device("cuda", 0)
torch.set_num_threads(1)
N_BATCH = 64


def build(graph=True):
    scene, model, step = bench.build_scene_step(args, 0, dev, 1, graph)
    return scene, model, step


def digest(step):
    h = hashlib.sha256()
    f = step.optimizer._flat
    for t in (f["param"], f["m"], f["v"]):
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def run_block(step, batches, start, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(start, start + n):
        losses, _ = step(*batches[i % N_BATCH])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, losses


def soak():
    import gc

    scene, model, step = build()
    batches = bench.make_batches(scene, N_BATCH, args.rays, 0, 1, dev, local=own.local)
    torch.manual_seed(1)
    half = (own.steps // 2 // own.block) * own.block
    rows, resumed_at = [], None
    tmp = os.path.join(tempfile.mkdtemp(), "soak_state.pth")
    i = 0
    gc.collect()
    gc.freeze()
    while i < own.steps:
        n = min(own.block, own.steps - i)
        ms, losses = run_block(step, batches, i, n)
        i += n
        st = step.optimizer._flat["state"].tolist()
        rows.append({"step": i, "ms_per_step": round(ms, 4), "loss": float(losses["loss"]), "rgb_loss": float(losses["rgb_loss"]),
                     "eikonal_loss": float(losses["eikonal_loss"]), "beta": float(model.density.get_beta()),
                     "lr": step.optimizer.param_groups[1]["lr"], "adam_t": st[0], "skipped_updates": st[1],
                     "alloc_high_water_MiB": round(torch.cuda.max_memory_allocated(dev) / 2**20, 1),
                     "reserved_high_water_MiB": round(torch.cuda.max_memory_reserved(dev) / 2**20, 1)})
        print(rows[-1], flush=True)
        if i == half and resumed_at is None:
            before = digest(step)
            torch.save(step.state_dict(), tmp)
            skipped_before = st[1]
            del step, model
            gc.collect()
            torch.cuda.empty_cache()
            _, model, step = build()
            step.load_state_dict(torch.load(tmp, map_location=dev))
            after = digest(step)
            resumed_at = {"step": i, "state_file_MiB": round(os.path.getsize(tmp) / 2**20, 1), "parameters_and_moments_identical_after_load": before == after,
                          "skipped_updates_before": skipped_before}
            print("resumed", resumed_at, flush=True)
    finite = all(np.isfinite(r["loss"]) for r in rows) and all(bool(torch.isfinite(p).all()) for p in model.parameters())
    ms = [r["ms_per_step"] for r in rows]
    return {"steps": own.steps, "block": own.block, "mode": "sync-free, hipGraph replay, 64 batches cycled" + (", local_data (feature-consistency term) on every step, three views in turn" if own.local else ""), "points": args.points, "rays": args.rays,
            "schedule": "CosineAnnealingLR(T_max=100000, eta_min=3e-4) from 5e-4", "resume": resumed_at, "blocks": rows, "all_finite": finite,
            "ms_per_step_first_block": ms[0], "ms_per_step_last_block": ms[-1], "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
            "skipped_updates_total_since_resume": rows[-1]["skipped_updates"],
            "alloc_high_water_MiB_first_block": rows[0]["alloc_high_water_MiB"], "alloc_high_water_MiB_last_block": rows[-1]["alloc_high_water_MiB"]}


def determinism(graph):
    ops.set_scatter_mode("fixed")
    try:
        n, mid = own.det_steps, own.det_steps // 2
        scene, model, step = build(graph)
        batches = bench.make_batches(scene, N_BATCH, args.rays, 0, 1, dev, local=own.local)
        torch.manual_seed(7)
        run_block(step, batches, 0, n)
        straight = digest(step)
        del step, model
        _, model, step = build(graph)
        torch.manual_seed(7)
        run_block(step, batches, 0, mid)
        blob = step.state_dict()
        buf = os.path.join(tempfile.mkdtemp(), "det_state.pth")
        torch.save(blob, buf)
        del step, model, blob
        torch.manual_seed(12345)                      # the resumed process knows nothing of the first one's generator
        _, model, step = build(graph)
        step.load_state_dict(torch.load(buf, map_location=dev))
        run_block(step, batches, mid, n - mid)
        resumed = digest(step)
        return {"scatter_mode": "fixed", "step_mode": "hipGraph replay" if graph else "eager sync-free", "steps": n, "checkpoint_at": mid, "sha256_uninterrupted": straight, "sha256_resumed": resumed,
                "bit_identical": straight == resumed}
    finally:
        ops.set_scatter_mode("atomic")


if __name__ == "__main__":
    res = {"what": "soak of the optimisation step with a mid-run checkpoint + rebuild + resume; determinism of the resume in fixed scatter mode",
           "csrc_sha256": bench.csrc_digest(), "device": torch.cuda.get_device_name(0)}
    res["determinism"] = determinism(False)
    print(res["determinism"], flush=True)
    try:
        res["determinism_graph"] = determinism(True)
    except Exception as e:  # noqa: BLE001 — recorded, the eager leg above is the claim
        res["determinism_graph"] = {"error": repr(e)[:300]}
    print(res["determinism_graph"], flush=True)
    res["soak"] = soak()
    os.makedirs(os.path.dirname(own.out) or ".", exist_ok=True)
    with open(own.out, "w") as f:
        json.dump(res, f, indent=1)
    s = res["soak"]
    print("soak: %d steps, ms/step first %.3f last %.3f (min %.3f max %.3f), finite %s, skipped %s, alloc high water %.0f -> %.0f MiB, resume bit-identical (fixed mode): %s"
          % (s["steps"], s["ms_per_step_first_block"], s["ms_per_step_last_block"], s["ms_per_step_min"], s["ms_per_step_max"], s["all_finite"],
             s["skipped_updates_total_since_resume"], s["alloc_high_water_MiB_first_block"], s["alloc_high_water_MiB_last_block"], res["determinism"]["bit_identical"]))
