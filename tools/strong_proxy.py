"""Strong-scaling proxy on ONE GPU: the per-rank workload an 8-GPU strong-scaled run of BASELINE.json configs[1] / configs[4] implies
(R / N rays of the batch per rank, full replica of the cloud), timed as the optimisation step before communication.

    python3 tools/strong_proxy.py [--out gpurun_out/strong_proxy.json] [--dense] [--steps 60]

Per ray count: ms/step eager (sync-free launches), ms/step with the forward + loss + backward replayed as one hipGraph (single stream, and
FORKED: independent passes as parallel branches — TrainStep(fork=True)), the host's time to
ENQUEUE an eager step (if that exceeds the GPU's step time the eager mode is host-bound).
--scenes S: the configs[3] proxy — S scenes stepped round-robin on two streams (MultiSceneTrainer), every scene with rays/8 rays of its batch
(what one rank of the 8-GPU run does), graph replays; against the same S scenes at the full batch.
"""
import argparse
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from spurfies_amd import ops  # noqa: E402
from spurfies_amd import synthetic as syn  # noqa: E402
from spurfies_amd.conf import default_model_conf  # noqa: E402
from spurfies_amd.model.pointneus_disent import PointVolSDF  # noqa: E402
from spurfies_amd.train import MultiSceneTrainer, TrainStep  # noqa: E402


def build(scene, device, **kw):
    st = scene["state"]
    conf = default_model_conf(near=0.5, grid_ranges=list(scene["ranges"]))
    model = PointVolSDF(conf, 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]}, device=device)
    model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}, strict=False)
    return model, TrainStep(model, **kw)


def timed(step, batches, warm, n, settle=1.0):
    """-> (ms/step over n steps right behind `warm` warm-up steps [round 4's figure: the chip's clock has not settled yet], host enqueue times,
    ms/step over n steps behind another `settle` seconds of the same step [the settled clock: what a long run sees; bench.py's --settle])."""
    import gc

    for i in range(warm):
        step(*batches[i % len(batches)])
    gc.collect()
    gc.disable()
    torch.cuda.synchronize()
    enq = []
    t0 = time.perf_counter()
    for i in range(warm, warm + n):
        t1 = time.perf_counter()
        step(*batches[i % len(batches)])
        enq.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    i = warm + n
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < settle:
        for _ in range(20):
            step(*batches[i % len(batches)])
            i += 1
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    for k in range(n):
        step(*batches[(i + k) % len(batches)])
    torch.cuda.synchronize()
    dts = time.perf_counter() - t2
    gc.enable()
    return dt / n * 1e3, 1e3 * float(np.median(enq[:3])), 1e3 * float(np.median(enq)), dts / n * 1e3


def multi_scene(a, dev):
    """configs[3] on one rank of an 8-GPU group: S scenes x 128 rays, graph-replayed forked steps on two streams; baseline: S x 1024 rays eager."""
    S = a.scenes
    scenes = [syn.make_scene(10000, seed=s, prior="fitted") for s in range(S)]
    rows = []
    for rays, kw, mode in ((1024, dict(sync_free=True), "eager"), (128, dict(sync_free=True, use_graph=True), "graph"),
                           (128, dict(sync_free=True, use_graph=True, fork=True), "graph_forked")):
        built = []
        for s, sc in enumerate(scenes):
            torch.manual_seed(1 + s)
            built.append(build(sc, dev, **kw))
        batches = [bench.make_batches(sc, 16, rays, 0, 1, dev, seed=12345 + s) for s, sc in enumerate(scenes)]
        multi = MultiSceneTrainer([st for _, st in built], n_streams=2, device=dev)
        import gc

        i = 0
        t1 = time.perf_counter()
        while i < 6 or time.perf_counter() - t1 < 1.0:                 # warm-up incl. captures, then ~1 s of settling
            multi.step([b[i % len(b)] for b in batches])
            i += 1
            if i % 4 == 0:
                torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            multi.step([b[(i + k) % len(b)] for b in batches])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        gc.enable()
        rows.append({"rays_per_scene": rays, "mode": mode, "ms_per_round": ms, "ray_samples_per_s": bench.SAMPLES_PER_RAY * rays * S / (ms * 1e-3)})
        print(json.dumps(rows[-1]), flush=True)
        del built, multi, batches
        torch.cuda.empty_cache()
    for r in rows:
        r["speedup_vs_full_batch"] = rows[0]["ms_per_round"] / r["ms_per_round"]
    res = {"what": f"strong-scaling proxy, configs[3] shape: {S} scenes round-robin on one MI355X (two streams), each with 128 rays per step = one rank's "
                   "share of 1024-ray batches over 8 GPUs, against the same scenes at 1024 rays; optimisation steps before communication",
           "scenes": S, "steps_timed": a.steps, "device": torch.cuda.get_device_name(0), "rows": rows}
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/strong_proxy.json")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--dense", action="store_true", help="configs[4]: 2e5 points, spacing 0.0125, 4096 rays per batch")
    ap.add_argument("--engine", default="h2")
    ap.add_argument("--scenes", type=int, default=0, help="configs[3] proxy: this many scenes round-robin on one GPU (two streams)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.set_num_threads(1)
    ops.set_geo_mode(a.engine)
    if a.dense:
        points, spacing, rays_list, name = 200000, 0.0125, [4096, 2048, 1024, 512], "configs[4]"
    else:
        points, spacing, rays_list, name = 10000, 0.025, [1024, 512, 256, 128], "configs[1]"
    if a.scenes > 0:
        return multi_scene(a, dev)
    scene = syn.make_scene(points, seed=0, spacing=spacing, prior="fitted")
    rows = []
    for rays in rays_list:
        batches = bench.make_batches(scene, 32, rays, 0, 1, dev)
        row = {"rays_per_rank": rays, "ranks_implied": rays_list[0] // rays}
        for mode, kw in (("eager", dict(sync_free=True)), ("graph", dict(sync_free=True, use_graph=True)),
                         ("graph_forked", dict(sync_free=True, use_graph=True, fork=True))):
            torch.manual_seed(1)
            model, step = build(scene, dev, **kw)
            ms, enq_first, enq_med, ms_settled = timed(step, batches, 12, a.steps)
            row[mode] = {"ms_per_step": ms, "ms_per_step_settled": ms_settled, "host_enqueue_ms_first3": enq_first, "host_enqueue_ms_median": enq_med,
                         "ray_samples_per_s": bench.SAMPLES_PER_RAY * rays / (ms * 1e-3)}
            counts = model.stats.get("counts")
            if counts is not None:
                row["valid_points"], row["pairs"] = int(counts[0].item()), int(counts[1].item())
            del model, step
            torch.cuda.empty_cache()
        rows.append(row)
        print(json.dumps(row), flush=True)
    base = rows[0]
    for r in rows:
        for mode in ("eager", "graph", "graph_forked"):
            r[mode]["speedup_vs_full_batch"] = base[mode]["ms_per_step"] / r[mode]["ms_per_step"]
            r[mode]["speedup_vs_full_batch_settled"] = base[mode]["ms_per_step_settled"] / r[mode]["ms_per_step_settled"]
    res = {"what": f"strong-scaling proxy, {name} shape: one rank's share of the batch on one MI355X, optimisation step before communication "
                   "(fwd + loss + bwd + clip + Adam), fitted prior", "neural_points": points, "engine": a.engine, "steps_timed": a.steps,
           "device": torch.cuda.get_device_name(0), "rows": rows}
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
