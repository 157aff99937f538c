"""Socket power next to clock and throughput for the kernels DESIGN.md section 5 explains with a "power envelope" (round-3 verdict item 4:
the explanation rested on clocks alone).  A side thread samples the GPU's power / clock sensors (hwmon sysfs, amdsmi python module or
rocm-smi, whichever this box offers to an unprivileged user) while ONE kernel at a time runs back to back for a few seconds:

    idle | tools/micro/x3_loop_rate (GEMM loop alone, 3 epilogue variants) | geometry main pass (both MFMA shapes) | colour forward (training
    stores) | colour backward | the whole optimisation step

    python3 tools/power_probe.py [--secs 4] [--out gpurun_out/r04_power.json]
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")


# ---------------------------------------------------------------------------------------------------------------- sensors
def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except Exception:
        return None


class Sensors:
    """Whatever this box lets an ordinary user read.  Preference: hwmon sysfs (cheap enough for 100 Hz), amdsmi, rocm-smi (subprocess, ~4 Hz)."""

    def __init__(self):
        self.kind, self.detail = None, {}
        self.power_file = self.sclk_file = self.cap_file = self.temp_file = None
        # the node holds several GPUs and the box shows one of them to HIP: take the card whose PCI address is HIP device 0's
        want = None
        try:
            pr = torch.cuda.get_device_properties(0)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception as e:
            self.detail["pci_error"] = repr(e)[:200]
        self.detail["hip_device_pci"] = want
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.detail["cards"] = {hw: os.path.basename(os.path.realpath(os.path.join(hw, "..", ".."))) for hw in cards}
        if want is not None:
            match = [hw for hw in cards if self.detail["cards"][hw].lower().startswith(want)]
            self.detail["matched"] = match
            cards = match or cards
        for hw in cards:
            names = sorted(os.listdir(hw))
            self.detail[hw] = {n: _read(os.path.join(hw, n)) for n in names if n.startswith(("power", "freq", "temp", "name", "in"))}
            for cand in ("power1_average", "power1_input"):
                v = _read(os.path.join(hw, cand))
                if self.power_file is None and v not in (None, "") and v.lstrip("-").isdigit():
                    self.power_file, self.kind = os.path.join(hw, cand), "hwmon:" + cand
                    self.sclk_file = os.path.join(hw, "freq1_input") if _read(os.path.join(hw, "freq1_input")) else None
                    self.cap_file = os.path.join(hw, "power1_cap") if _read(os.path.join(hw, "power1_cap")) else None
                    for t in ("temp2_input", "temp1_input"):
                        if _read(os.path.join(hw, t)):
                            self.temp_file = os.path.join(hw, t)
                            break
        self.smi = None
        if self.power_file is None:
            try:
                import amdsmi  # noqa: F401

                amdsmi.amdsmi_init()
                self.smi = (amdsmi, amdsmi.amdsmi_get_processor_handles()[0])
                self.kind = "amdsmi"
            except Exception as e:
                self.detail["amdsmi_error"] = repr(e)[:200]
        if self.kind is None:
            for tool in ("rocm-smi", "/opt/rocm/bin/rocm-smi"):
                try:
                    out = subprocess.run([tool, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
                    if out.returncode == 0 and out.stdout.strip().startswith("{"):
                        self.kind, self.tool = "rocm-smi", tool
                        self.detail["rocm_smi_sample"] = json.loads(out.stdout)
                        break
                    self.detail["rocm_smi_error"] = (out.stdout + out.stderr)[-300:]
                except Exception as e:
                    self.detail["rocm_smi_error"] = repr(e)[:200]

    def interval(self):
        return {"rocm-smi": 0.3, "amdsmi": 0.02}.get(self.kind, 0.01)

    def sample(self):
        """-> (watts | None, sclk_mhz | None)"""
        if self.power_file:
            p = _read(self.power_file)
            f = _read(self.sclk_file) if self.sclk_file else None
            return (int(p) / 1e6 if p else None), (int(f) / 1e6 if f else None)
        if self.kind == "amdsmi":
            smi, h = self.smi
            try:
                info = smi.amdsmi_get_power_info(h)
                w = info.get("current_socket_power") or info.get("average_socket_power")
                w = float(w) if w not in (None, "N/A") else None
            except Exception:
                w = None
            try:
                c = smi.amdsmi_get_clock_info(h, smi.AmdSmiClkType.GFX)
                f = float(c.get("clk") or c.get("cur_clk"))
            except Exception:
                f = None
            return w, f
        if self.kind == "rocm-smi":
            try:
                out = json.loads(subprocess.run([self.tool, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout)
                card = out[sorted(out)[0]]
                w = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k and str(v).replace(".", "", 1).isdigit()), None)
                f = next((float(str(v).strip("()Mhz")) for k, v in card.items() if k.startswith("sclk clock speed")), None)
                return w, f
            except Exception:
                return None, None
        return None, None


class Sampler(threading.Thread):
    def __init__(self, sensors):
        super().__init__(daemon=True)
        self.s, self.rows, self.on = sensors, [], True
        self.label = "start"

    def run(self):
        dt = self.s.interval()
        while self.on:
            w, f = self.s.sample()
            self.rows.append((time.perf_counter(), self.label, w, f))
            time.sleep(dt)

    def summary(self, label, t_from, t_to):
        r = [(w, f) for t, l, w, f in self.rows if l == label and t_from <= t <= t_to]
        ws, fs = [w for w, _ in r if w is not None], [f for _, f in r if f is not None]
        return {"samples": len(r), "power_w_mean": float(np.mean(ws)) if ws else None, "power_w_max": float(np.max(ws)) if ws else None,
                "power_w_p10": float(np.percentile(ws, 10)) if ws else None, "sclk_mhz_sensor_mean": float(np.mean(fs)) if fs else None}


# ---------------------------------------------------------------------------------------------------------------- loads
def run_for(fn, secs, sampler, label, ramp=1.0):
    """fn() enqueues one launch (or step); run it back to back for ramp + secs seconds; HIP events over the measured part."""
    sampler.label = label + ":ramp"
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    sampler.label = label
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t_from = time.perf_counter()
    e0.record()
    while time.perf_counter() - t_from < secs:
        for _ in range(8):
            fn()
        n += 8
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    t_to = time.perf_counter()
    sampler.label = "between"
    res = sampler.summary(label, t_from + 0.2, t_to)
    res.update({"launches": n, "ms_per_launch": e0.elapsed_time(e1) / n, "busy_fraction_of_wall": e0.elapsed_time(e1) * 1e-3 / (t_to - t_from)})
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--secs", type=float, default=4.0)
    ap.add_argument("--out", default="gpurun_out/r05_power.json")
    a = ap.parse_args()
    sensors = Sensors()
    sys.path.insert(0, ".")
    import bench  # noqa: E402  (csrc_digest: bench.py quotes this collection only while the kernel sources are the ones it was made on)

    res = {"csrc_sha256": bench.csrc_digest(), "sensor": sensors.kind, "sensor_files": {"power": sensors.power_file, "sclk": sensors.sclk_file},
           "power_cap_w": (int(_read(sensors.cap_file)) / 1e6 if sensors.cap_file else None), "loads": {}}
    res["sensor_detail"] = {k: sensors.detail.get(k) for k in ("hip_device_pci", "cards", "matched", "pci_error")}
    print("sensor:", sensors.kind, sensors.power_file, res["sensor_detail"], flush=True)
    if sensors.kind is None:
        res["sensor_detail"] = sensors.detail
        json.dump(res, open(a.out, "w"), indent=1)
        print(json.dumps(res)[:3000])
        return
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    smp = Sampler(sensors)
    smp.start()

    # idle
    smp.label = "idle"
    t0 = time.perf_counter()
    time.sleep(2.0)
    res["loads"]["idle"] = smp.summary("idle", t0, time.perf_counter())
    print("idle", res["loads"]["idle"], flush=True)

    # the GEMM loop alone (separate process on the same GPU; this process only samples)
    exe = os.path.join("tools", "micro", "x3_loop_rate")
    if os.path.exists(exe):
        for mode, what in ((0, "x3 GEMM loop + barrier only"), (1, "x3 GEMM loop + split-only epilogue"), (2, "x3 GEMM loop + bias / LeakyReLU / sign words / split"),
                           (3, "H2 GEMM loop (3 fp16 piece products, main + cross accumulators, combine) + barrier only"), (4, "H2 GEMM loop + split-only epilogue"),
                           (5, "H2 GEMM loop + bias / LeakyReLU / sign words / split")):
            label = f"x3_loop_{mode}" if mode < 3 else f"h2_loop_{mode - 3}"
            smp.label = label + ":ramp"
            t0 = time.perf_counter()
            proc = subprocess.Popen([exe, str(mode), str(a.secs)], stdout=subprocess.PIPE, text=True)
            time.sleep(1.5)                       # process start + its own 1 s of untimed load
            smp.label = label
            t_from = time.perf_counter()
            out, _ = proc.communicate(timeout=120)
            t_to = time.perf_counter()
            smp.label = "between"
            row = smp.summary(label, t_from, t_to - 0.1)
            try:
                row.update(json.loads(out.strip().splitlines()[-1]))
            except Exception:
                row["stdout"] = out[-300:]
            row["what"] = what
            res["loads"][label] = row
            print(label, row, flush=True)
            time.sleep(1.0)

    import bench
    from spurfies_amd import _lib, ops
    from spurfies_amd import synthetic as syn
    from spurfies_amd.torch_knnquery import VoxelGrid

    ops.geo_clock_enable(True)
    scene = syn.make_scene(10000, seed=0, prior="fitted")
    dev = {k: torch.as_tensor(np.asarray(v)).float().cuda() for k, v in scene["state"].items()}
    grid = VoxelGrid((0.025,) * 3, (3,) * 3, (3,) * 3, 26, 20000, scene["ranges"])
    grid.set_pointset(dev["neural_pts"].unsqueeze(0))
    packed = ops.pack_geometry_weights(dev)
    rng = np.random.default_rng(0)
    pts = scene["state"]["neural_pts"]
    n_r, n_s = 1000, 56                                     # queries along rays: neighbour sets overlap from sample to sample, as in a step
    o = pts[rng.integers(0, len(pts), n_r)] + rng.normal(0, 0.01, size=(n_r, 3))
    d = rng.normal(size=(n_r, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = (np.arange(n_s) - n_s / 2) * 0.0015
    x = torch.from_numpy((o[:, None, :] + t[None, :, None] * d[:, None, :]).reshape(-1, 3).astype(np.float32)).cuda()
    q = grid.query_dense(x.unsqueeze(1), 8, 2, 1)
    ps, _, n = ops.compact_points(q["slot_valid"])
    pl = ops.PairList(q["pidx"].reshape(-1, 8), ps, n)
    P, NP = pl.host_counts()
    res["pairs"], res["points"] = NP, P

    for mode, shape in (("h2", "32x32x16_f16"), ("split_w", "32x32x16_bf16"), ("split", "16x16x32_bf16")):
        ops.set_geo_mode(mode)
        ops.geo_clock(reset=True)
        row = run_for(lambda: ops.geo_forward(x, pl, dev["neural_pts"], dev["neural_feats_geometry"], packed, 45.0, with_grad=True), a.secs, smp, "geo_" + mode)
        clk = ops.geo_clock(reset=True).get(("split_w" if mode == "h2" else mode, True))          # (h2 = the 32x32x16 kernel: its counter slot)
        row["every_card_power_w_right_after"] = {hw: (int(_read(os.path.join(hw, "power1_input")) or 0) / 1e6) for hw in sensors.detail.get("cards", {})}
        row.update({"what": f"geo_pairs kernel, main-pass form (SDF + Jacobian sweep), v_mfma_f32_{shape}, {NP} pairs", "ghz_in_kernel": clk["ghz"] if clk else None,
                    "tflops_algorithmic": NP * (bench.F_FWD + bench.F_JAC) / (row["ms_per_launch"] * 1e-3) / 1e12})
        row["frac_of_peak"] = row["tflops_algorithmic"] / (bench.PEAK_BF16_MFMA_TFLOPS / 3.0 if mode == "h2" else bench.PEAK_SPLIT_TFLOPS)
        res["loads"]["geo_" + mode] = row
        print("geo_" + mode, row, flush=True)
    ops.set_geo_mode("h2")
    peak_c = lambda kern: bench.PEAK_BF16_MFMA_TFLOPS / 3.0 if ops._arith_of(0, kern) == 3 else bench.PEAK_SPLIT_TFLOPS      # H2 kernels: three piece products
    pieces = lambda kern: "H2: three fp16 piece products" if ops._arith_of(0, kern) == 3 else "six bf16 piece products"
    geo = ops.geo_forward(x, pl, dev["neural_pts"], dev["neural_feats_geometry"], packed, 45.0, with_grad=True)
    fcp = [dev[f"F_color.{i}.{w}"].clone().requires_grad_(True) for i in (0, 2, 4) for w in ("weight", "bias")]
    table = dev["neural_feats_color"].clone().requires_grad_(True)

    def fwd_train():
        return ops.ColorAgg.apply(table, *fcp, x, geo["wn"], pl, dev["neural_pts"], P, NP)

    row = run_for(fwd_train, a.secs, smp, "color_fwd")
    row.update({"what": f"color_forward_x3_kernel<true> (+ its pack launch), {pieces('color_fwd')}, {NP} pairs", "tflops_algorithmic": NP * bench.F_COLOR_FWD / (row["ms_per_launch"] * 1e-3) / 1e12})
    row["frac_of_peak"] = row["tflops_algorithmic"] / peak_c("color_fwd")
    res["loads"]["color_fwd"] = row
    print("color_fwd", row, flush=True)

    out = fwd_train()
    g = torch.randn_like(out)
    ctx_fn = out.grad_fn
    wn, pk, act0, act1, act2, masks = ctx_fn.saved_tensors
    rows = act1.shape[0]
    G = [torch.empty((rows, 256), device="cuda") for _ in range(3)]
    gb = torch.zeros((3, 256), device="cuda")
    gf = torch.zeros((table.shape[0], 64), device="cuda")

    def bwd_only():
        _lib.check(_lib.lib().spf_color_backward(_lib.ptr(g), _lib.ptr(pl.nbr), _lib.ptr(wn), _lib.ptr(pl.point_slot), _lib.ptr(pl.pair_off),
                                                 _lib.ptr(pl.pair_point), _lib.ptr(pl.n_pairs), NP, pl.k, _lib.ptr(pk), _lib.ptr(masks), _lib.ptr(G[0]),
                                                 _lib.ptr(G[1]), _lib.ptr(G[2]), _lib.ptr(gb[0]), _lib.ptr(gb[1]), _lib.ptr(gb[2]), _lib.ptr(gf), None,
                                                 ops._arith_of(ops._ARITH["color"], "color_bwd"), _lib.stream_ptr()), "bwd")

    try:
        row = run_for(bwd_only, a.secs, smp, "color_bwd")
        row.update({"what": f"color_backward_x3_kernel, {pieces('color_bwd')}, {NP} pairs", "tflops_algorithmic": NP * bench.F_COLOR_BWD / (row["ms_per_launch"] * 1e-3) / 1e12})
        row["frac_of_peak"] = row["tflops_algorithmic"] / peak_c("color_bwd")
        res["loads"]["color_bwd"] = row
        print("color_bwd", row, flush=True)
    except Exception as e:                       # the C signature moves with the rounds; the other rows stand on their own
        res["loads"]["color_bwd"] = {"error": repr(e)[:300]}
    # one weight-gradient GEMM of the colour trunk's shape (dW[256,256] += G^T A over NP rows, row-major operands)
    try:
        rows_w = (NP // 64) * 64
        Gw, Aw = torch.randn((rows_w, 256), device="cuda"), torch.randn((rows_w, 256), device="cuda")
        dW, db = torch.zeros((256, 256), device="cuda"), torch.zeros((256,), device="cuda")
        nr = torch.tensor([rows_w], dtype=torch.int32, device="cuda")
        row = run_for(lambda: ops.wgrad(Gw, Aw, nr, out=dW, dbias=db), a.secs, smp, "wgrad_256")
        row.update({"what": f"spf_wgrad C = 256 (GEMM + slab reduce launches), {pieces('wgrad')}, {rows_w} rows", "tflops_algorithmic": rows_w * 2.0 * 256 * 256 / (row["ms_per_launch"] * 1e-3) / 1e12})
        row["frac_of_peak"] = row["tflops_algorithmic"] / peak_c("wgrad")
        res["loads"]["wgrad_256"] = row
        print("wgrad_256", row, flush=True)
        del Gw, Aw
    except Exception as e:
        res["loads"]["wgrad_256"] = {"error": repr(e)[:300]}
    del G, gb, gf, out, act0, act1, act2

    # the whole optimisation step
    sys.argv = sys.argv[:1]
    args = bench.parse()
    devc = torch.device("cuda", 0)
    torch.set_num_threads(1)
    scene, model, step = bench.build_scene_step(args, 0, devc, 1, False)
    batches = bench.make_batches(scene, 32, args.rays, 0, 1, devc)
    it = [0]

    def one_step():
        step(*batches[it[0] % 32])
        it[0] += 1

    ops.geo_clock(reset=True)
    row = run_for(one_step, a.secs, smp, "step")
    clk = ops.geo_clock(reset=True).get(("split_w", True))
    row.update({"what": "the bench's optimisation step (1024 rays, 10^4 points), back to back", "ghz_in_geo_kernel": clk["ghz"] if clk else None})
    res["loads"]["step"] = row
    print("step", row, flush=True)
    smp.on = False
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
