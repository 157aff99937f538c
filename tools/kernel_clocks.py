"""The shader clock the chip HOLDS while each MLP / weight-gradient kernel runs inside the optimisation step (DVFS give-back,
MI355X_MICROARCH.md: in an MFMA-dense loop the clock sits well under the 2.4 GHz the peak numbers are quoted at).

Needs a -DSPF_CLOCK build of the library (python -m spurfies_amd.build --variant clock "-DSPF_CLOCK", then SPF_LIB_PATH=.../clock.so):
thread 0 of every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around the kernel; this tool runs the
default bench step back to back for a few seconds and prints sum(cycles) / sum(ticks) x 100 MHz per translation unit
(geometry = geo_pairs_x3_kernel, colour = trunk forward + backward, wgrad = every split8 weight-gradient launch).
Usage (GPU box, repo root): SPF_LIB_PATH=$PWD/spurfies_amd/lib/variants/clock.so python3 tools/kernel_clocks.py [seconds]"""
import ctypes
import json
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402
from spurfies_amd import _lib  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
sys.argv = sys.argv[:1]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.set_num_threads(1)
torch.manual_seed(0)
scene, model, step = bench.build_scene_step(args, 0, dev, 1, False)
batches = bench.make_batches(scene, 64, args.rays, 0, 1, dev)
torch.manual_seed(1)
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 32)()
entries = {"geometry (geo_pairs_x3_kernel)": "spf_debug_timing_geo", "colour trunk (forward + backward)": "spf_debug_timing_color",
           "weight gradients (wgrad_split8*)": "spf_debug_timing_wgrad"}
for e in entries.values():
    f = getattr(lib, e)
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
for i in range(20):
    step(*batches[i % 64])
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs / 2:          # bring the chip to its sustained state first
    step(*batches[n % 64])
    n += 1
torch.cuda.synchronize()
for e in entries.values():
    getattr(lib, e)(buf, 1)
t0 = time.perf_counter()
m = 0
while time.perf_counter() - t0 < secs / 2:
    step(*batches[(n + m) % 64])
    m += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out = {"steps": m, "ms_per_step": 1e3 * dt / m, "clocks_ghz": {}}
for name, e in entries.items():
    getattr(lib, e)(buf, 1)
    cyc, ticks, wgs = buf[0], buf[1], buf[2]
    out["clocks_ghz"][name] = {"ghz": round(cyc / max(ticks, 1) * 0.1, 3), "workgroups": int(wgs), "mean_kernel_us_per_workgroup": round(ticks / max(wgs, 1) / 100.0, 1)}
print(json.dumps(out))
