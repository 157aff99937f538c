"""Which host calls produce the step's memcpy launches (__amd_rocclr_copyBuffer in the kernel trace)?  torch profiler with stacks."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--no-cpu-baseline", "--sustained", "0"]
import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda", 0)
scene, model, step = bench.build_scene_step(args, 0, dev, 1, False)
batches = bench.make_batches(scene, 8, 1024, 0, 1, dev)
torch.manual_seed(1)
for i in range(4):
    step(*batches[i])
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(*batches[5])
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    if "emcpy" in ev.name or "copy_" in ev.name or "aten::to" == ev.name or "aten::fill_" == ev.name or "aten::zero_" == ev.name:
        st = [f for f in (ev.stack or []) if "spurfies_amd" in f or "bench.py" in f]
        key = (ev.name, st[0] if st else "?")
        seen[key] = seen.get(key, 0) + 1
for (name, where), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {name:28s} {where}")
