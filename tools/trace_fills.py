import sys, traceback, collections
sys.path.insert(0, ".")
import numpy as np, torch
from spurfies_amd import synthetic as syn
from spurfies_amd.conf import default_model_conf
from spurfies_amd.model.pointneus_disent import PointVolSDF
from spurfies_amd.train import TrainStep
import bench
dev = torch.device("cuda", 0)
scene = syn.make_scene(10000, seed=0)
st = scene["state"]
conf = default_model_conf(near=0.5, grid_ranges=list(scene["ranges"]))
model = PointVolSDF(conf, 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]}, device=dev)
model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}, strict=False)
step = TrainStep(model, sync_free=True)
batches = bench.make_batches(scene, 3, 1024, 0, 1, dev)
for i in range(2):
    step(*batches[i])
from torch.utils._python_dispatch import TorchDispatchMode
cnt = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        skip = ("view", "reshape", "empty", "detach", "alias", "slice", "select", "unsqueeze", "squeeze", "expand", "transpose", "t.default",
                "as_strided", "size", "stride", "is_", "_local_scalar", "split", "unbind", "permute", "lift_fresh", "_unsafe_view", "numel")
        if not any(k in name for k in skip):
            st_ = [f"{f.filename.split('/')[-1]}:{f.lineno}" for f in traceback.extract_stack() if "spurfies_amd" in f.filename or "bench.py" in f.filename][-3:]
            shape = tuple(args[0].shape) if args and hasattr(args[0], "shape") else (args[0] if args else None)
            cnt[(name, " < ".join(st_), str(shape))] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    step(*batches[2])
torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: kv[0][1]):
    print(v, k[0], "|", k[1], "|", k[2])
