"""Short-burst vs sustained timing of the colour / geometry kernels (clock behaviour under a continuous matrix-pipe load)."""
import sys
sys.path.insert(0, ".")
import torch  # noqa: E402
import tools.color_bench as cb  # noqa: E402
from tools.microbench import timeit  # noqa: E402
from spurfies_amd import ops  # noqa: E402

geo = lambda: ops.geo_forward(cb.x, cb.pl, cb.dev["neural_pts"], cb.dev["neural_feats_geometry"], cb.packed, 45.0, True)  # noqa: E731
for name, fn in (("colour bwd", cb.bwd_only), ("colour fwd", cb.fwd_train), ("geo fwd+jac", geo)):
    print(name, " burst(20) %.3f ms   sustained(600) %.3f ms   burst again %.3f ms" % (timeit(fn, 20), timeit(fn, 600), timeit(fn, 20)))
