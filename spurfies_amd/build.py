"""Build libspurfies_hip.so for gfx950 with hipcc (no torch headers, no cmake).

    python -m spurfies_amd.build [--force]

The shared library lands in spurfies_amd/lib/ (git-ignored, but it travels with gpurun).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libspurfies_hip.so")
SOURCES = ["grid.hip", "geo_mlp.hip", "color_mlp.hip", "rhead_mlp.hip", "wgrad.hip", "render.hip", "sampler.hip", "latents.hip", "camera.hip", "loss.hip", "optim.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "spurfies_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *os.environ.get("SPF_EXTRA_HIPCC_FLAGS", "").split(), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
