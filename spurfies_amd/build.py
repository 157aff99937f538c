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
SOURCES = ["grid.hip", "geo_mlp.hip", "color_mlp.hip", "rhead_mlp.hip", "wgrad.hip", "render.hip", "sampler.hip", "latents.hip", "camera.hip", "loss.hip", "local.hip", "optim.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "spurfies_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, flags: str, verbose: bool = False) -> str:
    """Side build for same-box A/B runs (tools/ab_prof.sh): lib/variants/<name>.so compiled with extra flags; select with SPF_LIB_PATH."""
    vdir = os.path.join(LIBDIR, "variants", name)
    os.makedirs(vdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(vdir, src.replace(".hip", ".o"))
        subprocess.run([hipcc, *FLAGS, *flags.split(), "-c", os.path.join(CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    out = os.path.join(LIBDIR, "variants", name + ".so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    if verbose:
        print(out)
    return out


def build(force: bool = False, verbose: bool = True, incremental: bool = False) -> str:
    """incremental (developer loop): recompile only the sources that are newer than their object (a header change recompiles everything)."""
    if not force and not incremental and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "spurfies_hip.h")]
    t_hdr = max(os.path.getmtime(h) for h in headers)
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if incremental and not force and os.path.exists(obj) and os.path.getmtime(obj) > max(t_hdr, os.path.getmtime(os.path.join(CSRC, src))):
            continue
        cmd = [hipcc, *FLAGS, *os.environ.get("SPF_EXTRA_HIPCC_FLAGS", "").split(), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:          # python -m spurfies_amd.build --variant NAME "-DFLAG ..."
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2] if len(sys.argv) > i + 2 else ""))
    else:
        build(force="--force" in sys.argv, incremental="--incremental" in sys.argv)
        print(LIB)
