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
# -amdgpu-mfma-vgpr-form (round 6): MFMA accumulators in architectural VGPRs.  The compiler's default put the H2 kernels' accumulators (main + cross:
# 128 registers per wave tile) into AGPRs and paid a v_accvgpr_read per element in every epilogue — 21 % of the geometry kernel's vector
# instructions, 12 - 14 % of the colour / head kernels' (static counts: geo main pass 12 445 -> 10 417 VALU, head backward 6 222 -> 5 467); the
# kernels are VALU- and power-bound beside the matrix pipe, so the step went 3.45 -> 3.29 ms on one box (A B B A B A).  No scratch in any kernel.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wall", "-Wno-unused-function"]
FLAGS_STAMP = os.path.join(LIBDIR, "build_flags.txt")      # objects built with other flags are stale


def _flags_text() -> str:
    return " ".join(FLAGS + os.environ.get("SPF_EXTRA_HIPCC_FLAGS", "").split())


def _stale() -> bool:
    if not os.path.exists(LIB) or (os.path.exists(FLAGS_STAMP) and open(FLAGS_STAMP).read() != _flags_text()):
        return True                                    # (a library that travelled without its stamp is taken as it is: dates decide)
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
    if incremental and not (os.path.exists(FLAGS_STAMP) and open(FLAGS_STAMP).read() == _flags_text()):
        force = True                                   # other (or unknown) flags than the objects on disk were built with: everything again
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
    with open(FLAGS_STAMP, "w") as f:
        f.write(_flags_text())
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:          # python -m spurfies_amd.build --variant NAME "-DFLAG ..."
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2] if len(sys.argv) > i + 2 else ""))
    else:
        build(force="--force" in sys.argv, incremental="--incremental" in sys.argv)
        print(LIB)
