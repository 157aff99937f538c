"""CPU: the library's HOST code under AddressSanitizer (the GPU side cannot be sanitised on this pool: no xnack).  A side build of
libspurfies_hip.so with `-fsanitize=address -fno-gpu-sanitize` (host pass instrumented, device code unchanged) is loaded under the ASan
runtime and tests/test_cabi.py runs against it in a subprocess: export / header consistency, argument validation of every entry point that
is called there, grid-handle life cycle and error-text paths — any heap / stack misuse in that host code aborts the run."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_suite_passes_under_an_asan_host_build():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not os.path.exists(hipcc) or not rt:
        pytest.skip("hipcc / the ASan runtime are not in this image")
    from spurfies_amd import build as b

    lib = b.build_variant("asan", "-fsanitize=address -fno-gpu-sanitize -shared-libsan -g")
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    assert "__asan_init" in syms, "the side build is not instrumented"
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=1", SPF_LIB_PATH=lib)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_cabi.py"), "-q", "-x", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0 and "AddressSanitizer" not in out.stderr + out.stdout, (out.stdout[-1500:], out.stderr[-1500:])
    assert " passed" in out.stdout
