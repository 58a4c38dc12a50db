"""The host side of the library - loader, lowering, translator, stage planner, C ABI - under AddressSanitizer +
UndefinedBehaviorSanitizer (`make -C fx8010-emulator-core_amd/csrc asan`: the same sources, device code untouched - GPU
sanitizers are not available on this pool, the device code has the hazard lint instead).  The CPU tests of the front-end, of
the translator and of the stage planner run against that build in a child process (python itself is not instrumented: the
runtime is preloaded).  First finding when this was set up: a memcpy from an empty vector's data() in fxp_translate."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fx8010-emulator-core_amd", "csrc")


def _runtime():
    found = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return found[-1] if found else None


def test_host_code_under_asan_and_ubsan():
    rt = _runtime()
    if rt is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no ROCm clang with an ASan runtime on this machine")
    subprocess.check_call(["make", "-s", "-C", CSRC, "asan"])
    lib = os.path.join(CSRC, "build", "asan", "libfx8010_amd.so")
    env = dict(os.environ, FX8010_AMD_LIB=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    tests = ["tests/test_frontend.py", "tests/test_xlate.py", "tests/test_stages.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-6000:]
    assert "passed" in r.stdout
