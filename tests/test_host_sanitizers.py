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
    subprocess.check_call(["make", "-s", "-j6", "-C", CSRC, "asan"])
    lib = os.path.join(CSRC, "build", "asan", "libfx8010_amd.so")
    env = dict(os.environ, FX8010_AMD_LIB=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    tests = ["tests/test_frontend.py", "tests/test_xlate.py", "tests/test_stages.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-6000:]
    assert "passed" in r.stdout


def _host_threads(target, env, runs):
    """build tests/hipstub/host_threads.cpp against the library's host sources (csrc/Makefile `tsan` / `stubasan`) and run it"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no ROCm clang on this machine")
    subprocess.check_call(["make", "-s", "-j6", "-C", CSRC, target])
    exe = os.path.join(CSRC, "build", target, "host_threads")
    for args in runs:
        r = subprocess.run([exe] + args, cwd=ROOT, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0 and "0 check(s) failed" in r.stdout, "%s %s\n%s\n%s" % (target, args, r.stdout[-2000:], r.stderr[-6000:])
        for name in ("controls", "queued", "hostpipe", "shards", "handles", "memory", "modules", "images"):
            assert "%-9s ok" % name in r.stdout, r.stdout
        assert "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]


def test_host_threads_under_tsan():
    """ThreadSanitizer over the library's own threads (the reference has none: /root/reference/include/FX8010.h:47-75) - the builder
    thread and the code cache it hands across, cache eviction while the caller keeps lowering, the launch-timing tuner with
    launches queued, drainBuilder during a load, a 3-shard handle with two host threads posting - without a GPU: the host
    sources are linked against tests/hipstub/ (streams are worker threads; kernels take time and touch their buffers), at three
    kernel durations for other interleavings.  First run (round 5) found: the builder reading instPerLane_ / iSlotsAlloc_ while
    the caller's own build wrote them (now a snapshot in BuildInputs), Sharded::lastError_ cleared in front of the mailbox lock
    (now one lock per call), and a failed block returning with its H2D copy still queued on the caller's buffer."""
    _host_threads("tsan", {"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1"}, [["--kernel-us=0"], ["--kernel-us=60"], ["--kernel-us=700"]])


def test_host_logic_and_error_paths_under_asan_without_a_device():
    """the same scenarios under AddressSanitizer + UBSan + LeakSanitizer: allocation failure (FX_E_MEMORY, handle stays usable,
    nothing leaks over 50 create / load / run / destroy cycles), module-load failures on the caller's and on the builder's thread,
    damaged state images in exactly-sized heap buffers (a read outside the image is a report)"""
    _host_threads("stubasan", {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}, [[]])


def test_random_host_call_sequences_under_asan_without_a_device():
    """tools/fuzz_api.py - random programs x random host call sequences: blocks of 1-40 samples, broadcast / per-instance / array
    register writes, control schedules (broadcast and per instance, up to five per block), noise seeds, reads, fxb_prepare, the
    whole state through an image into a NEW handle, PCM in pageable arrays or in pinned buffers of the library (processed in place,
    half of the blocks with input and output in ONE buffer) - through the batch engine under AddressSanitizer + UBSan, single handles and
    two-shard handles, with nothing compared (FX_FUZZ_NOCOMPARE: the stand-in kernel computes nothing; on the GPU the same
    sequences are compared with the oracle word for word).  What is under test is the host's memory discipline on paths only
    call SEQUENCES reach: rows that come and go, schedules re-armed, code cached and evicted, images loaded into fresh handles."""
    rt = _runtime()
    if rt is None or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no ROCm clang with an ASan runtime on this machine")
    subprocess.check_call(["make", "-s", "-j6", "-C", CSRC, "stubasanlib"])
    lib = os.path.join(CSRC, "build", "stubasan", "libfx8010_amd.so")
    base = {k: v for k, v in os.environ.items() if not k.startswith("FX_")}
    base.update(FX8010_AMD_LIB=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
                FX_FUZZ_NOCOMPARE="1", FXSTUB_ABORT_ON_BAD_PCM="1")
    for extra, first, count in (({}, 7000, 120), ({"FX_FUZZ_SHARDS": "2"}, 7100, 50), ({"FX_FUZZ_WILD": "1", "FX_BUILDER": "0"}, 7200, 40), ({"FX_FUZZ_PINNED": "1"}, 7300, 60), ({"FX_FUZZ_PANEL": "1"}, 7400, 60)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_api.py"), str(first), str(count)], cwd=ROOT, env=dict(base, **extra),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0 and "failures []" in r.stdout, r.stdout[-6000:]
