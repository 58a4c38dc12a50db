"""The library's HOST LOGIC through its C ABI without a GPU: the unchanged host sources linked against tests/hipstub/ (a stand-in
for the HIP runtime calls they make; `make -C fx8010-emulator-core_amd/csrc stublib`), driven by tests/hipstub/host_logic.py in a
child process (the binding reads FX8010_AMD_LIB once, at import).  Register API semantics (the reference's setRegisterValue /
getRegisterValue, source/FX8010.cpp:236-266), moving controls and the code cache, block-length classes, tier choice and its
reasons, state images across partitions, shard routing, error codes - decisions of the host engine that the GPU suite also
checks, here on every CPU run.  PCM results are NOT checked here (the stand-in kernel copies in to out): parity is `-m gpu`."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fx8010-emulator-core_amd", "csrc")


def test_host_logic_on_the_hip_stand_in():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no ROCm clang on this machine")
    subprocess.check_call(["make", "-s", "-C", CSRC, "stublib"])
    lib = os.path.join(CSRC, "build", "stub", "libfx8010_amd.so")
    env = {k: v for k, v in os.environ.items() if not k.startswith("FX_")}
    env["FX8010_AMD_LIB"] = lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipstub", "host_logic.py")], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "all scenarios passed" in r.stdout, r.stdout[-4000:]
    for name in ("registers", "controls, builder thread", "controls, no builder", "block classes", "tiers", "state images", "shards", "errors"):
        assert "%-26s ok" % name in r.stdout, r.stdout
    # the stand-in library must not have pulled the HIP runtime in
    out = subprocess.run(["ldd", lib], stdout=subprocess.PIPE, text=True).stdout
    assert "amdhip" not in out and "hsa-runtime" not in out, out
