"""The C restatement must not owe its bits to one compiler's instruction selection: the same source built at -O0 / -O3, with
clang, and under AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile `variants`) has to pass every fixture generated
from the compiled reference — NaN payloads included (nan_collisions.json) — and, where oracle/_ref exists, the live fuzz."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(ROOT, "oracle", "_variants")


@pytest.fixture(scope="module")
def variants():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "variants"])
    return VAR


def _asan_runtime():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.parametrize("variant", ["O0", "O3", "clang", "asan"])
def test_fixtures_pass_on_every_build(variants, variant):
    env = dict(os.environ, FXORACLE_SO=os.path.join(variants, "libfxoracle_%s.so" % variant), FXORACLE_LIVE_CASES="60")
    if variant == "asan":
        rt = _asan_runtime()
        if rt is None:
            pytest.skip("no libasan runtime for gcc on this machine")
        # python itself is not instrumented: preload the runtime, and do not report python's own (intentional) leaks
        env.update(LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    tests = ["tests/test_oracle_golden.py", "tests/test_oracle_vs_ref.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "passed" in r.stdout
