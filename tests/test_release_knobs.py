"""The release library and its environment: everything that changes results or pads generated code lives
behind -DFX_DIAGNOSTICS (csrc/fx_knobs.hpp, `make diag`); the shipped .so reads exactly the eight documented knobs, none of
which can change an output bit (the GPU half of that claim: tests/test_gpu_boundary.py::test_no_release_knob_changes_a_bit).
The contract being protected is the reference's bit-exact process() (/root/reference/source/FX8010.cpp:1023-1249)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fx8010-emulator-core_amd")
CSRC = os.path.join(PKG, "csrc")
RELEASE = ["FX_KERNEL", "FX_INST_PER_LANE", "FX_STAGES", "FX_STAGES_GROUP", "FX_STAGES_TUNE", "FX_BUILDER", "FX_XLATE_PRIO", "FX_HOST_PIPELINE"]


def _sources():
    out = {}
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".cpp", ".hpp", ".hip", ".h")):
            out[f] = open(os.path.join(CSRC, f)).read()
    return out


def _fx_names(path):
    data = open(path, "rb").read()
    return sorted(set(m.decode() for m in re.findall(rb"FX_[A-Z][A-Z0-9_]{2,}", data)))


def test_every_getenv_goes_through_the_knob_header():
    src = _sources()
    for name, text in src.items():
        if name == "fx_knobs.hpp":
            continue
        assert "getenv" not in text, "%s reads the environment directly: use fx::knob / FX_DIAG_KNOB (fx_knobs.hpp)" % name
    declared = re.search(r"kReleaseKnobs\[\] = \{([^}]*)\}", src["fx_knobs.hpp"]).group(1)
    assert sorted(re.findall(r'"(FX_[A-Z_]+)"', declared)) == sorted(RELEASE)
    # a release knob is read by fx::knob("..."), a diagnostic one only ever by FX_DIAG_KNOB("...")
    diag = set()
    for name, text in src.items():
        diag |= set(re.findall(r'FX_DIAG_KNOB\("(FX_[A-Z0-9_]+)"\)', text))
        for k in re.findall(r'(?<![A-Z_])knob\("(FX_[A-Z0-9_]+)"\)', text):
            assert k in RELEASE, "%s: %s is read as a release knob but is not documented as one" % (name, k)
    assert len(diag) >= 18 and not (diag & set(RELEASE))
    assert {"FX_XLATE_ENDSTAMP", "FX_XLATE_LUTPROBE_WRONG_RESULTS", "FX_XLATE_LOOPPAD", "FX_XLATE_LUTPAD", "FX_XLATE_LUTPRIO", "FX_STAGES_NO_RING",
            "FX_STAGES_BALANCE"} <= diag


def test_release_library_knows_no_diagnostic_knob():
    lib = os.path.join(PKG, "libfx8010_amd.so")
    assert os.path.exists(lib)
    names = [n for n in _fx_names(lib) if not n.startswith(("FX_E_", "FX_OPT_", "FXB_"))]
    assert sorted(names) == sorted(RELEASE), names
    blob = open(lib, "rb").read()
    for word in (b"WRONG_RESULTS", b"ENDSTAMP", b"LOOPPAD", b"LUTPAD", b"DIAGNOSTICS BUILD", b"fxb_diag_"):
        assert word not in blob, word


def test_integration_md_documents_exactly_the_release_knobs():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("### Environment knobs of the release library"):text.index("### The diagnostics build")]
    rows = re.findall(r"^\| `(FX_[A-Z_]+)` \|", section, flags=re.M)
    assert sorted(rows) == sorted(RELEASE)
    assert "none of them can\nchange an output bit" in section or "none of them can change an output bit" in section.replace("\n", " ")


def test_the_diagnostics_build_is_where_the_knobs_live(tmp_path):
    """`make diag` builds; a padding knob changes the code the DIAGNOSTICS build generates (fxp_code_hash, no device) and
    leaves the release library's untouched"""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this machine")
    subprocess.check_call(["make", "-s", "-j6", "-C", CSRC, "diag"])
    diag = os.path.join(CSRC, "build", "diag", "libfx8010_amd.so")
    assert "FX_XLATE_LUTPROBE_WRONG_RESULTS" in _fx_names(diag) and b"fxb_diag_read_end_stamps" in open(diag, "rb").read()
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import fx8010_amd as A, fx8010_programs as P\n"
            "p = A.FrontEnd(1); assert p.load_text(P.config4())\n"
            "print(p.code_hash(128, 1))\n") % (os.path.join(PKG, "python"), os.path.join(ROOT, "oracle"))

    def run(lib, **env):
        e = dict(os.environ, **env)
        for k in list(e):
            if k.startswith("FX_") and k not in env:
                del e[k]
        if lib:
            e["FX8010_AMD_LIB"] = lib
        else:
            e.pop("FX8010_AMD_LIB", None)
        return subprocess.run([sys.executable, "-c", code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stdout.strip()

    plain = run(None)
    assert run(None, FX_XLATE_LOOPPAD="4", FX_XLATE_LUTPROBE_WRONG_RESULTS="3", FX_XLATE_ENDSTAMP="1", FX_XLATE_CSE="0") == plain   # release: deaf
    assert run(diag) == plain                                                                # same sources, same code without knobs
    assert run(diag, FX_XLATE_LOOPPAD="4") != plain and run(diag, FX_XLATE_LUTPROBE_WRONG_RESULTS="3") != plain
