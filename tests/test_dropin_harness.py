"""The C++ drop-in boundary, checked with the reference's OWN caller: source/main.cpp + source/helpers.cpp are compiled
where they lie (nothing of them is copied) against fx8010-emulator-core_amd/host/FX8010.h and linked with
libfx8010_amd.so.  main.cpp needs everything the reference header hands its includers - `using namespace std`, PI,
DEBUG, AUDIOBLOCKSIZE, SAMPLERATE, <iostream>/<chrono>/<cmath> - besides the class itself.
Skipped where /root/reference is absent (the GPU box); there tests/test_gpu_boundary.py runs the prebuilt binary."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PKG = os.path.join(ROOT, "fx8010-emulator-core_amd")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "source")), reason="/root/reference not present")


@needs_ref
@pytest.mark.parametrize("guard", [[], ["-DFX8010_H"]], ids=["header_defines_guard", "guard_on_command_line"])
def test_reference_harness_compiles_and_links_against_the_dropin_header(tmp_path, guard):
    exe = str(tmp_path / "main_dropin")
    cmd = ["g++", "-std=c++17", "-O1", "-DFX8010_REFERENCE_COMPAT"] + guard + ["-include", os.path.join(PKG, "host", "FX8010.h"), "-I", os.path.join(ROOT, "include"),
           os.path.join(REF, "source", "main.cpp"), os.path.join(REF, "source", "helpers.cpp"), "-L", PKG, "-lfx8010_amd", "-Wl,-rpath," + PKG, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    errors = [l for l in r.stderr.splitlines() if "error" in l]
    assert r.returncode == 0 and not errors, "\n".join(errors[:20])
    assert os.path.exists(exe)
    # every FX8010 member main.cpp calls is bound to the library, not to the reference's FX8010.cpp (which is not linked)
    syms = subprocess.run(["nm", "-C", "--undefined-only", exe], capture_output=True, text=True).stdout
    for name in ("fx_create", "fx_load_file", "fx_process", "fx_set_register", "fx_get_register", "fx_instruction_counter"):
        assert name in syms, name


@needs_ref
def test_dropin_header_is_self_contained_and_exports_the_reference_macros(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#define FX8010_REFERENCE_COMPAT\n#include "FX8010.h"\n'
                   'int main() { Klangraum::FX8010* p = nullptr; (void)p; vector<float> v(AUDIOBLOCKSIZE); string s = "x";\n'
                   ' static_assert(SAMPLERATE == 48000 && AUDIOBLOCKSIZE == 32 && DEBUG == 0 && PRINT_REGISTERS == 0, "macros");\n'
                   ' static_assert(MAX_IDELAY_SIZE == 8192 && MAX_XDELAY_SIZE == 1048576, "tram");\n'
                   ' double a = PI * E; cout << sin(a) << setprecision(3) << endl; std::regex r("a"); std::map<int,int> m; std::array<int,2> q{};\n'
                   ' (void)r; (void)m; (void)q; return (int)v.size() - 32; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(PKG, "host"), "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[:2000]


def test_dropin_header_leaks_nothing_without_the_compat_switch(tmp_path):
    """a new includer (not the reference's own caller) gets the classes only: no `using namespace std`, no macros named E /
    DEBUG / PI, and the reference's include guard stays free"""
    src = tmp_path / "t.cpp"
    src.write_text('#define DEBUG 1\n'                                      # a -DDEBUG build
                   '#include "FX8010.h"\n'
                   'template <class E> struct Holder { E value; };\n'         # E as a template parameter
                   'static_assert(DEBUG == 1, "the includer\'s own DEBUG");\n'
                   '#if defined(PI) || defined(E) || defined(FX8010_H) || defined(AUDIOBLOCKSIZE)\n#error leaked\n#endif\n'
                   'struct vector {}; struct string {};\n'                   # would clash with a namespace dump
                   'int main() { Klangraum::FX8010* p = nullptr; Klangraum::FX8010Batch* q = nullptr; (void)p; (void)q; Holder<int> h{Klangraum::kAudioBlockSize};\n'
                   ' bool (Klangraum::FX8010::*load)(const std::string&) = &Klangraum::FX8010::load; (void)load; return h.value - 32; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(PKG, "host"), "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[:2000]
