// fx_knobs.hpp — every environment variable the library reads, in two classes.
//
// RELEASE knobs (kReleaseKnobs; documented in INTEGRATION.md): they choose between implementations that compute the SAME words -
// which tier runs a program, how many pipeline stages, whether a builder thread exists.  None of them can change an output bit
// (tests/test_gpu_boundary.py::test_no_release_knob_changes_a_bit runs config3 under every documented value), so an inherited
// environment can cost time, never correctness.  They are read where they are used, through knob().
//
// DIAGNOSTIC knobs exist only in the -DFX_DIAGNOSTICS build (`make -C fx8010-emulator-core_amd/csrc diag` ->
// csrc/build/diag/libfx8010_amd.so, which the probes under tools/ load through FX8010_AMD_LIB): code padding, pieces of the
// LOG / EXP code left out (WRONG results, timing only), per-wavefront end stamps, optimisations switched off one at a time.  In
// the release library FX_DIAG_KNOB(name) is a null pointer at compile time: the variable's NAME is not even in the binary
// (tests/test_release_knobs.py greps the .so), the code behind it is dead and removed.
#pragma once

#include <cstdlib>
#include <cstring>
#include <string>

namespace fx {

constexpr const char* const kReleaseKnobs[] = {"FX_KERNEL", "FX_INST_PER_LANE", "FX_STAGES", "FX_STAGES_GROUP", "FX_STAGES_TUNE",
                                               "FX_BUILDER", "FX_XLATE_PRIO", "FX_HOST_PIPELINE"};

inline const char* knob(const char* name) { return std::getenv(name); }
inline int knobInt(const char* value, int otherwise) { return value ? std::atoi(value) : otherwise; }

// The release knobs as ONE value, read once: a handle (fx::Batch) takes its copy when it is created and never looks at the
// environment again - its builder thread included, so a host that changes its environment while handles are alive (putenv is
// not thread-safe against getenv) cannot race with it, and two handles of one process can run under different settings.
struct ReleaseKnobs {
    std::string kernel;          // FX_KERNEL: "hip", "asm", "asm_lds", "asm_vNN", "xlate", "xlate_vNN"; empty: unset
    bool instPerLaneSet = false; // FX_INST_PER_LANE given at all (pins the HIP C++ tier)
    int instPerLane = 0;         // ... its value when it is 1, 2 or 4, else 0
    int stages = 0;              // FX_STAGES: number of pipeline stages asked for (clamped 1..16); 0: unset
    int stagesGroup = 0;         // FX_STAGES_GROUP: samples between two barriers (1, 2, 4); 0: unset
    bool stagesTune = true;      // FX_STAGES_TUNE=0: the planner's model decides, nothing is timed
    bool builder = true;         // FX_BUILDER=0: no builder thread
    int xlatePrio = -1;          // FX_XLATE_PRIO: 0 never / 1 always (unstaged) the priority turns; -1: unset
    bool hostPipeline = true;    // FX_HOST_PIPELINE=0: host blocks through staged copies in one piece (no in-place processing of pinned buffers, no pieces)

    bool kernelIs(const char* exact) const { return kernel == exact; }
    bool kernelStartsWith(const char* prefix) const { return kernel.compare(0, std::strlen(prefix), prefix) == 0; }
    static ReleaseKnobs fromEnvironment() {
        ReleaseKnobs k;
        if (const char* v = knob("FX_KERNEL")) k.kernel = v;
        if (const char* v = knob("FX_INST_PER_LANE")) {
            k.instPerLaneSet = true;
            const int n = std::atoi(v);
            k.instPerLane = (n == 1 || n == 2 || n == 4) ? n : 0;
        }
        if (const char* v = knob("FX_STAGES")) { const int n = std::atoi(v); k.stages = n < 1 ? 1 : (n > 16 ? 16 : n); }
        if (const char* v = knob("FX_STAGES_GROUP")) { const int g = std::atoi(v); k.stagesGroup = (g == 1 || g == 2 || g == 4) ? g : 0; }
        k.stagesTune = knobInt(knob("FX_STAGES_TUNE"), 1) != 0;
        k.builder = knobInt(knob("FX_BUILDER"), 1) != 0;
        if (const char* v = knob("FX_XLATE_PRIO")) k.xlatePrio = std::atoi(v) != 0 ? 1 : 0;
        k.hostPipeline = knobInt(knob("FX_HOST_PIPELINE"), 1) != 0;
        return k;
    }
};

#ifdef FX_DIAGNOSTICS
#define FX_DIAG_KNOB(name) (std::getenv(name))
constexpr bool kDiagnosticsBuild = true;
#else
#define FX_DIAG_KNOB(name) (static_cast<const char*>(nullptr))
constexpr bool kDiagnosticsBuild = false;
#endif

}  // namespace fx
