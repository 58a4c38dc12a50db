/* fx_diag.h - entry points that exist ONLY in the diagnostics build of the library (make -C fx8010-emulator-core_amd/csrc diag:
 * -DFX_DIAGNOSTICS, csrc/build/diag/libfx8010_amd.so; fx_knobs.hpp).  Not part of the drop-in boundary (include/fx8010_amd.h): the
 * probes under tools/ bind them through ctypes after loading that library via FX8010_AMD_LIB. */
#ifndef FX_DIAG_H
#define FX_DIAG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* FX_XLATE_ENDSTAMP=1: generated (unstaged) code stores, behind the last sample of a launch, the low word of the 100 MHz clock into
 * word [wavefront] of a buffer of the handle's own - never into an output element.  Copies up to n_words of it (one per 64
 * instances) to out; returns the number of words copied or a negative FX_E_* code.  Single-device handles only. */
int fxb_diag_read_end_stamps(void* handle, uint32_t* out, int64_t n_words);
/* 1: this library was built with -DFX_DIAGNOSTICS */
int fxb_diag_build(void);
#ifdef __cplusplus
}
#endif
#endif
