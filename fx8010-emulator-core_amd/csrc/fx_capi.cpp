// fx_capi.cpp — extern "C" surface declared in include/fx8010_amd.h.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <string>

#include "../../include/fx8010_amd.h"
#include <vector>

#include "fx_batch.hpp"
#include "fx_knobs.hpp"
#include "fx_shard.hpp"
#include "fx_xlate.hpp"

// a batch is one shard (fxb_create) or several (fxb_create_sharded / fxb_create_on_devices)
struct fxb_handle {
    fx::Sharded batch;
    fxb_handle(int64_t n, int ch, const std::vector<int>& devices) : batch(n, ch, devices) {}
};
struct fx_handle {
    fx::Batch batch;
    explicit fx_handle(int ch) : batch(1, ch, -1) {}
};

struct fxp_handle {
    fx::Program prog;
    fx::Lowered low;
    bool lowered = false;
    std::string err;
    std::vector<int> tracked;  // registers fxp_translate treats as trackable (fxp_track_register)
    explicit fxp_handle(int ch) : prog(ch) {}
    void noteError(const std::string& what) { err = what; }
};

namespace {
thread_local std::string g_createError;

template <class H, class... A>
H* create(A... args) {
    try {
        g_createError.clear();
        return new H(args...);
    } catch (const std::exception& e) {
        g_createError = e.what();
    } catch (...) {
        g_createError = "unknown error";
    }
    return nullptr;
}

int errorCount(fx::Batch& b) { return (int)b.program().errors.size(); }
const char* errorDesc(fx::Batch& b, int i) {
    const auto& e = b.program().errors;
    return (i < 0 || i >= (int)e.size()) ? "" : e[i].description.c_str();
}
int errorRow(fx::Batch& b, int i) {
    const auto& e = b.program().errors;
    return (i < 0 || i >= (int)e.size()) ? -1 : e[i].row;
}
int controlCount(fx::Batch& b) { return (int)b.program().controls.size(); }
const char* controlAt(fx::Batch& b, int i) {
    const auto& c = b.program().controls;
    return (i < 0 || i >= (int)c.size()) ? "" : c[i].c_str();
}
int metaGet(fx::Batch& b, const char* key, char* buf, int buflen) {
    if (!key) return 0;
    for (const auto& kv : b.program().meta)
        if (kv.first == key) {
            if (buf && buflen > 0) std::snprintf(buf, (size_t)buflen, "%s", kv.second.c_str());
            return 1;
        }
    return 0;
}


// No exception may cross the C boundary (the caller may be C, ctypes or a DAW host): every entry point runs its body
// through one of these; the message lands where fx*_last_error() finds it.
int codeOf(const std::exception& e) { return dynamic_cast<const std::bad_alloc*>(&e) ? FX_E_MEMORY : FX_E_PROGRAM; }
template <class H, class F>
auto guard(H* h, decltype(std::declval<F>()()) onError, F f) -> decltype(f()) {
    try {
        return f();
    } catch (const std::exception& e) {
        if (h) h->noteError(e.what());
    } catch (...) {
        if (h) h->noteError("unknown error");
    }
    return onError;
}
template <class H, class F>
int guardCode(H* h, F f) {
    try {
        return f();
    } catch (const std::exception& e) {
        if (h) h->noteError(e.what());
        return codeOf(e);
    } catch (...) {
        if (h) h->noteError("unknown error");
        return FX_E_PROGRAM;
    }
}
}  // namespace

extern "C" {

const char* fx_last_create_error(void) { return g_createError.c_str(); }
const char* fxb_version(void) { return "fx8010_amd 0.1 (gfx950)"; }
int fxb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* ---- single instance ---- */
fx_handle* fx_create(int num_channels) { return create<fx_handle>(num_channels); }
void fx_destroy(fx_handle* h) { delete h; }
int fx_load_file(fx_handle* h, const char* path) { return (h && path) ? guard(&h->batch, 0, [&] { return h->batch.loadFile(path) ? 1 : 0; }) : 0; }
int fx_process(fx_handle* h, const float* in, float* out) { return h ? guardCode(&h->batch, [&] { return h->batch.processHost(in, out, 1); }) : FX_E_ARG; }
int fx_process_block(fx_handle* h, const float* in, float* out, int n) { return h ? guardCode(&h->batch, [&] { return h->batch.processHost(in, out, n); }) : FX_E_ARG; }
int fx_set_register(fx_handle* h, const char* key, float v) { return (h && key) ? guardCode(&h->batch, [&] { return h->batch.setRegister(key, v); }) : 1; }
float fx_get_register(fx_handle* h, const char* key) { return (h && key) ? guard(&h->batch, 1.0f, [&] { return h->batch.getRegisterAt(key, 0); }) : 1.0f; }
int64_t fx_instruction_counter(fx_handle* h) { return h ? guard(&h->batch, (int64_t)0, [&] { return h->batch.instructionCounterAt(0); }) : 0; }
int fx_error_count(fx_handle* h) { return h ? errorCount(h->batch) : 0; }
const char* fx_error_desc(fx_handle* h, int i) { return h ? errorDesc(h->batch, i) : ""; }
int fx_error_row(fx_handle* h, int i) { return h ? errorRow(h->batch, i) : -1; }
int fx_control_count(fx_handle* h) { return h ? controlCount(h->batch) : 0; }
const char* fx_control_at(fx_handle* h, int i) { return h ? controlAt(h->batch, i) : ""; }
int fx_meta_get(fx_handle* h, const char* key, char* buf, int buflen) { return h ? guard(&h->batch, 0, [&] { return metaGet(h->batch, key, buf, buflen); }) : 0; }
void fx_set_channels(fx_handle* h, int c) { if (h) h->batch.setChannels(c); }
int fx_set_option(fx_handle* h, unsigned option, int on) { return h ? h->batch.setOption(option, on != 0) : FX_E_ARG; }
int fx_get_channels(fx_handle* h) { return h ? h->batch.loaderChannels() : 0; }
int fx_ready(fx_handle* h) { return (h && h->batch.program().ready) ? 1 : 0; }
const char* fx_last_error(fx_handle* h) { return h ? h->batch.lastError().c_str() : "null handle"; }

#ifdef FX_DIAGNOSTICS
/* ---- diagnostics build only (fx_diag.h) ---- */
extern "C" int fxb_diag_build(void) { return 1; }
extern "C" int fxb_diag_read_end_stamps(void* handle, uint32_t* out, int64_t n_words) {
    fxb_handle* h = static_cast<fxb_handle*>(handle);
    if (!h || h->batch.shards() != 1) return FX_E_ARG;
    return guard(&h->batch.front(), FX_E_PROGRAM, [&] { return h->batch.front().readEndStamps(out, n_words); });
}
#endif

/* ---- batch ---- */
fxb_handle* fxb_create(int64_t n, int ch, int device) { return create<fxb_handle>(n, ch, std::vector<int>{device}); }
fxb_handle* fxb_create_on_devices(int64_t n, int ch, const int* devices, int n_devices) {
    if (!devices || n_devices < 1) { g_createError = "no device given"; return nullptr; }
    return create<fxb_handle>(n, ch, std::vector<int>(devices, devices + n_devices));
}
fxb_handle* fxb_create_sharded(int64_t n, int ch, uint64_t device_mask) {
    std::vector<int> devices;
    for (int d = 0; d < 64; ++d)
        if (device_mask & (1ull << d)) devices.push_back(d);
    if (devices.empty()) { g_createError = "empty device mask"; return nullptr; }
    return create<fxb_handle>(n, ch, devices);
}
int fxb_shard_count(fxb_handle* h) { return h ? h->batch.shards() : 0; }
int fxb_shard_info(fxb_handle* h, int shard, int* device, int64_t* first_instance, int64_t* n_instances) {
    if (!h || shard < 0 || shard >= h->batch.shards()) return FX_E_ARG;
    if (device) *device = h->batch.deviceOf(shard);
    if (first_instance) *first_instance = h->batch.firstOf(shard);
    if (n_instances) *n_instances = h->batch.countOf(shard);
    return 0;
}
float fxb_shard_kernel_ms(fxb_handle* h, int shard) {
    if (!h || shard < 0 || shard >= h->batch.shards()) return -1.0f;
    return guard(&h->batch.front(), -1.0f, [&] { return h->batch.lastKernelMsOf(shard); });
}
int fxb_shard_plan(int64_t n, int n_shards, int64_t* first_instance, int64_t* n_instances_out) {
    if (n_shards < 1 || !first_instance || !n_instances_out) return FX_E_ARG;
    try {
        const auto ranges = fx::Sharded::plan(n, n_shards);
        for (int k = 0; k < n_shards; ++k) { first_instance[k] = ranges[(size_t)k].first; n_instances_out[k] = ranges[(size_t)k].second; }
        return 0;
    } catch (...) {
        return FX_E_ARG;
    }
}
void fxb_destroy(fxb_handle* h) { delete h; }
int fxb_set_option(fxb_handle* h, unsigned option, int on) { return h ? h->batch.setOption(option, on != 0) : FX_E_ARG; }
int fxb_load_file(fxb_handle* h, const char* path) { return (h && path) ? guard(&h->batch.front(), 0, [&] { return h->batch.loadFile(path) ? 1 : 0; }) : 0; }
int fxb_load_text(fxb_handle* h, const char* text) { return (h && text) ? guard(&h->batch.front(), 0, [&] { return h->batch.loadText(text) ? 1 : 0; }) : 0; }
int fxb_set_register(fxb_handle* h, const char* key, float v) { return (h && key) ? guardCode(&h->batch.front(), [&] { return h->batch.setRegister(key, v); }) : 1; }
int fxb_set_register_i(fxb_handle* h, const char* key, int64_t inst, float v) { return (h && key) ? guardCode(&h->batch.front(), [&] { return h->batch.setRegisterAt(key, inst, v); }) : 1; }
float fxb_get_register_i(fxb_handle* h, const char* key, int64_t inst) { return (h && key) ? guard(&h->batch.front(), 1.0f, [&] { return h->batch.getRegisterAt(key, inst); }) : 1.0f; }
int fxb_set_register_array(fxb_handle* h, const char* key, const float* values) { return (h && key) ? guardCode(&h->batch.front(), [&] { return h->batch.setRegisterArray(key, values); }) : 1; }
int fxb_get_register_array(fxb_handle* h, const char* key, float* values) { return (h && key) ? guardCode(&h->batch.front(), [&] { return h->batch.getRegisterArray(key, values); }) : 1; }
int fxb_set_register_track(fxb_handle* h, const char* key, const float* values, int n_steps, int period, int per_instance) {
    return (h && key) ? guardCode(&h->batch.front(), [&] { return h->batch.setRegisterTrack(key, values, n_steps, period, per_instance != 0); }) : 1;
}
int fxb_seed_noise_i(fxb_handle* h, int64_t inst, int32_t x1, int32_t x2) { return h ? guardCode(&h->batch.front(), [&] { return h->batch.seedNoiseAt(inst, x1, x2); }) : FX_E_ARG; }
void* fxb_host_alloc(int64_t bytes) {
    if (bytes <= 0) { g_createError = "fxb_host_alloc: bytes must be positive"; return nullptr; }
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        g_createError = std::string("fxb_host_alloc: ") + hipGetErrorString(e);
        return nullptr;
    }
    return p;
}
void fxb_host_free(void* p) {
    if (p && hipHostFree(p) != hipSuccess) (void)hipGetLastError();
}
int fxb_process_block(fxb_handle* h, const float* in, float* out, int n) { return h ? guardCode(&h->batch.front(), [&] { return h->batch.processHost(in, out, n); }) : FX_E_ARG; }
int fxb_process_block_dev(fxb_handle* h, const float* d_in, float* d_out, int n, void* stream) {
    return h ? guardCode(&h->batch.front(), [&] { return h->batch.processDevice(d_in, d_out, n, static_cast<hipStream_t>(stream)); }) : FX_E_ARG;
}
int fxb_process_block_dev_shards(fxb_handle* h, const float* const* d_in, float* const* d_out, int n) {
    return h ? guardCode(&h->batch.front(), [&] { return h->batch.processDeviceShards(d_in, d_out, n); }) : FX_E_ARG;
}
int fxb_sync(fxb_handle* h) { return h ? guardCode(&h->batch.front(), [&] { return h->batch.sync(); }) : FX_E_ARG; }
int fxb_prepare(fxb_handle* h, int n_samples, int wait) { return h ? guard(&h->batch.front(), FX_E_PROGRAM, [&] { return h->batch.prepare(n_samples, wait != 0); }) : FX_E_ARG; }
int64_t fxb_state_size(fxb_handle* h) { return h ? guard(&h->batch.front(), (int64_t)FX_E_PROGRAM, [&] { return h->batch.stateBytes(); }) : FX_E_ARG; }
int fxb_save_state(fxb_handle* h, void* buf, int64_t cap) { return h ? guard(&h->batch.front(), FX_E_PROGRAM, [&] { return h->batch.saveState(buf, cap); }) : FX_E_ARG; }
int fxb_load_state(fxb_handle* h, const void* buf, int64_t bytes) { return h ? guard(&h->batch.front(), FX_E_PROGRAM, [&] { return h->batch.loadState(buf, bytes); }) : FX_E_ARG; }
int fxb_get_tram_i(fxb_handle* h, int which, int64_t inst, float* out, int n_slots) {
    return h ? guard(&h->batch.front(), FX_E_PROGRAM, [&] { return h->batch.getTramAt(which, inst, out, n_slots); }) : FX_E_ARG;
}
int fxb_get_cursors_i(fxb_handle* h, int64_t inst, int32_t* out4) {
    return h ? guard(&h->batch.front(), FX_E_PROGRAM, [&] { return h->batch.getCursorsAt(inst, out4); }) : FX_E_ARG;
}
int64_t fxb_instruction_counter(fxb_handle* h) { return h ? guard(&h->batch.front(), (int64_t)-1, [&] { return h->batch.instructionCounter(); }) : 0; }
int64_t fxb_instruction_counter_i(fxb_handle* h, int64_t inst) { return h ? guard(&h->batch.front(), (int64_t)0, [&] { return h->batch.instructionCounterAt(inst); }) : 0; }
uint32_t fxb_ood_flags(fxb_handle* h) { return h ? guard(&h->batch.front(), ~0u, [&] { return h->batch.oodFlags(); }) : 0; }
int fxb_error_count(fxb_handle* h) { return h ? errorCount(h->batch.front()) : 0; }
const char* fxb_error_desc(fxb_handle* h, int i) { return h ? errorDesc(h->batch.front(), i) : ""; }
int fxb_error_row(fxb_handle* h, int i) { return h ? errorRow(h->batch.front(), i) : -1; }
int fxb_control_count(fxb_handle* h) { return h ? controlCount(h->batch.front()) : 0; }
const char* fxb_control_at(fxb_handle* h, int i) { return h ? controlAt(h->batch.front(), i) : ""; }
int fxb_meta_get(fxb_handle* h, const char* key, char* buf, int buflen) { return h ? guard(&h->batch.front(), 0, [&] { return metaGet(h->batch.front(), key, buf, buflen); }) : 0; }
int fxb_ready(fxb_handle* h) { return (h && h->batch.front().program().ready) ? 1 : 0; }
const char* fxb_last_error(fxb_handle* h) { return h ? h->batch.lastError().c_str() : "null handle"; }
float fxb_last_kernel_ms(fxb_handle* h) { return h ? guard(&h->batch.front(), -1.0f, [&] { return h->batch.lastKernelMs(); }) : -1.0f; }
int fxb_tier_note(fxb_handle* h, char* buf, int buflen) {
    if (!h) return FX_E_ARG;
    return guard(&h->batch.front(), (int)FX_E_PROGRAM, [&] {
        const std::string note = h->batch.front().tierNote();
        if (buf && buflen > 0) std::snprintf(buf, (size_t)buflen, "%s", note.c_str());
        return (int)note.size();
    });
}
int64_t fxb_info(fxb_handle* h, int what) { return h ? guard(&h->batch.front(), (int64_t)-1, [&] { return h->batch.info(what); }) : -1; }


/* ---- front-end only ---- */
fxp_handle* fxp_create(int ch) { return create<fxp_handle>(ch); }
void fxp_destroy(fxp_handle* h) { delete h; }
int fxp_set_option(fxp_handle* h, unsigned option, int on) {
    if (!h || (option & ~fx::kOptAll)) return FX_E_ARG;
    h->prog.options = on ? (h->prog.options | option) : (h->prog.options & ~option);
    return 0;
}
int fxp_load_file(fxp_handle* h, const char* path) { if (!h || !path) return 0; h->lowered = false; return guard(h, 0, [&] { return h->prog.loadFile(path) ? 1 : 0; }); }
int fxp_load_text(fxp_handle* h, const char* text) { if (!h || !text) return 0; h->lowered = false; return guard(h, 0, [&] { return h->prog.loadText(text) ? 1 : 0; }); }
int fxp_num_registers(fxp_handle* h) { return h ? (int)h->prog.regs.size() : 0; }
const char* fxp_register_name(fxp_handle* h, int i) { return (h && i >= 0 && i < (int)h->prog.regs.size()) ? h->prog.regs[i].name.c_str() : ""; }
int fxp_register_type(fxp_handle* h, int i) { return (h && i >= 0 && i < (int)h->prog.regs.size()) ? h->prog.regs[i].type : -1; }
int fxp_register_ioindex(fxp_handle* h, int i) { return (h && i >= 0 && i < (int)h->prog.regs.size()) ? h->prog.regs[i].io : -1; }
float fxp_register_value(fxp_handle* h, int i) { return (h && i >= 0 && i < (int)h->prog.regs.size()) ? h->prog.regs[i].value : 0.0f; }
int fxp_num_instructions(fxp_handle* h) { return h ? (int)h->prog.instrs.size() : 0; }
void fxp_instruction(fxp_handle* h, int i, int o[8]) {
    if (!h || !o || i < 0 || i >= (int)h->prog.instrs.size()) return;
    const fx::Instr& I = h->prog.instrs[i];
    o[0] = I.op; o[1] = I.r; o[2] = I.a; o[3] = I.x; o[4] = I.y; o[5] = I.hasInput; o[6] = I.hasOutput; o[7] = I.hasNoise;
}
int fxp_itram_size(fxp_handle* h) { return h ? h->prog.iTramSize : 0; }
int fxp_xtram_size(fxp_handle* h) { return h ? h->prog.xTramSize : 0; }
int fxp_error_count(fxp_handle* h) { return h ? (int)h->prog.errors.size() : 0; }
const char* fxp_error_desc(fxp_handle* h, int i) { return (h && i >= 0 && i < (int)h->prog.errors.size()) ? h->prog.errors[i].description.c_str() : ""; }
int fxp_error_row(fxp_handle* h, int i) { return (h && i >= 0 && i < (int)h->prog.errors.size()) ? h->prog.errors[i].row : -1; }
int fxp_control_count(fxp_handle* h) { return h ? (int)h->prog.controls.size() : 0; }
const char* fxp_control_at(fxp_handle* h, int i) { return (h && i >= 0 && i < (int)h->prog.controls.size()) ? h->prog.controls[i].c_str() : ""; }
int fxp_meta_get(fxp_handle* h, const char* key, char* buf, int buflen) {
    if (!h || !key) return 0;
    for (const auto& kv : h->prog.meta)
        if (kv.first == key) { if (buf && buflen > 0) std::snprintf(buf, (size_t)buflen, "%s", kv.second.c_str()); return 1; }
    return 0;
}
int fxp_ready(fxp_handle* h) { return (h && h->prog.ready) ? 1 : 0; }
const double* fxp_lut(int kind, int exponent) {
    static const fx::Luts luts;
    return kind ? luts.exp_[exponent & 31] : luts.log_[exponent & 31];
}
static int lowerImpl(fxp_handle* h);
int fxp_lower(fxp_handle* h) {
    if (!h) return FX_E_ARG;
    return guardCode(h, [&] { return lowerImpl(h); });
}
static int lowerImpl(fxp_handle* h) {
    if (!h->prog.ready) { h->err = "no program loaded"; return FX_E_NOTREADY; }
    std::vector<float> values(h->prog.regs.size());
    for (size_t r = 0; r < values.size(); ++r) values[r] = h->prog.regs[r].value;
    h->low = fx::lowerProgram(h->prog, values, std::vector<uint8_t>(values.size(), 0), 1);
    h->lowered = h->low.error.empty();
    h->err = h->low.error;
    return h->lowered ? 0 : FX_E_PROGRAM;
}
int64_t fxp_lower_info(fxp_handle* h, int what) {
    if (!h || !h->lowered) return -1;
    switch (what) {
        case FXB_INFO_NUM_INSTRUCTIONS: return (int64_t)h->prog.instrs.size();
        case FXB_INFO_NUM_REGISTERS: return (int64_t)h->prog.regs.size();
        case FXB_INFO_NUM_LANE_REGS: return h->low.nLaneRegs;
        case FXB_INFO_NUM_UNIFORM_REGS: return h->low.nUniformRegs;
        case FXB_INFO_LDS_BYTES_PER_WG: return (int64_t)h->low.nRows * 256;
        case FXB_INFO_NUM_MICROOPS: return (int64_t)h->low.steady.size();
        case FXB_INFO_ITRAM_SLOTS: return h->low.iSlots;
        case FXB_INFO_XTRAM_SLOTS: return h->low.xSlots;
        case FXB_INFO_TRAM_OPS: return h->low.tramOpsPerSample;
        case FXB_INFO_MULTIPASS: return h->low.multipass ? 1 : 0;
        case FXB_INFO_NUM_SHADOWED: return h->low.nShadowed;
        case FXB_INFO_NUM_CCR_LIVE: return h->low.nCcrLive;
        default: return -1;
    }
}
static int64_t translateImpl(fxp_handle* h, int vgprs, int stream, void* code, int64_t cap, char* listing, int64_t listing_cap, int stages, int stage,
                             int* stagesOut, int* info, int infoCap);
int fxp_track_register(fxp_handle* h, const char* key) {
    if (!h || !key) return FX_E_ARG;
    const int r = h->prog.findRegister(key);
    if (r < 0) return 1;
    if (std::find(h->tracked.begin(), h->tracked.end(), r) != h->tracked.end()) return 0;
    if (h->tracked.size() >= (size_t)fx::kMaxTracks) { h->err = "at most " + std::to_string(fx::kMaxTracks) + " registers can have schedules"; return FX_E_ARG; }
    h->tracked.push_back(r);
    return 0;
}
int64_t fxp_translate(fxp_handle* h, int vgprs, int stream, void* code, int64_t cap, char* listing, int64_t listing_cap) {
    if (!h) return FX_E_ARG;
    try {
        return translateImpl(h, vgprs, stream, code, cap, listing, listing_cap, 1, 0, nullptr, nullptr, 0);
    } catch (const std::exception& e) {
        h->err = e.what();
        return codeOf(e);
    } catch (...) {
        h->err = "unknown error";
        return FX_E_PROGRAM;
    }
}
static int64_t translateImpl(fxp_handle* h, int vgprs, int stream, void* code, int64_t cap, char* listing, int64_t listing_cap, int stages, int stage,
                             int* stagesOut, int* info, int infoCap) {
    if (!h->prog.ready) { h->err = "no program loaded"; return FX_E_NOTREADY; }
    if (listing && listing_cap > 0) listing[0] = 0;
    std::vector<float> values(h->prog.regs.size());
    for (size_t r = 0; r < values.size(); ++r) values[r] = h->prog.regs[r].value;
    // the lowering of the VGPR builds: one instance per lane, bookkeeping in VGPRs, rows are plain indices
    std::vector<uint8_t> perLane(values.size(), 0);
    for (int r : h->tracked) perLane[(size_t)r] = 1;   // (a register with a schedule has a row of its own, as in Batch::laneForced)
    fx::Lowered low = fx::lowerProgram(h->prog, values, perLane, 1, false, 1);
    if (!low.error.empty()) { h->err = low.error; return FX_E_PROGRAM; }
    std::string why;
    if (!fx::asmEligible(low, &why)) { h->err = "not eligible: " + why; return FX_E_PROGRAM; }
    if (low.multipass) { h->err = "not eligible: END can be skipped (multi-pass program: the interpreter runs the passes)"; return FX_E_PROGRAM; }
    int v = fx::ASM_V64;
    while (v < fx::ASM_V256 && low.nRows > fx::kAsmVgprRows[v]) ++v;
    if (vgprs != 0) {
        int want = -1;
        for (int q = fx::ASM_V64; q < fx::ASM_VARIANTS; ++q)
            if (fx::kAsmVgprRows[q] + 32 == vgprs) want = q;
        if (want < v) { h->err = "no such VGPR build, or too small for the program"; return FX_E_ARG; }
        v = want;
    }
    const fx::XlateTemplate* tmpl = fx::xlateTemplate((fx::AsmVariant)v, &h->err);
    if (!tmpl) return FX_E_PROGRAM;
    if (stream < 0 || stream > 4) { h->err = "stream: 0 steady fast, 1 steady exact, 2 last fast, 3 last exact, 4 run-once"; return FX_E_ARG; }
    std::vector<uint32_t> code4[5];
    std::string text4[5];
    fx::XlateImage plan;
    const std::vector<fx::MicroOp> steadyRecords = fx::encodeAsmStream(low.steady, nullptr, true), lastRecords = fx::encodeAsmStream(low.last, nullptr, true);
    std::vector<int> trackRows;
    for (int r : h->tracked) trackRows.push_back(low.rowOfReg[(size_t)r]);
    fx::XlateProgram xprog = fx::xlateProgramOf(steadyRecords, lastRecords, h->prog.iTramSize, h->prog.xTramSize, low.nRows, low.inRow, low.latchRow, trackRows);
    if (const int prio = fx::ReleaseKnobs::fromEnvironment().xlatePrio; prio >= 0) xprog.prioritySlices = prio != 0;   // (what a batch with that knob generates: tests)
    std::vector<std::vector<uint32_t>> stagedCode;
    std::vector<std::string> stagedText;
    if (stagesOut) *stagesOut = 1;
    bool staged = false;
    if (stages >= 2) {
        const fx::StagePlan sp = fx::planStages(steadyRecords, lastRecords, xprog, low.nRows, stages);
        if (!sp.cuts.empty()) {
            if (!fx::buildStagedImage(steadyRecords, lastRecords, *tmpl, xprog, sp, &plan, &stagedCode, &stagedText, &h->err, 144u * 1024u, fx::kStageGroupMax,
                                      fx::ReleaseKnobs::fromEnvironment().stagesGroup)) return FX_E_PROGRAM;
            staged = true;
            if (stagesOut) *stagesOut = plan.stages;
            // info: per cut its record index and the number of rows handed over, then the LDS bytes of a workgroup
            int q = 0;
            for (size_t c = 0; c < sp.cuts.size() && info && q + 2 <= infoCap; ++c) { info[q++] = sp.cuts[c]; info[q++] = (int)sp.live[c].size(); }
            if (info && q < infoCap) info[q++] = (int)plan.ldsBytes;
            // ... then the number of register-file rows and the stage that stores each of them
            if (info && q < infoCap) info[q++] = (int)sp.storeStage.size();
            for (size_t r = 0; r < sp.storeStage.size() && info && q < infoCap; ++r) info[q++] = sp.storeStage[r];
        } else {
            h->err = "not cut into stages: " + sp.why;
        }
    }
    if (staged && (stage < 0 || stage >= plan.stages)) { h->err = "no such stage"; return FX_E_ARG; }
    if (!staged && !fx::planXlate(steadyRecords, lastRecords, *tmpl, xprog, &plan, code4, text4, &h->err)) return FX_E_PROGRAM;
    const std::vector<uint32_t>& words = staged ? stagedCode[stream == 4 ? (size_t)plan.stages * 4 : (size_t)stage * 4 + (size_t)stream] : code4[stream];
    const std::string& text = staged ? stagedText[stream == 4 ? (size_t)plan.stages * 4 : (size_t)stage * 4 + (size_t)stream] : text4[stream];
    const int64_t bytes = (int64_t)words.size() * 4;
    if (code && cap > 0 && bytes > 0) std::memcpy(code, words.data(), (size_t)std::min<int64_t>(cap, bytes));   // (an empty stream has no data())
    if (listing && listing_cap > 0) {
        const size_t n = std::min<size_t>(text.size(), (size_t)listing_cap - 1);
        std::memcpy(listing, text.data(), n);
        listing[n] = 0;
    }
    return bytes;
}
int64_t fxp_translate_staged(fxp_handle* h, int vgprs, int stages, int stage, int stream, void* code, int64_t cap, char* listing, int64_t listing_cap,
                             int* stages_out, int* info, int info_cap) {
    if (!h) return FX_E_ARG;
    try {
        return translateImpl(h, vgprs, stream, code, cap, listing, listing_cap, stages, stage, stages_out, info, info_cap);
    } catch (const std::exception& e) {
        h->err = e.what();
        return codeOf(e);
    } catch (...) {
        h->err = "unknown error";
        return FX_E_PROGRAM;
    }
}
int64_t fxp_code_hash(fxp_handle* h, int vgprs, int stages, unsigned flags) {
    if (!h) return FX_E_ARG;
    try {
        if (!h->prog.ready) { h->err = "no program loaded"; return FX_E_NOTREADY; }
        std::vector<float> values(h->prog.regs.size());
        for (size_t r = 0; r < values.size(); ++r) values[r] = h->prog.regs[r].value;
        std::vector<uint8_t> perLane(values.size(), 0);
        for (int r : h->tracked) perLane[(size_t)r] = 1;
        fx::Lowered low = fx::lowerProgram(h->prog, values, perLane, 1, false, 1);
        if (!low.error.empty()) { h->err = low.error; return FX_E_PROGRAM; }
        std::string why;
        if (!fx::asmEligible(low, &why)) { h->err = "not eligible: " + why; return FX_E_PROGRAM; }
        if (low.multipass) { h->err = "not eligible: END can be skipped (multi-pass program: the interpreter runs the passes)"; return FX_E_PROGRAM; }
        int want = -1;
        for (int q = fx::ASM_V64; q < fx::ASM_VARIANTS; ++q)
            if (fx::kAsmVgprRows[q] + 32 == vgprs && low.nRows <= fx::kAsmVgprRows[q]) want = q;
        if (want < 0) { h->err = "no such VGPR build, or too small for the program"; return FX_E_ARG; }
        const fx::XlateTemplate* tmpl = fx::xlateTemplate((fx::AsmVariant)want, &h->err);
        if (!tmpl) return FX_E_PROGRAM;
        const std::vector<fx::MicroOp> steadyRecords = fx::encodeAsmStream(low.steady, nullptr, true), lastRecords = fx::encodeAsmStream(low.last, nullptr, true);
        std::vector<int> trackRows;
        for (int r : h->tracked) trackRows.push_back(low.rowOfReg[(size_t)r]);
        fx::XlateProgram xprog = fx::xlateProgramOf(steadyRecords, lastRecords, h->prog.iTramSize, h->prog.xTramSize, low.nRows, low.inRow, low.latchRow, trackRows);
        xprog.tramStreaming = (flags & 1u) != 0;
        xprog.prioritySlices = (flags & 2u) != 0 && stages < 2;
        fx::XlateImage image;
        bool built = false;
        if (stages >= 2) {
            const fx::StagePlan sp = fx::planStages(steadyRecords, lastRecords, xprog, low.nRows, stages);
            if (!sp.cuts.empty()) built = fx::buildStagedImage(steadyRecords, lastRecords, *tmpl, xprog, sp, &image, nullptr, nullptr, &h->err, 144u * 1024u, fx::kStageGroupMax,
                                                              fx::ReleaseKnobs::fromEnvironment().stagesGroup);
            if (!built) image = fx::XlateImage();
        }
        if (!built && !fx::buildXlateImage(steadyRecords, lastRecords, *tmpl, xprog, &image, &h->err)) return FX_E_PROGRAM;
        return (int64_t)fx::imageHash(image);
    } catch (const std::exception& e) {
        h->err = e.what();
        return codeOf(e);
    } catch (...) {
        h->err = "unknown error";
        return FX_E_PROGRAM;
    }
}
const char* fxp_last_error(fxp_handle* h) { return h ? h->err.c_str() : "null handle"; }

}  // extern "C"
