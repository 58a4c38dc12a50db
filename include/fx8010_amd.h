/* fx8010_amd.h — C ABI of the MI355X-native batch FX8010 interpreter (libfx8010_amd.so).
 *
 * This is the drop-in boundary for the one hot path this project accelerates: the
 * per-sample instruction loop Klangraum::FX8010::process() of easypx/FX8010-Emulator-Core
 * (reference: source/FX8010.cpp:1023-1249), together with the cold front-end it needs
 * (loadFile, the by-name register API).  The reference has no FFI of its own — its boundary
 * is the public surface of class Klangraum::FX8010 (include/FX8010.h:47-75) — so every entry
 * point below names the class member it replaces.  A header-only C++ class with the
 * reference's exact member names sits on top of this ABI:
 *   fx8010-emulator-core_amd/host/FX8010.h.
 *
 * Two families:
 *   fx_*   one emulated DSP, same call-per-sample convention as the reference class;
 *   fxb_*  N independent DSPs stepping one program in SIMT lockstep on one GPU
 *          ("batch"); fxb_process_block(S) ≡ calling process() S times on each of N objects.
 *
 * All compute runs in a hand-written HIP kernel on gfx950.  There is NO CPU fallback:
 * when no HIP device is usable, fx_create/fxb_create return NULL and fx_last_create_error()
 * says why; a failed launch returns a negative code and fxb_last_error() the text.
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer it passes; the
 * library owns the handle and all device state; a handle is not thread-safe - one thread at a
 * time, as with the reference's objects (include/FX8010.h:162-217: plain members, no lock); calls
 * that reach a multi-device handle from two threads are serialised, not made independent - and
 * distinct handles are independent.  A handle may own a builder thread of its own (code generated
 * ahead of time, FXB_INFO_XLATE_BACKGROUND_BUILDS); the environment's FX_* knobs are read when
 * code is generated: do not change the environment while a handle exists.  Return codes: the reference's own where one exists (noted per
 * function), otherwise 0 = ok and <0 = FX_E_*.
 */
#ifndef FX8010_AMD_H
#define FX8010_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FX_E_NODEVICE (-1) /* no usable HIP device / HIP call failed          */
#define FX_E_NOTREADY (-2) /* no program loaded (reference: getReadyStatus()) */
#define FX_E_ARG (-3)      /* bad argument                                    */
#define FX_E_PROGRAM (-4)  /* program cannot be lowered for the device (see fxb_last_error) */
#define FX_E_MEMORY (-5)   /* device allocation failed                        */

/* Options: behaviour BEYOND the reference, off by default - with every option off the library is the reference, bit for bit.
 * Set before loading a program (fx_set_option / fxb_set_option / fxp_set_option; 0 or FX_E_ARG).
 *
 * FX_OPT_TRAM_DANE  the delay-line model the reference's own design note asks for but does not implement
 *   (docs/TRAM Registermapping.pdf p.1-3; the reference advances a cursor per executed instruction and reads
 *   buf[(rpos - p) % size] with C's remainder, source/FX8010.cpp:909-967, so only offset 0 is defined):
 *     - one address counter per TRAM that steps DOWN once per sample period; a tap, read or write, addresses
 *       (counter + position) mod size (true ring): a value written at position pw is read pr - pw samples later at pr -
 *       any number of taps per delay line, the kX / DANE convention ("idelay write wrt at 0; idelay read rd at 1439");
 *     - the position of "idelay read, rd, at, 17" lives in a register of its own, "&rd" (the PDF's GPR "&rd", created with
 *       the literal as its value), which instructions may name as an operand and write: modulated delay lines;
 *     - no FXO_OOD flag for offsets (there is nothing out of the domain in a ring).
 * FX_OPT_TRAM_ADDR_SHIFT  position registers hold DANE addresses, the "Address-Shifting (0x800)" of the PDF: the FX8010
 *   keeps an address in a register as a 32-bit fixed-point fraction with 0x800 units per sample, so
 *   position = floatToInt(value) >> 11 = floor(value * 2^20) (floatToInt as source/FX8010.cpp:1016-1020), and a tap declared
 *   "at 1439" starts with &rd = 1439 * 2^-20.  That keeps addresses inside [-1, 1): MACS / MACSN / INTERP can compute
 *   them (kX's "macs t, &wrt1, rd_max_shifted, sin_abs"); the low 11 bits (the interpolation fraction) are dropped.
 * Implemented as generated code by the translated tier (taps whose position is the same in every instance), by the HIP C++
 * kernel tier (FXB_INFO_KERNEL 0; also taps with a per-instance position) and, as the checker, by oracle/ (FXO_OPT_*). */
#define FX_OPT_TRAM_DANE (1u << 0)
#define FX_OPT_TRAM_ADDR_SHIFT (1u << 1)
/* FX_OPT_TRAM_INTERP  (with FX_OPT_TRAM_ADDR_SHIFT) the interpolated read the reference's comment asks for ("To do linear
 *   Interpolation, you need to find the fractional part of the read position and interpolate between the two adjacent
 *   samples", source/FX8010.cpp:929-932; ":1192 (Y-2048) mit 11 Bit Shift"): a READ tap at DANE address a = floatToInt(value)
 *   returns x0 + f * (x1 - x0), x0 at position a >> 11, x1 at the position after it, f = (a & 0x7ff) / 2048 - four fp32
 *   operations in that order; f == 0 returns x0 itself.  Writes ignore the fraction. */
#define FX_OPT_TRAM_INTERP (1u << 2)

typedef struct fx_handle fx_handle;   /* one emulated DSP  */
typedef struct fxb_handle fxb_handle; /* a batch of N DSPs */

/* ------------------------------------------------------------------ single instance
 * Mirrors Klangraum::FX8010 one to one (batch of 1 on the current HIP device). */

/* FX8010::FX8010(int numChannels)            include/FX8010.h:51, source/FX8010.cpp:9-12 */
fx_handle* fx_create(int num_channels);
/* FX8010::~FX8010()                          include/FX8010.h:52 */
void fx_destroy(fx_handle* h);
/* bool loadFile(const string& path)          include/FX8010.h:62, source/FX8010.cpp:777-875
 * returns 1 = loaded, 0 = open failure or syntax errors (see fx_error_*). */
int fx_load_file(fx_handle* h, const char* path);
/* vector<float> process(const vector<float>&) include/FX8010.h:57, source/FX8010.cpp:1023-1249
 * in[num_channels] -> out[num_channels], one sample period.  0 or FX_E_*. */
int fx_process(fx_handle* h, const float* in, float* out);
/* S consecutive process() calls in one launch; in/out are [S][num_channels]. */
int fx_process_block(fx_handle* h, const float* in, float* out, int n_samples);
/* int setRegisterValue(const string&, float)  source/FX8010.cpp:236-253: 0 = found, 1 = not found */
int fx_set_register(fx_handle* h, const char* key, float value);
/* float getRegisterValue(const string&)       source/FX8010.cpp:256-266: 1.0f when not found */
float fx_get_register(fx_handle* h, const char* key);
/* int getInstructionCounter()                 source/FX8010.cpp:986-989 (64-bit here) */
int64_t fx_instruction_counter(fx_handle* h);
/* vector<MyError> getErrorList()              include/FX8010.h:63-68, source/FX8010.cpp:894-897
 * entry 0 is always {"Kein Fehler", 1}. */
int fx_error_count(fx_handle* h);
const char* fx_error_desc(fx_handle* h, int i);
int fx_error_row(fx_handle* h, int i);
/* vector<string> getControlRegisters()        source/FX8010.cpp:201-204 */
int fx_control_count(fx_handle* h);
const char* fx_control_at(fx_handle* h, int i);
/* unordered_map<string,string> getMetaData()  source/FX8010.cpp:1003-1006
 * keys name, copyright, created, engine, comment, guid; 1 = present, 0 = absent */
int fx_meta_get(fx_handle* h, const char* key, char* buf, int buflen);
/* setChannels / getChannels / getReadyStatus  include/FX8010.h:73-75 */
void fx_set_channels(fx_handle* h, int num_channels);
int fx_set_option(fx_handle* h, unsigned option, int on);
int fx_get_channels(fx_handle* h);
int fx_ready(fx_handle* h);
const char* fx_last_error(fx_handle* h);
/* why the last fx_create / fxb_create returned NULL (thread-local string) */
const char* fx_last_create_error(void);

/* ------------------------------------------------------------------ batch (the GPU path)
 * N independent instances of one program; instance n of sample s, channel c lives at
 * buf[(s * num_channels + c) * N + n]  — instance-fastest, so a wavefront (64 consecutive
 * instances) reads and writes 256 contiguous bytes. */

/* device: HIP ordinal, or -1 for the calling thread's current device. */
fxb_handle* fxb_create(int64_t n_instances, int num_channels, int device);
/* The same batch spread over several GPUs of one node (SURVEY.md section 8 b/e; the reference's README.md:13 "emulate
 * much more DSP's" on many cores): contiguous instance ranges, whole wavefronts per shard, one host thread + one HIP
 * stream per shard inside the library, program / tables / broadcast controls replicated, NO exchange between shards.
 * Every fxb_* call works on such a handle unchanged (instances keep their global numbers; host PCM buffers keep the
 * [sample][channel][all instances] layout and each shard copies its own columns); device-resident PCM is per device and
 * goes through fxb_process_block_dev_shards.
 *   fxb_create_sharded     one shard per set bit of device_mask (bit d = HIP ordinal d), in ordinal order
 *   fxb_create_on_devices  one shard per list entry; an ordinal may repeat (several shards on one GPU) */
fxb_handle* fxb_create_sharded(int64_t n_instances, int num_channels, uint64_t device_mask);
fxb_handle* fxb_create_on_devices(int64_t n_instances, int num_channels, const int* devices, int n_devices);
int fxb_shard_count(fxb_handle* h);
/* device ordinal, first global instance and instance count of a shard; 0 or FX_E_ARG.  Any out pointer may be NULL. */
int fxb_shard_info(fxb_handle* h, int shard, int* device, int64_t* first_instance, int64_t* n_instances);
/* HIP-event time of shard `shard`'s most recent launch in ms (fxb_last_kernel_ms is the slowest shard's); -1 when unknown.  A
 * scaling run reports every device with it ("report per-device times too"). */
float fxb_shard_kernel_ms(fxb_handle* h, int shard);
/* The partition fxb_create_sharded / fxb_create_on_devices use for n_instances over n_shards devices, without creating
 * anything (no device needed): first_instance[k], n_instances[k] for k < n_shards.  0, or FX_E_ARG when a shard would be
 * empty (fewer wavefronts than shards) or an argument is invalid. */
int fxb_shard_plan(int64_t n_instances, int n_shards, int64_t* first_instance, int64_t* n_instances_out);
void fxb_destroy(fxb_handle* h);
int fxb_set_option(fxb_handle* h, unsigned option, int on);   /* FX_OPT_*: before loading */
/* as fx_load_file; the program is parsed once on the host and lowered to the device
 * opcode stream.  fxb_load_text takes the program text itself. */
int fxb_load_file(fxb_handle* h, const char* path);
int fxb_load_text(fxb_handle* h, const char* text);
/* setRegisterValue on every instance / on one instance (0 found, 1 not found, <0 FX_E_*) */
int fxb_set_register(fxb_handle* h, const char* key, float value);
int fxb_set_register_i(fxb_handle* h, const char* key, int64_t instance, float value);
/* getRegisterValue of one instance (1.0f when not found) */
float fxb_get_register_i(fxb_handle* h, const char* key, int64_t instance);
/* setRegisterValue / getRegisterValue of every instance at once: values[n_instances], one per instance - what N
 * callers of the reference's setRegisterValue would do before a block (per-instance control automation).
 * 0 found, 1 not found, <0 FX_E_*.  One host-to-device copy; the register becomes per-instance. */
int fxb_set_register_array(fxb_handle* h, const char* key, const float* values);
int fxb_get_register_array(fxb_handle* h, const char* key, float* values);
/* Control track: a schedule of values for register `key` that the NEXT fxb_process_block* call applies by itself - at
 * sample s of that block, whenever s is a multiple of `period`, the register takes values[s / period] (per_instance:
 * values[(s / period) * n_instances + instance]) - exactly what a caller of the reference does with setRegisterValue()
 * between process() calls (the slider every 8 samples of source/main.cpp:107-114), in ONE launch instead of one per
 * change.  At most 16 registers can have tracks (one sorted list of events per block: the loop pays one compare per sample whatever their number); steps beyond the block are dropped; the register keeps its last value.
 * The translated program reads the schedule from device memory (no re-translation per schedule); the interpreter and
 * HIP C++ tiers cut the block at the change points.  0 found, 1 not found, <0 FX_E_*. */
int fxb_set_register_track(fxb_handle* h, const char* key, const float* values, int n_steps, int period, int per_instance);
/* white-noise generator seeds of one instance (reference: g_x1/g_x2, include/FX8010.h:290-291;
 * every instance starts with the reference's seeds) */
int fxb_seed_noise_i(fxb_handle* h, int64_t instance, int32_t x1, int32_t x2);
/* Generate the code blocks of n_samples samples will run, now: the translation (4-9 ms) otherwise falls into the first
 * fxb_process_block* call after a load.  wait != 0: also until the handle's builder thread has finished what follows a first
 * build (the variant with the declared controls in rows; stage counts on trial; the variant for the controls that rest, which
 * is then in force on return: FXB_INFO_CONTROL_ROWS).  For callers with a deadline per block (the
 * reference's caller has 667 us, include/FX8010.h:38): load, prepare, then start the stream.  0 or FX_E_*. */
int fxb_prepare(fxb_handle* h, int n_samples, int wait);
/* State snapshot.  The reference keeps all DSP state in plain members (include/FX8010.h:162-217, 288-291: register values, output
 * latches, smallDelayBuffer / largeDelayBuffer and their four positions, the LFSR words, the instruction counter); a batch's
 * state is the same per instance.  fxb_save_state writes fxb_state_size(h) bytes: a 64-byte header, then - by GLOBAL instance -
 * the state rows [row][instance], iTRAM [instance][slot] and xTRAM [instance][slot] (slots: as many as the program can reach).
 * fxb_load_state takes such an image into a batch of the same instance count with the same program loaded, whatever its
 * partition into shards (a sharded handle can be re-partitioned: save, destroy, create on other devices, load); the registers'
 * host side follows the image (a register every instance holds one value of counts as a broadcast write of that value).
 * Synchronous; 0 or FX_E_*.  Sizes: config5's 262 144 instances carry 8 GiB of xTRAM. */
int64_t fxb_state_size(fxb_handle* h);
int fxb_save_state(fxb_handle* h, void* buf, int64_t cap);
int fxb_load_state(fxb_handle* h, const void* buf, int64_t bytes);
/* one instance's delay memory as the reference holds it (which: 0 = smallDelayBuffer / iTRAM, 1 = largeDelayBuffer / xTRAM; the first
 * n_slots words; words the program cannot reach read 0) and its positions {iTRAM write, iTRAM read, xTRAM write, xTRAM read}
 * (reference smallDelayWritePos ... largeDelayReadPos, include/FX8010.h:214-217) */
int fxb_get_tram_i(fxb_handle* h, int which, int64_t instance, float* out, int n_slots);
int fxb_get_cursors_i(fxb_handle* h, int64_t instance, int32_t* out4);
/* S sample periods for all N instances.  Host buffers: synchronous (returns with `out` filled).  Caller buffers in PINNED host
 * memory (fxb_host_alloc below, hipHostMalloc / hipHostRegister, a torch pinned tensor) are processed IN PLACE: the kernel reads
 * and writes them over PCIe, no staging copies, one launch - what a real-time host wants (32-sample blocks of the 512-instruction
 * reverb: 163 840 instances inside 666.667 us; tools/realtime_capacity.py).  Pageable buffers: blocks of a few KB go through
 * pinned memory of the library, blocks of >= 32 MB are copied in, processed and copied out in overlapping pieces, everything else
 * is H2D, kernel, D2H in sequence.  `in` and `out` may be the same buffer; buffers that overlap in any other way, and buffers only
 * part of which is pinned, take the staged copies (the whole input is read before the first output is written; a buffer that
 * straddles the end of a hipHostRegister range is refused by the runtime's copy itself: FX_E_NODEVICE with its message;
 * the handle stays usable). */
int fxb_process_block(fxb_handle* h, const float* in, float* out, int n_samples);
/* Pinned, device-visible host memory for PCM buffers - for hosts that do not link the HIP runtime themselves (the reference's
 * callers keep their audio in plain vectors: include/FX8010.h:57; such a buffer is what to copy it into once per block).
 * NULL when the allocation fails (fx_last_create_error says why).  Free with fxb_host_free; both are thread-safe. */
void* fxb_host_alloc(int64_t bytes);
void fxb_host_free(void* p);
/* Same with device-resident buffers (hipMalloc'ed, on h's device); asynchronous on `stream`
 * (a hipStream_t, NULL = the handle's own stream).  Pair with fxb_sync(). */
int fxb_process_block_dev(fxb_handle* h, const float* d_in, float* d_out, int n_samples, void* stream);
/* Sharded batches: d_in[k] / d_out[k] are shard k's buffers on shard k's device, [n_samples][num_channels][n_instances of
 * the shard]; launched concurrently from the shards' own threads on their own streams.  Pair with fxb_sync(). */
int fxb_process_block_dev_shards(fxb_handle* h, const float* const* d_in, float* const* d_out, int n_samples);
int fxb_sync(fxb_handle* h);
/* executed instructions (reference counting: END and SKIP count, skipped ones do not):
 * summed over all instances / of one instance */
int64_t fxb_instruction_counter(fxb_handle* h);
int64_t fxb_instruction_counter_i(fxb_handle* h, int64_t instance);
/* OR of the per-instance "outside the parity domain" flags (0 = the whole batch stayed
 * inside the domain where the reference's behaviour is defined) */
uint32_t fxb_ood_flags(fxb_handle* h);
/* front-end results, as fx_* */
int fxb_error_count(fxb_handle* h);
const char* fxb_error_desc(fxb_handle* h, int i);
int fxb_error_row(fxb_handle* h, int i);
int fxb_control_count(fxb_handle* h);
const char* fxb_control_at(fxb_handle* h, int i);
int fxb_meta_get(fxb_handle* h, const char* key, char* buf, int buflen);
int fxb_ready(fxb_handle* h);
const char* fxb_last_error(fxb_handle* h);
/* HIP-event duration (ms) of the most recent interpreter-kernel launch, measured on the
 * stream it ran on; <0 if none.  Implies a sync on that stream. */
float fxb_last_kernel_ms(fxb_handle* h);

/* introspection of the lowered program (what the kernel actually runs) */
enum {
    FXB_INFO_NUM_INSTRUCTIONS = 0, /* reference instruction count, END included           */
    FXB_INFO_NUM_REGISTERS = 1,    /* reference register count                            */
    FXB_INFO_NUM_LANE_REGS = 2,    /* registers kept per instance (LDS rows)              */
    FXB_INFO_NUM_UNIFORM_REGS = 3, /* registers folded into the opcode stream             */
    FXB_INFO_LDS_BYTES_PER_WG = 4,
    FXB_INFO_WAVES_PER_WG = 5,
    FXB_INFO_NUM_MICROOPS = 6,     /* records in the device opcode stream                 */
    FXB_INFO_ITRAM_SLOTS = 7,      /* allocated slots per instance                        */
    FXB_INFO_XTRAM_SLOTS = 8,
    FXB_INFO_TRAM_OPS = 9,         /* delay reads+writes per sample (static)              */
    FXB_INFO_MULTIPASS = 10,       /* 1 if END can be skipped (generic pass loop in use)  */
    FXB_INFO_NUM_SHADOWED = 11,    /* instructions that can sit in a SKIP shadow          */
    FXB_INFO_NUM_CCR_LIVE = 12,    /* instructions whose CCR write is observable          */
    FXB_INFO_DEVICE = 13,
    FXB_INFO_GRID = 14,            /* workgroups of the last launch                       */
    FXB_INFO_INST_PER_LANE = 15,   /* instances one lane steps (kernel variant)           */
    FXB_INFO_KERNEL = 16,          /* 0 = HIP C++ kernel; hand-written gfx950 interpreter: 1 = register file in LDS,
                                      2..8 = register file in VGPRs (64/72/80/96/128/168/256-VGPR build);
                                      9..15 = program translated to gfx950 code, same seven VGPR builds */
    FXB_INFO_NUM_ROWS = 17,        /* rows of the per-instance register file               */
    FXB_INFO_XLATE_CODE_BYTES = 18,/* translated program: bytes of machine code (both streams), else 0 */
    FXB_INFO_XLATE_INLINED = 19,   /* records of the steady stream turned into straight-line code */
    FXB_INFO_XLATE_CALLED = 20,    /* records of the steady stream that call an interpreter handler */
    FXB_INFO_XLATE_UNSATURATED = 21,/* saturating instructions translated without a saturation (result provably in [-1, 1]) */
    FXB_INFO_XLATE_VALU = 22,       /* translated program: vector-ALU instructions per wavefront and sample period (steady fast stream) */
    FXB_INFO_XLATE_VALU_SLOW = 23,  /* ... those of the ~4-clock issue class (conversions, min/max/med3, compares, fp64, SGPR sources) */
    FXB_INFO_XLATE_VALU_CLOCKS = 24,/* ... modelled SIMD issue clocks of all of them per wavefront and sample period */
    FXB_INFO_XLATE_VGPR_CONSTANTS = 25, /* uniform constants the translated code keeps in spare VGPRs */
    FXB_INFO_XLATE_BUILDS = 26,    /* translations (code generation + module load) on the CALLER's thread since the handle was created: each one
                                      held a process call up for a few milliseconds */
    FXB_INFO_CODE_CACHE_HITS = 27, /* changes of code that were a pointer swap: the shape (block-length class, set of registers with rows, compiled-in
                                      values) had been generated before - by an earlier call or ahead of time by the builder thread */
    FXB_INFO_CODE_CACHED = 28,     /* generated code objects the handle holds (the one in force included) */
    FXB_INFO_XLATE_CODE_HASH = 30, /* fingerprint (63 bits) of the code object in force, 0 when the program is not translated: what a profile of a
                                      launch is a profile of (fxp_code_hash computes the same without a device) */
    FXB_INFO_STAGE_TRIALS = 31,    /* launches whose time went into the choice of the stage count (options the cost model cannot tell apart are timed
                                      on the caller's own blocks; FX_STAGES_TUNE=0 turns that off, FX_STAGES=n pins the count) */
    FXB_INFO_XLATE_BACKGROUND_BUILDS = 29, /* translations on the handle's builder thread (ahead of time: the variant with the declared controls in
                                      rows, the lean variant, code for another class of block lengths); FX_BUILDER=0 in the environment turns the thread off */
    FXB_INFO_CONTROL_ROWS = 32     /* declared controls that have a register row in the code in force: 0 until the host moves one (their values are
                                      folded into the code), then all of them (one change of code for the whole panel, generated ahead of time), then
                                      - a few blocks later, a pointer swap - only the ones that have been written lately (within 8192 sample periods;
                                      the others go back into the code, where a constant is cheaper than a row: e.g. INTERP with a constant X).  Same
                                      results in all three. */
};
int64_t fxb_info(fxb_handle* h, int what);
/* Which tier runs the program as it stands, in words - "translated to gfx950 code (fx_xlate_v128, 8 stages)", "interpreter
 * (fx_interp_v96): <why there is no translation>" (a SKIP that can jump over END: passes over the program; a register file above
 * 224 rows; controls that keep moving ...), "HIP C++ kernel (1 instance(s) per lane): <why no assembly tier takes the program>"
 * (a register file beyond every build, a literal LOG / EXP table number outside 0..31 ...) - so that a host that finds
 * FXB_INFO_KERNEL below 9 can say why.  Copies at most buflen-1 characters, returns the note's length
 * (negative FX_E_*).  Of shard 0 for a multi-device handle (every shard runs the same code).  Nothing in the reference. */
int fxb_tier_note(fxb_handle* h, char* buf, int buflen);


/* ------------------------------------------------------------------ front-end only (no device)
 * The host-side loader and lowering, usable without a GPU: what the .da text became.
 * Mirrors the reference's private model (include/FX8010.h:167-194) for inspection and tests. */
typedef struct fxp_handle fxp_handle;
fxp_handle* fxp_create(int num_channels);
void fxp_destroy(fxp_handle* h);
int fxp_set_option(fxp_handle* h, unsigned option, int on);
int fxp_load_file(fxp_handle* h, const char* path);
int fxp_load_text(fxp_handle* h, const char* text);
int fxp_num_registers(fxp_handle* h);
const char* fxp_register_name(fxp_handle* h, int i);
int fxp_register_type(fxp_handle* h, int i);    /* reference RegisterType numbering */
int fxp_register_ioindex(fxp_handle* h, int i);
float fxp_register_value(fxp_handle* h, int i);
int fxp_num_instructions(fxp_handle* h);
/* out8 = opcode (reference Opcode numbering), R, A, X, Y, hasInput, hasOutput, hasNoise */
void fxp_instruction(fxp_handle* h, int i, int out8[8]);
int fxp_itram_size(fxp_handle* h);
int fxp_xtram_size(fxp_handle* h);
int fxp_error_count(fxp_handle* h);
const char* fxp_error_desc(fxp_handle* h, int i);
int fxp_error_row(fxp_handle* h, int i);
int fxp_control_count(fxp_handle* h);
const char* fxp_control_at(fxp_handle* h, int i);
int fxp_meta_get(fxp_handle* h, const char* key, char* buf, int buflen);
int fxp_ready(fxp_handle* h);
/* LOG (kind 0) / EXP (kind 1) table of one exponent: 64 doubles (reference FX8010.cpp:63-105) */
const double* fxp_lut(int kind, int exponent);
/* lower for the device with the registers' initial values; 0 or FX_E_PROGRAM.  fxp_lower_info
 * takes the FXB_INFO_* selectors that describe the lowering. */
int fxp_lower(fxp_handle* h);
int64_t fxp_lower_info(fxp_handle* h, int what);
/* Translate the loaded program to gfx950 machine code as the batch path would (no device needed), for the VGPR
 * build with `vgprs` registers (64/72/80/96/128/168/256; 0 = the smallest build that holds the program).
 * stream: 0 = steady fast, 1 = steady exact, 2 = last-sample fast, 3 = last-sample exact (fast streams assume a
 * finite register file and leave for the exact one when a non-finite value appears; a program with a non-finite
 * uniform operand has no fast streams: size 0); 4 = run-once code (LDS tables).  Returns the code size in bytes (negative FX_E_* when the program
 * cannot be translated, see fxp_last_error) and copies at most `cap` bytes of code and at most listing_cap-1
 * characters of the assembler listing (one instruction per line). */
int64_t fxp_translate(fxp_handle* h, int vgprs, int stream, void* code, int64_t cap, char* listing, int64_t listing_cap);
/* ... with `key` among the registers that can have a control track (fxb_set_register_track): the code fxp_translate then
 * returns is what a batch runs after a track has been armed for that register.  0 found, 1 not found, FX_E_ARG beyond 16. */
int fxp_track_register(fxp_handle* h, const char* key);
/* The same for the program cut into (at most) `stages` stages run by the wavefronts of one workgroup - what the batch path
 * generates for small batches (fx_xlate.hpp StageInfo): the code of stream `stream` of stage `stage` (stream 4: the shared
 * run-once code).  *stages_out = the number of stages the program was cut into (1: not cut - the call then returns the
 * unstaged code and fxp_last_error says why); info (optional, info_cap ints): per cut {first record of the next stage, rows
 * handed over}, then the LDS bytes of a workgroup, the number of register-file rows and, per row, the stage that stores it. */
int64_t fxp_translate_staged(fxp_handle* h, int vgprs, int stages, int stage, int stream, void* code, int64_t cap, char* listing, int64_t listing_cap,
                             int* stages_out, int* info, int info_cap);
/* Fingerprint (63 bits, >= 0; negative FX_E_*) of the code object a batch would load for this program with its registers' initial
 * values: the `vgprs` build (64 ... 256), cut into at most `stages` stages (1: not cut; the default LDS budget and step length
 * of a small batch with long blocks), flags bit 0 = delay lines larger than the caches (non-temporal TRAM accesses), bit 1 =
 * the wavefronts of a SIMD take turns at the top priority (what a batch of two or more wavefronts per SIMD generates, on a build of
 * at most four wave slots: 128 registers and up).  Equals
 * fxb_info(FXB_INFO_XLATE_CODE_HASH) of a batch in that situation: tests and bench.py use it to tell whether a committed profile
 * still describes the code that is generated today. */
int64_t fxp_code_hash(fxp_handle* h, int vgprs, int stages, unsigned flags);
const char* fxp_last_error(fxp_handle* h);

/* library / device probe: number of HIP devices visible (0 if none), never throws */
int fxb_device_count(void);
const char* fxb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FX8010_AMD_H */
