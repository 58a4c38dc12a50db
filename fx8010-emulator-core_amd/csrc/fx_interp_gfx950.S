// fx_interp_gfx950.s — hand-written CDNA4 (gfx950) interpreter for the FX8010 opcode stream.
//
// One wavefront = 64 emulated DSPs (one per lane), one workgroup = one wavefront.  The host
// decoder (fx_decode.cpp + fx_asm.cpp encodeAsmStream) turns the program into 32-byte records; this kernel is a
// direct-threaded interpreter over them:
//
//   record  w0:w1 = absolute address of the handler (patched in by the host from the probe launch)
//           w2/w3/w4 = operand A/X/Y: offset (LDS build) or index (VGPR builds) of its row, or the IEEE
//                      bits of a uniform value; LOG/EXP: w3 = byte offset of the segment table
//           w5 = destination row R (offset / index)
//           w6 = flags for the generic handlers (bit0 A uniform, bit1 X uniform, bit2 Y uniform, bit3 write CCR)
//           w6:w7 = (1.0 - X) as a double for INTERP with a uniform X
//
//   fetch    one s_load_dwordx8 per record, issued one record ahead (s[24:31]) while s[16:23] executes
//   dispatch one s_setpc_b64 to the address in the record — no compare chain, no table hop;
//            MACS/MACSN/ACC3/INTERP have one handler per (operand kinds x CCR) so they test nothing
//   state    per-lane register file in LDS (row r at r*256 + lane*4, conflict-free); CCR = row 0;
//            skip counter, TRAM cursors, LFSR, flags in VGPRs; SKIP shadows run under EXEC
//
// Arithmetic follows the reference interpreter (source/FX8010.cpp:1023-1249) operation by
// operation: separate v_mul/v_add (never v_fma/v_mac), fp64 for INTERP and LOG/EXP, fp32
// denormals enabled, x86 cvttss2si semantics for every float->int conversion (CVTT).
//
// gfx950 hazards handled by hand: a VALU instruction that writes VCC/SGPRs needs 2 wait states
// before a VALU instruction reads them (s_nop 1 or two independent instructions).

// This file is assembled several times (clang -x assembler-with-cpp):
//   -DKNAME=fx_interp_lds                 register file in LDS (any size up to the LDS budget)
//   -DKNAME=fx_interp_vNN -DRF_VGPR -DNVGPR=NN    register file in VGPRs v32.. (VGPR index mode, M0): no LDS
//        traffic and no operand latency at all.  NN = 64 / 72 / 80 / 96 / 128 / 168 / 256 gives
//        32 / 40 / 48 / 64 / 96 / 136 / 224 rows at 8 / 7 / 6 / 5 / 4 / 3 / 2 wavefronts per SIMD.
// In the VGPR build a "row offset" in a record is simply the row index.
#ifndef KNAME
#define KNAME fx_interp_lds
#endif
#define PNAME_PASTE(a) a##_probe
#define PNAME_PASTE2(a) PNAME_PASTE(a)
#define PNAME PNAME_PASTE2(KNAME)
#define HOLE FX_HOLE_PASTE2(KNAME)
#define OTABLE FX_TAB_PASTE2(KNAME)
#define FX_TAB_PASTE(a) a##_table
#define FX_TAB_PASTE2(a) FX_TAB_PASTE(a)
#define FX_HOLE_PASTE(a) a##_hole
#define FX_HOLE_PASTE2(a) FX_HOLE_PASTE(a)
#ifndef HOLE_BYTES
#define HOLE_BYTES 262144
#endif
#ifndef NVGPR
#define NVGPR 32
#endif
#define RF v32   /* first register of the VGPR register file; v0..v31 belong to the interpreter */

	.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
	.amdhsa_code_object_version 6

// ---- kernarg layout (struct AsmArgs in fx_asm.hpp) ----
	.set KA_STEADY,   0x00
	.set KA_LAST,     0x08
	.set KA_ROWTAB,   0x10
	.set KA_STATE,    0x18
	.set KA_IN,       0x20
	.set KA_OUT,      0x28
	.set KA_ITRAM,    0x30
	.set KA_XTRAM,    0x38
	.set KA_LUT,      0x40
	.set KA_N,        0x48
	.set KA_NPAD,     0x50
	.set KA_NLOAD,    0x58
	.set KA_NSTORE,   0x5c
	.set KA_NSAMPLES, 0x60
	.set KA_CHANNELS, 0x64
	.set KA_INOFF,    0x68      // int[4]: LDS byte offset of the channel's input row, -1 unused
	.set KA_LATCHOFF, 0x78      // int[4]: LDS byte offset of the channel's output latch row
	.set KA_ISLOTS,   0x88
	.set KA_XSLOTS,   0x8c
	.set KA_ISIZE,    0x90
	.set KA_XSIZE,    0x94
	.set KA_CURSORROW,0x98      // state row of the 4 TRAM cursors
	.set KA_NOISEROW, 0x9c      // state row of g_x1, g_x2
	.set KA_OODROW,   0xa0
	.set KA_COUNTLO,  0xa4
	.set KA_COUNTHI,  0xa8
	.set KA_STATICCNT,0xac
	.set KA_LUTX1OFF, 0xb0      // byte offset of x1[] inside the LUT blob
	.set KA_INIT,     0xb4      // translated programs: offset of run-once code (LDS tables), 0 = none
	.set KA_TRACKS,   0xb8      // translated programs: control tracks of this block (fx_batch.hpp TrackHeader[3] + values), 0 = none
	.set KA_STAGES,   0xc0      // translated programs cut into stages (fx_xlate.hpp StageDescriptor[nStages], 32 bytes each), 0 = none
	.set KA_NSTAGES,  0xc8      // wavefronts per workgroup = stages of the program (0 or 1: one wavefront runs all of it)
	.set KA_TRAMDANE, 0xcc      // interpreter builds, bit 0: the opt-in DANE delay-line model (address counters step once per sample period);
	                            //  bit 1: multi-pass program (END inside a SKIP shadow: lanes that skipped it run the program again);
	                            //  bit 2: the wavefronts of a SIMD take turns at the top priority, turns of 2^(bits 12:8) ticks of 10 ns
	.set KA_SIZE,     0xd0

// ---- out-of-domain flag bits (fx_kernel.hpp) ----
	.set OOD_TRAM_READ_NEG, 1
	.set OOD_TRAM_WRITE_OOB, 2
	.set OOD_TRAM_SIZE0, 4
	.set OOD_LUT_TABLE, 8
	.set OOD_PASS_CAP, 32
	.set PASS_CAP, 64           // passes of a multi-pass program per sample period (fx_kernel.hpp kPassCap, the oracle's PASS_CAP)
	.set OOD_LUT_INDEX, 16

// ---- VGPRs ----
//  v0 lane  v1 lane*4  v2 a/result  v3 x  v4 y  v5 address  v6-v13 temporaries
//  v14 numSkip  v15 executed-shadowed count  v16 iw  v17 ir  v18 xw  v19 xr  v20 g1  v21 g2  v22 ood
//  v23-v26 prefetched input of channel 0-3   v27 instance*4 (byte offset into a PCM / state row)
//  v28 0x80000000   (v29-v31 spare)   v32.. register file of the VGPR builds
// ---- SGPRs ----
//  s[0:1] kernarg  s2 wave index  s3 sample  s[4:5] stream of this sample  s[6:7] steady  s8 fetch offset
//  s9 nSamples  s[10:11] state  s[12:13] in (this sample)  s[14:15] out (this sample)
//  s[16:23] current record  s[24:31] next record  s[32:33] branch table  s[34:35] jump target
//  s[36:37] iTRAM of this wave  s[38:39] xTRAM of this wave  s[40:41] LUT  s[42:43] last-sample stream
//  s44 channels  s45 bytes per sample (channels*N*4)  s46 iSlots  s47 xSlots  s[48:51] input row offsets
//  s[52:55] latch row offsets  s56 iSize  s57 xSize  s[58:59] lanes with instance < N  s60 state row pitch (nPad*4)
//  s61 byte offset of x1[] in the LUT blob  s62-s67 temporaries  s68 bytes per channel-sample (N*4)
//  s[72:73] row table  s74 nLoad  s75 nStore  s76 cursor state row  s77 LFSR state row
//  s70 (interpreter builds) bit 0 opt-in DANE delay-line model, bit 1 multi-pass program   s71 passes of this sample period
//  s[96:97] (interpreter builds) lanes that run the passes of this sample period   s[98:99] lanes that have executed END in it

	.text
	.globl	KNAME
	.p2align	8
	.type	KNAME,@function

// register-file access used by the prologue / per-sample I/O / epilogue (the handler sets have their own copies)
.macro LOADV dst, sreg
#ifdef RF_VGPR
	s_set_gpr_idx_on \sreg, 1
	v_mov_b32 \dst, RF
	s_set_gpr_idx_off
#else
	v_add_u32 \dst, \sreg, v1
	ds_read_b32 \dst, \dst
#endif
.endm
.macro STOREV sreg, src
#ifdef RF_VGPR
	s_set_gpr_idx_on \sreg, 8
	v_mov_b32 RF, \src
	s_set_gpr_idx_off
#else
	v_add_u32 v5, \sreg, v1
	ds_write_b32 v5, \src
#endif
.endm

// the prologue's batches: request the state row of table entry `entry` into `dst`; later check it (translated programs: taint,
// below) and put it into its register-file row
.macro ROW_FETCH entry, dst
	s_lshr_b32 s65, \entry, 16
	s_mul_i32 s66, s65, s60
	s_mul_hi_u32 s67, s65, s60
	s_add_u32 s66, s66, s10
	s_addc_u32 s67, s67, s11
	global_load_dword \dst, v27, s[66:67]
.endm
.macro ROW_TAKE entry, src
#ifdef XLATE
	v_cmp_class_f32 vcc, \src, v29
	s_bitcmp1_b32 \entry, 15                              // row of the BOUNDED class (fx_xlate.hpp)
	s_cbranch_scc0 1f
	v_cmp_nle_f32_e64 vcc, |\src|, 1.0
1:
	s_or_b64 s[78:79], s[78:79], vcc
	s_and_b32 s64, \entry, 0x7fff
#else
	s_and_b32 s64, \entry, 0xffff
#endif
#ifndef RF_VGPR
	s_lshl_b32 s64, s64, 8
#endif
	STOREV s64, \src
.endm
// ... and the epilogue's: register-file row of table entry `entry` -> `dst`, then to its state row
.macro ROW_READ entry, dst
	s_and_b32 s64, \entry, 0xffff
#ifndef RF_VGPR
	s_lshl_b32 s64, s64, 8
#endif
	LOADV \dst, s64
.endm
.macro ROW_STORE entry, src
	s_lshr_b32 s65, \entry, 16
	s_mul_i32 s66, s65, s60
	s_mul_hi_u32 s67, s65, s60
	s_add_u32 s66, s66, s10
	s_addc_u32 s67, s67, s11
	global_store_dword v27, \src, s[66:67]
.endm

// two-level cpp paste so that SFX expands
#define FX_PASTE(a, b) a##b
#define FX_PASTE2(a, b) FX_PASTE(a, b)
#define H(x) FX_PASTE2(x, SFX)
#define M(x) FX_PASTE2(x, SFX)
// ------------------------------------------------------------------------------------------ entry
KNAME:
#ifdef XLATE
	s_getpc_b64 s[32:33]                                  // KNAME + 4
	// a program cut into stages runs in workgroups of several wavefronts - wavefront k = stage k, all of them on the
	// workgroup's 64 instances (fx_xlate.hpp StageInfo); s52 = stage, v0 = lane either way
	v_readfirstlane_b32 s52, v0
	v_and_b32 v0, 63, v0
	s_lshr_b32 s52, s52, 6
#endif
	s_load_dwordx16 s[4:19], s[0:1], KA_STEADY            // steady last rowtab state in out itram xtram
	s_load_dwordx8  s[40:47], s[0:1], KA_LUT              // lut n npad nload nstore
	s_load_dwordx2  s[64:65], s[0:1], KA_NSAMPLES         // nSamples channels
	v_lshlrev_b32 v1, 2, v0
	s_lshl_b32 s62, s2, 6
	v_add_u32 v27, s62, v0                                // instance
	v_mov_b32 v28, 0x80000000
	v_mov_b32 v14, 0
	v_mov_b32 v15, 0
	v_mov_b32 v22, 0
	s_waitcnt lgkmcnt(0)
	s_mov_b64 s[72:73], s[8:9]                            // row table
	s_mov_b64 s[36:37], s[16:17]                          // itram
	s_mov_b64 s[38:39], s[18:19]                          // xtram
	s_mov_b64 s[70:71], s[6:7]                            // last-sample stream (moved to s[42:43] below)
	s_mov_b64 s[6:7], s[4:5]                              // steady
	// s[10:11] state, s[12:13] in, s[14:15] out are already in place
	// s[40:41] lut, s[42:43] n, s[44:45] nPad, s46 nLoad, s47 nStore
	v_cmp_gt_u32 s[58:59], s42, v27                       // lane has an instance (n < 2^31)
	s_lshl_b32 s60, s44, 2                                // state row pitch in bytes
	s_lshl_b32 s68, s42, 2                                // bytes of one channel of one sample (N*4)
	v_lshlrev_b32 v27, 2, v27                             // instance*4
	s_mov_b32 s9, s64                                     // nSamples
	s_mov_b32 s74, s46                                    // nLoad
	s_mov_b32 s75, s47                                    // nStore
	s_mov_b32 s44, s65                                    // channels
	s_mul_i32 s45, s42, s44
	s_lshl_b32 s45, s45, 2                                // bytes per sample of PCM = channels * N * 4
	s_mov_b64 s[42:43], s[70:71]                          // last-sample stream
#ifdef XLATE
	// translated programs: `steady` / `last` are byte offsets of the two code streams from the kernel entry
	// (each as two dwords: {fast stream, exact stream}, see "taint" below)
	s_sub_u32 s32, s32, 4
	s_subb_u32 s33, s33, 0                                // s[32:33] = address of the kernel entry
	// taint: lanes into whose register file a non-finite value (NaN, +-Inf) has come - from the state rows, the
	// PCM input, a TRAM read or a non-saturating instruction.  While no lane of the wave is tainted, no
	// saturating instruction can see a NaN, and the wave runs the fast stream, whose saturation is a bare
	// v_med3_f32; from the first tainted lane on it runs the exact stream (NaN passes the saturation,
	// FX8010.cpp:275-279).
	s_mov_b64 s[78:79], 0
	v_mov_b32 v29, 0x207                                  // v_cmp_class mask: sNaN | qNaN | -Inf | +Inf
	s_add_u32 s34, s32, (.Lepilogue-KNAME)
	s_addc_u32 s35, s33, 0                                // s[34:35] = epilogue (the translated last-sample stream ends with s_setpc_b64 s[34:35])
	s_mov_b32 s94, 0                                      // no TRAM read of the first sample is in flight yet
	s_mov_b32 s95, 0                                      // early TRAM reads: allowed only if the run-once code says so
#endif
#ifdef XLATE
	s_load_dwordx4  s[48:51], s[0:1], KA_INOFF            // inOff[4] (s52..s55: stage, stages, first entry of its store table, LDS scratch of the epilogue)
	s_load_dwordx2  s[62:63], s[0:1], KA_STAGES
	s_load_dword    s53, s[0:1], KA_NSTAGES
	s_mov_b32 s54, 0
#else
	s_load_dwordx8  s[48:55], s[0:1], KA_INOFF            // inOff[4] latchOff[4]
	s_load_dword    s70, s[0:1], KA_TRAMDANE
	s_mov_b64 s[96:97], s[58:59]                          // every instance runs the first pass
	s_mov_b64 s[98:99], 0
	s_mov_b32 s71, 0
#endif
	s_load_dwordx4  s[64:67], s[0:1], KA_ISLOTS           // iSlots xSlots iSize xSize
	s_load_dwordx2  s[76:77], s[0:1], KA_CURSORROW      // cursorRow noiseRow
	s_load_dword    s61, s[0:1], KA_LUTX1OFF
	s_waitcnt lgkmcnt(0)
	s_mov_b32 s46, s64
	s_mov_b32 s47, s65
	s_mov_b32 s56, s66
	s_mov_b32 s57, s67
#ifdef XLATE
	// staged: this wavefront's code streams and its slice of the store-row table
	s_cmp_lt_u32 s53, 2
	s_cbranch_scc1 .Lone_stage
	s_lshl_b32 s64, s52, 5
	s_load_dwordx8 s[80:87], s[62:63], s64                // steady {fast, exact}, last {fast, exact}, store first, store count, LDS scratch
	s_waitcnt lgkmcnt(0)
	s_mov_b64 s[6:7], s[80:81]
	s_mov_b64 s[42:43], s[82:83]
	s_mov_b32 s54, s84
	s_mov_b32 s75, s85
	s_mov_b32 s55, s86                                    // LDS offset of the epilogue's scratch area
.Lone_stage:
#endif
	// TRAM base of this wave: base + wave * slots * 256
	s_mul_i32 s62, s2, s46
	s_mul_hi_u32 s63, s2, s46
	s_lshl_b64 s[62:63], s[62:63], 8
	s_add_u32 s36, s36, s62
	s_addc_u32 s37, s37, s63
	s_mul_i32 s62, s2, s47
	s_mul_hi_u32 s63, s2, s47
	s_lshl_b64 s[62:63], s[62:63], 8
	s_add_u32 s38, s38, s62
	s_addc_u32 s39, s39, s63

	// ---- prologue: state rows -> LDS rows
	s_mov_b32 s62, 0
	s_cmp_eq_u32 s74, 0
	s_cbranch_scc1 .Lload_done
	// Eight rows at a time - ONE trip to the row table and ONE to the state rows per batch instead of two per row: a launch of
	// a few dozen samples (a real-time caller's block) is mostly prologue and epilogue.  Fewer than eight left: one by one.
.Lload_batch:
	s_add_u32 s63, s62, 8
	s_cmp_gt_u32 s63, s74
	s_cbranch_scc1 .Lload_loop
	s_lshl_b32 s63, s62, 2
	s_load_dwordx8 s[16:23], s[72:73], s63                // ldsRow | stateRow << 16, eight of them
	s_waitcnt lgkmcnt(0)
	ROW_FETCH s16, v6
	ROW_FETCH s17, v7
	ROW_FETCH s18, v8
	ROW_FETCH s19, v9
	ROW_FETCH s20, v10
	ROW_FETCH s21, v11
	ROW_FETCH s22, v12
	ROW_FETCH s23, v13
	s_waitcnt vmcnt(0)
	ROW_TAKE s16, v6
	ROW_TAKE s17, v7
	ROW_TAKE s18, v8
	ROW_TAKE s19, v9
	ROW_TAKE s20, v10
	ROW_TAKE s21, v11
	ROW_TAKE s22, v12
	ROW_TAKE s23, v13
	s_add_u32 s62, s62, 8
	s_cmp_lt_u32 s62, s74
	s_cbranch_scc1 .Lload_batch
	s_branch .Lload_done
.Lload_loop:
	s_lshl_b32 s63, s62, 2
	s_load_dword s64, s[72:73], s63                       // ldsRow | stateRow << 16
	s_waitcnt lgkmcnt(0)
	s_lshr_b32 s65, s64, 16
#ifdef XLATE
	s_bitcmp1_b32 s64, 15                                 // row of the BOUNDED class (fx_xlate.hpp)
	s_cselect_b32 s69, 1, 0
	s_and_b32 s64, s64, 0x7fff
#else
	s_and_b32 s64, s64, 0xffff
#endif
	s_mul_i32 s66, s65, s60
	s_mul_hi_u32 s67, s65, s60
	s_add_u32 s66, s66, s10
	s_addc_u32 s67, s67, s11
	global_load_dword v2, v27, s[66:67]
#ifndef RF_VGPR
	s_lshl_b32 s64, s64, 8
#endif
	s_waitcnt vmcnt(0)
#ifdef XLATE
	// taint: a BOUNDED row must start inside [-1, 1] (NaN fails too), any other row must start finite
	v_cmp_class_f32 vcc, v2, v29
	s_cmp_eq_u32 s69, 0
	s_cbranch_scc1 .Lrow_checked
	v_cmp_nle_f32_e64 vcc, |v2|, 1.0
.Lrow_checked:
	s_or_b64 s[78:79], s[78:79], vcc
#endif
	STOREV s64, v2
	s_add_u32 s62, s62, 1
	s_cmp_lt_u32 s62, s74
	s_cbranch_scc1 .Lload_loop
.Lload_done:
	// cursors and LFSR words straight into VGPRs
	s_mul_i32 s66, s76, s60
	s_mul_hi_u32 s67, s76, s60
	s_add_u32 s66, s66, s10
	s_addc_u32 s67, s67, s11
	global_load_dword v16, v27, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_load_dword v17, v27, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_load_dword v18, v27, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_load_dword v19, v27, s[66:67]
	s_mul_i32 s66, s77, s60
	s_mul_hi_u32 s67, s77, s60
	s_add_u32 s66, s66, s10
	s_addc_u32 s67, s67, s11
	global_load_dword v20, v27, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_load_dword v21, v27, s[66:67]
#ifdef XLATE
	// run-once code of the translated program (copies the LOG/EXP tables it uses into LDS, decides from the TRAM
	// cursors whether reads may be issued a sample ahead); returns through s[24:25]
	s_load_dword s62, s[0:1], KA_INIT
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_cmp_eq_u32 s62, 0
	s_cbranch_scc1 .Lno_init
	s_add_u32 s62, s62, s32
	s_addc_u32 s63, s33, 0
	s_getpc_b64 s[24:25]
.Lpc_init:
	s_add_u32 s24, s24, (.Lno_init-.Lpc_init)
	s_addc_u32 s25, s25, 0
	s_setpc_b64 s[62:63]
.Lno_init:
#endif
	// first sample's input
	v_mov_b32 v23, 0
	v_mov_b32 v24, 0
	v_mov_b32 v25, 0
	v_mov_b32 v26, 0
	s_mov_b32 s3, 0
	s_cmp_lt_i32 s9, 1
	s_cbranch_scc1 .Lepilogue
	s_mov_b64 s[62:63], s[12:13]
	s_mov_b64 exec, s[58:59]
	s_cmp_lt_i32 s48, 0
	s_cbranch_scc1 .Lp_in1
	global_load_dword v23, v27, s[62:63]
.Lp_in1:
	s_add_u32 s62, s62, s68
	s_addc_u32 s63, s63, 0
	s_cmp_lt_i32 s44, 2
	s_cbranch_scc1 .Lp_indone
	s_cmp_lt_i32 s49, 0
	s_cbranch_scc1 .Lp_in2
	global_load_dword v24, v27, s[62:63]
.Lp_in2:
	s_add_u32 s62, s62, s68
	s_addc_u32 s63, s63, 0
	s_cmp_lt_i32 s44, 3
	s_cbranch_scc1 .Lp_indone
	s_cmp_lt_i32 s50, 0
	s_cbranch_scc1 .Lp_in3
	global_load_dword v25, v27, s[62:63]
.Lp_in3:
	s_add_u32 s62, s62, s68
	s_addc_u32 s63, s63, 0
	s_cmp_lt_i32 s44, 4
	s_cbranch_scc1 .Lp_indone
	s_cmp_lt_i32 s51, 0
	s_cbranch_scc1 .Lp_indone
	global_load_dword v26, v27, s[62:63]
.Lp_indone:
	s_mov_b64 exec, -1
	s_waitcnt vmcnt(0)

#ifdef XLATE
	// The translated program IS the sample loop (fx_xlate.cpp: input rows, program, PCM out, next sample); it comes
	// back at the epilogue.  Cold entries: `steady` = {fast, exact} while more than one sample is left, else `last`;
	// a wave tainted by its state rows starts in the exact stream.
	s_cmp_eq_u32 s9, 1
	s_cselect_b32 s4, s42, s6
	s_cselect_b32 s5, s43, s7
	s_cmp_lg_u64 s[78:79], 0
	s_cselect_b32 s4, s5, s4
	s_add_u32 s4, s4, s32
	s_addc_u32 s5, s33, 0
	s_setpc_b64 s[4:5]
#else
	// ---- one sample period
.Lsample:
	// this sample's input -> LDS rows
	s_cmp_lt_i32 s48, 0
	s_cbranch_scc1 .Ls_w1
	STOREV s48, v23
.Ls_w1:
	s_cmp_lt_i32 s44, 2
	s_cbranch_scc1 .Ls_wdone
	s_cmp_lt_i32 s49, 0
	s_cbranch_scc1 .Ls_w2
	STOREV s49, v24
.Ls_w2:
	s_cmp_lt_i32 s44, 3
	s_cbranch_scc1 .Ls_wdone
	s_cmp_lt_i32 s50, 0
	s_cbranch_scc1 .Ls_w3
	STOREV s50, v25
.Ls_w3:
	s_cmp_lt_i32 s44, 4
	s_cbranch_scc1 .Ls_wdone
	s_cmp_lt_i32 s51, 0
	s_cbranch_scc1 .Ls_wdone
	STOREV s51, v26
.Ls_wdone:
	// prefetch the next sample's input (if any); its latency hides behind this sample's program
	s_add_u32 s62, s3, 1
	s_cmp_ge_i32 s62, s9
	s_cbranch_scc1 .Ls_pdone
	s_add_u32 s62, s12, s45
	s_addc_u32 s63, s13, 0
	s_mov_b64 exec, s[58:59]
	s_cmp_lt_i32 s48, 0
	s_cbranch_scc1 .Ls_p1
	global_load_dword v23, v27, s[62:63]
.Ls_p1:
	s_add_u32 s62, s62, s68
	s_addc_u32 s63, s63, 0
	s_cmp_lt_i32 s44, 2
	s_cbranch_scc1 .Ls_pend
	s_cmp_lt_i32 s49, 0
	s_cbranch_scc1 .Ls_p2
	global_load_dword v24, v27, s[62:63]
.Ls_p2:
	s_add_u32 s62, s62, s68
	s_addc_u32 s63, s63, 0
	s_cmp_lt_i32 s44, 3
	s_cbranch_scc1 .Ls_pend
	s_cmp_lt_i32 s50, 0
	s_cbranch_scc1 .Ls_p3
	global_load_dword v25, v27, s[62:63]
.Ls_p3:
	s_add_u32 s62, s62, s68
	s_addc_u32 s63, s63, 0
	s_cmp_lt_i32 s44, 4
	s_cbranch_scc1 .Ls_pend
	s_cmp_lt_i32 s51, 0
	s_cbranch_scc1 .Ls_pend
	global_load_dword v26, v27, s[62:63]
.Ls_pend:
	s_mov_b64 exec, -1
.Ls_pdone:
	// stream of this sample: the last sample materialises every CCR write
	s_add_u32 s62, s3, 1
	s_cmp_eq_u32 s62, s9
	s_cselect_b32 s4, s42, s6
	s_cselect_b32 s5, s43, s7
	v_mov_b32 v14, 0                                      // numSkip is local to process() (FX8010.cpp:1030)
	s_load_dwordx16 s[16:31], s[4:5], 0x0                 // records 0, 1
	s_load_dwordx16 s[80:95], s[4:5], 0x40                // records 2, 3
	s_mov_b32 s8, 128
	s_waitcnt lgkmcnt(0)
	s_setpc_b64 s[16:17]
#endif

// ------------------------------------------------------------------------------------------ handlers
// handler offsets (bytes from the kernel entry) of the four register sets, 84 slots each; read only by the
// probe mode, which turns them into absolute addresses for the host to put into the records
	.p2align 6
	.globl	OTABLE
OTABLE:
#define SFX _a
.macro M(JT2) op, kind
	.long h_\op\()_\kind\()_0\()SFX - KNAME
	.long h_\op\()_\kind\()_1\()SFX - KNAME
.endm
#include "fx_interp_table.inc"
#undef SFX
#ifdef XLATE
	.long HOLE - KNAME                                    // [84] the code hole the host fills (fx_xlate.cpp)
	.long HOLE_BYTES                                      // [85]
#else
#define SFX _b
.macro M(JT2) op, kind
	.long h_\op\()_\kind\()_0\()SFX - KNAME
	.long h_\op\()_\kind\()_1\()SFX - KNAME
.endm
#include "fx_interp_table.inc"
#undef SFX
#define SFX _c
.macro M(JT2) op, kind
	.long h_\op\()_\kind\()_0\()SFX - KNAME
	.long h_\op\()_\kind\()_1\()SFX - KNAME
.endm
#include "fx_interp_table.inc"
#undef SFX
#define SFX _d
.macro M(JT2) op, kind
	.long h_\op\()_\kind\()_0\()SFX - KNAME
	.long h_\op\()_\kind\()_1\()SFX - KNAME
.endm
#include "fx_interp_table.inc"
#undef SFX

#endif

// records 4n, 4n+1 live in window s[16:31], records 4n+2, 4n+3 in window s[80:95]
#define SFX _a
#define RA s18
#define RX s19
#define RY s20
#define RW1 s21
#define RFLG s22
#define ROMX s[22:23]
#define OWPC s[24:25]
#define WINDOW s[16:31]
#define NEXT_LOADS 0
#include "fx_interp_handlers.inc"
#undef SFX
#undef RA
#undef RX
#undef RY
#undef RW1
#undef RFLG
#undef ROMX
#undef OWPC
#undef WINDOW
#undef NEXT_LOADS
#ifndef XLATE
#define SFX _b
#define RA s26
#define RX s27
#define RY s28
#define RW1 s29
#define RFLG s30
#define ROMX s[30:31]
#define OWPC s[80:81]
#define WINDOW s[16:31]
#define NEXT_LOADS 1
#include "fx_interp_handlers.inc"
#undef SFX
#undef RA
#undef RX
#undef RY
#undef RW1
#undef RFLG
#undef ROMX
#undef OWPC
#undef WINDOW
#undef NEXT_LOADS
#define SFX _c
#define RA s82
#define RX s83
#define RY s84
#define RW1 s85
#define RFLG s86
#define ROMX s[86:87]
#define OWPC s[88:89]
#define WINDOW s[80:95]
#define NEXT_LOADS 0
#include "fx_interp_handlers.inc"
#undef SFX
#undef RA
#undef RX
#undef RY
#undef RW1
#undef RFLG
#undef ROMX
#undef OWPC
#undef WINDOW
#undef NEXT_LOADS
#define SFX _d
#define RA s90
#define RX s91
#define RY s92
#define RW1 s93
#define RFLG s94
#define ROMX s[94:95]
#define OWPC s[16:17]
#define WINDOW s[80:95]
#define NEXT_LOADS 1
#include "fx_interp_handlers.inc"
#undef SFX
#undef RA
#undef RX
#undef RY
#undef RW1
#undef RFLG
#undef ROMX
#undef OWPC
#undef WINDOW
#undef NEXT_LOADS

#endif

// ---- end of the program for this sample: latch rows -> PCM out, next sample (interpreter builds; a translated
// program has its own)
#ifdef XLATE
h_endsample_a:
#else
h_endsample_a:
h_endsample_b:
h_endsample_c:
h_endsample_d:
	// A multi-pass program (FX8010.cpp:1033,1243: `do { ... } while (!isEND)` - every instruction lies in some SKIP's shadow, END
	// included): the lanes that did not execute END run the program again with the skip count they have left, at most PASS_CAP
	// passes per sample period (then: flagged, as the oracle and the HIP C++ kernel define it).
	s_bitcmp1_b32 s70, 1
	s_cbranch_scc0 .Le_single
	s_andn2_b64 s[96:97], s[96:97], s[98:99]
	s_add_u32 s71, s71, 1
	s_cmp_eq_u64 s[96:97], 0
	s_cbranch_scc1 .Le_passes_done
	s_cmp_lt_u32 s71, PASS_CAP
	s_cbranch_scc0 .Le_cap
	s_waitcnt lgkmcnt(0)                                  // (the record prefetch of the pass that ends here may still be on its way into a window)
	s_load_dwordx16 s[16:31], s[4:5], 0x0                 // records 0, 1
	s_load_dwordx16 s[80:95], s[4:5], 0x40                // records 2, 3
	s_mov_b32 s8, 128
	s_waitcnt lgkmcnt(0)
	s_setpc_b64 s[16:17]
.Le_cap:
#ifdef RF_VGPR
	s_set_gpr_idx_off
	s_nop 3
#endif
	s_mov_b64 exec, s[96:97]
	v_or_b32 v22, OOD_PASS_CAP, v22
.Le_passes_done:
	s_mov_b64 s[96:97], s[58:59]
	s_mov_b64 s[98:99], 0
	s_mov_b32 s71, 0
.Le_single:
	s_mov_b64 exec, s[58:59]
	LOADV v2, s52
	s_mov_b64 s[62:63], s[14:15]
	s_waitcnt lgkmcnt(0)
	global_store_dword v27, v2, s[62:63]
	s_cmp_lt_i32 s44, 2
	s_cbranch_scc1 .Le_done
	LOADV v2, s53
	s_add_u32 s62, s62, s68
	s_addc_u32 s63, s63, 0
	s_waitcnt lgkmcnt(0)
	global_store_dword v27, v2, s[62:63]
	s_cmp_lt_i32 s44, 3
	s_cbranch_scc1 .Le_done
	LOADV v2, s54
	s_add_u32 s62, s62, s68
	s_addc_u32 s63, s63, 0
	s_waitcnt lgkmcnt(0)
	global_store_dword v27, v2, s[62:63]
	s_cmp_lt_i32 s44, 4
	s_cbranch_scc1 .Le_done
	LOADV v2, s55
	s_add_u32 s62, s62, s68
	s_addc_u32 s63, s63, 0
	s_waitcnt lgkmcnt(0)
	global_store_dword v27, v2, s[62:63]
.Le_done:
	s_mov_b64 exec, -1
	s_add_u32 s12, s12, s45
	s_addc_u32 s13, s13, 0
	s_add_u32 s14, s14, s45
	s_addc_u32 s15, s15, 0
	// opt-in DANE delay-line model (fx_kernel.hip daneStep): both address counters step down once per sample period, every lane
	s_bitcmp1_b32 s70, 0
	s_cbranch_scc0 .Le_counters
	s_cmp_lt_i32 s56, 1
	s_cbranch_scc1 .Le_xcounter
	s_sub_i32 s62, s56, 1
	v_cmp_ge_i32 vcc, 0, v16
	v_subrev_u32 v5, 1, v16
	v_mov_b32 v6, s62
	s_nop 1
	v_cndmask_b32 v16, v5, v6, vcc
.Le_xcounter:
	s_cmp_lt_i32 s57, 1
	s_cbranch_scc1 .Le_counters
	s_sub_i32 s62, s57, 1
	v_cmp_ge_i32 vcc, 0, v18
	v_subrev_u32 v5, 1, v18
	v_mov_b32 v6, s62
	s_nop 1
	v_cndmask_b32 v18, v5, v6, vcc
.Le_counters:
	// A launch that fills the wave slots once: the wavefronts of a SIMD take turns at the top priority (the arbiter serves the
	// oldest first, and they would finish one after the other - DESIGN.md section 5).  Mode bit 2; every fourth sample the priority
	// becomes ((100 MHz clock >> mode bits 12:8) + wave-buffer slot) & 3.
	s_bitcmp1_b32 s70, 2
	s_cbranch_scc0 .Le_noturn
	s_and_b32 s62, s3, 3
	s_cmp_eq_u32 s62, 3
	s_cbranch_scc0 .Le_noturn
	s_memrealtime s[62:63]
	s_getreg_b32 s64, hwreg(HW_REG_HW_ID, 0, 4)
	s_bfe_u32 s65, s70, 0x50008
	s_waitcnt lgkmcnt(0)
	s_lshr_b32 s62, s62, s65
	s_add_u32 s62, s62, s64
	s_and_b32 s62, s62, 3
	s_cmp_eq_u32 s62, 0
	s_cbranch_scc0 .Le_turn1
	s_setprio 0
	s_branch .Le_noturn
.Le_turn1:
	s_cmp_eq_u32 s62, 1
	s_cbranch_scc0 .Le_turn2
	s_setprio 1
	s_branch .Le_noturn
.Le_turn2:
	s_cmp_eq_u32 s62, 2
	s_cbranch_scc0 .Le_turn3
	s_setprio 2
	s_branch .Le_noturn
.Le_turn3:
	s_setprio 3
.Le_noturn:
	s_add_u32 s3, s3, 1
	s_cmp_lt_i32 s3, s9
	s_waitcnt vmcnt(0)
	s_cbranch_scc1 .Lsample
#endif

	// ---- epilogue: LDS rows and VGPR state -> state rows
.Lepilogue:
#ifdef RF_VGPR
	s_set_gpr_idx_off                                     // entered from a handler / from generated code: whatever index mode that left
#endif
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_mov_b32 s62, 0
	s_cmp_eq_u32 s75, 0
	s_cbranch_scc1 .Lstore_done
	// (eight rows per trip to the row table, like the prologue)
.Lstore_batch:
	s_add_u32 s63, s62, 8
	s_cmp_gt_u32 s63, s75
	s_cbranch_scc1 .Lstore_loop
	s_add_u32 s63, s62, s74
#ifdef XLATE
	s_add_u32 s63, s63, s54                               // (staged: this stage's rows)
#endif
	s_lshl_b32 s63, s63, 2
	s_load_dwordx8 s[16:23], s[72:73], s63
	s_waitcnt lgkmcnt(0)
	ROW_READ s16, v6
	ROW_READ s17, v7
	ROW_READ s18, v8
	ROW_READ s19, v9
	ROW_READ s20, v10
	ROW_READ s21, v11
	ROW_READ s22, v12
	ROW_READ s23, v13
	s_waitcnt lgkmcnt(0)
	ROW_STORE s16, v6
	ROW_STORE s17, v7
	ROW_STORE s18, v8
	ROW_STORE s19, v9
	ROW_STORE s20, v10
	ROW_STORE s21, v11
	ROW_STORE s22, v12
	ROW_STORE s23, v13
	s_add_u32 s62, s62, 8
	s_cmp_lt_u32 s62, s75
	s_cbranch_scc1 .Lstore_batch
	s_branch .Lstore_done
.Lstore_loop:
	s_add_u32 s63, s62, s74
#ifdef XLATE
	s_add_u32 s63, s63, s54                               // (staged: this stage's rows)
#endif
	s_lshl_b32 s63, s63, 2
	s_load_dword s64, s[72:73], s63
	s_waitcnt lgkmcnt(0)
	s_lshr_b32 s65, s64, 16
	s_and_b32 s64, s64, 0xffff
#ifndef RF_VGPR
	s_lshl_b32 s64, s64, 8
#endif
	LOADV v2, s64
	s_mul_i32 s66, s65, s60
	s_mul_hi_u32 s67, s65, s60
	s_add_u32 s66, s66, s10
	s_addc_u32 s67, s67, s11
	s_waitcnt lgkmcnt(0)
	global_store_dword v27, v2, s[66:67]
	s_add_u32 s62, s62, 1
	s_cmp_lt_u32 s62, s75
	s_cbranch_scc1 .Lstore_loop
.Lstore_done:
#ifdef XLATE
	// staged: every wavefront has stored the rows it owns; the delay-line cursors, the LFSR words, the out-of-domain flags
	// and the instruction counter are stage 0's - with the other stages' shadowed-instruction counts and flags, which
	// come through LDS ([stage][2][lane] from s55 on: an area of its own - an early stage gets here while the later ones
	// still read tables and packets)
	s_cmp_lt_u32 s53, 2
	s_cbranch_scc1 .Lsolo
	s_lshl_b32 s62, s52, 9
	s_add_u32 s62, s62, s55
	v_add_u32 v5, s62, v1
	ds_write_b32 v5, v15
	ds_write_b32 v5, v22 offset:256
	s_waitcnt lgkmcnt(0)
	s_barrier
	s_cmp_lg_u32 s52, 0
	s_cbranch_scc1 .Lstaged_done
	s_mov_b32 s62, 1
.Lgather:
	s_lshl_b32 s63, s62, 9
	s_add_u32 s63, s63, s55
	v_add_u32 v5, s63, v1
	ds_read_b32 v6, v5
	ds_read_b32 v7, v5 offset:256
	s_waitcnt lgkmcnt(0)
	v_add_u32 v15, v15, v6
	v_or_b32 v22, v22, v7
	s_add_u32 s62, s62, 1
	s_cmp_lt_u32 s62, s53
	s_cbranch_scc1 .Lgather
.Lsolo:
#endif
	s_mul_i32 s66, s76, s60
	s_mul_hi_u32 s67, s76, s60
	s_add_u32 s66, s66, s10
	s_addc_u32 s67, s67, s11
	global_store_dword v27, v16, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_store_dword v27, v17, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_store_dword v27, v18, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_store_dword v27, v19, s[66:67]
	s_mul_i32 s66, s77, s60
	s_mul_hi_u32 s67, s77, s60
	s_add_u32 s66, s66, s10
	s_addc_u32 s67, s67, s11
	global_store_dword v27, v20, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_store_dword v27, v21, s[66:67]
	// ood |= , counter += staticCount * nSamples + executed shadowed
	s_load_dwordx4 s[64:67], s[0:1], KA_OODROW            // oodRow countLo countHi staticCount
	s_waitcnt lgkmcnt(0)
	s_mul_i32 s68, s64, s60
	s_mul_hi_u32 s69, s64, s60
	s_add_u32 s68, s68, s10
	s_addc_u32 s69, s69, s11
	global_load_dword v2, v27, s[68:69]
	s_mul_i32 s70, s65, s60
	s_mul_hi_u32 s71, s65, s60
	s_add_u32 s70, s70, s10
	s_addc_u32 s71, s71, s11
	global_load_dword v3, v27, s[70:71]
	s_mul_i32 s20, s66, s60
	s_mul_hi_u32 s21, s66, s60
	s_add_u32 s20, s20, s10
	s_addc_u32 s21, s21, s11
	global_load_dword v4, v27, s[20:21]
	s_mul_i32 s22, s67, s9                                // staticCount * nSamples (low)
	s_mul_hi_u32 s23, s67, s9
	s_waitcnt vmcnt(2)
	v_or_b32 v2, v2, v22
	global_store_dword v27, v2, s[68:69]
	v_mov_b32 v6, s23
	s_waitcnt vmcnt(1)
	v_add_co_u32 v3, vcc, s22, v3
	s_nop 1
	v_addc_co_u32 v4, vcc, v4, v6, vcc
	v_add_co_u32 v3, vcc, v3, v15
	s_nop 1
	v_addc_co_u32 v4, vcc, 0, v4, vcc
	global_store_dword v27, v3, s[70:71]
	global_store_dword v27, v4, s[20:21]
	s_endpgm
#ifdef XLATE
.Lstaged_done:
	s_endpgm
#endif
.Lfunc_end0:
	.size	KNAME, .Lfunc_end0-KNAME


#ifdef XLATE
// ------------------------------------------------------------------------------------------ code hole
// The host writes the translated program (two streams of plain gfx950 code, fx_xlate.cpp) over this filler in
// its copy of the code object before loading it.
	.p2align	8
	.globl	HOLE
HOLE:
	.fill	(HOLE_BYTES / 4), 4, 0xbf800000                 // s_nop 0
	s_endpgm
#else
// ------------------------------------------------------------------------------------------ probe kernel
// One wavefront: writes the absolute address of every handler of this build (4 sets x 84 slots, 64-bit
// each) to `out`; the host puts these into the records it encodes (fx_asm.cpp asmHandlerTable). A kernel
// of its own so that it does not show up among the interpreter's launches in a kernel trace.
	.globl	PNAME
	.p2align	8
	.type	PNAME,@function
PNAME:
	s_load_dwordx2 s[14:15], s[0:1], KA_OUT
	s_getpc_b64 s[62:63]
.Lpc1:
	s_sub_u32 s62, s62, (.Lpc1-KNAME)                     // address of the interpreter's entry
	s_subb_u32 s63, s63, 0
	s_add_u32 s64, s62, (OTABLE-KNAME)
	s_addc_u32 s65, s63, 0
	v_mov_b32 v4, v0                                      // slot index handled by this lane
	s_waitcnt lgkmcnt(0)
.Lprobe_loop:
	v_cmp_gt_u32 vcc, 336, v4
	s_and_saveexec_b64 s[66:67], vcc
	s_cbranch_execz .Lprobe_done
	v_lshlrev_b32 v5, 2, v4
	global_load_dword v2, v5, s[64:65]
	v_mov_b32 v3, s63
	s_waitcnt vmcnt(0)
	v_add_co_u32 v2, vcc, s62, v2
	s_nop 1
	v_addc_co_u32 v3, vcc, 0, v3, vcc
	v_lshlrev_b32 v5, 3, v4
	global_store_dwordx2 v5, v[2:3], s[14:15]
	v_add_u32 v4, 64, v4
	s_branch .Lprobe_loop
.Lprobe_done:
	s_waitcnt vmcnt(0)
	s_endpgm
.Lfunc_end1:
	.size	PNAME, .Lfunc_end1-PNAME
#endif

	.rodata
	.p2align	6, 0x0
	.amdhsa_kernel KNAME
		.amdhsa_group_segment_fixed_size 0
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size KA_SIZE
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr NVGPR
#ifdef XLATE
		.amdhsa_next_free_sgpr 96
#else
		.amdhsa_next_free_sgpr 100
#endif
		.amdhsa_accum_offset NVGPR
		.amdhsa_reserve_vcc 1
		.amdhsa_float_round_mode_32 0
		.amdhsa_float_round_mode_16_64 0
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
	.end_amdhsa_kernel
#ifndef XLATE
	.p2align	6, 0x0
	.amdhsa_kernel PNAME
		.amdhsa_group_segment_fixed_size 0
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size KA_SIZE
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr 8
		.amdhsa_next_free_sgpr 72
		.amdhsa_accum_offset 8
		.amdhsa_reserve_vcc 1
	.end_amdhsa_kernel
#endif

	.amdgpu_metadata
---
amdhsa.kernels:
  - .args:
      - .offset: 0
        .size: 208
        .value_kind: by_value
    .group_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .kernarg_segment_size: 208
#ifdef XLATE
    .max_flat_workgroup_size: 1024
#else
    .max_flat_workgroup_size: 64
#endif
    .name: KNAME
    .private_segment_fixed_size: 0
    .sgpr_count: 102
    .symbol: KNAME.kd
    .vgpr_count: NVGPR
    .wavefront_size: 64
#ifndef XLATE
  - .args:
      - .offset: 0
        .size: 208
        .value_kind: by_value
    .group_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .kernarg_segment_size: 208
    .max_flat_workgroup_size: 64
    .name: PNAME
    .private_segment_fixed_size: 0
    .sgpr_count: 80
    .symbol: PNAME.kd
    .vgpr_count: 8
    .wavefront_size: 64
#endif
amdhsa.target: amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
	.end_amdgpu_metadata
