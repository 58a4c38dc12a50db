// fx_interp_gfx950.s — hand-written CDNA4 (gfx950) interpreter for the FX8010 opcode stream.
//
// One wavefront = 64 emulated DSPs (one per lane), one workgroup = one wavefront.  The host
// decoder (fx_asm_stream.cpp) turns the program into 32-byte records; this kernel is a
// direct-threaded interpreter over them:
//
//   record  w0 = handler slot * 4 (byte offset into the branch table below)
//           w1 = LDS byte offset of the destination row R
//           w2/w3/w4 = operand A/X/Y: LDS byte offset of its row, or the IEEE bits of a uniform value
//           w5 = flags for the generic handlers (bit0 A uniform, bit1 X uniform, bit2 Y uniform, bit3 write CCR)
//                / byte offset of the LOG/EXP segment table
//           w6:w7 = (1.0 - X) as a double for INTERP with a uniform X
//
//   fetch    one s_load_dwordx8 per record, issued one record ahead (s[24:31]) while s[16:23] executes
//   dispatch s_setpc_b64 into a table of s_branch — no compare chain, the record names its handler;
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
//   -DKNAME=fx_interp_v64  -DRF_VGPR -DNVGPR=64    register file in VGPRs v40.. (VGPR index mode, M0),
//   -DKNAME=fx_interp_v128 -DRF_VGPR -DNVGPR=128   no LDS traffic and no operand latency at all;
//   -DKNAME=fx_interp_v256 -DRF_VGPR -DNVGPR=256   24 / 88 / 216 rows at 8 / 4 / 2 wavefronts per SIMD
// In the VGPR build a "row offset" in a record is simply the row index.
#ifndef KNAME
#define KNAME fx_interp_lds
#endif
#ifndef NVGPR
#define NVGPR 40
#endif

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
	.set KA_SIZE,     0xb8

// ---- out-of-domain flag bits (fx_kernel.hpp) ----
	.set OOD_TRAM_READ_NEG, 1
	.set OOD_TRAM_WRITE_OOB, 2
	.set OOD_TRAM_SIZE0, 4
	.set OOD_LUT_INDEX, 16

// ---- VGPRs ----
//  v0 lane  v1 lane*4  v2 a/result  v3 x  v4 y  v5 address  v6-v13 temporaries
//  v14 numSkip  v15 executed-shadowed count  v16 iw  v17 ir  v18 xw  v19 xr  v20 g1  v21 g2  v22 ood
//  v23-v26 prefetched input of channel 0-3   v27 instance*4 (byte offset into a PCM / state row)
//  v28 0x80000000  v29 20.0  v30 16.0  v31 6.0  v32 8.0   v33-v39 temporaries
// ---- SGPRs ----
//  s[0:1] kernarg  s2 wave index  s3 sample  s[4:5] stream of this sample  s[6:7] steady  s8 fetch offset
//  s9 nSamples  s[10:11] state  s[12:13] in (this sample)  s[14:15] out (this sample)
//  s[16:23] current record  s[24:31] next record  s[32:33] branch table  s[34:35] jump target
//  s[36:37] iTRAM of this wave  s[38:39] xTRAM of this wave  s[40:41] LUT  s[42:43] last-sample stream
//  s44 channels  s45 bytes per sample (channels*N*4)  s46 iSlots  s47 xSlots  s[48:51] input row offsets
//  s[52:55] latch row offsets  s56 iSize  s57 xSize  s[58:59] lanes with instance < N  s60 state row pitch (nPad*4)
//  s61 byte offset of x1[] in the LUT blob  s62-s67 temporaries  s68 bytes per channel-sample (N*4)
//  s[72:73] row table  s82 nLoad  s83 nStore  s88 cursor state row  s89 LFSR state row

	.text
	.globl	KNAME
	.p2align	8
	.type	KNAME,@function

// ------------------------------------------------------------------------------------------ macros
// NEXT: go to the next record.  The record prefetch (SMEM) shares lgkmcnt with LDS and returns out of
// order, so it must have been drained by a full lgkmcnt(0) since it was issued.  Handlers that waited
// for LDS operands already did that and use NEXT (their own ds_write may stay in flight: LDS executes
// a wave's accesses in order); handlers with no such wait use NEXT_W.
.macro NEXT_W
#ifndef RF_VGPR
	s_waitcnt lgkmcnt(0)
#endif
	NEXT
.endm

// register-file access: row whose offset (LDS build) / index (VGPR build) is in \sreg
.macro LOADV dst, sreg
#ifdef RF_VGPR
	s_set_gpr_idx_on \sreg, 1
	v_mov_b32 \dst, v40
	s_set_gpr_idx_off
#else
	v_add_u32 \dst, \sreg, v1
	ds_read_b32 \dst, \dst
#endif
.endm
.macro STOREV sreg, src
#ifdef RF_VGPR
	s_set_gpr_idx_on \sreg, 8
	v_mov_b32 v40, \src
	s_set_gpr_idx_off
#else
	v_add_u32 v5, \sreg, v1
	ds_write_b32 v5, \src
#endif
.endm
// wait for LOADV results (LDS build only; VGPR loads are plain moves)
.macro WAITOPS
#ifndef RF_VGPR
	s_waitcnt lgkmcnt(0)
#endif
.endm

.macro NEXT
#ifdef RF_VGPR
	s_waitcnt lgkmcnt(0)
#endif
	s_mov_b64 s[16:17], s[24:25]
	s_mov_b64 s[18:19], s[26:27]
	s_mov_b64 s[20:21], s[28:29]
	s_mov_b64 s[22:23], s[30:31]
	s_load_dwordx8 s[24:31], s[4:5], s8
	s_add_u32 s8, s8, 32
	s_add_u32 s34, s32, s16
	s_addc_u32 s35, s33, 0
	s_setpc_b64 s[34:35]
.endm

// x86 cvttss2si: truncation, 0x80000000 for NaN and anything outside int32
.macro CVTT dst, src
	v_cvt_i32_f32 \dst, \src
	v_cmp_ngt_f32 vcc, 0x4f000000, \src
	s_nop 1
	v_cndmask_b32 \dst, \dst, v28, vcc
.endm

// reference saturate(x, 1.0f) (FX8010.cpp:275-279); NaN passes through
.macro SAT reg
	v_med3_f32 v5, \reg, -1.0, 1.0
	v_cmp_u_f32 vcc, \reg, \reg
	s_nop 1
	v_cndmask_b32 \reg, v5, \reg, vcc
.endm

// reference wrapAround (FX8010.cpp:299-328): v >= 1 ? v-2 : (v < -1 ? v+2 : v)
.macro WRAP reg
	v_add_f32 v5, -2.0, \reg
	v_add_f32 v6, 2.0, \reg
	v_cmp_gt_f32 vcc, -1.0, \reg
	s_nop 1
	v_cndmask_b32 v6, \reg, v6, vcc
	v_cmp_le_f32 vcc, 1.0, \reg
	s_nop 1
	v_cndmask_b32 \reg, v6, v5, vcc
.endm

// reference setCCR (FX8010.cpp:211-232): CCR row (row 0) <- flags of \reg
.macro CCR_FROM reg
	v_mov_b32 v6, 0
	v_cmp_eq_f32 vcc, -1.0, \reg
	v_cmp_eq_f32 s[62:63], 1.0, \reg
	v_cmp_eq_f32 s[64:65], 0, \reg
	v_cndmask_b32 v6, v6, v29, vcc
	v_cndmask_b32 v6, v6, v30, s[62:63]
	v_cndmask_b32 v6, v6, v32, s[64:65]
	v_cmp_lt_f32 vcc, 0, \reg
	v_cmp_gt_f32 s[62:63], 1.0, \reg
	v_cmp_gt_f32 s[64:65], 0, \reg
	v_cmp_lt_f32 s[66:67], -1.0, \reg
	s_and_b64 vcc, vcc, s[62:63]
	s_and_b64 s[64:65], s[64:65], s[66:67]
	v_cndmask_b32 v6, v6, 2.0, vcc
	v_cndmask_b32 v6, v6, v31, s[64:65]
#ifdef RF_VGPR
	v_mov_b32 v40, v6
#else
	ds_write_b32 v1, v6
#endif
.endm

// result in v2 -> row R
.macro STORE_R
	STOREV s17, v2
.endm

// operand fetch of the specialised handlers: \kind bit0/1/2 = A/X/Y is uniform
.macro FETCH3 kind
#ifdef RF_VGPR
	.if ((\kind) & 1) == 0
	s_set_gpr_idx_on s18, 1
	v_mov_b32 v2, v40
	.endif
	.if ((\kind) & 2) == 0
	s_set_gpr_idx_on s19, 1
	v_mov_b32 v3, v40
	.endif
	.if ((\kind) & 4) == 0
	s_set_gpr_idx_on s20, 1
	v_mov_b32 v4, v40
	.endif
	.if (\kind) != 7
	s_set_gpr_idx_off
	.endif
	.if ((\kind) & 1)
	v_mov_b32 v2, s18
	.endif
	.if ((\kind) & 2)
	v_mov_b32 v3, s19
	.endif
	.if ((\kind) & 4)
	v_mov_b32 v4, s20
	.endif
#else
	.if ((\kind) & 1) == 0
	v_add_u32 v2, s18, v1
	ds_read_b32 v2, v2
	.else
	v_mov_b32 v2, s18
	.endif
	.if ((\kind) & 2) == 0
	v_add_u32 v3, s19, v1
	ds_read_b32 v3, v3
	.else
	v_mov_b32 v3, s19
	.endif
	.if ((\kind) & 4) == 0
	v_add_u32 v4, s20, v1
	ds_read_b32 v4, v4
	.else
	v_mov_b32 v4, s20
	.endif
	.if (\kind) != 7
	s_waitcnt lgkmcnt(0)
	.endif
#endif
.endm

// operand fetch of the generic handlers: kinds from the flag word w5 (s21)
.macro GFETCH reg, sreg, bit
	v_mov_b32 \reg, \sreg
	s_bitcmp1_b32 s21, \bit
	s_cbranch_scc1 .Lgf\@
	LOADV \reg, \sreg
.Lgf\@:
.endm

// generic result store: row R, and CCR when flag bit 3 is set
.macro GSTORE
	STORE_R
	s_bitcmp0_b32 s21, 3
	s_cbranch_scc1 .Lgs\@
	CCR_FROM v2
.Lgs\@:
.endm

.macro HOT_MACS kind, ccr, neg
	FETCH3 \kind
	v_mul_f32 v3, v3, v4
	.if \neg
	v_sub_f32 v2, v2, v3
	.else
	v_add_f32 v2, v2, v3
	.endif
	SAT v2
	STORE_R
	.if \ccr
	CCR_FROM v2
	.endif
	.if (\kind) == 7
	NEXT_W
	.else
	NEXT
	.endif
.endm

.macro HOT_ACC3 kind, ccr
	FETCH3 \kind
	v_add_f32 v2, v2, v3
	v_add_f32 v2, v2, v4
	SAT v2
	STORE_R
	.if \ccr
	CCR_FROM v2
	.endif
	.if (\kind) == 7
	NEXT_W
	.else
	NEXT
	.endif
.endm

// INTERP (FX8010.cpp:1180-1187): R = sat((float)((1.0 - (double)X) * (double)A + (double)(X*Y)))
.macro HOT_INTERP kind, ccr
	FETCH3 \kind
	v_mul_f32 v4, v3, v4
	v_cvt_f64_f32 v[8:9], v2
	.if ((\kind) & 2)
	v_mul_f64 v[6:7], s[22:23], v[8:9]
	.else
	v_cvt_f64_f32 v[6:7], v3
	v_add_f64 v[6:7], 1.0, -v[6:7]
	v_mul_f64 v[6:7], v[6:7], v[8:9]
	.endif
	v_cvt_f64_f32 v[8:9], v4
	v_add_f64 v[6:7], v[6:7], v[8:9]
	v_cvt_f32_f64 v2, v[6:7]
	SAT v2
	STORE_R
	.if \ccr
	CCR_FROM v2
	.endif
	.if (\kind) == 7
	NEXT_W
	.else
	NEXT
	.endif
.endm

// TRAM read (FX8010.cpp:934-967): row R <- buffer[(rpos - p) % size]; rpos = (rpos+1) % size
.macro TRAM_READ size, slots, baselo, basehi, cur
	GFETCH v4, s20, 2
	WAITOPS
	v_mov_b32 v2, 0
	s_cmp_lt_i32 \size, 1
	s_cbranch_scc1 .Ltr0\@
	CVTT v6, v4
	s_sub_i32 s62, \size, 1
	v_min_i32 v6, s62, v6
	v_max_i32 v6, 0, v6
	v_sub_u32 v6, \cur, v6
	v_cmp_gt_i32 vcc, 0, v6
	v_add_u32 v7, \size, v6
	v_add_u32 v8, 1, \cur
	v_cndmask_b32 v6, v6, v7, vcc
	v_cndmask_b32 v7, 0, 1, vcc
	v_or_b32 v22, v22, v7
	v_cmp_gt_i32 vcc, \slots, v6
	v_lshlrev_b32 v7, 8, v6
	v_add_u32 v7, v7, v1
	s_and_saveexec_b64 s[64:65], vcc
	global_load_dword v2, v7, s[\baselo:\basehi]
	s_mov_b64 exec, s[64:65]
	v_cmp_le_i32 vcc, \size, v8
	s_nop 1
	v_cndmask_b32 \cur, v8, 0, vcc
	s_waitcnt vmcnt(0)
	s_branch .Ltr1\@
.Ltr0\@:
	v_or_b32 v22, OOD_TRAM_SIZE0, v22
.Ltr1\@:
	STORE_R
	NEXT
.endm

// TRAM write (FX8010.cpp:909-927): buffer[wpos + p] <- A (no modulo); wpos = (wpos+1) % size
.macro TRAM_WRITE size, slots, cap, baselo, basehi, cur
	GFETCH v2, s18, 0
	GFETCH v4, s20, 2
	WAITOPS
	s_cmp_lt_i32 \size, 1
	s_cbranch_scc1 .Ltw0\@
	CVTT v6, v4
	s_sub_i32 s62, \size, 1
	v_min_i32 v6, s62, v6
	v_max_i32 v6, 0, v6
	v_add_u32 v6, \cur, v6
	s_min_i32 s63, \slots, \cap
	v_add_u32 v8, 1, \cur
	v_cmp_gt_i32 vcc, s63, v6
	v_lshlrev_b32 v7, 8, v6
	v_add_u32 v7, v7, v1
	v_cndmask_b32 v9, OOD_TRAM_WRITE_OOB, 0, vcc
	v_or_b32 v22, v22, v9
	s_and_saveexec_b64 s[64:65], vcc
	global_store_dword v7, v2, s[\baselo:\basehi]
	s_mov_b64 exec, s[64:65]
	v_cmp_le_i32 vcc, \size, v8
	s_nop 1
	v_cndmask_b32 \cur, v8, 0, vcc
	s_branch .Ltw1\@
.Ltw0\@:
	v_or_b32 v22, OOD_TRAM_SIZE0, v22
.Ltw1\@:
	NEXT
.endm

.macro JT2 op, kind
	s_branch h_\op\()_\kind\()_0
	s_branch h_\op\()_\kind\()_1
.endm

.macro DEF_HOT kind
h_macs_\kind\()_0:
	HOT_MACS \kind, 0, 0
h_macs_\kind\()_1:
	HOT_MACS \kind, 1, 0
h_macsn_\kind\()_0:
	HOT_MACS \kind, 0, 1
h_macsn_\kind\()_1:
	HOT_MACS \kind, 1, 1
h_acc3_\kind\()_0:
	HOT_ACC3 \kind, 0
h_acc3_\kind\()_1:
	HOT_ACC3 \kind, 1
h_interp_\kind\()_0:
	HOT_INTERP \kind, 0
h_interp_\kind\()_1:
	HOT_INTERP \kind, 1
.endm

// ------------------------------------------------------------------------------------------ entry
KNAME:
	s_load_dwordx16 s[4:19], s[0:1], KA_STEADY            // steady last rowtab state in out itram xtram
	s_load_dwordx8  s[40:47], s[0:1], KA_LUT              // lut n npad nload nstore
	s_load_dwordx2  s[80:81], s[0:1], KA_NSAMPLES         // nSamples channels
	v_lshlrev_b32 v1, 2, v0
	s_lshl_b32 s62, s2, 6
	v_add_u32 v27, s62, v0                                // instance
	v_mov_b32 v28, 0x80000000
	v_mov_b32 v29, 0x41a00000                             // 20.0
	v_mov_b32 v30, 0x41800000                             // 16.0
	v_mov_b32 v31, 0x40c00000                             // 6.0
	v_mov_b32 v32, 0x41000000                             // 8.0
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
	s_mov_b32 s9, s80                                     // nSamples
	s_mov_b32 s82, s46                                    // nLoad
	s_mov_b32 s83, s47                                    // nStore
	s_mov_b32 s44, s81                                    // channels
	s_mul_i32 s45, s42, s44
	s_lshl_b32 s45, s45, 2                                // bytes per sample of PCM = channels * N * 4
	s_mov_b64 s[42:43], s[70:71]                          // last-sample stream
	s_load_dwordx8  s[48:55], s[0:1], KA_INOFF            // inOff[4] latchOff[4]
	s_load_dwordx4  s[84:87], s[0:1], KA_ISLOTS           // iSlots xSlots iSize xSize
	s_load_dwordx2  s[88:89], s[0:1], KA_CURSORROW        // cursorRow noiseRow
	s_load_dword    s61, s[0:1], KA_LUTX1OFF
	s_waitcnt lgkmcnt(0)
	s_mov_b32 s46, s84
	s_mov_b32 s47, s85
	s_mov_b32 s56, s86
	s_mov_b32 s57, s87
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
	// branch table address
	s_getpc_b64 s[32:33]
.Lpc0:
	s_add_u32 s32, s32, (jump_table-.Lpc0)
	s_addc_u32 s33, s33, 0

	// ---- prologue: state rows -> LDS rows
	s_mov_b32 s62, 0
	s_cmp_eq_u32 s82, 0
	s_cbranch_scc1 .Lload_done
.Lload_loop:
	s_lshl_b32 s63, s62, 2
	s_load_dword s64, s[72:73], s63                       // ldsRow | stateRow << 16
	s_waitcnt lgkmcnt(0)
	s_lshr_b32 s65, s64, 16
	s_and_b32 s64, s64, 0xffff
	s_mul_i32 s66, s65, s60
	s_mul_hi_u32 s67, s65, s60
	s_add_u32 s66, s66, s10
	s_addc_u32 s67, s67, s11
	global_load_dword v2, v27, s[66:67]
#ifndef RF_VGPR
	s_lshl_b32 s64, s64, 8
#endif
	s_waitcnt vmcnt(0)
	STOREV s64, v2
	s_add_u32 s62, s62, 1
	s_cmp_lt_u32 s62, s82
	s_cbranch_scc1 .Lload_loop
.Lload_done:
	// cursors and LFSR words straight into VGPRs
	s_mul_i32 s66, s88, s60
	s_mul_hi_u32 s67, s88, s60
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
	s_mul_i32 s66, s89, s60
	s_mul_hi_u32 s67, s89, s60
	s_add_u32 s66, s66, s10
	s_addc_u32 s67, s67, s11
	global_load_dword v20, v27, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_load_dword v21, v27, s[66:67]
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
	s_load_dwordx8 s[16:23], s[4:5], 0x0
	s_load_dwordx8 s[24:31], s[4:5], 0x20
	s_mov_b32 s8, 64
	s_waitcnt lgkmcnt(0)
	s_add_u32 s34, s32, s16
	s_addc_u32 s35, s33, 0
	s_setpc_b64 s[34:35]

// ------------------------------------------------------------------------------------------ handlers
	.p2align 6
jump_table:
	s_branch h_endsample      // 0
	s_branch h_nop            // 1
	s_branch h_pred           // 2
	s_branch h_unpred         // 3
	s_branch h_mov            // 4
	s_branch h_macw           // 5
	s_branch h_macwn          // 6
	s_branch h_macintw        // 7
	s_branch h_andxor         // 8
	s_branch h_tstneg         // 9
	s_branch h_limit          // 10
	s_branch h_limitn         // 11
	s_branch h_lut            // 12
	s_branch h_skip           // 13
	s_branch h_tram_ir        // 14
	s_branch h_tram_iw        // 15
	s_branch h_tram_xr        // 16
	s_branch h_tram_xw        // 17
	s_branch h_noise          // 18
	s_branch h_nop            // 19 (reserved)
	// 20..35 MACS, 36..51 MACSN, 52..67 ACC3, 68..83 INTERP: slot = base + kind*2 + ccr
	.irp kind, 0, 1, 2, 3, 4, 5, 6, 7
	JT2 macs, \kind
	.endr
	.irp kind, 0, 1, 2, 3, 4, 5, 6, 7
	JT2 macsn, \kind
	.endr
	.irp kind, 0, 1, 2, 3, 4, 5, 6, 7
	JT2 acc3, \kind
	.endr
	.irp kind, 0, 1, 2, 3, 4, 5, 6, 7
	JT2 interp, \kind
	.endr

h_nop:
	NEXT_W

h_unpred:
	s_mov_b64 exec, -1
	NEXT_W

// start of an instruction inside a SKIP shadow (FX8010.cpp:1037,1235-1241): lanes with numSkip == 0
// execute it (EXEC), the others count their skip down
h_pred:
	s_mov_b64 exec, -1
	v_cmp_eq_u32 vcc, 0, v14
	v_max_i32 v5, 1, v14
	v_add_u32 v14, -1, v5
	v_cndmask_b32 v5, 0, 1, vcc
	v_add_u32 v15, v15, v5
	s_mov_b64 exec, vcc
	NEXT_W

h_mov:
	GFETCH v2, s18, 0
	WAITOPS
	GSTORE
	NEXT

h_macw:                                                   // R = A + wrap(X*Y)   FX8010.cpp:1126-1131
	GFETCH v2, s18, 0
	GFETCH v3, s19, 1
	GFETCH v4, s20, 2
	WAITOPS
	v_mul_f32 v3, v3, v4
	WRAP v3
	v_add_f32 v2, v2, v3
	GSTORE
	NEXT

h_macwn:                                                  // R = A - wrap(X*Y)   :1132-1137
	GFETCH v2, s18, 0
	GFETCH v3, s19, 1
	GFETCH v4, s20, 2
	WAITOPS
	v_mul_f32 v3, v3, v4
	WRAP v3
	v_sub_f32 v2, v2, v3
	GSTORE
	NEXT

h_macintw:                                                // R = wrap(A + X*Y)   :1138-1143
	GFETCH v2, s18, 0
	GFETCH v3, s19, 1
	GFETCH v4, s20, 2
	WAITOPS
	v_mul_f32 v3, v3, v4
	v_add_f32 v2, v2, v3
	WRAP v2
	GSTORE
	NEXT

h_andxor:                                                 // logicOps, FX8010.cpp:330-360, :1150-1154
	GFETCH v2, s18, 0
	GFETCH v3, s19, 1
	GFETCH v4, s20, 2
	WAITOPS
	CVTT v8, v2                                           // A
	CVTT v9, v3                                           // X
	CVTT v10, v4                                          // Y
	v_and_b32 v2, v8, v9
	v_xor_b32 v2, v2, v10                                 // (A & X) ^ Y
	v_not_b32 v11, v8                                     // ~A
	v_not_b32 v12, v9                                     // ~X
	v_mov_b32 v13, 0xffffff
	v_cmp_eq_u32 vcc, v10, v13                            // Y == 0xFFFFFF
	v_and_b32 v33, v11, v9                                // ~A & X
	s_nop 0
	v_cndmask_b32 v2, v2, v33, vcc
	v_cmp_eq_u32 s[62:63], v10, v12                       // Y == ~X
	v_or_b32 v33, v8, v10
	s_nop 0
	v_cndmask_b32 v2, v2, v33, s[62:63]
	v_mov_b32 v34, 0xfffffff
	v_cmp_eq_u32 s[62:63], v9, v34                        // X == 0xFFFFFFF
	s_and_b64 s[62:63], s[62:63], vcc                     //   && Y == 0xFFFFFF
	v_cndmask_b32 v2, v2, v11, s[62:63]
	v_cmp_eq_u32 vcc, v9, v13                             // X == 0xFFFFFF
	v_xor_b32 v33, v8, v10
	s_nop 0
	v_cndmask_b32 v2, v2, v33, vcc
	v_cmp_eq_u32 vcc, 0, v10                              // Y == 0
	v_and_b32 v33, v8, v9
	s_nop 0
	v_cndmask_b32 v2, v2, v33, vcc
	v_cvt_f32_i32 v2, v2
	GSTORE
	NEXT

h_tstneg:                                                 // R = A >= Y ? X : intToFloat(~floatToInt(X))   :1155-1162
	GFETCH v2, s18, 0
	GFETCH v3, s19, 1
	GFETCH v4, s20, 2
	WAITOPS
	v_mul_f32 v6, 0x4f000000, v3                          // X * 2^31
	CVTT v7, v6
	v_not_b32 v7, v7
	v_cvt_f32_i32 v7, v7
	v_mul_f32 v7, 0x30000000, v7                          // exact / 2^31
	v_cmp_ge_f32 vcc, v2, v4
	s_nop 1
	v_cndmask_b32 v2, v7, v3, vcc
	GSTORE
	NEXT

h_limit:                                                  // R = A >= Y ? X : Y   :1163-1168
	GFETCH v2, s18, 0
	GFETCH v3, s19, 1
	GFETCH v4, s20, 2
	WAITOPS
	v_cmp_ge_f32 vcc, v2, v4
	s_nop 1
	v_cndmask_b32 v2, v4, v3, vcc
	GSTORE
	NEXT

h_limitn:                                                 // R = A < Y ? X : Y   :1169-1174
	GFETCH v2, s18, 0
	GFETCH v3, s19, 1
	GFETCH v4, s20, 2
	WAITOPS
	v_cmp_lt_f32 vcc, v2, v4
	s_nop 1
	v_cndmask_b32 v2, v4, v3, vcc
	GSTORE
	NEXT

// LOG / EXP with a uniform table (FX8010.cpp:1113-1125, linearInterpolate :283-296) from the
// precomputed thresholds and segments (fx_model.hpp LutDevice): w3 = byte offset of the table's
// {slope, y1} array in the blob.  idx = #{k >= 1 : t >= thr[k]}, guessed from t*31.5 and corrected.
h_lut:
	GFETCH v2, s18, 0
	WAITOPS
	v_cvt_f64_f32 v[6:7], v2                              // x
	v_add_f64 v[8:9], v[6:7], 1.0                         // t = x - -1.0
	v_mov_b32 v10, 0
	v_mov_b32 v11, 0x403f8000                             // 31.5
	v_mul_f64 v[10:11], v[8:9], v[10:11]
	v_cvt_i32_f64 v12, v[10:11]                           // saturating, NaN -> 0
	v_med3_i32 v12, v12, 0, 63
	v_lshlrev_b32 v13, 3, v12
	global_load_dwordx4 v[34:37], v13, s[40:41]           // thr[idx], thr[idx+1]
	s_waitcnt vmcnt(0)
	v_cmp_ge_f64 vcc, v[8:9], v[36:37]
	v_cmp_lt_f64 s[62:63], v[8:9], v[34:35]
	s_nop 0
	v_cndmask_b32 v13, 0, 1, vcc
	v_add_u32 v12, v12, v13
	v_cndmask_b32 v13, 0, 1, s[62:63]
	v_sub_u32 v12, v12, v13
	v_med3_i32 v12, v12, 0, 63
	// out-of-domain note: t outside [-step, 64*step) or NaN
	v_mov_b32 v10, 0x10410410
	v_mov_b32 v11, 0xbfa04104                             // -step = -2/63
	v_cmp_nge_f64 vcc, v[8:9], v[10:11]
	v_mov_b32 v10, 0x10410410
	v_mov_b32 v11, 0x40004104                             // 64*step
	v_cmp_nlt_f64 s[62:63], v[8:9], v[10:11]
	s_or_b64 vcc, vcc, s[62:63]
	v_cndmask_b32 v13, 0, OOD_LUT_INDEX, vcc
	v_or_b32 v22, v22, v13
	v_lshlrev_b32 v13, 3, v12
	v_add_u32 v13, s61, v13
	global_load_dwordx2 v[38:39], v13, s[40:41]           // x1[idx]
	v_lshlrev_b32 v13, 4, v12
	v_add_u32 v13, s19, v13
	global_load_dwordx4 v[34:37], v13, s[40:41]           // slope, y1
	s_waitcnt vmcnt(1)
	v_add_f64 v[6:7], v[6:7], -v[38:39]                   // x - x1
	s_waitcnt vmcnt(0)
	v_mul_f64 v[6:7], v[34:35], v[6:7]
	v_add_f64 v[6:7], v[6:7], v[36:37]
	v_cvt_f32_f64 v2, v[6:7]
	GSTORE
	NEXT

h_skip:                                                   // if ((float)(int)X == CCR) numSkip = (int)Y   :1175-1179
	GFETCH v3, s19, 1
	GFETCH v4, s20, 2
#ifdef RF_VGPR
	v_mov_b32 v6, v40                                     // CCR row
#else
	ds_read_b32 v6, v1                                    // CCR row
	s_waitcnt lgkmcnt(0)
#endif
	CVTT v7, v3
	v_cvt_f32_i32 v7, v7
	CVTT v8, v4
	v_cmp_eq_f32 vcc, v7, v6
	s_nop 1
	v_cndmask_b32 v14, v14, v8, vcc
	NEXT

h_tram_ir:
	TRAM_READ s56, s46, 36, 37, v17
h_tram_iw:
	TRAM_WRITE s56, s46, 8192, 36, 37, v16
h_tram_xr:
	TRAM_READ s57, s47, 38, 39, v19
h_tram_xw:
	TRAM_WRITE s57, s47, 1048576, 38, 39, v18

h_noise:                                                  // whitenoise(), FX8010.cpp:993-1000
	v_xor_b32 v20, v20, v21
	v_cvt_f32_i32 v2, v21
	v_mul_f32 v2, 0x30000000, v2                          // * 2^-31
	v_add_u32 v21, v21, v20
	STORE_R
	NEXT_W

	.irp kind, 0, 1, 2, 3, 4, 5, 6, 7
	DEF_HOT \kind
	.endr

// ---- end of the program for this sample: latch rows -> PCM out, next sample
h_endsample:
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
	s_add_u32 s3, s3, 1
	s_cmp_lt_i32 s3, s9
	s_waitcnt vmcnt(0)
	s_cbranch_scc1 .Lsample

	// ---- epilogue: LDS rows and VGPR state -> state rows
.Lepilogue:
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_mov_b32 s62, 0
	s_cmp_eq_u32 s83, 0
	s_cbranch_scc1 .Lstore_done
.Lstore_loop:
	s_add_u32 s63, s62, s82
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
	s_cmp_lt_u32 s62, s83
	s_cbranch_scc1 .Lstore_loop
.Lstore_done:
	s_mul_i32 s66, s88, s60
	s_mul_hi_u32 s67, s88, s60
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
	s_mul_i32 s66, s89, s60
	s_mul_hi_u32 s67, s89, s60
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
	s_mul_i32 s74, s66, s60
	s_mul_hi_u32 s75, s66, s60
	s_add_u32 s74, s74, s10
	s_addc_u32 s75, s75, s11
	global_load_dword v4, v27, s[74:75]
	s_mul_i32 s76, s67, s9                                // staticCount * nSamples (low)
	s_mul_hi_u32 s77, s67, s9
	s_waitcnt vmcnt(2)
	v_or_b32 v2, v2, v22
	global_store_dword v27, v2, s[68:69]
	v_mov_b32 v6, s77
	s_waitcnt vmcnt(1)
	v_add_co_u32 v3, vcc, s76, v3
	s_nop 1
	v_addc_co_u32 v4, vcc, v4, v6, vcc
	v_add_co_u32 v3, vcc, v3, v15
	s_nop 1
	v_addc_co_u32 v4, vcc, 0, v4, vcc
	global_store_dword v27, v3, s[70:71]
	global_store_dword v27, v4, s[74:75]
	s_endpgm
.Lfunc_end0:
	.size	KNAME, .Lfunc_end0-KNAME

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
		.amdhsa_next_free_sgpr 96
		.amdhsa_accum_offset NVGPR
		.amdhsa_reserve_vcc 1
		.amdhsa_float_round_mode_32 0
		.amdhsa_float_round_mode_16_64 0
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
	.end_amdhsa_kernel

	.amdgpu_metadata
---
amdhsa.kernels:
  - .args:
      - .offset: 0
        .size: 184
        .value_kind: by_value
    .group_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .kernarg_segment_size: 184
    .max_flat_workgroup_size: 64
    .name: KNAME
    .private_segment_fixed_size: 0
    .sgpr_count: 102
    .symbol: KNAME.kd
    .vgpr_count: NVGPR
    .wavefront_size: 64
amdhsa.target: amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
	.end_amdgpu_metadata
