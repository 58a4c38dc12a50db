// fx_pair_gfx950.S — frame of a translated FX8010 program that steps TWO instances per lane (gfx950, wave64).
//
// The translated code of fx_xlate.cpp is bound by the fp32 VALU pipe.  Half of its fp32 instructions are the
// multiply and the add of MACS-like operations, and gfx950 has packed forms of exactly those (v_pk_mul_f32,
// v_pk_add_f32: two fp32 results per lane at the cost of one).  This frame gives every lane two instances so
// that the translator can use them: a "pair wave" does the work of two ordinary wavefronts,
//     half 0 = instances  wg*128      + lane,      half 1 = instances  wg*128 + 64 + lane,
// i.e. exactly the instances, state rows, PCM columns and TRAM tiles of wavefronts 2*wg and 2*wg+1 of the
// one-instance-per-lane kernels (fx_interp_gfx950.S): every memory layout is unchanged, half 1 is simply
// +256 bytes (+ one TRAM tile).  Register-file row r lives in the VGPR pair v[32+2r : 33+2r] = {half 0, half 1},
// the operand form of the packed instructions.
//
// Only programs that translate without handler calls and without SKIP run here (fx_xlate.cpp decides), so this
// file has no handlers: prologue, per-sample frame, epilogue and the hole for the generated code.
//
// Build: -DKNAME=fx_pair_vNN -DNVGPR=NN.  Kernel arguments: struct AsmArgs (fx_asm.hpp), as the interpreter's.
#ifndef KNAME
#define KNAME fx_pair_v168
#endif
#ifndef NVGPR
#define NVGPR 168
#endif
#ifndef HOLE_BYTES
#define HOLE_BYTES 262144
#endif
#define FX_PASTE(a, b) a##b
#define FX_PASTE2(a, b) FX_PASTE(a, b)
#define HOLE FX_PASTE2(KNAME, _hole)
#define OTABLE FX_PASTE2(KNAME, _table)

	.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
	.amdhsa_code_object_version 6

	.set KA_STEADY,   0x00
	.set KA_LUT,      0x40
	.set KA_NSAMPLES, 0x60
	.set KA_INOFF,    0x68
	.set KA_ISLOTS,   0x88
	.set KA_CURSORROW,0x98
	.set KA_OODROW,   0xa0
	.set KA_INIT,     0xb4
	.set KA_SIZE,     0xb8

// ---- VGPRs ----
//  v0 lane  v1 lane*4  v2-v13 temporaries of the generated code
//  v14 / v15 out-of-domain flags of half 0 / 1   v16-v19 / v20-v23 prefetched input of channel 0-3, half 0 / 1
//  v24-v27 TRAM cursors (equal in every lane: programs with a TRAM instruction in a SKIP shadow do not run here)
//  v28 (instance of half 0)*4   v29 v_cmp_class mask (NaN, +-Inf)   v30, v31 spare   v32.. register file
// ---- SGPRs ----  (as fx_interp_gfx950.S where the generated code relies on them)
//  s[0:1] kernarg  s2 pair-wave index  s3 sample  s[4:5] stream entry  s[6:7] steady {fast, exact} offsets
//  s9 nSamples  s[10:11] state  s[12:13] in  s[14:15] out  s[16:17] lanes of half 1 with an instance
//  s18-s25 record words / scratch of the generated code  s26 / s27 bytes of one iTRAM / xTRAM tile
//  s[28:31] scratch of the generated code  s[32:33] kernel entry  s[34:35] end-of-sample frame
//  s[36:37] / s[38:39] iTRAM / xTRAM tile of half 0  s[40:41] LUT  s[42:43] last-sample {fast, exact} offsets
//  s44 channels  s45 bytes per sample of PCM  s46/s47 iSlots/xSlots  s[48:51] input rows  s[52:55] latch rows
//  s56/s57 iSize/xSize  s[58:59] lanes of half 0 with an instance  s60 state row pitch  s62-s67, s69-s71 temporaries
//  s68 bytes per channel-sample (N*4)  s[72:73] row table  s74/s75 nLoad/nStore  s76 cursor state row
//  s[78:79] taint  s80-s93 generated code (TRAM cursors, LUT bases)

	.text
	.globl	KNAME
	.p2align	8
	.type	KNAME,@function

// taint (fx_xlate.hpp): a BOUNDED row (s69 != 0) must hold a value inside [-1, 1], any other a finite one
.macro TAINT_ROW reg
	v_cmp_class_f32 vcc, \reg, v29
	s_cmp_eq_u32 s69, 0
	s_cbranch_scc1 .Ltr\@
	v_cmp_nle_f32_e64 vcc, |\reg|, 1.0
.Ltr\@:
	s_or_b64 s[78:79], s[78:79], vcc
.endm
.macro TAINT_FINITE reg
	v_cmp_class_f32 vcc, \reg, v29
	s_or_b64 s[78:79], s[78:79], vcc
.endm

// PCM of one channel of one sample for both halves; s[62:63] = address of the channel's row, advanced to the next
.macro IN_CH chan, v0reg, v1reg, offreg
	s_cmp_lt_i32 s44, \chan + 1
	s_cbranch_scc1 .Linskip\@
	s_cmp_lt_i32 \offreg, 0
	s_cbranch_scc1 .Linnone\@
	s_mov_b64 exec, s[58:59]
	global_load_dword \v0reg, v28, s[62:63]
	s_mov_b64 exec, s[16:17]
	global_load_dword \v1reg, v28, s[62:63] offset:256
.Linnone\@:
	s_add_u32 s62, s62, s68
	s_addc_u32 s63, s63, 0
.Linskip\@:
.endm
.macro INPUT_LOADS
	IN_CH 0, v16, v20, s48
	IN_CH 1, v17, v21, s49
	IN_CH 2, v18, v22, s50
	IN_CH 3, v19, v23, s51
	s_mov_b64 exec, -1
.endm
// this sample's input of one channel -> its register-file row (both halves)
.macro IN_ROW chan, v0reg, v1reg, offreg
	s_cmp_lt_i32 s44, \chan + 1
	s_cbranch_scc1 .Lrowskip\@
	s_cmp_lt_i32 \offreg, 0
	s_cbranch_scc1 .Lrowskip\@
	s_lshl_b32 s62, \offreg, 1
	s_set_gpr_idx_on s62, 8
	v_mov_b32 v32, \v0reg
	v_mov_b32 v33, \v1reg
	s_set_gpr_idx_off
.Lrowskip\@:
.endm
// output latch row of one channel -> PCM (both halves); s[62:63] = address of the channel's row, advanced
.macro OUT_CH chan, offreg
	s_cmp_lt_i32 s44, \chan + 1
	s_cbranch_scc1 .Loutskip\@
	s_lshl_b32 s64, \offreg, 1
	s_set_gpr_idx_on s64, 1
	v_mov_b32 v2, v32
	v_mov_b32 v3, v33
	s_set_gpr_idx_off
	s_nop 3
	s_mov_b64 exec, s[58:59]
	global_store_dword v28, v2, s[62:63]
	s_mov_b64 exec, s[16:17]
	global_store_dword v28, v3, s[62:63] offset:256
	s_mov_b64 exec, -1
	s_add_u32 s62, s62, s68
	s_addc_u32 s63, s63, 0
.Loutskip\@:
.endm
// s[66:67] = address of state row \rowreg
.macro STATE_ROW rowreg
	s_mul_i32 s66, \rowreg, s60
	s_mul_hi_u32 s67, \rowreg, s60
	s_add_u32 s66, s66, s10
	s_addc_u32 s67, s67, s11
.endm

// ------------------------------------------------------------------------------------------ entry
KNAME:
	s_getpc_b64 s[32:33]                                  // KNAME + 4
	s_load_dwordx16 s[4:19], s[0:1], KA_STEADY            // steady last rowtab state in out itram xtram
	s_load_dwordx8  s[40:47], s[0:1], KA_LUT              // lut n npad nload nstore
	s_load_dwordx2  s[64:65], s[0:1], KA_NSAMPLES         // nSamples channels
	v_lshlrev_b32 v1, 2, v0
	s_lshl_b32 s62, s2, 7
	v_add_u32 v28, s62, v0                                // instance of half 0
	v_add_u32 v2, 64, v28                                 // instance of half 1
	v_mov_b32 v14, 0
	v_mov_b32 v15, 0
	v_mov_b32 v29, 0x207                                  // v_cmp_class mask: sNaN | qNaN | -Inf | +Inf
	s_waitcnt lgkmcnt(0)
	s_mov_b64 s[72:73], s[8:9]                            // row table
	s_mov_b64 s[36:37], s[16:17]                          // itram
	s_mov_b64 s[38:39], s[18:19]                          // xtram
	s_mov_b64 s[70:71], s[6:7]                            // last-sample stream offsets
	s_mov_b64 s[6:7], s[4:5]                              // steady stream offsets
	v_cmp_gt_u32 s[58:59], s42, v28                       // lanes whose half-0 instance exists (n < 2^31)
	v_cmp_gt_u32 s[16:17], s42, v2                        // ... half-1 instance
	s_lshl_b32 s60, s44, 2                                // state row pitch in bytes
	s_lshl_b32 s68, s42, 2                                // bytes of one channel of one sample (N*4)
	v_lshlrev_b32 v28, 2, v28
	s_mov_b32 s9, s64                                     // nSamples
	s_mov_b32 s74, s46                                    // nLoad
	s_mov_b32 s75, s47                                    // nStore
	s_mov_b32 s44, s65                                    // channels
	s_mul_i32 s45, s42, s44
	s_lshl_b32 s45, s45, 2                                // bytes per sample of PCM
	s_mov_b64 s[42:43], s[70:71]
	s_sub_u32 s32, s32, 4
	s_subb_u32 s33, s33, 0                                // s[32:33] = address of the kernel entry
	s_add_u32 s34, s32, (pair_endsample-KNAME)
	s_addc_u32 s35, s33, 0
	s_mov_b64 s[78:79], 0
	s_load_dwordx8  s[48:55], s[0:1], KA_INOFF            // inOff[4] latchOff[4]
	s_load_dwordx4  s[64:67], s[0:1], KA_ISLOTS           // iSlots xSlots iSize xSize
	s_load_dword    s76, s[0:1], KA_CURSORROW
	s_waitcnt lgkmcnt(0)
	s_mov_b32 s46, s64
	s_mov_b32 s47, s65
	s_mov_b32 s56, s66
	s_mov_b32 s57, s67
	// TRAM: the tiles of wavefronts 2*wg (half 0) and 2*wg+1 (half 1, one tile further)
	s_lshl_b32 s62, s2, 1
	s_mul_i32 s64, s62, s46
	s_mul_hi_u32 s65, s62, s46
	s_lshl_b64 s[64:65], s[64:65], 8
	s_add_u32 s36, s36, s64
	s_addc_u32 s37, s37, s65
	s_mul_i32 s64, s62, s47
	s_mul_hi_u32 s65, s62, s47
	s_lshl_b64 s[64:65], s[64:65], 8
	s_add_u32 s38, s38, s64
	s_addc_u32 s39, s39, s65
	s_lshl_b32 s26, s46, 8
	s_lshl_b32 s27, s47, 8

	// ---- prologue: state rows -> register file
	s_mov_b32 s62, 0
	s_cmp_eq_u32 s74, 0
	s_cbranch_scc1 .Lload_done
.Lload_loop:
	s_lshl_b32 s63, s62, 2
	s_load_dword s64, s[72:73], s63                       // row | BOUNDED << 15 | stateRow << 16
	s_waitcnt lgkmcnt(0)
	s_lshr_b32 s65, s64, 16
	s_bitcmp1_b32 s64, 15
	s_cselect_b32 s69, 1, 0
	s_and_b32 s64, s64, 0x7fff
	s_lshl_b32 s64, s64, 1                                // VGPR index of the row's pair
	STATE_ROW s65
	global_load_dword v2, v28, s[66:67]
	global_load_dword v3, v28, s[66:67] offset:256
	s_waitcnt vmcnt(0)
	TAINT_ROW v2
	TAINT_ROW v3
	s_set_gpr_idx_on s64, 8
	v_mov_b32 v32, v2
	v_mov_b32 v33, v3
	s_set_gpr_idx_off
	s_add_u32 s62, s62, 1
	s_cmp_lt_u32 s62, s74
	s_cbranch_scc1 .Lload_loop
.Lload_done:
	// the four TRAM cursors (equal in all instances: read half 0's)
	STATE_ROW s76
	global_load_dword v24, v28, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_load_dword v25, v28, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_load_dword v26, v28, s[66:67]
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_load_dword v27, v28, s[66:67]
	s_waitcnt vmcnt(0)
	// run-once code of the translated program (LOG/EXP tables -> LDS); returns through s[24:25]
	s_load_dword s62, s[0:1], KA_INIT
	s_waitcnt lgkmcnt(0)
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
	// first sample's input
	v_mov_b32 v16, 0
	v_mov_b32 v17, 0
	v_mov_b32 v18, 0
	v_mov_b32 v19, 0
	v_mov_b32 v20, 0
	v_mov_b32 v21, 0
	v_mov_b32 v22, 0
	v_mov_b32 v23, 0
	s_mov_b32 s3, 0
	s_cmp_lt_i32 s9, 1
	s_cbranch_scc1 .Lepilogue
	s_mov_b64 s[62:63], s[12:13]
	INPUT_LOADS
	s_waitcnt vmcnt(0)

	// ---- one sample period
.Lsample:
	TAINT_FINITE v16                                      // non-finite PCM input taints (unused channels hold 0)
	TAINT_FINITE v17
	TAINT_FINITE v18
	TAINT_FINITE v19
	TAINT_FINITE v20
	TAINT_FINITE v21
	TAINT_FINITE v22
	TAINT_FINITE v23
	IN_ROW 0, v16, v20, s48
	IN_ROW 1, v17, v21, s49
	IN_ROW 2, v18, v22, s50
	IN_ROW 3, v19, v23, s51
	s_nop 3
	// prefetch the next sample's input (if any); its latency hides behind this sample's program
	s_add_u32 s62, s3, 1
	s_cmp_ge_i32 s62, s9
	s_cbranch_scc1 .Ls_pdone
	s_add_u32 s62, s12, s45
	s_addc_u32 s63, s13, 0
	INPUT_LOADS
.Ls_pdone:
	// stream of this sample: the last sample materialises every CCR write; tainted wave: exact stream (high dword)
	s_add_u32 s62, s3, 1
	s_cmp_eq_u32 s62, s9
	s_cselect_b32 s4, s42, s6
	s_cselect_b32 s5, s43, s7
	s_cmp_lg_u64 s[78:79], 0
	s_cselect_b32 s4, s5, s4
	s_add_u32 s4, s4, s32
	s_addc_u32 s5, s33, 0
	s_setpc_b64 s[4:5]

	// ---- end of the program for this sample: latch rows -> PCM out, next sample
pair_endsample:
	s_mov_b64 exec, -1
	s_mov_b64 s[62:63], s[14:15]
	OUT_CH 0, s52
	OUT_CH 1, s53
	OUT_CH 2, s54
	OUT_CH 3, s55
	s_add_u32 s12, s12, s45
	s_addc_u32 s13, s13, 0
	s_add_u32 s14, s14, s45
	s_addc_u32 s15, s15, 0
	s_add_u32 s3, s3, 1
	s_cmp_lt_i32 s3, s9
	s_waitcnt vmcnt(0)
	s_cbranch_scc1 .Lsample

	// ---- epilogue: register file and bookkeeping -> state rows
.Lepilogue:
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_mov_b64 exec, -1
	s_mov_b32 s62, 0
	s_cmp_eq_u32 s75, 0
	s_cbranch_scc1 .Lstore_done
.Lstore_loop:
	s_add_u32 s63, s62, s74
	s_lshl_b32 s63, s63, 2
	s_load_dword s64, s[72:73], s63
	s_waitcnt lgkmcnt(0)
	s_lshr_b32 s65, s64, 16
	s_and_b32 s64, s64, 0x7fff
	s_lshl_b32 s64, s64, 1
	s_set_gpr_idx_on s64, 1
	v_mov_b32 v2, v32
	v_mov_b32 v3, v33
	s_set_gpr_idx_off
	s_nop 3
	STATE_ROW s65
	global_store_dword v28, v2, s[66:67]
	global_store_dword v28, v3, s[66:67] offset:256
	s_add_u32 s62, s62, 1
	s_cmp_lt_u32 s62, s75
	s_cbranch_scc1 .Lstore_loop
.Lstore_done:
	STATE_ROW s76
	global_store_dword v28, v24, s[66:67]
	global_store_dword v28, v24, s[66:67] offset:256
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_store_dword v28, v25, s[66:67]
	global_store_dword v28, v25, s[66:67] offset:256
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_store_dword v28, v26, s[66:67]
	global_store_dword v28, v26, s[66:67] offset:256
	s_add_u32 s66, s66, s60
	s_addc_u32 s67, s67, 0
	global_store_dword v28, v27, s[66:67]
	global_store_dword v28, v27, s[66:67] offset:256
	// ood |= flags ; counter += staticCount * nSamples   (no SKIP here: every instruction of every sample ran)
	s_load_dwordx4 s[64:67], s[0:1], KA_OODROW            // oodRow countLo countHi staticCount
	s_waitcnt lgkmcnt(0)
	s_mul_i32 s22, s67, s9                                // staticCount * nSamples (low)
	s_mul_hi_u32 s23, s67, s9
	s_mov_b32 s24, s65
	s_mov_b32 s25, s66
	STATE_ROW s64
	global_load_dword v2, v28, s[66:67]
	global_load_dword v3, v28, s[66:67] offset:256
	s_waitcnt vmcnt(0)
	v_or_b32 v2, v2, v14
	v_or_b32 v3, v3, v15
	global_store_dword v28, v2, s[66:67]
	global_store_dword v28, v3, s[66:67] offset:256
	STATE_ROW s24
	s_mov_b64 s[70:71], s[66:67]                          // countLo row
	STATE_ROW s25                                         // countHi row
	global_load_dword v4, v28, s[70:71]
	global_load_dword v5, v28, s[66:67]
	global_load_dword v6, v28, s[70:71] offset:256
	global_load_dword v7, v28, s[66:67] offset:256
	v_mov_b32 v8, s23
	s_waitcnt vmcnt(0)
	v_add_co_u32 v4, vcc, s22, v4
	s_nop 1
	v_addc_co_u32 v5, vcc, v5, v8, vcc
	v_add_co_u32 v6, vcc, s22, v6
	s_nop 1
	v_addc_co_u32 v7, vcc, v7, v8, vcc
	global_store_dword v28, v4, s[70:71]
	global_store_dword v28, v5, s[66:67]
	global_store_dword v28, v6, s[70:71] offset:256
	global_store_dword v28, v7, s[66:67] offset:256
	s_endpgm
.Lfunc_end0:
	.size	KNAME, .Lfunc_end0-KNAME

// what the host reads from the image (fx_xlate.cpp xlateTemplate): 84 handler slots (none here), the hole
	.p2align 6
	.globl	OTABLE
OTABLE:
	.fill	84, 4, 0
	.long HOLE - KNAME
	.long HOLE_BYTES

	.p2align	8
	.globl	HOLE
HOLE:
	.fill	(HOLE_BYTES / 4), 4, 0xbf800000                 // s_nop 0
	s_endpgm

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
