// idxmode_probe.S — directed experiment for the VGPR-index-mode question of DESIGN.md section 4.5:
// does a plain VALU instruction that directly follows s_set_gpr_idx_off ever still execute under the
// index mode (DST or SRC0) the previous instruction stream left on?  Round 1 saw sporadic memory faults in
// interpreter builds whose hot handlers began with "s_set_gpr_idx_off ; v_mov_b32 v2, sN" after a handler
// that ended in DST mode, at >= 4 waves per SIMD next to fp64 instructions, and avoided the pattern without
// explaining it.  If the mode write could lag, that v_mov would write v(2 + M0) - for M0 = 25 that is v27,
// the instance byte offset every PCM / state access uses as its address.
//
// This kernel reproduces exactly those instruction sequences, millions of times per wave at 8 waves per
// SIMD with four variants co-resident on every SIMD, but on registers that feed NO address: a stale mode
// shows up as a count, not as a fault.  Per lane: v60 counts iterations in which the indexed alias was hit
// or the plain destination was missed.
//
//   variant 0  dst-mode write ; s_set_gpr_idx_off ; v_mov_b32 v2, s12
//   variant 1  fp64 chain ; dst-mode write ; s_set_gpr_idx_off ; v_mov_b32 v2, s12
//   variant 2  as 1, with an s_setpc_b64 (handler dispatch) between the dst-mode write and the idx_off
//   variant 3  src0-mode read ; s_set_gpr_idx_off ; v_cvt_f64_f32 v[6:7], v3   (must convert v3, not v(3+M0))
//
//   control    (kernarg flag = 1, every wave) variant 0 WITHOUT the s_set_gpr_idx_off: every iteration must count -
//              shows that the check does see a plain write that executes under DST mode
//
// kernarg: u32* out (one dword per lane of every wave), u32 iters, u32 control
	.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
	.amdhsa_code_object_version 6
	.text
	.globl	idxmode_probe
	.p2align	8
	.type	idxmode_probe,@function
idxmode_probe:
	s_load_dwordx2 s[4:5], s[0:1], 0x0
	s_load_dwordx2 s[6:7], s[0:1], 0x8
	v_lshlrev_b32 v1, 2, v0
	v_mov_b32 v60, 0
	v_mov_b32 v40, 0x5e5e5e5e                             // sentinel
	s_mov_b32 s10, 25                                     // M0 index: alias of v2 is v27, alias of v3 is v28
	s_mov_b32 s13, 0                                      // iteration
	s_and_b32 s11, s2, 3                                  // variant = wave index mod 4
	s_getpc_b64 s[20:21]
.Lpc:
	s_add_u32 s22, s20, (.Lafter2-.Lpc)
	s_addc_u32 s23, s21, 0
	v_mov_b32 v3, 1.0
	v_mov_b32 v28, 2.0                                    // alias of v3 under M0 = 25
	v_mov_b32 v4, 0x3e99999a
	s_waitcnt lgkmcnt(0)
	s_cmp_eq_u32 s7, 1
	s_cbranch_scc1 .Lctl
	s_cmp_eq_u32 s11, 1
	s_cbranch_scc1 .Lv1
	s_cmp_eq_u32 s11, 2
	s_cbranch_scc1 .Lv2
	s_cmp_eq_u32 s11, 3
	s_cbranch_scc1 .Lv3

.macro CHECK_DST
	// v2 must hold s12, v27 must still hold the sentinel
	v_cmp_ne_u32 vcc, s12, v2
	v_cmp_ne_u32 s[14:15], v27, v40
	s_nop 1
	s_or_b64 vcc, vcc, s[14:15]
	s_nop 1
	v_cndmask_b32 v61, 0, 1, vcc
	v_add_u32 v60, v60, v61
	v_mov_b32 v27, v40
	v_mov_b32 v2, 0
.endm

.Lv0:
	v_mov_b32 v27, v40
.Lv0_loop:
	s_or_b32 s12, s13, 0x40000000
	v_mov_b32 v5, s13
	v_cmp_u_f32_e64 vcc, 0, v5
	v_med3_f32 v6, -1.0, v5, 1.0
	s_set_gpr_idx_on s10, 8
	v_cndmask_b32 v32, v6, v5, vcc                        // writes v57
	s_set_gpr_idx_off
	v_mov_b32 v2, s12
	CHECK_DST
	s_add_u32 s13, s13, 1
	s_cmp_lt_u32 s13, s6
	s_cbranch_scc1 .Lv0_loop
	s_branch .Ldone

.Lctl:
	v_mov_b32 v27, v40
.Lctl_loop:
	s_or_b32 s12, s13, 0x40000000
	v_mov_b32 v5, s13
	v_cmp_u_f32_e64 vcc, 0, v5
	v_med3_f32 v6, -1.0, v5, 1.0
	s_set_gpr_idx_on s10, 8
	v_cndmask_b32 v32, v6, v5, vcc
	s_nop 0                                               // (no s_set_gpr_idx_off)
	v_mov_b32 v2, s12                                     // lands in v27
	s_set_gpr_idx_off
	CHECK_DST
	s_add_u32 s13, s13, 1
	s_cmp_lt_u32 s13, s6
	s_cbranch_scc1 .Lctl_loop
	s_branch .Ldone

.Lv1:
	v_mov_b32 v27, v40
.Lv1_loop:
	s_or_b32 s12, s13, 0x40000000
	v_cvt_f64_f32 v[8:9], v4
	v_cvt_f64_f32 v[10:11], v3
	v_fma_f64 v[6:7], v[8:9], v[8:9], v[10:11]
	v_cvt_f32_f64 v5, v[6:7]
	v_cmp_u_f32_e64 vcc, 0, v5
	v_med3_f32 v6, -1.0, v5, 1.0
	s_set_gpr_idx_on s10, 8
	v_cndmask_b32 v32, v6, v5, vcc
	s_set_gpr_idx_off
	v_mov_b32 v2, s12
	CHECK_DST
	s_add_u32 s13, s13, 1
	s_cmp_lt_u32 s13, s6
	s_cbranch_scc1 .Lv1_loop
	s_branch .Ldone

.Lv2:
	v_mov_b32 v27, v40
.Lv2_loop:
	s_or_b32 s12, s13, 0x40000000
	v_cvt_f64_f32 v[8:9], v4
	v_cvt_f64_f32 v[10:11], v3
	v_fma_f64 v[6:7], v[8:9], v[8:9], v[10:11]
	v_cvt_f32_f64 v5, v[6:7]
	v_cmp_u_f32_e64 vcc, 0, v5
	v_med3_f32 v6, -1.0, v5, 1.0
	s_set_gpr_idx_on s10, 8
	v_cndmask_b32 v32, v6, v5, vcc
	s_setpc_b64 s[22:23]
	s_nop 0
	s_nop 0
.Lafter2:
	s_set_gpr_idx_off
	v_mov_b32 v2, s12
	CHECK_DST
	s_add_u32 s13, s13, 1
	s_cmp_lt_u32 s13, s6
	s_cbranch_scc1 .Lv2_loop
	s_branch .Ldone

.Lv3:
.Lv3_loop:
	v_cvt_f32_u32 v3, s13                                 // plain operand, changes every iteration
	s_set_gpr_idx_on s10, 1
	v_cvt_f64_f32 v[8:9], v32                             // reads v57 (src0 mode)
	s_set_gpr_idx_off
	v_cvt_f64_f32 v[6:7], v3                              // must read v3 - under a stale mode it would read v28 = 2.0
	v_cvt_f32_f64 v5, v[6:7]
	v_cmp_neq_f32 vcc, v5, v3
	s_nop 1
	v_cndmask_b32 v61, 0, 1, vcc
	v_add_u32 v60, v60, v61
	s_add_u32 s13, s13, 1
	s_cmp_lt_u32 s13, s6
	s_cbranch_scc1 .Lv3_loop

.Ldone:
	s_lshl_b32 s16, s2, 8                                 // wave * 256 bytes
	s_add_u32 s4, s4, s16
	s_addc_u32 s5, s5, 0
	global_store_dword v1, v60, s[4:5]
	s_waitcnt vmcnt(0)
	s_endpgm
.Lend:
	.size	idxmode_probe, .Lend-idxmode_probe

	.rodata
	.p2align	6, 0x0
	.amdhsa_kernel idxmode_probe
		.amdhsa_group_segment_fixed_size 0
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size 16
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr 64
		.amdhsa_next_free_sgpr 32
		.amdhsa_accum_offset 64
		.amdhsa_reserve_vcc 1
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
        .size: 16
        .value_kind: by_value
    .group_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .kernarg_segment_size: 16
    .max_flat_workgroup_size: 64
    .name: idxmode_probe
    .private_segment_fixed_size: 0
    .sgpr_count: 38
    .symbol: idxmode_probe.kd
    .vgpr_count: 64
    .wavefront_size: 64
amdhsa.target: amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
	.end_amdgpu_metadata
