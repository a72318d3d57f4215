/* Lab build of libf5tts_hip.so (F5_LAB=1 bash f5_tts_mlx_amd/csrc/build.sh): the product library plus the kernels and schedules
 * that were measured and superseded or rejected, and the hooks that select them.  Nothing here is reachable from sample() in
 * the product build; DESIGN.md records what each experiment measured.  f5_lab_build() (f5tts_hip.h) returns 1 for this build. */
#ifndef F5TTS_HIP_LAB_H
#define F5TTS_HIP_LAB_H
#include "f5tts_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* large-shape GEMM kernel: 2 = 256x256 role-split schedule (product), 3 = 128x256 of round 1 (two workgroups per CU; stagger
 * = initial delay of the second workgroup in cycles, < 0 auto), 4 = 256x256 lock-step schedule of rounds 1-2, 5 = 128x256 with
 * in-wave fragment prefetch (round 3, gemm128.hip) */
int f5_debug_set_gemm_big_kernel(int v, int stagger_cycles);
int f5_debug_set_gemm128_pad(int bytes);    /* gemm128: extra dynamic LDS per workgroup; > 8 KB leaves one workgroup per CU */
/* 128x256 round-1 kernel: issue priority 0 = MFMA clusters, 1 = none, 2 = epilogue */
int f5_debug_set_gemm_v3_prio(int v);
/* lock-step 256x256 kernel: 0 = one tile per workgroup, 1 = stream-K (persistent workgroup per CU over contiguous K-step ranges),
 * 2 = hybrid (lockstep rounds + stream-K tail); f5_debug_gemm_streamk_error() returns 1 if a partial-tile hand-off timed out */
int f5_debug_set_gemm_streamk(int v);
int f5_debug_gemm_streamk_error(void);
/* attention: 1 = register-staged kernel of round 1, 2 = global_load_lds ring kernels (product), 3 / 4 = in-wave software
 * pipelining, 5 / 6 = pipelined 8- / 4-wave experiment */
int f5_debug_set_attn_version(int v);
/* experiment bits: 1 = single-issue softmax VALU (v5 / v6), 2 = one workgroup per CU, 4 = plain 2-D block numbering, 8 = eager O
 * rescale, 16 = per-tile maximum (v2w / non-fast split kernels), 32 = the role-split kernel re-reads Q from LDS.
 * f5_debug_set_attn_wide(2) (lab build only) selects the role-split large-grid kernel f5_attn2r_kernel */
int f5_debug_set_attn_variant(int bits);
int f5_debug_set_attn_ablation(int v);      /* timing-only ablations of the mid-size ring kernel (results wrong unless 0) */
/* large-grid kernel with a per-tile maximum: which phase holds SIMD issue priority: 0 MFMA clusters, 1 none, 2 softmax section */
int f5_debug_set_attn_prio(int v);
#ifdef __cplusplus
}
#endif
#endif
