/*
 * gigapose_hip_probe.h -- what libgigapose_hip_probe.so exports ON TOP of gigapose_hip.h: the same sources compiled with -DGP_PROBES
 * (gigapose_amd/csrc/Makefile).  Nothing here is part of the product: these are the A/B switches behind the bit-identity tests (a
 * variant on / off must give equal bits), time-stamped probe builds of the hot kernels (tools/probe_*.py), test-only epilogues and the
 * readers of the stream-K scratch's error word.  All switches are PROCESS-GLOBAL and not thread-safe: set, measure, reset.
 * Python: `with _lib.probe_library(): ...` (gigapose_amd/_lib.py) routes the calls of a block to this library.
 */
#ifndef GIGAPOSE_HIP_PROBE_H
#define GIGAPOSE_HIP_PROBE_H
#include "gigapose_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- matcher ---- */
/* probe build of gp_match_tiles_split_dir (two-plane bank, tar2src) writing 8 time stamps per tile, see gp_match.hip */
int gp_match_tiles_split_trace(const void* q_hi, const void* q_lo, const void* b_hi, const void* b_lo, const float* qmask,
                               const float* bmask, const int* labels, int B, int O, int N, int C, float sim_threshold,
                               float patch_threshold, uint8_t* idx_t2s, float* score_t2s, float* mask_all, float* sim_avg,
                               unsigned long long* trace, void* stream);
/* 1 (default) = tiles are built from the live (mask != 0) patches only, per-wave rectangles of 32 x 32 blocks; 0 = every patch counts
 * as live (the full 2 x 4 per wave).  Outputs are bit-identical. */
int gp_match_split_set_compact(int on);

/* ---- f32 (chain) GEMM ---- */
void gp_gemm_set_streamk(int mode); /* 0 = one workgroup per tile even with a scratch, 1 = by the built-in rule (default), 2 = split
                                       whenever the tile count allows (results identical) */
void gp_gemm_set_group(int g);      /* tiles are ordered in bands of g i-tiles (default 8; results identical) */
int gp_gemm_streamk_error(const float* scratch, void* stream);  /* synchronises; the scratch's error word (0 = every hand-off arrived) */
int gp_gemm_probe(int variant, const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I, int J, int K, void* stream);
int gp_gemm_probe_occupancy(void);
int gp_gemm_product_occupancy(int streamk);
int gp_gemm_probe_clock(unsigned long long* host4);

/* ---- split GEMMs ---- */
int gp_gemm_split_set_trace(unsigned long long* dev_buf);
int gp_gemm_split_timing(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J, int K,
                         unsigned long long* out20 /* 23 entries */, void* stream);
int gp_gemm_split256_timing(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J, int K,
                            float* scratch, unsigned long long* out6, void* stream);
int gp_gemm_split256_error(const float* scratch, void* stream); /* scratch error word of gp_gemm_split256 / gp_gemm_planes256_scaled */
/* gemm_planes256_kernel: bit 0 (default 1): data-parallel rounds before the stream-K remainder; bit 1: TEST hook, head fragments are
 * never published (every waiter times out -> GP_STATUS_HANDOFF_SPLIT) */
int gp_gemm_planes256_set_dp(int mode);
/* default 1: shapes with 8 <= tiles < 256 (ViT-L below 64 crops) run with the slots of a tile splitting its K in parallel; 0: such
 * shapes are refused and the ViT falls back to the 128 x 128 kernels; n >= 2: at least n k-steps per slot of a split tile */
int gp_gemm_planes256_set_par(int on);
/* default 1: launches whose 256 x 256 tiles fill at most half of the 256 slots run on 256 x 128 tiles; 0: always 256 x 256 */
int gp_gemm_planes256_set_half_tiles(int on);
/* gp_gemm_planes256_scaled (epilogues 0, 3, 6, 7; plane scale 8) from a build with per-slot time stamps, see gp_split256.hip */
int gp_gemm_planes256_trace(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, float* D, int ldd, void* out_hi,
                            void* out_lo, int ldo, int I, int J, int J_valid, int K, int epilogue, const float* bias, const float* scale,
                            const float* residual, int ldr, float out_scale, float* scratch, unsigned long long* trace, void* stream);
int gp_gemm_planes256_timing(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, float* D, int ldd, int I, int J,
                             int K, float* scratch, unsigned long long* out8, void* stream);

/* ---- ViT ---- */
void gp_attention_set_nq(int nq); /* chain attention: 1 (default) / 2 = register-resident kernel with 1 / 2 query tiles per wave; 0 = K/V
                                     shared through LDS (results identical) */
void gp_vit_set_ln_reg(int mode); /* plane path's LayerNorm: 1 (default) 32-token blocks, 16-token blocks when the launch has at most 128 of
                                     them; 2 always 32-token blocks; 0 the first-generation three-pass kernel (results identical) */
void gp_vit_set_planes(int mode); /* 2 (default) = planes + attention in split numerics, 1 = planes + f32 attention, 0 = f32 activations */

/* ---- IST convolutions ---- */
void gp_conv_set_direct(int on);                                  /* 0 = always the generic gather kernel (results identical) */
void gp_conv2d_planes_set_trace(unsigned long long* device_buf);  /* per slot segments / k-steps / ticks, NULL = off */
/* 1 (default) = 3 x 3 / stride 1 layers take conv_halo_kernel and its parallel split below 256 tiles; 0 = neither; 5 = halo without the
 * parallel split; 1 + 16 n: at least n channel blocks per slot of a split tile */
int gp_conv2d_planes_set_halo(int on);

#ifdef __cplusplus
}
#endif
#endif /* GIGAPOSE_HIP_PROBE_H */
