/*
 * gigapose_hip.h -- C-ABI of libgigapose_hip.so: the MI355X (gfx950) implementation of the
 * GigaPose coarse-pose hot path (reference: nv-nguyen/gigapose, src/models/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; the caller owns all buffers,
 *     kernels never allocate; inputs are never modified;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous;
 *   - return value: 0 = ok, -1 = invalid argument, -2 = launch failure; gp_last_error() returns a
 *     thread-local message for the last failure (the reference raises Python exceptions /
 *     asserts at the same places, cited per function);
 *   - tensors are dense, row-major in the order written, f32 unless stated otherwise;
 *   - P = 256 patches (16x16 grid of 14-px patches on a 224x224 crop).
 * This header is the PRODUCT interface: what gigapose_amd/*.py, bench.py and a host integration call.  A/B switches, time-stamped probe
 * builds, test-only epilogues and error-word readers live in gigapose_hip_probe.h and exist only in libgigapose_hip_probe.so (the same
 * sources compiled with -DGP_PROBES; tests and tools/ load it where they need a hook).
 */
#ifndef GIGAPOSE_HIP_H
#define GIGAPOSE_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int gp_abi_version(void);
const char* gp_last_error(void);

/* Guard rails: conditions that the reference turns into Python exceptions (IndexError at `ae_features[label - 1]`,
 * gigaPose.py:520) or that have no counterpart there (a lost stream-K accumulator hand-over; an activation outside the
 * range of the split-f16 planes, |x| >= 8190 or non-finite) are OR-ed by the kernels into ONE device int32 owned by the
 * caller PER DEVICE.  The host reads it at its next synchronisation point (gigapose_amd/_lib.py: check_status) and raises.
 * gp_set_status_buffer registers the word of the CURRENT HIP device (hipGetDevice) in a per-device table; every launch takes the word of
 * the device it is issued on, so several GPUs driven from one process (or from several threads) never share or re-point a word.
 * device_word == NULL switches the reporting off for that device.  Bits: */
#define GP_STATUS_HANDOFF_SPLIT 1 /* split GEMM: a hand-over timed out, the tile it fed is garbage */
#define GP_STATUS_HANDOFF_CHAIN 2 /* f32 (chain) GEMM: same */
#define GP_STATUS_SPLIT_RANGE 4   /* split numerics: plane value out of range / NaN (use numerics "chain" or GIGAPOSE_SPLIT_GEMM=128) */
#define GP_STATUS_LABEL_RANGE 8   /* label >= O or template id >= N (clamped to 0 so nothing reads out of bounds) */
#define GP_STATUS_SPLIT_RANGE_CONV 16 /* GP_STATUS_SPLIT_RANGE raised by the IST convolution planes (fallback: the 128 x 128 two-accumulator
                                         convolution, GIGAPOSE_SPLIT_CONV=128; gigapose_amd/gigaPose.py widens automatically) */
int gp_set_status_buffer(int* device_word);

/* Optional timing of kernel families with HIP events on the launch stream (used by bench.py for the
 * roofline figure; not part of the reference interface).  gp_prof_begin() starts recording;
 * gp_prof_end() stops, synchronises and returns per-kind totals: ms[k], work[k] (flops, or bytes for
 * layernorm), launches[k]; its return value is the number of kinds; names via gp_prof_kind_name(). */
/* kind < 0: every launch of every family.  kind >= 0: light instrumentation for a timed region -- events only around every `stride`-th
 * launch of THAT kernel family (an event pair per launch costs ~3 us of queue time: 1.4 ms on a 46 ms step when all ~210 launches are
 * bracketed); choose a stride coprime with the family's launches per layer so that the sample cycles through its shapes. */
void gp_prof_begin(int kind, int stride);
int gp_prof_end(int max_kinds, double* ms, double* work, long long* launches);
const char* gp_prof_kind_name(int kind);

/* ---- detection pre-processing (the step before the hot path) ------------- */

/* CropResizePad.__call__(xyxy_boxes, images) (src/utils/crop.py:11-61): per detection d, crop images[d] to its
 * box (clamped at the frame border like the reference's slicing), nearest resize by 224/max(w,h), zero-pad to
 * target x target, nearest resize to target; M[d] = M_resize_pad @ M_crop.  One gather per output pixel with
 * ATen's nearest index arithmetic (bit-exact vs the reference).
 *   images (D,C,H,W) f32, boxes (D,4) int64 xyxy -> out (D,C,target,target) f32, M (D,3,3) f32.
 * A box that is empty, starts outside the frame or scales to nothing (the reference raises / produces an empty
 * tensor there) leaves its outputs untouched and stores d+1 in *err_flag (device int, caller zeroes it). */
int gp_crop_resize_pad(const float* images, const long long* boxes, int D, int C, int H, int W, int target, float* out,
                       float* M, int* err_flag, void* stream);

/* process_real + normalize fused (src/dataloader/train.py:80-123, src/dataloader/test.py:295-315,
 * configs/data/transform.yaml): tar_img = ((rgb/255) * mask cropped as above - mean) / std, tar_mask = cropped mask.
 *   rgb (n_img,3,H,W) u8 full frames, masks (D,H,W) f32 {0,1}, boxes (D,4) int64 xyxy, im_id (D) int32 frame of
 *   each detection, mean3/std3: HOST arrays of 3 floats -> tar_img (D,3,target,target), tar_mask (D,target,target),
 *   M (D,3,3).  err_flag as above. */
int gp_preprocess_detections(const uint8_t* rgb, const float* masks, const long long* boxes, const int* im_id, int n_img,
                             int D, int H, int W, int target, const float* mean3_host, const float* std3_host,
                             float* tar_img, float* tar_mask, float* M, int* err_flag, void* stream);

/* ---- template matching: LocalSimilarity.test (src/models/matching.py:188-316) ------------- */

/* F.normalize(x, dim=C) for x (rows, C, 256).  Replaces matching.py:224,229 and ae_net.py:69. */
int gp_l2norm_cp(const float* x, float* out, int rows, int C, void* stream);

/* Fused similarity + masks + threshold + bidirectional argmax + cycle check + template score
 * for every (detection b, template n) pair.  Replaces matching.py:233-278 and
 * find_consistency_patches (:80-113); the (B,N,256,256) `sim` tensor is never written.
 *   query (B,C,256), bank (O,N,C,256): features already normalised by gp_l2norm_cp
 *   qmask (B,256), bmask (O,N,256): patch-grid masks (nearest sample of the 224x224 masks,
 *                                    matching.py:222,227)
 *   labels (B) int32: 0-based object index per detection (reference: label-1, gigaPose.py:520)
 * outputs: idx_t2s u8 (B,N,256), score_t2s (B,N,256), mask_all (B,N,256), sim_avg (B,N).
 * Requires C % 16 == 0.
 * search_direction = the reference's ctor argument (matching.py:18, :239-244): 0 = "tar2src" (its default), 1 = "src2tar" (the row /
 * column argmax pairs exchange roles; masks stay positional as in the reference).
 * patch_threshold <= 0 in any of the match entry points = no cycle check (matching.py:256-257: mask_cycle = ones). */
int gp_match_tiles_dir(const float* query, const float* bank, const float* qmask, const float* bmask,
                       const int* labels, int B, int O, int N, int C, float sim_threshold,
                       float patch_threshold, int search_direction, uint8_t* idx_t2s, float* score_t2s, float* mask_all,
                       float* sim_avg, void* stream);

/* Split-f16 numerics of the matcher (same outputs as gp_match_tiles up to f32 round-off: the similarity
 * tile is computed as 3 f16 MFMAs per k-block on operands split into f16 halves, f32 accumulation; gp_split.hip).
 * gp_l2norm_split: x (rows, C, 256) f32 -> F.normalize(x, dim=C) * 32 as two f16 planes hi / lo, each
 * (rows, 256, Cp), Cp = round_up(C, 32), zero padded (value ~= (hi + lo) / 32).
 * gp_match_tiles_split: query planes (B,256,Cp), bank planes (O,N,256,Cp); C here = Cp; everything else as
 * gp_match_tiles. */
/* gp_l2norm_split_mask: the split normalisation and, in the same launch, the rows' 16 x 16 patch masks: patch_mask (rows, 256) f32 =
 * mask_img (rows, mask_h, mask_w) f32 sampled at pixel (i mask_h / 16, j mask_w / 16) = F.interpolate(mask, (16, 16)) nearest
 * (matching.py:222, 227); mask_img = patch_mask = NULL: the normalisation alone. */
int gp_l2norm_split_mask(const float* x, void* hi, void* lo, int rows, int C, const float* mask_img, int mask_h, int mask_w,
                         float* patch_mask, void* stream);
int gp_match_tiles_split_dir(const void* q_hi, const void* q_lo, const void* b_hi, const void* b_lo, const float* qmask,
                             const float* bmask, const int* labels, int B, int O, int N, int C, float sim_threshold,
                             float patch_threshold, int search_direction, uint8_t* idx_t2s, float* score_t2s, float* mask_all,
                             float* sim_avg, void* stream);
/* torch.topk(sim_avg, k, dim=1) (matching.py:279); ties: lower template index first.
 * Fails (-1) when k > N, like torch.topk raises. ids int32 (B,k), scores (B,k). */
int gp_topk(const float* sim_avg, int B, int N, int k, int* ids, float* scores, void* stream);

/* Gather the per-patch records of the selected templates (matching.py:282-285):
 * rec_idx u8 (B,k,256), rec_score (B,k,256) [= score_pts], rec_mask (B,k,256). */
int gp_gather_records(const int* ids, const uint8_t* idx_t2s, const float* score_t2s,
                      const float* mask_all, int B, int N, int k, uint8_t* rec_idx, float* rec_score,
                      float* rec_mask, void* stream);

/* format_prediction + convert_index2location (matching.py:29-61, 63-68):
 * tar_pts, src_pts int64 (rows,256,2) as (x,y), -1 where rec_mask == 0.  rows = B*k. */
int gp_format_points(const uint8_t* rec_idx, const float* rec_mask, int rows, long long* tar_pts,
                     long long* src_pts, void* stream);
/* gp_topk + gp_gather_records + gp_format_points in ONE launch (what GigaPose's hot loop calls; the three above stay as stage entries
 * and for the template-sharded path, which exchanges records between the top-k and the formatting): ids (B,k) int64 as the reference
 * returns them (matching.py:279-316), scores (B,k), rec_score (B,k,256), tar_pts / src_pts (B,k,256,2) int64 (-1 = invalid). */
int gp_select_topk(const float* sim_avg, const uint8_t* idx_t2s, const float* score_t2s, const float* mask_all, int B, int N, int k,
                   long long* ids, float* scores, float* rec_score, long long* tar_pts, long long* src_pts, void* stream);

/* ---- dense layers: k-major f32 MFMA GEMM ---------------------------------------------------- */

/* D[i][j] = epi( sum_k A[k][i] * B[k][j] ), A (K,lda), B (K,ldb), D (I,ldd); accumulation is the
 * sequential fmaf chain over k.  Stands for torch.nn.Linear on TRANSPOSED activations
 * (A = W^T [in][out], B = X^T [in][tokens] -> D = Y^T [out][tokens]): DINOv2 qkv/proj/fc1/fc2 (the
 * un-vendored backbone called at src/models/network/ae_net.py:44-47) and the IST regressor MLPs
 * (src/models/network/ist_net.py:140-155).
 * epilogue: 0 none | 1 +bias[i] | 2 gelu_erf(+bias[i]) | 3 residual[i][j] + scale[i]*(acc+bias[i])
 *           (D may alias residual) | 4 +bias[j] | 5 relu(+bias[i]).
 * Requires I % 128 == 0, J % 128 == 0, K % 16 == 0, lda/ldb % 4 == 0, 16-byte aligned A/B. */
int gp_gemm_kmajor(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I, int J,
                   int K, int epilogue, const float* bias, const float* scale, const float* residual,
                   int ldr, void* stream);

/* Same contraction with a device scratch that lets the library balance tile counts that are not a multiple
 * of the resident workgroups ("chain-preserving stream-K", gp_gemm.hip): a tile split between two workgroups
 * is handed over as an accumulator fragment, so the per-output fmaf chain -- and every output bit -- is the
 * same as gp_gemm_kmajor's.  scratch: gp_gemm_streamk_workspace_bytes() bytes, 16-byte aligned, one per
 * stream; call gp_gemm_streamk_reset() on it once before first use (zeroes the hand-off flags).  A hand-over that never
 * arrives raises GP_STATUS_HANDOFF_CHAIN in the status word. */
size_t gp_gemm_streamk_workspace_bytes(void);
int gp_gemm_streamk_reset(float* scratch, void* stream);
int gp_gemm_kmajor_sk(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I, int J,
                      int K, int epilogue, const float* bias, const float* scale, const float* residual,
                      int ldr, float* scratch, size_t scratch_bytes, void* stream);

/* Split-f16 numerics of the dense layers (opt-in): D = epi(W . X) with every f32 operand x ~= hi + lo * 2^-11
 * (hi, lo f16) and three v_mfma_f32_32x32x16_f16 per k-block, f32 accumulation -- f32-equivalent accuracy
 * (tests/test_gpu_split.py: error vs f64 not above the fmaf chain's) at ~2x the speed of gp_gemm_kmajor.
 * Not bit-identical to gp_gemm_kmajor.
 *   gp_split_weights: W^T (K, n) f32 k-major (row stride ldw) -> planes hi, lo (n, K) f16.
 *   gp_gemm_split: act (K, ld_act) f32 k-major activations; whi / wlo (n_w, K) pre-split weights;
 *     act_is_b != 0: D[i][j] = sum_k W[i][k] act[k][j];  act_is_b == 0: D[i][j] = sum_k act[k][i] W[j][k].
 *     epilogues as gp_gemm_kmajor.  Requires I, J % 128 == 0, K % 32 == 0. */
int gp_split_weights(const float* Wt, int K, int n, int ldw, void* hi, void* lo, void* stream);
int gp_gemm_split(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J, int K,
                  int act_is_b, int epilogue, const float* bias, const float* scale, const float* residual, int ldr,
                  void* stream);

/* ---- DINOv2 ViT patch features: AENet.forward (src/models/network/ae_net.py:44-73) ---------- */

/* Bytes of device workspace gp_vit_forward needs for B crops. */
size_t gp_vit_workspace_bytes(int B, int dim, int mlp_dim);

/* images (B,3,224,224) f32 (CLIP-normalised crops) -> out_features (B, dim, 256):
 * `forward_features(x)["x_prenorm"][:, 1:, :]` rearranged "b (h w) c -> b c h w" and, if
 * normalize != 0, F.normalize(dim=1) (ae_net.py:64-69).  The final LayerNorm is not applied.
 * patch 14, 257 tokens, head dim 64 (dim == 64*heads), GELU(erf), LayerScale, pre-norm, ln_eps 1e-6.
 * `weights`: HOST array of n_weights = 4 + 16*depth DEVICE pointers, all f32, pre-transposed:
 *   [0] patch_w^T (592, dim)  rows k = ci*196 + dy*14 + dx, rows 588..591 zero
 *   [1] patch_b (dim)   [2] cls_token + pos_embed[0] (dim)   [3] pos_embed[1:]^T (dim, 256)
 *   per layer l at 4 + 16*l:  ln1_g, ln1_b, Wqk^T (dim, 2dim), bqk (2dim), Wv^T (dim, dim), bv,
 *                             Wproj^T (dim, dim), bproj, ls1, ln2_g, ln2_b, Wfc1^T (dim, mlp),
 *                             bfc1, Wfc2^T (mlp, dim), bfc2, ls2
 * stop_after_layers: < 0 = all layers (otherwise run only that many blocks; test hook).
 * After the call the workspace's first dim*Mpad floats hold x_prenorm^T (dim, Mpad),
 * column b*257 + t, Mpad = round_up(257*B, 256). */
int gp_vit_forward(const float* images, int B, int dim, int depth, int heads, int mlp_dim, float ln_eps,
                   const float* const* weights, int n_weights, float* workspace, size_t workspace_bytes,
                   float* out_features, int normalize, int stop_after_layers, void* stream);

/* Second-generation split GEMM (gp_split256.hip): 256 x 256 tiles, ONE accumulator on operands pre-scaled by powers
 * of two (activations x 8 while staging, weights x 64 in the planes made by gp_split256_weights: hi = f16(64 w),
 * lo = f16(64 w - hi)), work balanced by stream-K with deterministic accumulator hand-offs.  Same arguments as
 * gp_gemm_split plus a device scratch of gp_gemm_split256_workspace_bytes() bytes (one per stream); requires
 * I, J % 256 == 0, K % 32 == 0, (I/256)*(J/256) >= 256, |activation| < 8190; epilogues 1-4 (0 and 5 in the probe library).  A lost
 * hand-over raises GP_STATUS_HANDOFF_SPLIT in the status word. */
size_t gp_gemm_split256_workspace_bytes(void);
int gp_split256_weights(const float* W, size_t count, void* hi, void* lo, void* stream);
int gp_gemm_split256(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J, int K,
                     int act_is_b, int epilogue, const float* bias, const float* scale, const float* residual, int ldr,
                     float* scratch, size_t scratch_bytes, void* stream);

/* Third generation (gp_split256.hip, gemm_planes256_kernel): BOTH operands as pre-split f16 planes, A [I][K] and B [J][K] (k contiguous;
 * gp_split_planes makes them: hi = f16(scale x), lo = f16(scale x - hi); weights use scale 64, activations a power of two s, 8 by
 * default), the two wave groups of a workgroup running half a k-step apart (one issues MFMAs while the other stages).
 *   D[i][j] = epi(out_scale * sum_k A[i][k] B[j][k]),  out_scale = 1 / (scale_a * scale_b);
 *   epilogue 3 = residual[i][j] + scale[i] * (acc + bias[i]) as f32 D (ldd; D may alias residual); 6 = bias along i + GELU, 7 = bias
 *   along i, both written as activation planes out_hi / out_lo [j][i] (x plane_scale, row stride ldo) for the next kernel; the probe
 *   library adds the plain f32 epilogues 0, 1, 2, 4, 5.
 * Ragged J (257 tokens per crop are never a multiple of 256): only rows < J_valid of B carry data (J, the padded row count of the
 * buffers, stays a multiple of 256).  The 256 x 256 tiles cover floor(J_valid / 256) * 256 rows -- at B = 64 crops exactly one / three /
 * four whole tiles per CU, no stream-K hand-over -- and the remaining < 256 rows are computed as 32 x 32 fragments with the same
 * per-accumulator instruction sequence (bit-identical to tiled results).  Rows >= round_up(J_valid, 32) of the outputs are not written.
 * Needs (I / 256) * (J_valid / 256) >= 8 tiles and at least one k-step per slot; below 256 tiles the slots of a tile split its K in
 * parallel (fixed-order reduction), below 128 the tiles are 256 x 128.
 * plane_scale: the power of two the OUTPUT planes of epilogues 6 / 7 carry (the consumer GEMM then runs with out_scale = 1 / (64 s));
 * amax: NULL, or (calibration passes) a device float that receives max |x| of the planes written (atomic max on the f32 bits). */
int gp_split_planes(const float* X, size_t count, float scale, void* hi, void* lo, void* stream);
int gp_gemm_planes256_scaled(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, float* D, int ldd, void* out_hi,
                             void* out_lo, int ldo, int I, int J, int J_valid, int K, int epilogue, const float* bias, const float* scale,
                             const float* residual, int ldr, float out_scale, float plane_scale, float* amax, float* scratch,
                             size_t scratch_bytes, void* stream);
/* LayerNorm over C of X [C][Mpad] f32 (channel-major, as the residual stream is kept) -> token-major activation planes hi / lo
 * [Mpad][C] (x 8); HF modeling_dinov2.py:342-380 norm1 / norm2.  Stage entry of the plane path. */
int gp_layernorm_planes(const float* X, void* out_hi, void* out_lo, const float* gamma, const float* beta, int C, int Mpad, float eps,
                        void* stream);
/* softmax(q k^T / 8) v per (image, head) in split numerics (attention_split_kernel): qkv_hi/lo = f16 planes [Mpad][3 dim] of Q | K | V
 * (x qkv_scale, a power of two; token b*257 + t in row order), out_hi/lo = planes [Mpad][dim] (x qkv_scale) of the attention output;
 * replaces HF modeling_dinov2.py:207-229 inside AENet.forward for the split mode. */
int gp_attention_split_scaled(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, int B, int heads, int dim, int Mpad,
                              float qkv_scale, void* stream);

/* gp_vit_forward with the linear layers in split-f16 numerics: `split` = HOST array of n_split = 10*depth DEVICE pointers, per layer:
 * qk_hi, qk_lo (2dim, dim), v_hi, v_lo (dim, dim), proj_hi, proj_lo (dim, dim), fc1_hi, fc1_lo (mlp, dim), fc2_hi, fc2_lo (dim, mlp) --
 * f16 planes of the PyTorch-native [out][in] weights, w ~= hi + lo * 2^-11.  With n_split = 20*depth each layer carries ten more
 * pointers: the same five weights as gp_split256_weights planes; GEMMs whose shape fills the chip with 256 x 256 tiles then use
 * gp_gemm_split256.  When all five GEMMs of a layer fit the plane kernel (ViT-L from 8 crops; dim a multiple of 256), the activations
 * between the kernels travel as token-major f16 planes written by LayerNorm, attention and fc1's GELU epilogue, and every GEMM is
 * gemm_planes256_kernel.  split == NULL: identical to gp_vit_forward.  (Patch embedding, LayerNorm statistics and the feature epilogue
 * are the same f32 arithmetic in both modes.)
 * Per-tensor plane scales (round 5; DINOv2's massive activations; arithmetic: HF modeling_dinov2.py:342-380, the same forward).  The
 * plane path keeps four activation tensors per layer as f16 hi / lo planes of s x: [0] LayerNorm-1 output, [1] q | k | v and the
 * attention output, [2] LayerNorm-2 output, [3] GELU output; s is a power of two PER (layer, tensor):
 *   plane_scales  HOST array [depth][4] or NULL (= all 8); each in [2^-10, 64];
 *   plane_amax    DEVICE array [depth][4] f32 or NULL; non-NULL = calibration pass: every plane producer records max |x| of what it
 *                 wrote (atomic max on the f32 bits; the caller zeroes it) -- gigapose_amd/vit.py picks s from it with headroom.
 * Consumers undo the scale exactly (out_scale = 1 / (64 s)); the range guard (GP_STATUS_SPLIT_RANGE) stays |s x| <= 65504. */
int gp_vit_forward_split2(const float* images, int B, int dim, int depth, int heads, int mlp_dim, float ln_eps,
                          const float* const* weights, int n_weights, const void* const* split, int n_split,
                          float* workspace, size_t workspace_bytes, float* out_features, int normalize,
                          int stop_after_layers, const float* plane_scales, float* plane_amax, void* stream);

/* ---- IST backbone: ResNet.forward (src/models/network/resnet.py:364-381, BasicBlock :26-50) -- */

/* F.interpolate(x, (S,S), mode="bilinear", align_corners=True) (resnet.py:366-368):
 * images (B,C,IH,IW) NCHW -> out channel-major [C][B][S][S]. */
int gp_resize_bilinear_cm(const float* images, float* out, int B, int C, int IH, int IW, int S, void* stream);

/* Conv2d(bias=False) + eval BatchNorm + optional residual + optional ReLU on channel-major
 * activations X [Cin][B][H][W] -> Y [Cout][B][OH][OW] (or NCHW (B,Cout,OH,OW) when nchw_out):
 *   y = relu( residual + (conv(x) * alpha[co] + beta[co]) ),   alpha = gamma/sqrt(var+eps),
 *   beta = bias - mean*alpha (NULL alpha/beta = no BN; NULL residual = none).
 * Wt: (Kpad, Cout), row k = ci*KH*KW + dy*KW + dx, Kpad = round_up(Cin*KH*KW, 16), extra rows zero.
 * Implicit GEMM on the f32 matrix core; accumulation = sequential fmaf over k.
 * Requires Cout % 64 == 0 and B*OH*OW % 256 == 0. */
int gp_conv2d_cm(const float* X, const float* Wt, float* Y, const float* alpha, const float* beta,
                 const float* residual, int Cin, int B, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                 int relu, int nchw_out, void* stream);

/* Split-f16 numerics of the IST convolutions (opt-in; see gp_gemm_split): Conv2d(bias=False) + eval BatchNorm +
 * optional residual + optional ReLU on channel-LAST activations stored as f16 planes hi / lo (value ~= hi + lo * 2^-11):
 *   x_hi/x_lo (B,H,W,Cin); w_hi/w_lo (round_up(Cout,128), KH*KW*Cin) with k = (dy*KW + dx)*Cin + ci, padded rows zero;
 *   alpha/beta (Cout) folded BN or NULL; res_hi/res_lo (B,OH,OW,Cout) or NULL;
 *   output: planes out_hi/out_lo (B,OH,OW,Cout), or -- when out_f32_nchw != NULL -- f32 (B,Cout,OH,OW).
 * Requires Cin % 32 == 0, Cout % 64 == 0, B*OH*OW % 128 == 0.  (A CNHW f32 tensor [C][npix] becomes planes [npix][C]
 * with gp_split_weights(x, C, npix, npix, hi, lo).) */
int gp_conv2d_nhwc_split(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* alpha,
                         const float* beta, const void* res_hi, const void* res_lo, int B, int H, int W, int Cin, int Cout,
                         int KH, int KW, int stride, int pad, int relu, void* out_hi, void* out_lo, float* out_f32_nchw,
                         void* stream);

/* Second-generation split convolution (gp_conv256.hip, conv_planes_kernel): the plane GEMM's structure as an implicit
 * GEMM -- 256-pixel x (128 | 192 | 256)-channel tiles, ONE accumulator, half-step-offset wave groups, stream-K remainder,
 * epilogue through LDS.  Same operation and argument meaning as gp_conv2d_nhwc_split, but the planes follow the
 * single-accumulator convention of gp_gemm_planes256: activations / residual / output hi = f16(8 x), lo = f16(8 x - hi)
 * (|x| < 8190, guarded through the status word), weights (Cout, KH*KW*Cin) planes of 64 w (gp_split256_weights), k =
 * (dy*KW + dx)*Cin + ci.  Requires Cin % 32 == 0, Cout % 64 == 0, KH*KW <= 9, B*OH*OW % 256 == 0; scratch of
 * gp_conv2d_planes_workspace_bytes() bytes (stream-K hand-overs; zeroed ONCE by the caller at allocation -- launches tag their
 * hand-off flags with a per-launch epoch and never reset them).  gp_planes_from_cm: f32 [C][npix] -> such planes [npix][C]. */
size_t gp_conv2d_planes_workspace_bytes(void);
int gp_planes_from_cm(const float* X, int C, int npix, void* hi, void* lo, void* stream);
/* 3 x 3 / stride 1 / pad 1 convolutions on images whose sides are multiples of 16 run conv_halo_kernel (16 x 16 pixel blocks whose
 * 18 x 18 halo is staged in LDS once per 32 input channels; the nine taps read it at shifted rows) -- same arguments, results equal
 * to conv_planes_kernel's to f32 round-off (summation order (channel block, tap) instead of (tap, channel block)). */
int gp_conv2d_planes(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* alpha, const float* beta,
                     const void* res_hi, const void* res_lo, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                     int relu, void* out_hi, void* out_lo, float* out_f32_nchw, float* scratch, size_t scratch_bytes, void* stream);

/* The ResNet stem (reference resnet.py:333-337, 366-370: bilinear resize to S x S, Conv2d(3 -> Cout, 7 x 7, stride 2, padding 3, no
 * bias) + BatchNorm + ReLU) in split numerics: gp_resize_stem_planes writes the resized crops as 4-channel planes (B, S + 6, S + 8,
 * 4) x 8 inside a frame of zeros (the caller zeroes the buffers ONCE: the frame is never written) so that a kernel row of 8 taps is
 * 64 contiguous bytes and no tap needs a range check; gp_conv2d_stem_planes runs conv_planes_kernel over them, one kernel row per
 * k-step: k = dy * 32 + dx * 4 + ci, K = 224, weight planes (Cout, 224) of 64 w with zeros at dx = 7 and ci = 3; output planes
 * (B * (S/2)^2, Cout) x 8. */
int gp_resize_stem_planes(const float* images, void* hi, void* lo, int B, int IH, int IW, int S, void* stream);
int gp_conv2d_stem_planes(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* alpha, const float* beta,
                          int B, int S, int Cout, int relu, void* out_hi, void* out_lo, float* scratch, size_t scratch_bytes, void* stream);

/* ---- IST regressor: ISTNet.inference (src/models/network/ist_net.py:97-120) ----------------- */

size_t gp_ist_workspace_bytes(int B, int k, int D, int H);

/* For every (detection b, hypothesis j, patch t) row: gather tar_feat[b][:, tar_pt] and
 * src_bank[labels[b]][id_src[b][j]][:, src_pt] (src/utils/batch.py:46-73, index = y*16+x), concat
 * [tar, src] (2D), run scale MLP 2D->2H->H->1 and in-plane MLP 2D->2H->H->2 (+tanh)
 * (ist_net.py:140-155).  Rows whose points are -1 get -1000 (ist_net.py:109-119).
 *   tar_feat (B,D,256)  src_bank (O,N,D,256)  labels (B) int32  id_src (B,k) int64
 *   tar_pts, src_pts (B,k,256,2) int64   ->  scales (B,k,256)  cos_sin (B,k,256,2)
 * weights: HOST array of 12 DEVICE pointers: for scale then in-plane head:
 *   W1^T (2D,2H), b1 (2H), W2^T (2H,H), b2 (H), W3 (nout,H), b3 (nout).
 * Requires 2D % 16 == 0, H % 128 == 0.  n_weights = 20: eight more pointers select the split numerics for the hidden layers
 * (per head: W1 hi, lo ([2H][2D]), W2 hi, lo ([H][2H]) f16 planes of the PyTorch-native [out][in] weights, w ~= hi + lo 2^-11). */
int gp_ist_regress(const float* tar_feat, const float* src_bank, const int* labels, const long long* id_src,
                   const long long* tar_pts, const long long* src_pts, int B, int O, int N, int k, int D,
                   int H, const float* const* weights, int n_weights, int use_tanh, float* workspace,
                   size_t workspace_bytes, float* scales, float* cos_sin, void* stream);

/* ---- 2-D similarity voting: RANSAC.forward (src/models/ransac.py:108-172) -------------------- */

/* R independent problems of 256 padded correspondences (valid where src_pts.x != -1).  Each valid
 * correspondence proposes M = [s*R | t]; inliers = other correspondences within pixel_threshold;
 * first maximum wins.  Outputs: M (R,3,3); failed (R) u8 = (best count == 0); the winner's inliers
 * packed at the front of inl_src/inl_tar (R,256,2) int64 (pad -1) and inl_score (R,256) int64 (pad 0).
 * No valid correspondence: M = I, failed = 0.
 * RANSAC.forward(batch, scores=...) (ransac.py:108-121, 98): `score` (R,256) f32 weights every correspondence (NULL = ones, the plain
 * forward above): a candidate's score is the f32 sum, in ascending order, of the weights of the other correspondences within the
 * threshold; failed = (best score == 0); inl_score = the winners' weights cast to int64 as the reference's assignment does. */
int gp_ransac_scored(const long long* src_pts, const long long* tar_pts, const float* rel_scale,
                     const float* rel_inplane, const float* score, int R, float patch_size, float pixel_threshold, float* M,
                     unsigned char* failed, long long* inl_src, long long* inl_tar, long long* inl_score, void* stream);

/* ---- pose recovery: ObjectPoseRecovery.forward_recovery (src/models/poses.py:26-122) --------- */

/* labels (B) int32 0-based, tar_K, tar_M (B,3,3), id_src (B,k) int64, pred_M (B,k,3,3),
 * tmpl_K (O,3,3), tmpl_M (O,N,3,3), tmpl_pose (O,N,4,4) -> poses (B,k,4,4).
 * *bad_crop_M (device int, caller zeroes it) is OR-ed with 1 when a tar_M violates the reference's
 * assert (src/lib3d/torch.py:54-55: isotropic scale + translation). */
int gp_recover_poses(const int* labels, const float* tar_K, const float* tar_M, const long long* id_src,
                     const float* pred_M, const float* tmpl_K, const float* tmpl_M, const float* tmpl_pose, int B,
                     int O, int N, int k, float* poses, int* bad_crop_M, void* stream);

/* ---- hypothesis ranking: GigaPose.eval_retrieval's score + sort (src/models/gigaPose.py:588-594) ----
 * inl_score (B,k,P) int64 = RANSAC inlier scores -> scores (B,k) = sum / P in float32, written in ranked order; order (B,k)
 * int64 = the hypothesis index at each rank (sort != 0: descending score, ties keep the lower index; sort == 0: identity).
 * Each of the n_tensors (<= 16) tensors src[t] (B,k,row_bytes[t] bytes) is copied to dst[t] with its k rows in ranked
 * order (dst != src).  Replaces torch.sum + torch.argsort + `v[rows, order]` per tensor. */
int gp_rank_hypotheses(const long long* inl_score, int B, int k, int P, int sort, float* scores, long long* order, int n_tensors,
                       const void* const* src, void* const* dst, const int* row_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GIGAPOSE_HIP_H */
