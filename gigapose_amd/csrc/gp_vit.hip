// DINOv2 ViT forward for gfx950, f32, activations kept channel-major (X^T: [C][tokens]).
//
// Replaces the un-vendored backbone the reference calls at ae_net.py:44-47
// (`dinov2_model.forward_features(x)["x_prenorm"]`, wired in configs/model/ae_net/dinov2_l.yaml);
// block structure per HF transformers modeling_dinov2.py:97-112 (embeddings), :199-229
// (attention), :272-299 (LayerScale, MLP), :342-380 (pre-norm block), eps 1e-6.
// The final LayerNorm is NOT applied (GigaPose consumes x_prenorm); the epilogue drops CLS and
// L2-normalises over C (ae_net.py:64-69) straight into the (B, C, 16, 16) layout the matcher reads.
//
// Token column of image b, token t:  m = b * T + t   (T = 257, no per-image padding);
// all activation matrices have Mpad = round_up(B*T, 256) columns.
#include "gp_common.h"

int gp_gemm_launch(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I, int J,
                   int K, int epilogue, const float* bias, const float* scale, const float* res, int ldr,
                   float* sk_ws, hipStream_t st);
int gp_gemm_split_launch(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J,
                         int K, int act_is_b, int epilogue, const float* bias, const float* scale, const float* res, int ldr,
                         hipStream_t st);
bool gp_gemm_split256_usable(int I, int J, int K);
int gp_gemm_split256_launch(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J,
                            int K, int act_is_b, int epilogue, const float* bias, const float* scale, const float* res, int ldr,
                            float* scratch, hipStream_t st);
size_t gp_gemm_streamk_bytes();
int gp_gemm_streamk_reset_launch(float* sk_ws, hipStream_t st);
bool gp_gemm_split256_usable(int I, int J, int K);
bool gp_gemm_planes256_usable(int I, int J, int J_valid, int K);
int gp_gemm_planes256_launch(const void* ahi, const void* alo, const void* bhi, const void* blo, float* D, int ldd, void* ohi,
                             void* olo, int ldo, int I, int J, int J_valid, int K, int epilogue, const float* bias, const float* scale,
                             const float* res, int ldr, float out_scale, float* scratch, hipStream_t st, unsigned long long* trace = nullptr,
                             const GpPlaneOut* po = nullptr);

namespace {

constexpr int PATCH = 14, IMG = 224, KPE = 588, KPE_PAD = 592, T_TOK = 257;

// ---- patch-embed operand: col[k][b*256+p] = img[b][ci][14py+dy][14px+dx], k = ci*196+dy*14+dx
// (flatten order of Conv2d weight (C,3,14,14); HF modeling_dinov2.py:139).  Rows 588..591 = 0.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, float* __restrict__ col, int B)
{
    const int k = blockIdx.x, b = blockIdx.y, p = threadIdx.x;
    float v = 0.f;
    if (k < KPE) {
        const int ci = k / 196, dy = (k % 196) / PATCH, dx = k % PATCH;
        const int py = p >> 4, px = p & 15;
        v = img[(((size_t)b * 3 + ci) * IMG + (py * PATCH + dy)) * IMG + px * PATCH + dx];
    }
    col[(size_t)k * (B * GP_P) + (size_t)b * GP_P + p] = v;
}

// ---- tokens = cat(cls, patches) + pos  (HF modeling_dinov2.py:108-112); zero the pad columns
__global__ __launch_bounds__(320) void embed_kernel(const float* __restrict__ pe /*[C][B*256]*/,
                                                     const float* __restrict__ cls_pos /*[C]*/,
                                                     const float* __restrict__ pos_t /*[C][256]*/,
                                                     float* __restrict__ X, int B, int Mpad)
{
    const int c = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    if (b == B) {  // extra block row: clear padding columns [B*T, Mpad)
        for (int m = B * T_TOK + t; m < Mpad; m += 320) X[(size_t)c * Mpad + m] = 0.f;
        return;
    }
    if (t >= T_TOK) return;
    float v;
    if (t == 0) v = cls_pos[c];
    else v = pe[(size_t)c * (B * GP_P) + (size_t)b * GP_P + (t - 1)] + pos_t[(size_t)c * GP_P + (t - 1)];
    X[(size_t)c * Mpad + (size_t)b * T_TOK + t] = v;
}

// ---- LayerNorm over C of channel-major X [C][Mpad]; thread = token column (coalesced)
__global__ __launch_bounds__(128) void layernorm_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int C, int Mpad,
                                                         float eps)
{
    const int m = blockIdx.x * 128 + threadIdx.x;
    const float* x = X + m;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += x[(size_t)c * Mpad];
    const float mean = s / (float)C;
    float v = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = x[(size_t)c * Mpad] - mean;
        v = __builtin_fmaf(d, d, v);
    }
    const float rstd = 1.0f / __builtin_sqrtf(v / (float)C + eps);
    for (int c = 0; c < C; ++c)
        Y[(size_t)c * Mpad + m] = (x[(size_t)c * Mpad] - mean) * rstd * gamma[c] + beta[c];
}

// ---- LayerNorm, wide: block = 64 token columns x 16 channel slices (16 waves, 4128 waves in flight at
// B=64 instead of 258), three short passes; the block's 64 x C tile is re-read from L2.
__global__ __launch_bounds__(1024) void layernorm_wide_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int C, int Mpad,
                                                               float eps)
{
    __shared__ float red[16][64];
    const int tok = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int per = C >> 4, c0 = sl * per;
    const float* x = X + (size_t)c0 * Mpad + (size_t)blockIdx.x * 64 + tok;
    float* y = Y + (size_t)c0 * Mpad + (size_t)blockIdx.x * 64 + tok;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < per; ++i) s += x[(size_t)i * Mpad];
    red[sl][tok] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) tot += red[j][tok];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.f;
#pragma unroll 8
    for (int i = 0; i < per; ++i) {
        const float d = x[(size_t)i * Mpad] - mean;
        q = __builtin_fmaf(d, d, q);
    }
    red[sl][tok] = q;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) tot += red[j][tok];
    const float rstd = 1.0f / __builtin_sqrtf(tot / (float)C + eps);
#pragma unroll 8
    for (int i = 0; i < per; ++i)
        y[(size_t)i * Mpad] = (x[(size_t)i * Mpad] - mean) * rstd * gamma[c0 + i] + beta[c0 + i];
}

int launch_layernorm(const float* X, float* Y, const float* g, const float* b, int C, int Mpad, float eps,
                     hipStream_t st)
{
    GpProfScope prof(GP_PROF_LN, 8.0 * C * Mpad, st);
    if (C % 16 == 0)
        hipLaunchKernelGGL(layernorm_wide_kernel, dim3(Mpad / 64), dim3(1024), 0, st, X, Y, g, b, C, Mpad, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel, dim3(Mpad / 128), dim3(128), 0, st, X, Y, g, b, C, Mpad, eps);
    return 0;
}


// ---- activation planes (split numerics, gp_split256.hip's plane x plane GEMM): the consumer wants its activations
// token-major as two f16 planes hi = f16(8 y), lo = f16(8 y - hi).  Same block shape and the same arithmetic as
// layernorm_wide_kernel for the statistics (so y is bit-identical); the last pass transposes through LDS so that the
// plane stores are 256 contiguous bytes per token row.
typedef _Float16 v16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 v16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int au32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kPlaneScale = 8.0f;  // == kActScale of gp_split256.hip

__global__ __launch_bounds__(1024) void layernorm_planes_kernel(const float* __restrict__ X, _Float16* __restrict__ Yhi,
                                                                 _Float16* __restrict__ Ylo, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int C, int Mpad, float eps,
                                                                 int* __restrict__ status, float plane_scale, float* __restrict__ amax)
{
    __shared__ float red[16][64];
    const int tok = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int per = C >> 4, c0 = sl * per;
    const float* x = X + (size_t)c0 * Mpad + (size_t)blockIdx.x * 64 + tok;
    int bad = 0;  // range guard of the x 8 planes (gp_common.h: GP_ST_SPLIT_RANGE)
    float vmax = 0.f;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < per; ++i) s += x[(size_t)i * Mpad];
    red[sl][tok] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) tot += red[j][tok];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.f;
#pragma unroll 8
    for (int i = 0; i < per; ++i) {
        const float d = x[(size_t)i * Mpad] - mean;
        q = __builtin_fmaf(d, d, q);
    }
    red[sl][tok] = q;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) tot += red[j][tok];
    const float rstd = 1.0f / __builtin_sqrtf(tot / (float)C + eps);
    // pass 3, re-sliced: chunks of 128 channels; wave sl computes channels 8 sl .. 8 sl + 7 of the chunk for its 64 tokens
    // (coalesced reads along tokens), packs (hi, lo) into one word and writes tile[channel][token]; after the barrier the
    // block reads it transposed: wave sl = (token octet sl & 7, channel half sl >> 3), lane = (token lane & 7, channel
    // group lane >> 3) -> 8 lanes x 16 bytes cover 128 contiguous bytes of a token row per plane (bank = 8 g + t + e:
    // two lanes per bank, the minimum for 64 lanes).  Two tiles, one barrier per chunk.
    __shared__ unsigned int tile[2][128 * 65];
    const float* xb = X + (size_t)blockIdx.x * 64 + tok;
    const int t2 = 8 * (sl & 7) + (tok & 7), g2 = 8 * (sl >> 3) + (tok >> 3);
    const size_t row = ((size_t)blockIdx.x * 64 + t2) * C + 8 * g2;
    const int nchunk = C >> 7;
    for (int k = 0; k < nchunk; ++k) {
        unsigned int* T = tile[k & 1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cl = 8 * sl + e, c = 128 * k + cl;
            const float y = (xb[(size_t)c * Mpad] - mean) * rstd * gamma[c] + beta[c];
            const float v = y * plane_scale;
            const _Float16 hh = (_Float16)v;
            const _Float16 ll = (_Float16)(v - (float)hh);
            bad |= !(fabsf(v) <= kSplitPlaneLimit);  // !(<=): a NaN / inf residual stream counts
            vmax = __builtin_elementwise_maximum(vmax, __builtin_fabsf(v));
            T[cl * 65 + tok] = (unsigned int)__builtin_bit_cast(unsigned short, hh) |
                               ((unsigned int)__builtin_bit_cast(unsigned short, ll) << 16);
        }
        __syncthreads();
        v16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned int w = T[(8 * g2 + e) * 65 + t2];
            h[e] = __builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
            l[e] = __builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
        }
        *reinterpret_cast<v16x8*>(Yhi + row + 128 * k) = h;
        *reinterpret_cast<v16x8*>(Ylo + row + 128 * k) = l;
    }
    if (bad) gp_raise(status, GP_ST_SPLIT_RANGE);
    if (amax) gp_record_amax(amax, vmax, 1.0f / plane_scale);
}

// Second version: X is read ONCE.  Block = 32 tokens x 16 channel slices (512 threads, two blocks per CU); thread (tok, sl)
// keeps its C / 16 values -- channels 128 k + 8 sl + e, the ones it later converts -- in registers through both statistics
// passes and the conversion (the kernel above re-reads X from L2 for each of its three passes: 72 us per launch = 1.9 TB/s
// for a 136 MB job, 6 % of the step).  Two-pass mean / variance as before (the per-thread summation order differs from
// layernorm_wide_kernel's contiguous slices: y agrees with it to f32 round-off, not bit for bit).  Conversion: each thread
// packs (hi, lo) of its 8 channels of chunk k into one 32-byte LDS row piece; after the barrier 16 consecutive lanes read
// the 16 pieces of one token and store 256 contiguous bytes per plane.
// TOK (round 5): tokens per block.  32 by default; 16 when 32-token blocks would cover at most half the CUs (ViT-L up to 15 crops): 16
// tokens x 32 channel slices -- twice the blocks, half the values per thread (a chunk is then 256 channels, a token's store 512
// contiguous bytes per plane).  Measured (tools/gpu_r05_ln16.sh, two alternating rounds on one box): 8 crops 12.1 -> 10.6 us per launch
// (the 8-crop step 8.93 -> 8.82 ms); at 16 / 24 crops (129 / 193 blocks) the 16-token form is SLOWER (13.0 -> 15.6, 15.1 -> 17.3 us: twice
// the blocks with 64-byte token segments) -- hence the threshold at 128 blocks.  The per-thread summation order of the statistics differs
// between the two forms: they agree to f32 round-off, not bit for bit -- like two batch shapes.
template <int NK, int TOK = 32>
__global__ __launch_bounds__(512) void layernorm_planes_reg_kernel(const float* __restrict__ X, _Float16* __restrict__ Yhi,
                                                                    _Float16* __restrict__ Ylo, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int Mpad, float eps,
                                                                    int* __restrict__ status, float plane_scale, float* __restrict__ amax)
{
    constexpr int C = 128 * NK, SL = 512 / TOK, CH = 8 * SL, NC = C / CH;  // slices, channels per chunk, chunks
    constexpr int TP = CH + 4;  // LDS row pitch in words: 16-byte aligned pieces, consecutive tokens one 16-byte slot off the bank period
    static_assert(C % CH == 0 && (TOK == 32 || TOK == 16), "layernorm_planes_reg_kernel: C must be a multiple of the chunk width");
    __shared__ float red[SL][TOK];
    __shared__ __attribute__((aligned(16))) unsigned int tile[2][TOK * TP];
    // gamma | beta once per block through LDS (requested with the activations, visible after the first barrier): read from global
    // memory inside the chunk loop they were eight dependent L2 round trips per block -- hidden by the other blocks of a CU at 64
    // crops, 4 of the 13 us of a launch at 8 (65 blocks on 256 CUs)
    __shared__ __attribute__((aligned(16))) float gb[2][C];
    const int tok = threadIdx.x & (TOK - 1), sl = threadIdx.x / TOK;
    const size_t tok0 = (size_t)blockIdx.x * TOK;
    float xv[NC][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xv[k][e] = X[(size_t)(CH * k + 8 * sl + e) * Mpad + tok0 + tok];
            s += xv[k][e];
        }
    for (int c = threadIdx.x; c < C; c += 512) {
        gb[0][c] = gamma[c];
        gb[1][c] = beta[c];
    }
    red[sl][tok] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < SL; ++j) tot += red[j][tok];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = xv[k][e] - mean;
            q = __builtin_fmaf(d, d, q);
        }
    red[sl][tok] = q;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int j = 0; j < SL; ++j) tot += red[j][tok];
    const float rstd = 1.0f / __builtin_sqrtf(tot / (float)C + eps);
    const int g2 = threadIdx.x & (SL - 1), t2 = threadIdx.x / SL;  // store phase: SL lanes = the SL eight-channel pieces of token t2
    const size_t row = (tok0 + t2) * C + 8 * g2;
    int bad = 0;
    float vmax = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        unsigned int* T = tile[k & 1];
        const int c = CH * k + 8 * sl;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gb[0] + c), g1 = *reinterpret_cast<const f32x4*>(gb[0] + c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(gb[1] + c), b1 = *reinterpret_cast<const f32x4*>(gb[1] + c + 4);
        unsigned int w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float y = (xv[k][e] - mean) * rstd * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
            const float v = y * plane_scale;   // the tensor's power-of-two scale (8 unless calibrated otherwise): exact
            const _Float16 hh = (_Float16)v;
            const _Float16 ll = (_Float16)(v - (float)hh);
            bad |= !(fabsf(v) <= kSplitPlaneLimit);  // !(<=): a NaN / inf residual stream counts
            if (amax) vmax = __builtin_elementwise_maximum(vmax, __builtin_fabsf(v));   // calibration passes only (uniform branch)
            w[e] = (unsigned int)__builtin_bit_cast(unsigned short, hh) | ((unsigned int)__builtin_bit_cast(unsigned short, ll) << 16);
        }
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        u4 w0 = {w[0], w[1], w[2], w[3]}, w1 = {w[4], w[5], w[6], w[7]};
        *reinterpret_cast<u4*>(T + tok * TP + 8 * sl) = w0;
        *reinterpret_cast<u4*>(T + tok * TP + 8 * sl + 4) = w1;
        __syncthreads();  // one barrier per chunk: the other tile is rewritten only after every thread passed this one
        const u4 r0 = *reinterpret_cast<const u4*>(T + t2 * TP + 8 * g2), r1 = *reinterpret_cast<const u4*>(T + t2 * TP + 8 * g2 + 4);
        v16x8 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = __builtin_bit_cast(_Float16, (unsigned short)(r0[e] & 0xffffu));
            l[e] = __builtin_bit_cast(_Float16, (unsigned short)(r0[e] >> 16));
            h[4 + e] = __builtin_bit_cast(_Float16, (unsigned short)(r1[e] & 0xffffu));
            l[4 + e] = __builtin_bit_cast(_Float16, (unsigned short)(r1[e] >> 16));
        }
        *reinterpret_cast<v16x8*>(Yhi + row + CH * k) = h;
        *reinterpret_cast<v16x8*>(Ylo + row + CH * k) = l;
    }
    if (bad) gp_raise(status, GP_ST_SPLIT_RANGE);
    if (amax) gp_record_amax(amax, vmax, 1.0f / plane_scale);
}


static int g_ln_planes_reg = 1;  // A/B hook (gp_vit_set_ln_reg): 0 = the three-pass kernel, 2 = always the 32-token blocks
#ifdef GP_PROBES
extern "C" void gp_vit_set_ln_reg(int on) { g_ln_planes_reg = (on >= 0 && on <= 2) ? on : 1; }
#endif

int launch_layernorm_planes(const float* X, _Float16* hi, _Float16* lo, const float* g, const float* b, int C, int Mpad, float eps,
                            hipStream_t st, float plane_scale = kPlaneScale, float* amax = nullptr)
{
    GpProfScope prof(GP_PROF_LN, 8.0 * C * Mpad, st);
    if (g_ln_planes_reg && C == 1024 && Mpad % 32 == 0 && Mpad / 32 <= 128 && g_ln_planes_reg != 2)   // at most half a block per CU: 16-token blocks
        hipLaunchKernelGGL((layernorm_planes_reg_kernel<8, 16>), dim3(Mpad / 16), dim3(512), 0, st, X, hi, lo, g, b, Mpad, eps, gp_status_buffer(), plane_scale, amax);
    else if (g_ln_planes_reg && C == 1024 && Mpad % 32 == 0)
        hipLaunchKernelGGL(layernorm_planes_reg_kernel<8>, dim3(Mpad / 32), dim3(512), 0, st, X, hi, lo, g, b, Mpad, eps, gp_status_buffer(), plane_scale, amax);
    else if (g_ln_planes_reg && C == 768 && Mpad % 32 == 0)
        hipLaunchKernelGGL(layernorm_planes_reg_kernel<6>, dim3(Mpad / 32), dim3(512), 0, st, X, hi, lo, g, b, Mpad, eps, gp_status_buffer(), plane_scale, amax);
    else
        hipLaunchKernelGGL(layernorm_planes_kernel, dim3(Mpad / 64), dim3(1024), 0, st, X, hi, lo, g, b, C, Mpad, eps, gp_status_buffer(), plane_scale, amax);
    return 0;
}

// ---- attention, one wave per (image, head, 32-query block); everything in registers.
// S^T tile trick: compute D[i=key][j=query] = sum_d K[d][key] * Q[d][query] so that a lane owns ONE
// query column and 16 key rows per tile: softmax over keys is in-register (+1 cross-half shuffle),
// and the P values sit exactly where the P.V MFMA wants its B operand (k-slot = lane>>5), so P never
// moves.  V is token-major (Vt[token][C]) so its A-operand loads are coalesced.
// HF modeling_dinov2.py:207-229: softmax(q k^T * 64^-0.5) v   (0.125 is exact, so scaling after the
// dot product equals scaling q first).
constexpr int NKT = 9;  // ceil(257 / 32) key tiles

// NQ query tiles of 32 per wave share every K / V operand load (one 4-byte global load feeds NQ MFMAs).
// PLANES: the output goes to token-major activation planes (Ohi / Olo pre-offset to this image's first token and this
// head's first channel; row stride C) instead of channel-major f32 -- the same values o * inv, split as 8 x.
template <int NQ, bool PLANES = false>
__device__ __forceinline__ void attention_body(const float* __restrict__ Qp, const float* __restrict__ Kp,
                                               const float* __restrict__ Vp, float* __restrict__ Op, int q0, int C,
                                               int Mpad, float scale, _Float16* __restrict__ Ohi = nullptr,
                                               _Float16* __restrict__ Olo = nullptr, unsigned int* Tile = nullptr)
{
    const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
    int tq[NQ], tq_c[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        tq[u] = q0 + 32 * u + l31;
        tq_c[u] = tq[u] < T_TOK ? tq[u] : T_TOK - 1;
    }
    // Online softmax over 3 chunks of 3 key tiles (96 keys).
    f32x16 o0[NQ], o1[NQ];
    float m_run[NQ], l_part[NQ];  // l_part: this lane-half's share of the softmax denominator
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[u][r] = 0.f; o1[u][r] = 0.f; }
        m_run[u] = -INFINITY;
        l_part[u] = 0.f;
    }

#pragma unroll 1
    for (int ch = 0; ch < 3; ++ch) {
        f32x16 s[3][NQ];
        int tk_c[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int u = 0; u < NQ; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][u][r] = 0.f;
            const int tk = (ch * 3 + t) * 32 + l31;
            tk_c[t] = tk < T_TOK ? tk : T_TOK - 1;
        }
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const size_t drow = (size_t)(2 * kk + half) * Mpad;
            float qv[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) qv[u] = Qp[drow + tq_c[u]];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float kv = Kp[drow + tk_c[t]];
#pragma unroll
                for (int u = 0; u < NQ; ++u) s[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qv[u], s[t][u], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            float cmax = -INFINITY;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tk = (ch * 3 + t) * 32 + frag_row(r, lane);
                    const float v = (tk < T_TOK) ? s[t][u][r] * scale : -INFINITY;
                    s[t][u][r] = v;
                    cmax = fmaxf(cmax, v);
                }
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
            const float m_new = fmaxf(m_run[u], cmax);
            const float alpha = expf(m_run[u] - m_new);  // first chunk: exp(-inf) = 0
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = expf(s[t][u][r] - m_new);
                    s[t][u][r] = p;
                    psum += p;
                }
            l_part[u] = l_part[u] * alpha + psum;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[u][r] *= alpha; o1[u][r] *= alpha; }
            m_run[u] = m_new;
        }
        // O[d][tq] += sum_tk V[tk][d] * P[tk][tq]; the A-operand lane (i = d, k-slot = half) reads
        // V[tk = tile*32 + frag_row(r, lane)][d] -- the key this lane's P register r belongs to.
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int tk = (ch * 3 + t) * 32 + frag_row(r, lane);
                tk = tk < T_TOK ? tk : T_TOK - 1;  // P is 0 there; keep the load in bounds / finite
                const float* vrow = Vp + (size_t)tk * C + l31;
                const float v0 = vrow[0], v1 = vrow[32];
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    o0[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[t][u][r], o0[u], 0, 0, 0);
                    o1[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[t][u][r], o1[u], 0, 0, 0);
                }
            }
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const float inv = 1.0f / (l_part[u] + __shfl_xor(l_part[u], 32));
        if (PLANES) {
            // transpose through the wave's LDS tile [32 queries][64 channels (+1)] of packed (hi, lo) words, then 16-byte
            // stores: 8 lanes cover the 128 contiguous bytes of a token row's head slice
#pragma unroll
            for (int half_d = 0; half_d < 2; ++half_d)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = ((half_d ? o1[u][r] : o0[u][r]) * inv) * kPlaneScale;
                    const _Float16 hh = (_Float16)v;
                    const _Float16 ll = (_Float16)(v - (float)hh);
                    Tile[l31 * 65 + 32 * half_d + frag_row(r, lane)] =
                        (unsigned int)__builtin_bit_cast(unsigned short, hh) | ((unsigned int)__builtin_bit_cast(unsigned short, ll) << 16);
                }
            __syncthreads();
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int tok = 8 * pass + (lane >> 3), g = lane & 7;
                v16x8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned int w = Tile[tok * 65 + 8 * g + e];
                    h[e] = __builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
                    l[e] = __builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
                }
                if (q0 + 32 * u + tok < T_TOK) {
                    const size_t o = (size_t)(q0 + 32 * u + tok) * C + 8 * g;
                    *reinterpret_cast<v16x8*>(Ohi + o) = h;
                    *reinterpret_cast<v16x8*>(Olo + o) = l;
                }
            }
        } else if (tq[u] < T_TOK) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = frag_row(r, lane);
                Op[(size_t)d * Mpad + tq[u]] = o0[u][r] * inv;
                Op[(size_t)(32 + d) * Mpad + tq[u]] = o1[u][r] * inv;
            }
        }
    }
}

// One wave per (image, head, query block).  NQ = 1: nine blocks of 32 queries (the first version; kept as the A/B
// reference, results are bit-identical).  NQ = 2: four blocks of 64 queries + one of 32 (the 257th token), so eight
// of the nine query tiles share their K / V operand loads pairwise.
template <int NQ>
__global__ __launch_bounds__(64, NQ == 1 ? 4 : 2) void attention_kernel(const float* __restrict__ QK /*[2C][Mpad]*/,
                                                                         const float* __restrict__ Vt /*[Mpad][C]*/,
                                                                         float* __restrict__ O /*[C][Mpad]*/, int B, int H,
                                                                         int C, int Mpad, float scale)
{
    constexpr int NB = NQ == 1 ? NKT : 5;
    const int q = xcd_chunked_tile(blockIdx.x, B * H * NB);
    if (q < 0) return;
    const int qb = q % NB, bh = q / NB, h = bh % H, b = bh / H;
    const float* Qp = QK + (size_t)(h * 64) * Mpad + (size_t)b * T_TOK;
    const float* Kp = QK + (size_t)(C + h * 64) * Mpad + (size_t)b * T_TOK;
    const float* Vp = Vt + (size_t)b * T_TOK * C + h * 64;
    float* Op = O + (size_t)(h * 64) * Mpad + (size_t)b * T_TOK;
    if (NQ == 1) attention_body<1>(Qp, Kp, Vp, Op, qb * 32, C, Mpad, scale);
    else if (qb < 4) attention_body<2>(Qp, Kp, Vp, Op, qb * 64, C, Mpad, scale);
    else attention_body<1>(Qp, Kp, Vp, Op, 256, C, Mpad, scale);
}

// NQ = 1 with the output written as activation planes [Mpad][C] (split numerics, plane x plane GEMMs).  Direct 8-byte
// stores from the accumulator layout instead of the LDS transpose measured the same end to end (966-969 crops/s).
__global__ __launch_bounds__(64, 4) void attention_planes_kernel(const float* __restrict__ QK, const float* __restrict__ Vt,
                                                                  _Float16* __restrict__ Ohi, _Float16* __restrict__ Olo, int B, int H,
                                                                  int C, int Mpad, float scale)
{
    const int q = xcd_chunked_tile(blockIdx.x, B * H * NKT);
    if (q < 0) return;
    const int qb = q % NKT, bh = q / NKT, h = bh % H, b = bh / H;
    const float* Qp = QK + (size_t)(h * 64) * Mpad + (size_t)b * T_TOK;
    const float* Kp = QK + (size_t)(C + h * 64) * Mpad + (size_t)b * T_TOK;
    const float* Vp = Vt + (size_t)b * T_TOK * C + h * 64;
    const size_t o = (size_t)b * T_TOK * C + h * 64;
    __shared__ unsigned int tile[32 * 65];
    attention_body<1, true>(Qp, Kp, Vp, nullptr, qb * 32, C, Mpad, scale, Ohi + o, Olo + o, tile);
}

// ---------------------------------------------------------------------------------------------------------------
// Attention in split numerics.  Inputs: Q | K | V of every token as f16 planes [Mpad][3C] (x 8; written by the qk and v
// GEMMs' plane epilogue), output: activation planes [Mpad][C] for proj.  One workgroup per (image, head), eight waves =
// the eight full 32-query tiles (query 256 takes a vector-ALU path first); K [288 keys][64] and V^T [64][288 keys] (both
// planes) are staged in LDS ONCE and shared by the waves (the f32 kernel re-reads them from L2 per query tile).  Per wave the S^T trick of attention_body:
//   S^T[key][query] = sum_d K[key][d] Q[query][d]   as  3 x v_mfma_f32_32x32x16_f16 per 16-d block (hi hi, hi lo, lo hi),
// so a lane owns one query column, the softmax over keys is in-register, and its P registers 8m .. 8m+7 ARE the B
// fragment of the P.V MFMA once the hardware k index (lane half, element e) is read as key 16m + 8 (e >> 2) + 4 half
// + (e & 3) -- the A fragment (V^T row d) fetches the same keys as two 8-byte LDS reads.  P is split as 2^15 p = hi + lo
// (the low half of a probability above 8e-6 stays a NORMAL f16: subnormal MFMA inputs are flushed).
// All scalings are powers of two and undone exactly.  exp: exp2 on the hardware unit with the argument's rounding
// error fed back (<= 2 ulp; libm's expf costs a third of the f32 kernel's time).
constexpr int AKS = 72;   // K row stride (halfs): 64 + 8 -> rows 4 banks apart
constexpr int AVS = 292;  // V^T row stride (halfs): 288 + 4 -> 8-byte aligned, rows 18 banks apart
constexpr int AKEYS = 288;

__device__ __forceinline__ float exp_neg(float x)  // x <= 0 (or -inf)
{
    x = fmaxf(x, -10000.0f);
    const float t = x * 1.4426950216293335f;
    float e = __builtin_fmaf(x, 1.4426950216293335f, -t);
    e = __builtin_fmaf(x, 1.925963033500011e-08f, e);
    const float r = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(r, e * 0.6931471824645996f, r);
}

typedef const __attribute__((address_space(3))) v16x8* lds_v16x8p;
typedef const __attribute__((address_space(3))) v16x4* lds_v16x4p;
__device__ __forceinline__ unsigned lds_addr(const _Float16* p)  // byte address inside the workgroup's LDS
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) _Float16*)p;
}

constexpr int ATH = 512;  // threads: eight waves = the eight full 32-query tiles (two waves per SIMD)

// Pair conversions of the plane producers (as the GEMM epilogues, gp_split256.hip): two f32 values -> their packed f16 hi pair in one
// v_cvt_pk_f16_f32, and the packed lo pair f16(v - hi) in one v_fma_mixlo_f16 + one v_fma_mixhi_f16 that read the f16 hi straight
// from the packed register (v - hi is exact in f32, so the mixed fma's single rounding equals subtraction + conversion).
typedef _Float16 av16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned attn_hi_pair(float v0, float v1)
{
    av16x2 h;
    h[0] = (_Float16)v0;
    h[1] = (_Float16)v1;
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ unsigned attn_lo_pair(float v0, float v1, unsigned hi)
{
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"   // the pair may be the next instruction's MFMA operand / store data: hipcc does not pad hazards behind an asm producer
        : "=&v"(lo)
        : "v"(v0), "v"(v1), "v"(hi));
    return lo;
}
__device__ __forceinline__ float attn_max3(float a, float b, float c)
{   // v_maximum3_f32 through the builtin, NOT inline asm: the operands are accumulator registers fresh from the matrix core, and
    // hipcc pads MFMA -> VALU read hazards for its own instructions only (an asm v_max3_f32 here read stale registers: run-to-run
    // differences in round 6's first build)
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}

__global__ __launch_bounds__(ATH) void attention_split_kernel(const _Float16* __restrict__ QKVhi, const _Float16* __restrict__ QKVlo,
                                                               _Float16* __restrict__ Ohi, _Float16* __restrict__ Olo, int B, int H,
                                                               int C, int Mpad, float inv_s2)
{   // inv_s2 = 1 / s^2, s = the power-of-two scale the q | k | v planes carry (1 / 64 for the default x 8); the output planes carry s too
    __shared__ __attribute__((aligned(16))) _Float16 sm[2 * AKEYS * AKS + 2 * 64 * AVS];  // 157,696 bytes
    __shared__ __attribute__((aligned(16))) float xq[64], xs[T_TOK + 3], xred[2][8], xo[8][64];  // the 257th query's VALU path
    _Float16* sKh = sm;
    _Float16* sKl = sKh + AKEYS * AKS;
    _Float16* sVh = sKl + AKEYS * AKS;
    _Float16* sVl = sVh + 64 * AVS;
    const int bh = xcd_chunked_tile(blockIdx.x, B * H);
    if (bh < 0) return;
    const int h = bh % H, b = bh / H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const size_t ld = 3 * (size_t)C, tok0 = (size_t)b * T_TOK;

    // ---- staging of K (rows = keys) and V^T (rows = channels), BY KEY CHUNK (round 6).  Threads 0..255 take the hi planes, 256..511
    // the lo planes; 8 lanes copy the 128 bytes of one key (coalesced); one "iteration" = 32 keys, three iterations = one chunk of 96
    // keys = what one pass of the matrix loop below consumes.  K rows go to LDS as they are.  V is transposed on the way: lanes L and
    // L ^ 8 hold keys 2j and 2j + 1 of the same 8 channels -- they swap half of their data so that each writes FOUR dwords (key 2j |
    // key 2j + 1 of one channel row) instead of eight scattered halfwords (8-way bank conflicts, tools/probe_attn_split.py).
    // Everything is staged before the first matrix pass.  Round 6 also built the pipelined order -- only chunk 0 up front, chunk c + 1
    // loaded (global -> registers) before the pass over chunk c and written to its own LDS region after it, one barrier per chunk -- to
    // hide the 24 % of a launch that staging takes with ONE workgroup per CU: measured 5-9 % SLOWER in the same binary (97-109 us against
    // 92-100 isolated, profiles/r06_attention.txt: the in-flight loads and LDS writes land inside the matrix passes, where the vector /
    // LDS pipes are the busy ones) and removed.  The 257th query runs last.
    const int pl = tid >> 8, t8 = tid & 255;
    const _Float16* Ksrc = (pl ? QKVlo : QKVhi) + tok0 * ld + C + h * 64;
    _Float16* sKp = pl ? sKl : sKh;
    unsigned* sVp = reinterpret_cast<unsigned*>(pl ? sVl : sVh);
    const bool odd = (lane & 8) != 0;  // key parity: piece c = t8 + 256 i -> key = c >> 3
    // two register sets (round 6): chunk c + 1 is requested before chunk c is written to LDS -- three dependent round trips became one and a bit
    au32x4 kvA[3], vvA[3], kvB[3], vvB[3];
    auto load_chunk = [&](int c3, au32x4 (&kv)[3], au32x4 (&vv)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = t8 + 256 * (3 * c3 + u), key = c >> 3, ch8 = c & 7;
            kv[u] = au32x4{0u, 0u, 0u, 0u};
            vv[u] = au32x4{0u, 0u, 0u, 0u};
            if (key < T_TOK) {
                const _Float16* src = Ksrc + (size_t)key * ld + 8 * ch8;
                kv[u] = *reinterpret_cast<const au32x4*>(src);
                vv[u] = *reinterpret_cast<const au32x4*>(src + C);
            }
        }
    };
    auto store_chunk = [&](int c3, au32x4 (&kv)[3], au32x4 (&vv)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = t8 + 256 * (3 * c3 + u), key = c >> 3, ch8 = c & 7;
            // every lane takes part in the exchange (pieces past the plane carry zeros)
            const unsigned r0 = __shfl_xor(odd ? vv[u][0] : vv[u][2], 8), r1 = __shfl_xor(odd ? vv[u][1] : vv[u][3], 8);
            if (key < T_TOK + 1) {   // key 257 = zeros (key 256's partner in the transposition)
                *reinterpret_cast<au32x4*>(sKp + key * AKS + 8 * ch8) = kv[u];
                const unsigned e0 = odd ? r0 : vv[u][0], e1 = odd ? r1 : vv[u][1];  // the even key's channels 4 odd .. 4 odd + 3
                const unsigned o0 = odd ? vv[u][2] : r0, o1 = odd ? vv[u][3] : r1;  // the odd key's
                unsigned* dst = sVp + ((8 * ch8 + (odd ? 4 : 0)) * AVS + (key & ~1)) / 2;
                dst[0 * (AVS / 2)] = (e0 & 0xffffu) | (o0 << 16);
                dst[1 * (AVS / 2)] = (e0 >> 16) | (o0 & 0xffff0000u);
                dst[2 * (AVS / 2)] = (e1 & 0xffffu) | (o1 << 16);
                dst[3 * (AVS / 2)] = (e1 >> 16) | (o1 & 0xffff0000u);
            }
        }
    };
    load_chunk(0, kvA, vvA);
    load_chunk(1, kvB, vvB);
    for (int c = tid; c < 2 * (AKEYS - T_TOK - 1) * 8; c += ATH) {  // keys 258..287: zero rows (masked below)
        const int plane = c >= (AKEYS - T_TOK - 1) * 8 ? 1 : 0, r = c - plane * (AKEYS - T_TOK - 1) * 8;
        *reinterpret_cast<au32x4*>((plane ? sKl : sKh) + (T_TOK + 1 + (r >> 3)) * AKS + 8 * (r & 7)) = au32x4{0u, 0u, 0u, 0u};
    }
    {
        constexpr int ZW = (AVS - T_TOK - 1) / 2;  // V^T columns 258..291 as dwords: zero (finite x p = 0)
        for (int c = tid; c < 2 * 64 * ZW; c += ATH) {
            const int plane = c >= 64 * ZW ? 1 : 0, r = c - plane * 64 * ZW;
            reinterpret_cast<unsigned*>(plane ? sVl : sVh)[((r / ZW) * AVS + T_TOK + 1) / 2 + r % ZW] = 0u;
        }
    }
    // ---- this wave's query tile: Q fragments (B operand: column = query, k = 8 half + e inside each 16-d block)
    const int tq = wave * 32 + l31;  // < 256: the eight waves cover queries 0..255
    const size_t qo = (tok0 + tq) * ld + h * 64 + 8 * half;
    if (tid < 64) {  // query 256 (257 = 8 x 32 + 1): the exact f32 value hi + lo of its 64 channels (x 8)
        const size_t q256 = (tok0 + (T_TOK - 1)) * ld + h * 64 + tid;
        xq[tid] = (float)QKVhi[q256] + (float)QKVlo[q256];
    }
    store_chunk(0, kvA, vvA);
    load_chunk(2, kvA, vvA);
    store_chunk(1, kvB, vvB);
    store_chunk(2, kvA, vvA);
    __syncthreads();

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_part = 0.f;
    // scores in log2 units: softmax scale 64^-0.5 (Q and K planes carry x 8 each) x log2(e), so p = exp2(v - max) is ONE
    // hardware exp2 per element (v_exp_f32, 1 ulp).  The argument v = s * c carries one f32 rounding (<= 30 x 2^-24 in log2
    // units at the largest |v| that matters, i.e. ~1e-6 relative in p) -- the same error the reference's f32 softmax has in
    // x = (q.k) * scale before its exp; the compensated exp_neg of the f32 kernels costs 6 more VALU instructions per
    // element, and this loop is bound by VALU issue (SQ_ACTIVE_INST_VALU 52 % vs matrix pipe 26 %, profiles/r02_pmc_attention.txt).
    const float s_scale = (0.125f * inv_s2) * 1.4426950408889634f;

#pragma unroll 1
    for (int ch = 0; ch < 3; ++ch) {
        f32x16 s[3];
        v16x8 qh[4], ql[4];  // re-loaded per chunk (L1-hot): not live across the softmax / P.V phase
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            qh[kb] = *reinterpret_cast<const v16x8*>(QKVhi + qo + 16 * kb);
            ql[kb] = *reinterpret_cast<const v16x8*>(QKVlo + qo + 16 * kb);
        }
        // one LDS base address per plane and chunk, made opaque so that every fragment read below is base + a 16-bit
        // instruction offset (left alone, hipcc folds the plane's position into a per-read constant > 64 KB: one v_add
        // per read, 55 per chunk in a loop that is bound by vector-ALU issue)
        unsigned akh = lds_addr(sKh) + 2u * ((ch * 96 + l31) * AKS + 8 * half), akl = akh + 2u * (AKEYS * AKS);
        unsigned avh = lds_addr(sVh) + 2u * (l31 * AVS + ch * 96 + 4 * half), avl = avh + 2u * (64 * AVS);
        asm volatile("" : "+v"(akh), "+v"(akl), "+v"(avh), "+v"(avl));
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const v16x8 kh = *(lds_v16x8p)(size_t)(akh + 2u * (t * 32 * AKS + 16 * kb));
                const v16x8 kl = *(lds_v16x8p)(size_t)(akl + 2u * (t * 32 * AKS + 16 * kb));
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[kb], s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[kb], s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[kb], s[t], 0, 0, 0);
            }
        }
        // online softmax over this chunk of 96 keys.  Round 6 (vector instructions removed, not stalls): the maximum is taken over the
        // RAW scores (s_scale > 0: max(c s) = c max(s)) with one v_max3_f32 per pair; scale and reference are folded into ONE fma per
        // element, d = fma(s, c, -mb) (one rounding instead of the product's and the difference's); 9.5 -> 7 instructions per pair
        float cmax = -INFINITY;
        if (ch == 2) {  // only the last tile of the last chunk holds keys >= 257 (keys 256..287): a wave-uniform branch, not 24 selects per chunk
#pragma unroll
            for (int r = 0; r < 16; ++r) s[2][r] = (256 + frag_row(r, lane) < T_TOK) ? s[2][r] : -INFINITY;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; r += 2) cmax = attn_max3(cmax, s[t][r], s[t][r + 1]);
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32)) * s_scale;
        // running reference mb = max - 15: P carries x 2^15 for its f16 split (and so does l_part); the SAME rounded value
        // serves the exponentials of this chunk and the rescaling of the earlier ones
        const float mb = fmaxf(m_run, cmax - 15.0f);
        const float alpha = __builtin_amdgcn_exp2f(m_run - mb);     // first chunk: exp2(-inf) = 0
        const float nmb = -mb;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pp = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], s_scale, nmb));
                s[t][r] = pp;
                psum += pp;
            }
        l_part = l_part * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        m_run = mb;
        // O^T[d][query] += sum_key V^T[d][key] P^T[key][query], 16 keys per MFMA block (m)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                // P = hi + lo per pair: v_cvt_pk_f16_f32 (hi, rounded) + v_fma_mixlo / mixhi_f16 (lo = f16(p - hi)): 1.5 instructions per
                // element (round 5: mask + subtract + two conversions, 2.5)
                au32x4 phu, plu;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float px = s[t][8 * m + e], py = s[t][8 * m + e + 1];
                    const unsigned hp = attn_hi_pair(px, py);
                    phu[e >> 1] = hp;
                    plu[e >> 1] = attn_lo_pair(px, py, hp);
                }
                const v16x8 ph = __builtin_bit_cast(v16x8, phu), pl = __builtin_bit_cast(v16x8, plu);
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    const unsigned vo = 2u * (32 * dh * AVS + t * 32 + 16 * m);
                    const v16x4 a0 = *(lds_v16x4p)(size_t)(avh + vo), a1 = *(lds_v16x4p)(size_t)(avh + vo + 16u);
                    const v16x4 b0 = *(lds_v16x4p)(size_t)(avl + vo), b1 = *(lds_v16x4p)(size_t)(avl + vo + 16u);
                    v16x8 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vh[e] = a0[e]; vh[4 + e] = a1[e]; vl[e] = b0[e]; vl[4 + e] = b1[e]; }
                    if (dh == 0) {
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o0, 0, 0, 0);
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o0, 0, 0, 0);
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o0, 0, 0, 0);
                    } else {
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o1, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o1, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o1, 0, 0, 0);
                    }
                }
            }
    }
    // planes out: 8 * O = acc / l  (acc carries 2^15 p x 8 v, l carries 2^15); lane = (query, half): 4 consecutive channels per r4
    const float sc = 1.0f / (l_part + __shfl_xor(l_part, 32));
    {
        const size_t row = (tok0 + tq) * C + h * 64;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                typedef unsigned int au32x2 __attribute__((ext_vector_type(2)));
                au32x2 hv, lv;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    float v0 = (dh ? o1[4 * r4 + e] : o0[4 * r4 + e]) * sc, v1 = (dh ? o1[4 * r4 + e + 1] : o0[4 * r4 + e + 1]) * sc;
                    // keep the rounded f32 products: hipcc otherwise folds multiply + conversion into v_fma_mixlo_f16 (ONE rounding of
                    // the exact product) for hi while lo subtracts from the f32-rounded product -- at an f16 tie the two disagree by
                    // one f16 ulp of hi (seen as sparse 2^-k errors)
                    asm volatile("" : "+v"(v0), "+v"(v1));
                    const unsigned hp = attn_hi_pair(v0, v1);
                    hv[e >> 1] = hp;
                    lv[e >> 1] = attn_lo_pair(v0, v1, hp);
                }
                const size_t o = row + 32 * dh + frag_row(4 * r4, lane);
                *reinterpret_cast<au32x2*>(Ohi + o) = hv;
                *reinterpret_cast<au32x2*>(Olo + o) = lv;
            }
    }
    // ---- query 256 (257 = 8 x 32 + 1) on the vector ALU, all eight waves, AFTER the matrix loop (every key is resident by now; round 5
    // ran it first, behind the whole staging).  A ninth wave would put three waves on one SIMD and its 31 idle query columns would cost
    // as much as a full tile.  The planes are exact f32 values (hi + lo), so plain f32 FMAs over them are at least as accurate as the
    // split products.
    {
        float sv = -INFINITY;
        if (tid < T_TOK) {  // score of key `tid`
            const _Float16* kh = sKh + tid * AKS;
            const _Float16* kl = sKl + tid * AKS;
            float a = 0.f;
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const v16x8 h8 = *reinterpret_cast<const v16x8*>(kh + 8 * c8), l8 = *reinterpret_cast<const v16x8*>(kl + 8 * c8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {   // hi and lo as two mixed-precision fmas (v_fma_mix_f32 reads the f16 operand in place)
                    a = __builtin_fmaf((float)h8[e], xq[8 * c8 + e], a);
                    a = __builtin_fmaf((float)l8[e], xq[8 * c8 + e], a);
                }
            }
            sv = a * (0.125f * inv_s2);
        }
        float mx = sv;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if (lane == 0) xred[0][wave] = mx;
        __syncthreads();
        mx = xred[0][0];
#pragma unroll
        for (int w = 1; w < 8; ++w) mx = fmaxf(mx, xred[0][w]);
        const float pr = tid < T_TOK ? exp_neg(sv - mx) : 0.f;
        if (tid < T_TOK) xs[tid] = pr;
        float ps = pr;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ps += __shfl_xor(ps, off);
        if (lane == 0) xred[1][wave] = ps;
        __syncthreads();
        // out[d] = sum_key p[key] v[key][d]: thread = (d = lane, key slice = wave: keys 32 w .. 32 w + 31; wave 0 adds key 256)
        {
            const _Float16* vh = sVh + lane * AVS + 32 * wave;
            const _Float16* vl = sVl + lane * AVS + 32 * wave;
            v16x4 h4[8], l4[8];
            f32x4 p4[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                h4[u] = *reinterpret_cast<const v16x4*>(vh + 4 * u);
                l4[u] = *reinterpret_cast<const v16x4*>(vl + 4 * u);
                p4[u] = *reinterpret_cast<const f32x4*>(xs + 32 * wave + 4 * u);
            }
            float a = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a = __builtin_fmaf((float)h4[u][e], p4[u][e], a);
                    a = __builtin_fmaf((float)l4[u][e], p4[u][e], a);
                }
            if (wave == 0) {
                a = __builtin_fmaf((float)sVh[lane * AVS + T_TOK - 1], xs[T_TOK - 1], a);
                a = __builtin_fmaf((float)sVl[lane * AVS + T_TOK - 1], xs[T_TOK - 1], a);
            }
            xo[wave][lane] = a;
        }
        __syncthreads();
        if (tid < 64) {
            float l = xred[1][0], a = xo[0][tid];
#pragma unroll
            for (int w = 1; w < 8; ++w) { l += xred[1][w]; a += xo[w][tid]; }
            float v = a * (1.0f / l);  // = 8 x O (the V planes carry x 8)
            asm volatile("" : "+v"(v));
            const _Float16 hh = (_Float16)v;
            const size_t o = (tok0 + (T_TOK - 1)) * C + h * 64 + tid;
            Ohi[o] = hh;
            Olo[o] = (_Float16)(v - (float)hh);
        }
    }
}

// K and V of one (image, head) staged in LDS ONCE and shared by the nine query-tile waves of the workgroup (the
// register-resident kernel above re-reads them from L2 with one 4-byte load per MFMA, nine times per (image, head)).
//   sK [64 d][288 keys] (keys >= 257 zero), sV [288 keys][64 d]: 147 KB -> one workgroup per CU, 9 waves.
// Operand reads are conflict-free ds_read_b32 (lanes = consecutive keys / consecutive d).  Q fragments are loaded once
// per wave (32 registers).  Arithmetic per (key, query) is the same MFMA chain and the same 3-chunk online softmax
// as attention_body, so the result is bit-identical.
constexpr int AK = 288;  // padded key count
__global__ __launch_bounds__(576, 1) void attention_lds_kernel(const float* __restrict__ QK, const float* __restrict__ Vt,
                                                                float* __restrict__ O, int B, int H, int C, int Mpad,
                                                                float scale)
{
    __shared__ float sK[64 * AK];
    __shared__ float sV[AK * 64];
    const int bh = xcd_chunked_tile(blockIdx.x, B * H);
    if (bh < 0) return;
    const int h = bh % H, b = bh / H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const float* Qp = QK + (size_t)(h * 64) * Mpad + (size_t)b * T_TOK;
    const float* Kp = QK + (size_t)(C + h * 64) * Mpad + (size_t)b * T_TOK;
    const float* Vp = Vt + (size_t)b * T_TOK * C + h * 64;
    float* Op = O + (size_t)(h * 64) * Mpad + (size_t)b * T_TOK;
    for (int e = tid; e < 64 * AK; e += 576) {
        const int d = e / AK, key = e - d * AK;
        sK[e] = key < T_TOK ? Kp[(size_t)d * Mpad + key] : 0.f;
    }
    for (int e = tid; e < AK * 64; e += 576) {
        const int key = e >> 6, d = e & 63;
        sV[e] = key < T_TOK ? Vp[(size_t)key * C + d] : 0.f;
    }
    const int tq = wave * 32 + l31;
    const int tq_c = tq < T_TOK ? tq : T_TOK - 1;
    float qreg[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) qreg[kk] = Qp[(size_t)(2 * kk + half) * Mpad + tq_c];
    __syncthreads();

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_part = 0.f;
#pragma unroll 1
    for (int ch = 0; ch < 3; ++ch) {
        f32x16 s[3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const float* krow = sK + (2 * kk + half) * AK + ch * 96 + l31;
#pragma unroll
            for (int t = 0; t < 3; ++t) s[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[32 * t], qreg[kk], s[t], 0, 0, 0);
        }
        float cmax = -INFINITY;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tk = (ch * 3 + t) * 32 + frag_row(r, lane);
                const float v = (tk < T_TOK) ? s[t][r] * scale : -INFINITY;
                s[t][r] = v;
                cmax = fmaxf(cmax, v);
            }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        const float m_new = fmaxf(m_run, cmax);
        const float alpha = expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = expf(s[t][r] - m_new);
                s[t][r] = p;
                psum += p;
            }
        l_part = l_part * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tk = (ch * 3 + t) * 32 + frag_row(r, lane);  // < 288; rows >= 257 are zero and P is 0 there
                const float* vrow = sV + tk * 64 + l31;
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], s[t][r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], s[t][r], o1, 0, 0, 0);
            }
    }
    const float inv = 1.0f / (l_part + __shfl_xor(l_part, 32));
    if (tq < T_TOK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = frag_row(r, lane);
            Op[(size_t)d * Mpad + tq] = o0[r] * inv;
            Op[(size_t)(32 + d) * Mpad + tq] = o1[r] * inv;
        }
    }
}

// Measured on ViT-L, B=64 (tools/probe_attn.py, whole forward in split numerics): register-resident with one query tile
// per wave 58.2 ms; two tiles per wave (operand loads shared, 226 VGPR -> 2 waves/SIMD) 60.4 ms; K/V through LDS (one
// 147 KB workgroup per CU, staging not overlapped) 60.8 ms.  Occupancy wins: the default stays 1.
// split numerics: activation planes + plane x plane GEMMs when every GEMM of a layer fits them.  2 (default): attention in
// split numerics too (Q | K | V as planes, attention_split_kernel); 1: f32 attention on f32 Q, K, V (bit-identical to 0);
// 0: f32 activations and the lock-step kernels (kept for A/B runs and the bit-identity test)
static int g_vit_planes = 2;
#ifdef GP_PROBES
extern "C" void gp_vit_set_planes(int mode) { g_vit_planes = (mode >= 0 && mode <= 2) ? mode : 2; }
#endif
static int g_attn_nq = 1;  // 0: LDS-shared K/V kernel; 1 / 2: register-resident kernel with 1 / 2 query tiles per wave
#ifdef GP_PROBES
extern "C" void gp_attention_set_nq(int nq) { g_attn_nq = (nq >= 0 && nq <= 2) ? nq : 1; }
#endif

// ---- x_prenorm[:, 1:] -> (B, C, 256), F.normalize over C (ae_net.py:64-69); fixed fmaf order
__global__ __launch_bounds__(256) void features_kernel(const float* __restrict__ X, float* __restrict__ out,
                                                        int C, int Mpad, int normalize, int* __restrict__ status)
{
    // grid (B, nchunk): every block recomputes the full norm of its 256 patches (same sequential fma chain; the nchunk-fold
    // re-read is L2 traffic) and writes its share of the channels -- B blocks alone cannot fill 256 CUs, one block per 32
    // channels (the first version) re-read X 32 times: 0.26 ms per step.
    const int b = blockIdx.x, p = threadIdx.x;
    const int c0 = (int)((long long)C * blockIdx.y / gridDim.y), c1 = (int)((long long)C * (blockIdx.y + 1) / gridDim.y);
    const float* x = X + (size_t)b * T_TOK + 1 + p;
    float d = 1.f;
    if (normalize) {
        float ss = 0.f;
        int c = 0;
        for (; c + 16 <= C; c += 16) {  // 16 independent loads in flight, then the fmas IN ORDER (the chain is unchanged; one
            float v[16];                // load per dependent fma left the loop latency-bound: 0.3 ms for a 67 MB pass)
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = x[(size_t)(c + u) * Mpad];
#pragma unroll
            for (int u = 0; u < 16; ++u) ss = __builtin_fmaf(v[u], v[u], ss);
        }
        for (; c < C; ++c) {
            const float v = x[(size_t)c * Mpad];
            ss = __builtin_fmaf(v, v, ss);
        }
        d = fmaxf(__builtin_sqrtf(ss), 1e-12f);
        // last line of defence of every numerics mode (the plane producers have their own range guard): a NaN / inf anywhere in
        // this token's residual stream ends up in its norm -- never hand non-finite features to the matcher silently
        if (blockIdx.y == 0 && !(ss <= 3.0e38f)) gp_raise(status, GP_ST_SPLIT_RANGE);
    }
    float* o = out + (size_t)b * C * GP_P + p;
    for (int c = c0; c < c1; ++c) o[(size_t)c * GP_P] = normalize ? x[(size_t)c * Mpad] / d : x[(size_t)c * Mpad];
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace

// weight table layout (host array of device pointers), see include/gigapose_hip.h
enum { W_PATCH_WT = 0, W_PATCH_B, W_CLS_POS, W_POS_T, W_HEADER = 4 };
enum { L_LN1_G = 0, L_LN1_B, L_QK_WT, L_QK_B, L_V_WT, L_V_B, L_PROJ_WT, L_PROJ_B, L_LS1, L_LN2_G, L_LN2_B,
       L_FC1_WT, L_FC1_B, L_FC2_WT, L_FC2_B, L_LS2, L_PER_LAYER = 16 };

extern "C" {

size_t gp_vit_workspace_bytes(int B, int dim, int mlp_dim)
{
    if (B <= 0 || dim <= 0 || mlp_dim <= 0) return 0;
    const size_t Mpad = (size_t)round_up(B * T_TOK, 256);
    // X, H (C each), QK (2C), Vt (C), F (mlp_dim; also hosts im2col + patch-embed output)
    size_t f = (size_t)mlp_dim * Mpad;
    const size_t pe_need = (size_t)KPE_PAD * B * GP_P + (size_t)dim * B * GP_P;
    if (pe_need > f) f = pe_need;
    // + the stream-K scratch of the GEMMs (gp_gemm.hip), behind the activations
    return sizeof(float) * ((size_t)5 * dim * Mpad + f) + gp_gemm_streamk_bytes();
}

// per-layer table of pre-split weight planes (f16 hi / lo, PyTorch-native [out][in]) for the split-f16 mode
// (entries 10..19, optional: the same five weights as x64 single-accumulator planes for the 256 x 256 kernel of gp_split256.hip)
enum { S_QK_HI = 0, S_QK_LO, S_V_HI, S_V_LO, S_PROJ_HI, S_PROJ_LO, S_FC1_HI, S_FC1_LO, S_FC2_HI, S_FC2_LO, S_PER_LAYER = 10 };
int gp_vit_forward_split2(const float* images, int B, int dim, int depth, int heads, int mlp_dim, float ln_eps,
                          const float* const* weights, int n_weights, const void* const* split, int n_split,
                          float* workspace, size_t workspace_bytes, float* out_features, int normalize,
                          int stop_after_layers, const float* plane_scales, float* plane_amax, void* stream);

int gp_attention_split_scaled(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, int B, int heads, int dim, int Mpad,
                              float qkv_scale, void* stream);
/* q | k | v planes (and the output planes) carry the power-of-two `qkv_scale` instead of the default 8 */
int gp_attention_split_scaled(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, int B, int heads, int dim, int Mpad,
                              float qkv_scale, void* stream)
{
    GP_REQUIRE(qkv_scale > 0.f, "gp_attention_split: bad plane scale");
    GP_REQUIRE(qkv_hi && qkv_lo && out_hi && out_lo && B > 0 && heads > 0 && dim == heads * 64 && Mpad >= B * T_TOK,
               "gp_attention_split: bad arguments");
    hipLaunchKernelGGL(attention_split_kernel, dim3(xcd_chunked_grid(B * heads)), dim3(ATH), 0, (hipStream_t)stream,
                       (const _Float16*)qkv_hi, (const _Float16*)qkv_lo, (_Float16*)out_hi, (_Float16*)out_lo, B, heads, dim, Mpad, 1.0f / (qkv_scale * qkv_scale));
    GP_CHECK_LAUNCH("gp_attention_split");
    return GP_OK;
}

/* stage entry (tests / tools/probe_stage_errors.py): the LayerNorm of the plane path on its own.  X channel-major [C][Mpad] f32 ->
 * token-major activation planes hi / lo [Mpad][C] (x 8), exactly what gp_vit_forward_split launches before qkv / fc1. */
int gp_layernorm_planes(const float* X, void* out_hi, void* out_lo, const float* gamma, const float* beta, int C, int Mpad, float eps,
                        void* stream)
{
    GP_REQUIRE(X && out_hi && out_lo && gamma && beta && C > 0 && C % 128 == 0 && Mpad > 0 && Mpad % 64 == 0, "gp_layernorm_planes: bad arguments");
    launch_layernorm_planes(X, (_Float16*)out_hi, (_Float16*)out_lo, gamma, beta, C, Mpad, eps, (hipStream_t)stream);
    GP_CHECK_LAUNCH("gp_layernorm_planes");
    return GP_OK;
}

int gp_vit_forward(const float* images, int B, int dim, int depth, int heads, int mlp_dim, float ln_eps,
                   const float* const* weights, int n_weights, float* workspace, size_t workspace_bytes,
                   float* out_features, int normalize, int stop_after_layers, void* stream)
{
    return gp_vit_forward_split2(images, B, dim, depth, heads, mlp_dim, ln_eps, weights, n_weights, nullptr, 0, workspace,
                                 workspace_bytes, out_features, normalize, stop_after_layers, nullptr, nullptr, stream);
}

/* gp_vit_forward_split with PER-TENSOR plane scales (round 5).  The plane path keeps four activation tensors per layer as f16 hi / lo
 * planes of s x: LayerNorm-1 output (PS_LN1), q | k | v and the attention output (PS_QKV: the attention output is a convex combination
 * of v rows, so it shares their range), LayerNorm-2 output (PS_LN2), GELU output (PS_GELU).  s was a compile-time 8 (|x| < 8190): one
 * outlier anywhere in a real checkpoint -- DINOv2 is known for a few massive activations -- sent the WHOLE ViT to the two-accumulator
 * 128 x 128 kernels (-43 %).  Now s is a power of two per (layer, tensor):
 *   plane_scales  host array [depth][4] (PS_* order), or null = all 8 (bit-identical to gp_vit_forward_split);
 *   plane_amax    device array [depth][4] of f32 or null: a CALIBRATION pass -- every plane producer records max |x| of what it wrote
 *                 (atomic max on the f32 bits; zero it first), from which the host picks the scales (vit.py: calibrate_plane_scales).
 * A smaller scale only moves the f16 subnormal floor of the lo plane up (absolute error 2^-25 / s per element instead of 2^-28); the
 * consumers undo it exactly (out_scale = 1 / (64 s)), so with all scales 8 nothing changes.  Shapes that do not take the plane path
 * (f32 activations, 128 x 128 kernels: range 65504) ignore both arrays. */
int gp_vit_forward_split2(const float* images, int B, int dim, int depth, int heads, int mlp_dim, float ln_eps,
                          const float* const* weights, int n_weights, const void* const* split, int n_split,
                          float* workspace, size_t workspace_bytes, float* out_features, int normalize,
                          int stop_after_layers, const float* plane_scales, float* plane_amax, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    enum { PS_LN1 = 0, PS_QKV, PS_LN2, PS_GELU, PS_PER_LAYER = 4 };
    bool default_scales = true;
    if (plane_scales)
        for (int i = 0; i < depth * PS_PER_LAYER; ++i) {
            int ex = 0;
            GP_REQUIRE(plane_scales[i] > 0.f && frexpf(plane_scales[i], &ex) == 0.5f && plane_scales[i] <= 64.f && plane_scales[i] >= 1.0f / 1024.0f,
                       "gp_vit_forward_split2: plane scale %d = %g is not a power of two in [2^-10, 64]", i, (double)plane_scales[i]);
            default_scales = default_scales && plane_scales[i] == kPlaneScale;
        }
    GP_REQUIRE(B >= 0 && dim > 0 && depth > 0 && heads > 0, "gp_vit_forward: bad config");
    if (B == 0) return GP_OK;
    GP_REQUIRE(dim == heads * 64, "gp_vit_forward: head dim must be 64 (dim=%d heads=%d)", dim, heads);
    GP_REQUIRE(dim % 128 == 0 && mlp_dim % 128 == 0, "gp_vit_forward: dim/mlp_dim must be multiples of 128");
    GP_REQUIRE(n_weights == W_HEADER + depth * L_PER_LAYER, "gp_vit_forward: expected %d weight pointers, got %d",
               W_HEADER + depth * L_PER_LAYER, n_weights);
    GP_REQUIRE(images && weights && workspace && out_features, "gp_vit_forward: null pointer");
    GP_REQUIRE(workspace_bytes >= gp_vit_workspace_bytes(B, dim, mlp_dim), "gp_vit_forward: workspace too small");
    for (int i = 0; i < n_weights; ++i) GP_REQUIRE(weights[i], "gp_vit_forward: weight pointer %d is null", i);
    GP_REQUIRE(split == nullptr || n_split == depth * S_PER_LAYER || n_split == 2 * depth * S_PER_LAYER,
               "gp_vit_forward_split: expected %d (or %d) split planes, got %d", depth * S_PER_LAYER, 2 * depth * S_PER_LAYER, n_split);
    const int sp_stride = (split && n_split == 2 * depth * S_PER_LAYER) ? 2 * S_PER_LAYER : S_PER_LAYER;
    const bool have_x64 = split && sp_stride >= 2 * S_PER_LAYER;
    if (split) {
        GP_REQUIRE(dim % 32 == 0 && mlp_dim % 32 == 0, "gp_vit_forward_split: dim / mlp_dim must be multiples of 32");
        for (int i = 0; i < n_split; ++i) GP_REQUIRE(split[i], "gp_vit_forward_split: split plane %d is null", i);
    }

    const int C = dim, Mpad = round_up(B * T_TOK, 256), BP = B * GP_P;
    float* X = workspace;
    float* Hn = X + (size_t)C * Mpad;
    float* QK = Hn + (size_t)C * Mpad;
    float* Vt = QK + (size_t)2 * C * Mpad;
    float* F = Vt + (size_t)C * Mpad;
    size_t f_floats = (size_t)mlp_dim * Mpad;
    if ((size_t)KPE_PAD * B * GP_P + (size_t)dim * B * GP_P > f_floats) f_floats = (size_t)KPE_PAD * B * GP_P + (size_t)dim * B * GP_P;
    float* SK = F + f_floats;                  // stream-K scratch (flags zeroed once per forward)
    float* col = F;                            // [592][B*256]
    float* pe = F + (size_t)KPE_PAD * BP;      // [C][B*256]
    int rc;

    if ((rc = gp_gemm_streamk_reset_launch(SK, st))) return rc;
    hipLaunchKernelGGL(im2col_kernel, dim3(KPE_PAD, B), dim3(256), 0, st, images, col, B);
    GP_CHECK_LAUNCH("gp_vit_forward/im2col");
    // BP = B*256 is a multiple of 128 only for even B... (256 is) -> always a multiple of 128
    if ((rc = gp_gemm_launch(weights[W_PATCH_WT], C, col, BP, pe, BP, C, BP, KPE_PAD, 1 /*BIAS_I*/,
                             weights[W_PATCH_B], nullptr, nullptr, 0, SK, st)))
        return rc;
    hipLaunchKernelGGL(embed_kernel, dim3(C, B + 1), dim3(320), 0, st, pe, weights[W_CLS_POS], weights[W_POS_T],
                       X, B, Mpad);
    GP_CHECK_LAUNCH("gp_vit_forward/embed");

    const int nl = (stop_after_layers >= 0 && stop_after_layers < depth) ? stop_after_layers : depth;
    // Split numerics, third generation: when all five GEMMs of a layer fill the chip with 256 x 256 tiles, the
    // activations between the kernels travel as token-major f16 planes (written by LayerNorm, attention and fc1's GELU
    // epilogue; the f32 residual stream X, Q/K and V stay as they are) and every GEMM is gemm_planes256_kernel.  The
    // planes alias the f32 buffers they replace (2 planes x 2 bytes = 4 bytes per element).  Bit-identical to the
    // f32-activation kernels (same values, same split, same k order).
    const int Mtok = B * T_TOK;  // rows that carry tokens: the plane GEMMs tile floor(Mtok / 256) * 256 of them + a strip
    const bool planes = have_x64 && g_vit_planes && gp_gemm_planes256_usable(2 * C, Mpad, Mtok, C) &&
                        gp_gemm_planes256_usable(C, Mpad, Mtok, C) && gp_gemm_planes256_usable(mlp_dim, Mpad, Mtok, C) &&
                        gp_gemm_planes256_usable(C, Mpad, Mtok, mlp_dim) && (g_vit_planes == 2 || gp_gemm_split256_usable(Mpad, C, C));
    GP_REQUIRE(g_vit_planes == 2 || !planes || (default_scales && !plane_amax), "gp_vit_forward_split2: plane scales need the full plane path (gp_vit_set_planes(2))");
    if (planes) {
        _Float16* Hhi = reinterpret_cast<_Float16*>(Hn);
        _Float16* Hlo = Hhi + (size_t)C * Mpad;
        _Float16* Fhi = reinterpret_cast<_Float16*>(F);
        _Float16* Flo = Fhi + (size_t)mlp_dim * Mpad;
        const float ps_default[PS_PER_LAYER] = {kPlaneScale, kPlaneScale, kPlaneScale, kPlaneScale};
        for (int l = 0; l < nl; ++l) {
            const float* const* w = weights + W_HEADER + l * L_PER_LAYER;
            const void* const* sq = split + l * sp_stride + S_PER_LAYER;  // x64 weight planes [out][in]
            // this layer's plane scales (powers of two) and, in a calibration pass, where each tensor's max |x| goes
            const float* ps = plane_scales ? plane_scales + l * PS_PER_LAYER : ps_default;
            float* am = plane_amax ? plane_amax + l * PS_PER_LAYER : nullptr;
            const float os = 1.0f / (kPlaneScale * 64.0f);  // g_vit_planes == 1 (probe path): activations x 8, weights x 64
            const float os_ln1 = 1.0f / (ps[PS_LN1] * 64.0f), os_qkv = 1.0f / (ps[PS_QKV] * 64.0f), os_ln2 = 1.0f / (ps[PS_LN2] * 64.0f),
                        os_gelu = 1.0f / (ps[PS_GELU] * 64.0f);   // a consumer undoes its B operand's scale and the weights' x 64, exactly
            const GpPlaneOut po_qkv{ps[PS_QKV], am ? am + PS_QKV : nullptr}, po_gelu{ps[PS_GELU], am ? am + PS_GELU : nullptr};
            launch_layernorm_planes(X, Hhi, Hlo, w[L_LN1_G], w[L_LN1_B], C, Mpad, ln_eps, st, ps[PS_LN1], am ? am + PS_LN1 : nullptr);
            GP_CHECK_LAUNCH("gp_vit_forward/layernorm_planes");
            if (g_vit_planes == 2) {
                // Q | K | V as planes [Mpad][3C] (aliasing the f32 QK + Vt buffers): W_qk / W_v (A) x tokens (B), plane epilogue
                _Float16* Ahi = reinterpret_cast<_Float16*>(QK);
                _Float16* Alo = Ahi + (size_t)3 * C * Mpad;
                // one launch when the host packed q|k and v contiguously (planes and biases are views of one tensor, vit.py): 12 x 64 =
                // 768 tiles = 3 per slot instead of 512 + 256 in two launches (one strip, one launch boundary less per layer)
                const bool fused_qkv = (const char*)sq[S_V_HI] == (const char*)sq[S_QK_HI] + (size_t)2 * C * C * 2 &&
                                       (const char*)sq[S_V_LO] == (const char*)sq[S_QK_LO] + (size_t)2 * C * C * 2 && w[L_V_B] == w[L_QK_B] + 2 * C &&
                                       gp_gemm_planes256_usable(3 * C, Mpad, Mtok, C);
                if (fused_qkv) {
                    if ((rc = gp_gemm_planes256_launch(sq[S_QK_HI], sq[S_QK_LO], Hhi, Hlo, nullptr, 0, Ahi, Alo, 3 * C, 3 * C, Mpad, Mtok, C,
                                                       7 /*BIAS_I -> planes*/, w[L_QK_B], nullptr, nullptr, 0, os_ln1, SK, st, nullptr, &po_qkv)))
                        return rc;
                } else {
                    if ((rc = gp_gemm_planes256_launch(sq[S_QK_HI], sq[S_QK_LO], Hhi, Hlo, nullptr, 0, Ahi, Alo, 3 * C, 2 * C, Mpad, Mtok, C,
                                                       7 /*BIAS_I -> planes*/, w[L_QK_B], nullptr, nullptr, 0, os_ln1, SK, st, nullptr, &po_qkv)))
                        return rc;
                    if ((rc = gp_gemm_planes256_launch(sq[S_V_HI], sq[S_V_LO], Hhi, Hlo, nullptr, 0, Ahi + 2 * C, Alo + 2 * C, 3 * C, C, Mpad, Mtok, C,
                                                       7 /*BIAS_I -> planes*/, w[L_V_B], nullptr, nullptr, 0, os_ln1, SK, st, nullptr, &po_qkv)))
                        return rc;
                }
                {
                    GpProfScope prof(GP_PROF_ATTN, 4.0 * B * heads * 257.0 * 257.0 * 64.0, st);
                    hipLaunchKernelGGL(attention_split_kernel, dim3(xcd_chunked_grid(B * heads)), dim3(ATH), 0, st, Ahi, Alo, Hhi, Hlo, B,
                                       heads, C, Mpad, 1.0f / (ps[PS_QKV] * ps[PS_QKV]));
                }
                GP_CHECK_LAUNCH("gp_vit_forward/attention_split");
            } else {
            // Q,K channel-major [2C][Mpad] = W_qk (A) x tokens (B)
            if ((rc = gp_gemm_planes256_launch(sq[S_QK_HI], sq[S_QK_LO], Hhi, Hlo, QK, Mpad, nullptr, nullptr, 0, 2 * C, Mpad, Mtok, C,
                                               1 /*BIAS_I*/, w[L_QK_B], nullptr, nullptr, 0, os, SK, st)))
                return rc;
            // V token-major [Mpad][C] = tokens (A) x W_v (B), bias along j
            if ((rc = gp_gemm_planes256_launch(Hhi, Hlo, sq[S_V_HI], sq[S_V_LO], Vt, C, nullptr, nullptr, 0, Mpad, C, C, C,
                                               4 /*BIAS_J*/, w[L_V_B], nullptr, nullptr, 0, os, SK, st)))
                return rc;
            {
                GpProfScope prof(GP_PROF_ATTN, 4.0 * B * heads * 257.0 * 257.0 * 64.0, st);
                hipLaunchKernelGGL(attention_planes_kernel, dim3(xcd_chunked_grid(B * heads * NKT)), dim3(64), 0, st, QK, Vt, Hhi, Hlo,
                                   B, heads, C, Mpad, 0.125f);
            }
            GP_CHECK_LAUNCH("gp_vit_forward/attention_planes");
            }
            // x = x + ls1 * proj(attn)   (the attention output carries the q | k | v scale; g_vit_planes == 1: the default x 8)
            if ((rc = gp_gemm_planes256_launch(sq[S_PROJ_HI], sq[S_PROJ_LO], Hhi, Hlo, X, Mpad, nullptr, nullptr, 0, C, Mpad, Mtok, C,
                                               3 /*BIAS_I_SCALE_RES*/, w[L_PROJ_B], w[L_LS1], X, Mpad, g_vit_planes == 2 ? os_qkv : os, SK, st)))
                return rc;
            launch_layernorm_planes(X, Hhi, Hlo, w[L_LN2_G], w[L_LN2_B], C, Mpad, ln_eps, st, ps[PS_LN2], am ? am + PS_LN2 : nullptr);
            GP_CHECK_LAUNCH("gp_vit_forward/layernorm_planes");
            // gelu(fc1(.)) straight to planes [Mpad][mlp_dim]
            if ((rc = gp_gemm_planes256_launch(sq[S_FC1_HI], sq[S_FC1_LO], Hhi, Hlo, nullptr, 0, Fhi, Flo, mlp_dim, mlp_dim, Mpad, Mtok, C,
                                               6 /*GELU -> planes*/, w[L_FC1_B], nullptr, nullptr, 0, os_ln2, SK, st, nullptr, &po_gelu)))
                return rc;
            // x = x + ls2 * fc2(.)
            if ((rc = gp_gemm_planes256_launch(sq[S_FC2_HI], sq[S_FC2_LO], Fhi, Flo, X, Mpad, nullptr, nullptr, 0, C, Mpad, Mtok, mlp_dim,
                                               3 /*BIAS_I_SCALE_RES*/, w[L_FC2_B], w[L_LS2], X, Mpad, os_gelu, SK, st)))
                return rc;
        }
    }
    for (int l = planes ? nl : 0; l < nl; ++l) {
        const float* const* w = weights + W_HEADER + l * L_PER_LAYER;
        launch_layernorm(X, Hn, w[L_LN1_G], w[L_LN1_B], C, Mpad, ln_eps, st);
        GP_CHECK_LAUNCH("gp_vit_forward/layernorm");
        const void* const* sp = split ? split + l * sp_stride : nullptr;
        const void* const* sq = (sp && have_x64) ? sp + S_PER_LAYER : nullptr;  // x64 planes (256-tile kernel)
        // split GEMM dispatch: 256 x 256 stream-K kernel when the shape fills the chip with 256-tiles, else 128 x 128
        auto sgemm = [&](const float* act, int ld_act, int which, float* D, int ldd, int I, int J, int K, int act_is_b, int epi,
                         const float* bias, const float* scale, const float* res, int ldr) -> int {
            if (sq && gp_gemm_split256_usable(I, J, K))
                return gp_gemm_split256_launch(act, ld_act, sq[which], sq[which + 1], D, ldd, I, J, K, act_is_b, epi, bias, scale, res,
                                               ldr, SK, st);
            return gp_gemm_split_launch(act, ld_act, sp[which], sp[which + 1], D, ldd, I, J, K, act_is_b, epi, bias, scale, res, ldr, st);
        };
        // Q,K channel-major [2C][Mpad]
        if (sp) rc = sgemm(Hn, Mpad, S_QK_HI, QK, Mpad, 2 * C, Mpad, C, 1, 1, w[L_QK_B], nullptr, nullptr, 0);
        else rc = gp_gemm_launch(w[L_QK_WT], 2 * C, Hn, Mpad, QK, Mpad, 2 * C, Mpad, C, 1, w[L_QK_B], nullptr, nullptr, 0, SK, st);
        if (rc) return rc;
        // V token-major [Mpad][C]: swap operand roles (A = activations, B = weights), bias along j
        if (sp) rc = sgemm(Hn, Mpad, S_V_HI, Vt, C, Mpad, C, C, 0, 4 /*BIAS_J*/, w[L_V_B], nullptr, nullptr, 0);
        else rc = gp_gemm_launch(Hn, Mpad, w[L_V_WT], C, Vt, C, Mpad, C, C, 4 /*BIAS_J*/, w[L_V_B], nullptr, nullptr, 0, SK, st);
        if (rc) return rc;
        {
            GpProfScope prof(GP_PROF_ATTN, 4.0 * B * heads * 257.0 * 257.0 * 64.0, st);
            if (g_attn_nq == 0)
                hipLaunchKernelGGL(attention_lds_kernel, dim3(xcd_chunked_grid(B * heads)), dim3(576), 0, st, QK, Vt, Hn, B, heads, C,
                                   Mpad, 0.125f);
            else if (g_attn_nq == 1)
                hipLaunchKernelGGL(attention_kernel<1>, dim3(xcd_chunked_grid(B * heads * NKT)), dim3(64), 0, st, QK, Vt, Hn, B,
                                   heads, C, Mpad, 0.125f);
            else
                hipLaunchKernelGGL(attention_kernel<2>, dim3(xcd_chunked_grid(B * heads * 5)), dim3(64), 0, st, QK, Vt, Hn, B,
                                   heads, C, Mpad, 0.125f);
        }
        GP_CHECK_LAUNCH("gp_vit_forward/attention");
        // x = x + ls1 * proj(attn)
        if (sp) rc = sgemm(Hn, Mpad, S_PROJ_HI, X, Mpad, C, Mpad, C, 1, 3, w[L_PROJ_B], w[L_LS1], X, Mpad);
        else rc = gp_gemm_launch(w[L_PROJ_WT], C, Hn, Mpad, X, Mpad, C, Mpad, C, 3, w[L_PROJ_B], w[L_LS1], X, Mpad, SK, st);
        if (rc) return rc;
        launch_layernorm(X, Hn, w[L_LN2_G], w[L_LN2_B], C, Mpad, ln_eps, st);
        GP_CHECK_LAUNCH("gp_vit_forward/layernorm");
        if (sp) rc = sgemm(Hn, Mpad, S_FC1_HI, F, Mpad, mlp_dim, Mpad, C, 1, 2 /*GELU*/, w[L_FC1_B], nullptr, nullptr, 0);
        else rc = gp_gemm_launch(w[L_FC1_WT], mlp_dim, Hn, Mpad, F, Mpad, mlp_dim, Mpad, C, 2 /*GELU*/, w[L_FC1_B], nullptr,
                                 nullptr, 0, SK, st);
        if (rc) return rc;
        // x = x + ls2 * fc2(gelu(fc1))
        if (sp) rc = sgemm(F, Mpad, S_FC2_HI, X, Mpad, C, Mpad, mlp_dim, 1, 3, w[L_FC2_B], w[L_LS2], X, Mpad);
        else rc = gp_gemm_launch(w[L_FC2_WT], C, F, Mpad, X, Mpad, C, Mpad, mlp_dim, 3, w[L_FC2_B], w[L_LS2], X, Mpad, SK, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(features_kernel, dim3(B, B >= 64 ? 4 : (B >= 16 ? 16 : 32)), dim3(256), 0, st, X, out_features, C, Mpad, normalize,
                       gp_status_buffer());
    GP_CHECK_LAUNCH("gp_vit_forward/features");
    return GP_OK;
}

}  // extern "C"
