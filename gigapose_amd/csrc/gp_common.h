// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of gigapose_amd.
//
// Everything here is written for wave64 + the f32-input matrix core
// (v_mfma_f32_32x32x2_f32): exact f32, bit-for-bit a k-ordered fmaf chain when k-pairs are
// issued in ascending order -- which is what makes index outputs reproducible against the
// CPU oracle (oracle/gp_oracle.c).  Build with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GP_OK 0
#define GP_EINVAL -1
#define GP_ELAUNCH -2

#define GP_P 256  // patches per crop (16x16)
#define GP_G 16

// Device-side status word (guard rails): kernels OR a bit into *gp_status_buffer() when something that must not pass
// silently happens; the host reads it at its next synchronisation point (gigapose_amd/_lib.py: check_status) and raises.
// The buffer is the caller's (gp_set_status_buffer, one int32 on the device); NULL = reporting off.
enum { GP_ST_HANDOFF_SPLIT = 1,   // a stream-K accumulator hand-over of the split GEMMs timed out: that tile is garbage
       GP_ST_HANDOFF_CHAIN = 2,   // same, f32 (chain) GEMM
       GP_ST_SPLIT_RANGE = 4,     // an activation left the range of the single-accumulator split planes (|x| >= 8190) or is not finite
       GP_ST_LABEL_RANGE = 8,     // a detection label / template id outside the onboarded bank
       GP_ST_SPLIT_RANGE_CONV = 16 };  // GP_ST_SPLIT_RANGE raised by the IST planes (conv_planes_kernel and its producers): the host's
                                  // automatic fallback widens only the network that needs it (gigaPose.py: _widen_split_range)
int* gp_status_buffer();
__device__ __forceinline__ void gp_raise(int* status, int bit)
{
    if (status) atomicOr(status, bit);
}
constexpr float kSplitPlaneLimit = 65504.0f;  // largest finite f16: |8 x| beyond it would store inf in the hi plane

// Plane producers (LayerNorm, the plane epilogues 6 / 7 of gp_split256.hip, attention): the power of two a tensor is multiplied
// by before it is split into its f16 hi / lo planes (default 8: |x| < 8190), and -- calibration passes only -- where to record
// max |x| of what was written (f32 bits, atomic max; gp_vit_forward_split2).  Per tensor, chosen by the host (vit.py).
struct GpPlaneOut {
    float scale;   // power of two
    float* amax;   // device float or null
};
__device__ __forceinline__ void gp_record_amax(float* amax, float mx_scaled, float inv_scale)
{
    // mx_scaled >= 0 or NaN.  Wave maximum first (one atomic per wave); non-negative floats order like their bit patterns, a NaN's
    // pattern lies above every finite one
    float m = mx_scaled;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(m, off);
        m = (o > m || o != o) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m * inv_scale));
}

// set by every entry point on failure; read through gp_last_error()
void gp_set_error(const char* fmt, ...);

#define GP_REQUIRE(cond, ...)             \
    do {                                  \
        if (!(cond)) {                    \
            gp_set_error(__VA_ARGS__);    \
            return GP_EINVAL;             \
        }                                 \
    } while (0)

#define GP_CHECK_LAUNCH(name)                                                   \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) {                                                 \
            gp_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return GP_ELAUNCH;                                                  \
        }                                                                       \
    } while (0)

// bench-only timing scope: records HIP events around a launch when gp_prof_begin() is active
enum { GP_PROF_GEMM = 0, GP_PROF_MATCH = 1, GP_PROF_ATTN = 2, GP_PROF_LN = 3, GP_PROF_CONV = 4, GP_PROF_OTHER = 5, GP_PROF_GEMM_SPLIT = 6,
       GP_PROF_MATCH_SPLIT = 7, GP_PROF_KINDS = 8 };
struct GpProfScope {
    GpProfScope(int kind, double work, hipStream_t st);
    ~GpProfScope();
    int idx_;
    hipStream_t st_;
};

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) of the split kernels' epilogues (fc1), scaled by 2 hs (hs = 0.5: GELU itself; hs = half the plane
// scale folds an activation plane's power-of-two scaling into it, bit for bit).  Form (round 6): erfc(|x| / sqrt 2) = 2^-Q(z), z = min(|x|, 9),
// Q a degree-8 polynomial without constant term fitted (weighted minimax, tools/fit_gelu.py) to -log2 erfc so that the error of GELU is uniform
// in |x| + 1; then 2 hs GELU = (a + |a|) - |a| E with a = hs x, E = 2^-Q: ONE fma covers both signs -- x >= 0: 2a - aE (a single rounding of the
// exact value), x < 0: a E, no cancellation.  12 vector instructions + v_exp_f32 per element where the round-2 form (erfc = t P9(t) exp(-z^2),
// t = 1 / (1 + 0.3275911 z): 20 + v_rcp_f32 + v_exp_f32) spent the fc1 tile epilogue's 16.6 us on the vector pipes; max |error| / (|x| + 1) over
// [-12, 12] in f32: 4.9e-8 (the round-2 form: 7.0e-8, f32 0.5 x (1 + erf) with an exact erf: 8.2e-8), rms on N(0, 2) inputs 3.6e-8 (4.5e-8).
// z is clamped, so |x| > 9 returns x or -|x| 2^-64; NaN and inf come out as NaN (the plane epilogues' range guard counts both).
__device__ __forceinline__ float gp_gelu_scaled(float x, float hs)
{
    const float z = __builtin_fminf(__builtin_fabsf(x), 9.0f);
    float p = 2.3157908378749728e-06f;
    p = __builtin_fmaf(p, z, -3.3171934239110258e-05f);
    p = __builtin_fmaf(p, z, 0.00015670828855028176f);
    p = __builtin_fmaf(p, z, 0.00020827152112870692f);
    p = __builtin_fmaf(p, z, -0.0071571907367062922f);
    p = __builtin_fmaf(p, z, 0.05256192828655367f);
    p = __builtin_fmaf(p, z, 0.45918599081344252f);
    p = __builtin_fmaf(p, z, 1.151107890162735f);
    const float e = __builtin_amdgcn_exp2f(-(p * z));   // erfc(|x| / sqrt 2), Q <= 64: a normal number
    const float a = hs * x;
    return __builtin_fmaf(-__builtin_fabsf(a), e, a + __builtin_fabsf(a));
}

// The same for two elements, bit for bit, written on 2-vectors so that the polynomial issues as v_pk_fma_f32 (two lanes' worth per issue slot:
// 7.5 + v_exp_f32 instructions per element instead of 12.5; the hot caller is the fc1 tile epilogue of gp_split256.hip).
typedef float gp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gp_f32x2 gp_gelu_scaled2(gp_f32x2 x, float hs)
{
    gp_f32x2 z;
    z[0] = __builtin_fminf(__builtin_fabsf(x[0]), 9.0f);
    z[1] = __builtin_fminf(__builtin_fabsf(x[1]), 9.0f);
    const auto k = [](float c) { return gp_f32x2{c, c}; };
    gp_f32x2 p = k(2.3157908378749728e-06f);
    p = __builtin_elementwise_fma(p, z, k(-3.3171934239110258e-05f));
    p = __builtin_elementwise_fma(p, z, k(0.00015670828855028176f));
    p = __builtin_elementwise_fma(p, z, k(0.00020827152112870692f));
    p = __builtin_elementwise_fma(p, z, k(-0.0071571907367062922f));
    p = __builtin_elementwise_fma(p, z, k(0.05256192828655367f));
    p = __builtin_elementwise_fma(p, z, k(0.45918599081344252f));
    p = __builtin_elementwise_fma(p, z, k(1.151107890162735f));
    const gp_f32x2 q = p * z;
    const gp_f32x2 a = x * hs;
    gp_f32x2 r;
    r[0] = __builtin_fmaf(-__builtin_fabsf(a[0]), __builtin_amdgcn_exp2f(-q[0]), a[0] + __builtin_fabsf(a[0]));
    r[1] = __builtin_fmaf(-__builtin_fabsf(a[1]), __builtin_amdgcn_exp2f(-q[1]), a[1] + __builtin_fabsf(a[1]));
    return r;
}

// MFMA 32x32x2 f32 C/D fragment map (guide section 3): lane l, register r in [0,16):
//   col = l & 31,  row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// XCD-aware, bijective-by-construction tile order: hardware places block id on XCD id % 8,
// so give each XCD one contiguous chunk of the tile list (neighbours share operand panels in
// that XCD's private L2).  Returns -1 for padding blocks (grid = 8 * ceil(total / 8)).
__device__ __forceinline__ int xcd_chunked_tile(int block_id, int total)
{
    const int per = (total + 7) >> 3;
    const int q = (block_id & 7) * per + (block_id >> 3);
    return (q < total && (block_id >> 3) < per) ? q : -1;
}
static inline int xcd_chunked_grid(int total) { return 8 * ((total + 7) / 8); }

// ---------------------------------------------------------------------------------------------
// k-major f32 MFMA main loop.
//   D[i][j] = sum_k A[k][i] * B[k][j],  A: [K][lda] (i contiguous), B: [K][ldb] (j contiguous)
// Block tile BM x BN = (WM*MI*32) x (WN*NI*32); WM*WN waves; wave (wm, wn) owns MI x NI MFMA
// tiles.  K is consumed in slabs of KS rows staged through LDS (double buffered, one barrier per
// slab); global loads of slab s+1 are in flight while slab s feeds the matrix core.
// Per lane the k index of an MFMA operand is 2*kk + (lane >> 5): ascending k-pairs => the
// accumulator is exactly  acc = fmaf(A[k][i], B[k][j], acc)  for k = 0..K-1.
// Requirements: K % KS == 0, lda/ldb % 4 == 0, A/B 16-byte aligned, tile fully in bounds.
// LDS needed: 2 * KS * (BM + BN) floats.
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int MI, int NI, int KS, bool PIPE = false>
struct KMajor {
    static constexpr int WM_ = WM, WN_ = WN, KS_ = KS;
    static constexpr int BM = WM * MI * 32;
    static constexpr int BN = WN * NI * 32;
    static constexpr int NT = WM * WN * 64;
    static constexpr int AF4 = KS * BM / 4, BF4 = KS * BN / 4;  // float4s per slab
    static constexpr int A4 = (AF4 + NT - 1) / NT;  // float4 loads per thread per slab (A)
    static constexpr int B4 = (BF4 + NT - 1) / NT;
    static constexpr bool AG = (AF4 % NT) != 0, BG = (BF4 % NT) != 0;  // guarded (not every thread loads)
    static constexpr int LDS_FLOATS = 2 * KS * (BM + BN);
    static_assert(KS % 2 == 0, "KS must be even");


    __device__ static __forceinline__ void gload(const float* __restrict__ A, int lda,
                                                  const float* __restrict__ B, int ldb, int k0, int tid,
                                                  f32x4 (&ra)[A4], f32x4 (&rb)[B4])
    {
#pragma unroll
        for (int u = 0; u < A4; ++u) {
            const int f = tid + u * NT;
            const int r = f / (BM / 4), c4 = f % (BM / 4);
            if (!AG || f < AF4) ra[u] = *reinterpret_cast<const f32x4*>(A + (size_t)(k0 + r) * lda + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < B4; ++u) {
            const int f = tid + u * NT;
            const int r = f / (BN / 4), c4 = f % (BN / 4);
            if (!BG || f < BF4) rb[u] = *reinterpret_cast<const f32x4*>(B + (size_t)(k0 + r) * ldb + c4 * 4);
        }
    }
    __device__ static __forceinline__ void swrite(float* __restrict__ sA, float* __restrict__ sB, int buf,
                                                   int tid, const f32x4 (&ra)[A4], const f32x4 (&rb)[B4])
    {
#pragma unroll
        for (int u = 0; u < A4; ++u)
            if (!AG || tid + u * NT < AF4) *reinterpret_cast<f32x4*>(sA + buf * KS * BM + (tid + u * NT) * 4) = ra[u];
#pragma unroll
        for (int u = 0; u < B4; ++u)
            if (!BG || tid + u * NT < BF4) *reinterpret_cast<f32x4*>(sB + buf * KS * BN + (tid + u * NT) * 4) = rb[u];
    }

    __device__ static __forceinline__ void run(const float* __restrict__ A, int lda,
                                                const float* __restrict__ B, int ldb, int K,
                                                float* __restrict__ smem, f32x16 (&acc)[MI][NI])
    {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        run_acc(A, lda, B, ldb, K, smem, acc);
    }

    // continues the chain: acc holds the partial sums of the k rows BEFORE A/B (or zeros)
    __device__ static __forceinline__ void run_acc(const float* __restrict__ A, int lda,
                                                    const float* __restrict__ B, int ldb, int K,
                                                    float* __restrict__ smem, f32x16 (&acc)[MI][NI])
    {
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        float* sA = smem;                // [2][KS][BM]
        float* sB = smem + 2 * KS * BM;  // [2][KS][BN]

        f32x4 ra[A4], rb[B4];
        const int nslab = K / KS;
        gload(A, lda, B, ldb, 0, tid, ra, rb);
        swrite(sA, sB, 0, tid, ra, rb);
        __syncthreads();
        const int arow = wm * MI * 32 + (lane & 31);
        const int bcol = wn * NI * 32 + (lane & 31);
        const int khalf = lane >> 5;
        for (int s = 0; s < nslab; ++s) {
            const int buf = s & 1;
            if (s + 1 < nslab) gload(A, lda, B, ldb, (s + 1) * KS, tid, ra, rb);
            const float* cA = sA + buf * KS * BM;
            const float* cB = sB + buf * KS * BN;
            if (!PIPE) {
#pragma unroll
                for (int kk = 0; kk < KS / 2; ++kk) {
                    const int k = 2 * kk + khalf;
                    float a[MI], b[NI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) a[mi] = cA[k * BM + arow + mi * 32];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) b[ni] = cB[k * BN + bcol + ni * 32];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                }
            } else {  // operands of k-pair kk+1 are read from LDS before the MFMAs of k-pair kk issue
                float a[2][MI], b[2][NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[0][mi] = cA[khalf * BM + arow + mi * 32];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[0][ni] = cB[khalf * BN + bcol + ni * 32];
#pragma unroll
                for (int kk = 0; kk < KS / 2; ++kk) {
                    if (kk + 1 < KS / 2) {
                        const int k = 2 * (kk + 1) + khalf;
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) a[(kk + 1) & 1][mi] = cA[k * BM + arow + mi * 32];
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) b[(kk + 1) & 1][ni] = cB[k * BN + bcol + ni * 32];
                    }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][mi], b[kk & 1][ni], acc[mi][ni], 0, 0, 0);
                }
            }
            if (s + 1 < nslab) swrite(sA, sB, buf ^ 1, tid, ra, rb);
            __syncthreads();
        }
    }
};
