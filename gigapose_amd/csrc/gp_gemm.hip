// k-major f32 MFMA GEMM with fused epilogues (gfx950).
//
//   D[i][j] = epi( sum_k A[k][i] * B[k][j] )        A: [K][lda], B: [K][ldb], D: [I][ldd]
//
// Every dense layer of the hot path is this contraction with activations kept TRANSPOSED
// ("channel-major": X^T [features][tokens]), so operand loads and result stores are coalesced
// without any transpose pass:
//   ViT linear layers (DINOv2 block: qkv / proj / fc1 / fc2; HF modeling_dinov2.py:199-297,
//   reference call site ae_net.py:44-47):   A = W^T [in][out] (pre-transposed once at load),
//   B = X^T [in][tokens]  ->  D = Y^T [out][tokens]
//   IST regressor MLPs (reference ist_net.py:140-155): same, tokens = correspondences.
// Accumulation order is the sequential fmaf chain over k of KMajor (gp_common.h).
#include "gp_common.h"

namespace {

enum {
    EPI_NONE = 0,
    EPI_BIAS_I = 1,            // + bias[i]
    EPI_BIAS_I_GELU = 2,       // gelu_erf(acc + bias[i])                      (DINOv2 fc1)
    EPI_BIAS_I_SCALE_RES = 3,  // res[i][j] + scale[i] * (acc + bias[i])       (LayerScale + residual)
    EPI_BIAS_J = 4,            // + bias[j]                                    (token-major V)
    EPI_BIAS_I_RELU = 5,       // relu(acc + bias[i])                          (IST MLP)
};

using GM = KMajor<2, 4, 2, 1, 16>;  // main tile: 128 x 128, 8 waves (2 x 4), 32 accumulators/lane
using GT = KMajor<1, 2, 1, 1, 16>;  // tail tile:  32 x 64, 2 waves (1/8 of a main tile's work)

__device__ __forceinline__ float gelu_erf(float x)
{
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// One kernel body for both tile shapes.  Columns [j_begin, j_begin + tiles_j * CFG::BN) are covered;
// the per-output accumulation chain does not depend on the tiling, so main and tail tiles (and any
// other split) give bit-identical results.
template <int EPI, class CFG, int MINW>
__global__ __launch_bounds__(CFG::NT, MINW) void gemm_kmajor_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* D,
    int ldd, int tiles_i, int tiles_j, int j_begin, int K, const float* __restrict__ bias,
    const float* __restrict__ scale, const float* res /* may alias D (in-place residual) */, int ldr)
{
    __shared__ float smem[CFG::LDS_FLOATS];
    const int q = xcd_chunked_tile(blockIdx.x, tiles_i * tiles_j);
    if (q < 0) return;
    // i fastest: the blocks of one XCD chunk share the B (activation) panel in that XCD's L2
    const int ti = q % tiles_i, tj = q / tiles_i;
    const int i0 = ti * CFG::BM, j0 = j_begin + tj * CFG::BN;
    constexpr int MI = CFG::BM / 32 / CFG::WM_, NI = CFG::BN / 32 / CFG::WN_;
    f32x16 acc[MI][NI];
    CFG::run(A + i0, lda, B + j0, ldb, K, smem, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / CFG::WN_, wn = wave % CFG::WN_;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int j = j0 + wn * NI * 32 + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wm * MI * 32 + mi * 32 + frag_row(r, lane);
                float v = acc[mi][ni][r];
                if (EPI == EPI_BIAS_I || EPI == EPI_BIAS_I_GELU || EPI == EPI_BIAS_I_SCALE_RES ||
                    EPI == EPI_BIAS_I_RELU)
                    v = v + bias[i];
                if (EPI == EPI_BIAS_J) v = v + bias[j];
                if (EPI == EPI_BIAS_I_GELU) v = gelu_erf(v);
                if (EPI == EPI_BIAS_I_RELU) v = fmaxf(v, 0.f);
                // 32-bit element offsets (I * ld < 2^31 is checked by the launcher): half the address VGPRs
                if (EPI == EPI_BIAS_I_SCALE_RES) v = res[(unsigned)i * (unsigned)ldr + (unsigned)j] + scale[i] * v;
                D[(unsigned)i * (unsigned)ldd + (unsigned)j] = v;
            }
            // keep the next tile's residual loads below this tile's stores (register pressure)
            if (EPI == EPI_BIAS_I_SCALE_RES) __builtin_amdgcn_sched_barrier(0);
        }
}

// Residency of the main kernel: 4 workgroups per CU (<= 128 VGPR, 32 KiB LDS) x 256 CUs.  When the
// tile count is just over a multiple of that (ViT-L at B=64: 8 x 129 = 1032 tiles on 1024 slots), the
// last few main tiles would run alone after everything else has finished; instead the trailing j-tiles
// are "peeled" into 32x64 tail tiles (8x more, 1/8 of the work each) that spread over the whole chip.
constexpr int kResidentSlots = 1024;
static bool g_tail_peel = true;

template <int EPI>
int launch(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I, int J, int K,
           const float* bias, const float* scale, const float* res, int ldr, hipStream_t st)
{
    const int ti = I / GM::BM;
    int tj = J / GM::BN;
    const int rem = (ti * tj) % kResidentSlots;
    int peel = 0;  // number of trailing j-tiles handed to the tail kernel
    if (g_tail_peel && ti * tj > kResidentSlots && rem > 0 && rem <= kResidentSlots / 4 && rem % ti == 0 && rem / ti < tj)
        peel = rem / ti;
    tj -= peel;
    hipLaunchKernelGGL((gemm_kmajor_kernel<EPI, GM, 2>), dim3(xcd_chunked_grid(ti * tj)), dim3(GM::NT), 0, st, A, lda,
                       B, ldb, D, ldd, ti, tj, 0, K, bias, scale, res, ldr);
    if (peel) {
        const int tti = I / GT::BM, ttj = peel * GM::BN / GT::BN;
        hipLaunchKernelGGL((gemm_kmajor_kernel<EPI, GT, 2>), dim3(xcd_chunked_grid(tti * ttj)), dim3(GT::NT), 0, st, A,
                           lda, B, ldb, D, ldd, tti, ttj, tj * GM::BN, K, bias, scale, res, ldr);
    }
    return 0;
}

}  // namespace

// internal C++ entry used by gp_vit.hip / gp_ist.hip
int gp_gemm_launch(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I, int J,
                   int K, int epilogue, const float* bias, const float* scale, const float* res, int ldr,
                   hipStream_t st)
{
    GP_REQUIRE(I > 0 && J > 0 && K > 0, "gp_gemm_kmajor: empty problem (I=%d J=%d K=%d)", I, J, K);
    GP_REQUIRE(I % 128 == 0 && J % 128 == 0 && K % 16 == 0,
               "gp_gemm_kmajor: I=%d, J=%d must be multiples of 128 and K=%d of 16 (pad the operands)", I, J, K);
    GP_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= I && ldb >= J && ldd >= J,
               "gp_gemm_kmajor: bad leading dimensions lda=%d ldb=%d ldd=%d", lda, ldb, ldd);
    GP_REQUIRE((long long)I * ldd < (1ll << 31) && (long long)I * (ldr > 0 ? ldr : 1) < (1ll << 31),
               "gp_gemm_kmajor: output larger than 2^31 elements");
    GP_REQUIRE(A && B && D, "gp_gemm_kmajor: null pointer");
    GP_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0), "gp_gemm_kmajor: operands must be 16-byte aligned");
    GpProfScope prof(GP_PROF_GEMM, 2.0 * I * J * K, st);
    switch (epilogue) {
        case EPI_NONE: launch<EPI_NONE>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, st); break;
        case EPI_BIAS_I:
            GP_REQUIRE(bias, "gp_gemm_kmajor: bias required");
            launch<EPI_BIAS_I>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, st);
            break;
        case EPI_BIAS_I_GELU:
            GP_REQUIRE(bias, "gp_gemm_kmajor: bias required");
            launch<EPI_BIAS_I_GELU>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, st);
            break;
        case EPI_BIAS_I_SCALE_RES:
            GP_REQUIRE(bias && scale && res && ldr >= J, "gp_gemm_kmajor: bias/scale/residual required");
            launch<EPI_BIAS_I_SCALE_RES>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, st);
            break;
        case EPI_BIAS_J:
            GP_REQUIRE(bias, "gp_gemm_kmajor: bias required");
            launch<EPI_BIAS_J>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, st);
            break;
        case EPI_BIAS_I_RELU:
            GP_REQUIRE(bias, "gp_gemm_kmajor: bias required");
            launch<EPI_BIAS_I_RELU>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, st);
            break;
        default: GP_REQUIRE(false, "gp_gemm_kmajor: unknown epilogue %d", epilogue);
    }
    GP_CHECK_LAUNCH("gp_gemm_kmajor");
    return GP_OK;
}

// ---- tile-shape experiments (tools/probe_gemm.py); not part of the product path -------------------
namespace {
template <class CFG, int MINW>
__global__ __launch_bounds__(CFG::NT, MINW) void gemm_probe_kernel(const float* __restrict__ A, int lda,
                                                                   const float* __restrict__ B, int ldb, float* D,
                                                                   int ldd, int tiles_i, int tiles_j, int K)
{
    __shared__ float smem[CFG::LDS_FLOATS];
    const int q = xcd_chunked_tile(blockIdx.x, tiles_i * tiles_j);
    if (q < 0) return;
    const int ti = q % tiles_i, tj = q / tiles_i;
    const int i0 = ti * CFG::BM, j0 = tj * CFG::BN;
    constexpr int MI = CFG::BM / 32 / CFG::WM_, NI = CFG::BN / 32 / CFG::WN_;
    f32x16 acc[MI][NI];
    CFG::run(A + i0, lda, B + j0, ldb, K, smem, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / CFG::WN_, wn = wave % CFG::WN_;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int j = j0 + wn * NI * 32 + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wm * MI * 32 + mi * 32 + frag_row(r, lane);
                D[(size_t)i * ldd + j] = acc[mi][ni][r];
            }
        }
}
template <class CFG, int MINW>
int probe_launch(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I, int J, int K, hipStream_t st)
{
    if (I % CFG::BM || J % CFG::BN || K % 32) return GP_EINVAL;
    const int ti = I / CFG::BM, tj = J / CFG::BN;
    hipLaunchKernelGGL((gemm_probe_kernel<CFG, MINW>), dim3(xcd_chunked_grid(ti * tj)), dim3(CFG::NT), 0, st, A, lda, B,
                       ldb, D, ldd, ti, tj, K);
    return hipGetLastError() == hipSuccess ? GP_OK : GP_ELAUNCH;
}
}  // namespace

extern "C" void gp_gemm_set_tail_peel(int on) { g_tail_peel = on != 0; }

extern "C" int gp_gemm_probe(int variant, const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I,
                             int J, int K, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 0: return probe_launch<KMajor<2, 2, 2, 2, 16>, 2>(A, lda, B, ldb, D, ldd, I, J, K, st);  // 128x128 (product)
        case 1: return probe_launch<KMajor<2, 2, 2, 2, 32>, 2>(A, lda, B, ldb, D, ldd, I, J, K, st);  // KS=32
        case 2: return probe_launch<KMajor<2, 4, 2, 2, 16>, 2>(A, lda, B, ldb, D, ldd, I, J, K, st);  // 128x256, 8 waves
        case 3: return probe_launch<KMajor<4, 2, 2, 2, 16>, 2>(A, lda, B, ldb, D, ldd, I, J, K, st);  // 256x128, 8 waves
        case 4: return probe_launch<KMajor<4, 2, 2, 4, 16>, 2>(A, lda, B, ldb, D, ldd, I, J, K, st);  // 256x256 (matcher)
        case 5: return probe_launch<KMajor<2, 2, 4, 2, 16>, 1>(A, lda, B, ldb, D, ldd, I, J, K, st);  // 256x128, 4 waves
        case 6: return probe_launch<KMajor<2, 2, 2, 4, 16>, 1>(A, lda, B, ldb, D, ldd, I, J, K, st);  // 128x256, 4 waves
        case 7: return probe_launch<KMajor<2, 2, 2, 2, 8>, 2>(A, lda, B, ldb, D, ldd, I, J, K, st);   // KS=8
        case 8: return probe_launch<KMajor<2, 2, 2, 2, 16, true>, 2>(A, lda, B, ldb, D, ldd, I, J, K, st);  // + LDS operand prefetch
        case 9: return probe_launch<KMajor<2, 4, 2, 1, 16>, 2>(A, lda, B, ldb, D, ldd, I, J, K, st);   // 128x128, 8 waves
        case 10: return probe_launch<KMajor<2, 2, 2, 2, 8, true>, 2>(A, lda, B, ldb, D, ldd, I, J, K, st);  // KS=8 + prefetch
        case 11: return probe_launch<KMajor<4, 2, 2, 4, 16, true>, 2>(A, lda, B, ldb, D, ldd, I, J, K, st); // 256x256 + prefetch
        default: return GP_EINVAL;
    }
}

extern "C" int gp_gemm_kmajor(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I,
                              int J, int K, int epilogue, const float* bias, const float* scale,
                              const float* residual, int ldr, void* stream)
{
    return gp_gemm_launch(A, lda, B, ldb, D, ldd, I, J, K, epilogue, bias, scale, residual, ldr,
                          (hipStream_t)stream);
}
