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

using GM = KMajor<2, 4, 2, 1, 16>;  // tile: 128 x 128, 8 waves (2 x 4), 32 accumulators/lane

__device__ __forceinline__ float gelu_erf(float x)
{
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// stream-K scratch (device): [flags: kMaxSlots ints][error word ... padded to 8 KiB][kMaxSlots partial tiles]
constexpr int kMaxSlots = 1024;  // resident workgroups: 4 per CU (<= 64 VGPR, 32 KiB LDS) x 256 CUs
constexpr size_t kSkHeaderBytes = 8192;
constexpr size_t kSkPartialFloats = (size_t)GM::BM * GM::BN;
constexpr int kSpinLimit = 400000;  // x ~0.4 us: a lost hand-off ends in an error word, never in a hang

struct GemmArgs {
    const float* A; int lda;
    const float* B; int ldb;
    float* D; int ldd;
    int K;
    const float* bias; const float* scale;
    const float* res /* may alias D (in-place residual) */; int ldr;
    int tiles_i, tiles_j;
    int group;       // tile order: bands of `group` i-tiles, i fastest inside a band
    int* flags;      // stream-K only
    float* partial;  // stream-K only
    int epoch;       // stream-K only: value a published flag carries in this launch (never 0)
    int* status;     // guard rails (gp_common.h)
};

// tile list position q -> tile origin.  The ~128 tiles an XCD runs concurrently form a compact
// group x (128/group) rectangle whose A and B panels are shared through that XCD's L2.
__device__ __forceinline__ void tile_origin(const GemmArgs& a, int q, int& i0, int& j0)
{
    const int per_band = a.group * a.tiles_j;
    const int band = q / per_band, r = q - band * per_band;
    const int first_i = band * a.group;
    const int gsz = min(a.group, a.tiles_i - first_i);
    i0 = (first_i + r % gsz) * GM::BM;
    j0 = (r / gsz) * GM::BN;
}

template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, int i0, int j0, const f32x16 (&acc)[2][1])
{
    int tid_ = threadIdx.x;
    // opaque: inside the persistent kernel's segment loop the compiler would otherwise hoist all 32 row / address
    // computations out of the loop and spill them
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6;
    const int wm = wave / GM::WN_, wn = wave % GM::WN_;
    const float* __restrict__ bias = a.bias;
    const float* __restrict__ scale = a.scale;
    const float* res = a.res;
    float* D = a.D;
    const int j = j0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + wm * 64 + mi * 32 + frag_row(r, lane);
            float v = acc[mi][0][r];
            if (EPI == EPI_BIAS_I || EPI == EPI_BIAS_I_GELU || EPI == EPI_BIAS_I_SCALE_RES || EPI == EPI_BIAS_I_RELU)
                v = v + bias[i];
            if (EPI == EPI_BIAS_J) v = v + bias[j];
            if (EPI == EPI_BIAS_I_GELU) v = gelu_erf(v);
            if (EPI == EPI_BIAS_I_RELU) v = fmaxf(v, 0.f);
            // 32-bit element offsets (I * ld < 2^31 is checked by the launcher): half the address VGPRs
            if (EPI == EPI_BIAS_I_SCALE_RES) v = res[(unsigned)i * (unsigned)a.ldr + (unsigned)j] + scale[i] * v;
            D[(unsigned)i * (unsigned)a.ldd + (unsigned)j] = v;
        }
        // keep the next tile's residual loads below this tile's stores (register pressure)
        if (EPI == EPI_BIAS_I_SCALE_RES) __builtin_amdgcn_sched_barrier(0);
    }
}

// One workgroup per 128x128 tile (tile count <= resident slots, or no scratch given).
template <int EPI>
__global__ __launch_bounds__(GM::NT, 8) void gemm_kmajor_kernel(const GemmArgs a)
{
    __shared__ float smem[GM::LDS_FLOATS];
    const int q = xcd_chunked_tile(blockIdx.x, a.tiles_i * a.tiles_j);
    if (q < 0) return;
    int i0, j0;
    tile_origin(a, q, i0, j0);
    f32x16 acc[2][1];
    GM::run(a.A + i0, a.lda, a.B + j0, a.ldb, a.K, smem, acc);
    gemm_epilogue<EPI>(a, i0, j0, acc);
}

// Chain-preserving stream-K.  When the tile count is not a multiple of the resident slots (ViT-L at B=64:
// 8 x 129 = 1032 tiles on 1024 slots), one-tile-per-workgroup scheduling leaves a few CUs with one tile more
// than all the others (+25 % wall time).  Here the grid is exactly the resident slots; the (tile, k-slab)
// iteration space of each XCD's tile chunk is cut into equal contiguous ranges, one per slot.  A range ends
// inside a tile ("head": slabs [0, s)) and starts inside another ("rest": slabs [s, nslab)).  Because the
// contraction is a sequential fmaf chain, a split tile is NOT reduced from two partial sums: the slot owning
// the head computes it FIRST, publishes the accumulator fragment, and the next slot LOADS it as the initial
// accumulator of the rest -- the per-output chain k = 0..K-1 is unchanged, results are bit-identical to the
// one-workgroup-per-tile kernel.
// Progress: slot n of an XCD only ever waits for slot n-1 of the same XCD (block id - 8), which publishes
// before it waits for anything itself; a lower block id is never dispatched later, so there is no circular
// wait whatever the residency.  Hand-off: MI355X guide, Guideline 16 (agent-scope release / acquire); the
// spin is bounded and reports through the error word.
template <int EPI>
__global__ __launch_bounds__(GM::NT, 8) void gemm_streamk_kernel(const GemmArgs a)
{
    __shared__ float smem[GM::LDS_FLOATS];
    const int tid = threadIdx.x;
    const int p = blockIdx.x, x = p & 7, n = p >> 3, slots_x = gridDim.x >> 3;
    const int T = a.tiles_i * a.tiles_j;
    const int per = (T + 7) >> 3;
    const int t_lo = x * per;
    const int n_t = min(T - t_lo, per);  // tiles of this XCD's chunk (launcher guarantees n_t >= slots_x)
    const int nslab = a.K / GM::KS_;
    const long long U = (long long)n_t * nslab;
    const long long u0 = U * n / slots_x, u1 = U * (n + 1) / slots_x;
    const int ta = (int)(u0 / nslab), sa = (int)(u0 % nslab);  // range starts at slab sa of tile ta
    const int tb = (int)(u1 / nslab), sb = (int)(u1 % nslab);  // and ends before slab sb of tile tb
    // segments in execution order: [head of tb] [whole tiles ta(+1) .. tb-1] [rest of ta]; ONE instance of the
    // main loop serves all three kinds (register allocation = the plain kernel's)
    const int n_head = sb > 0 ? 1 : 0, n_rest = sa > 0 ? 1 : 0;
    const int first_whole = ta + n_rest;
    const int n_seg = n_head + (tb - first_whole) + n_rest;
    for (int seg = 0; seg < n_seg; ++seg) {
        const bool is_head = seg < n_head;
        const bool is_rest = n_rest && seg == n_seg - 1;
        const int t = is_head ? tb : (is_rest ? ta : first_whole + seg - n_head);
        const int k0 = is_rest ? sa * GM::KS_ : 0;
        const int k1 = is_head ? sb * GM::KS_ : a.K;
        int i0, j0;
        tile_origin(a, t_lo + t, i0, j0);
        f32x16 acc[2][1];
        if (is_rest) {  // continue the chain slot n-1 started: wait for its fragment, load it as the accumulator
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(a.flags + (p - 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
                    __builtin_amdgcn_s_sleep(16);
                    if (++spins > kSpinLimit) {
                        __hip_atomic_store(a.flags + kMaxSlots, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gp_raise(a.status, GP_ST_HANDOFF_CHAIN);
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            // fragment layout: 32 consecutive floats per thread (one base pointer, immediate offsets)
            const f32x4* w = reinterpret_cast<const f32x4*>(a.partial + (size_t)(p - 8) * kSkPartialFloats) + tid * 8;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 v = w[mi * 4 + r4];
                    acc[mi][0][r4 * 4 + 0] = v[0]; acc[mi][0][r4 * 4 + 1] = v[1];
                    acc[mi][0][r4 * 4 + 2] = v[2]; acc[mi][0][r4 * 4 + 3] = v[3];
                }
        } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][0][r] = 0.f;
        }
        GM::run_acc(a.A + (size_t)k0 * a.lda + i0, a.lda, a.B + (size_t)k0 * a.ldb + j0, a.ldb, k1 - k0, smem, acc);
        if (is_head) {  // publish the fragment for slot n+1
            // plain stores + agent-scope release by one lane (guide, Guideline 16).  Write-through (sc1) stores
            // without the release fence were measured too: 2 % slower on the ViT-L layer.
            f32x4* w = reinterpret_cast<f32x4*>(a.partial + (size_t)p * kSkPartialFloats) + tid * 8;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 v;
                    v[0] = acc[mi][0][r4 * 4 + 0]; v[1] = acc[mi][0][r4 * 4 + 1];
                    v[2] = acc[mi][0][r4 * 4 + 2]; v[3] = acc[mi][0][r4 * 4 + 3];
                    w[mi * 4 + r4] = v;
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.flags + p, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            gemm_epilogue<EPI>(a, i0, j0, acc);
        }
    }
}

static int g_streamk = 1;  // 0 off, 1 by the rule in launch(), 2 whenever possible (tests)
static int g_group = 8;
static unsigned g_epoch = 0;

template <int EPI>
int launch(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I, int J, int K,
           const float* bias, const float* scale, const float* res, int ldr, float* sk_ws, hipStream_t st)
{
    GemmArgs a{A, lda, B, ldb, D, ldd, K, bias, scale, res, ldr, I / GM::BM, J / GM::BN, g_group, nullptr, nullptr, 0, gp_status_buffer()};
    const int T = a.tiles_i * a.tiles_j;
    // Measured on ViT-L at B=64 (tools/probe_vit.py): with fewer than 4 tiles per slot the balanced split wins
    // (fc2 1.32 -> 1.13 ms, V 0.34 -> 0.30, QK 0.60 -> 0.58, proj 0.35 -> 0.34); from 4 tiles per slot on, the
    // few left-over tiles of one-workgroup-per-tile scheduling run alone on their CUs at several times the shared
    // speed and cost less than the hand-offs (fc1, 4128 tiles: 1.15 vs 1.19 ms).
    if (g_streamk && sk_ws && T > kMaxSlots && (T < 4 * kMaxSlots || g_streamk == 2) && T % kMaxSlots != 0) {
        const int per = (T + 7) / 8, last = T - 7 * per;  // tiles of the smallest XCD chunk
        const int slots_x = last < kMaxSlots / 8 ? last : kMaxSlots / 8;
        if (slots_x > 0) {
            a.flags = reinterpret_cast<int*>(sk_ws);
            a.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(sk_ws) + kSkHeaderBytes);
            if (++g_epoch == 0) ++g_epoch;
            a.epoch = (int)g_epoch;
            hipLaunchKernelGGL((gemm_streamk_kernel<EPI>), dim3(8 * slots_x), dim3(GM::NT), 0, st, a);
            return 0;
        }
    }
    hipLaunchKernelGGL((gemm_kmajor_kernel<EPI>), dim3(xcd_chunked_grid(T)), dim3(GM::NT), 0, st, a);
    return 0;
}

}  // namespace

// internal C++ entry used by gp_vit.hip / gp_ist.hip.  sk_ws: optional stream-K scratch
// (gp_gemm_streamk_workspace_bytes(), flags zeroed by gp_gemm_streamk_reset()); one scratch per stream.
int gp_gemm_launch(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I, int J,
                   int K, int epilogue, const float* bias, const float* scale, const float* res, int ldr,
                   float* sk_ws, hipStream_t st)
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
    GP_REQUIRE(((uintptr_t)sk_ws % 16 == 0), "gp_gemm_kmajor: stream-K scratch must be 16-byte aligned");
    GpProfScope prof(GP_PROF_GEMM, 2.0 * I * J * K, st);
    switch (epilogue) {
        case EPI_NONE: launch<EPI_NONE>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, sk_ws, st); break;
        case EPI_BIAS_I:
            GP_REQUIRE(bias, "gp_gemm_kmajor: bias required");
            launch<EPI_BIAS_I>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, sk_ws, st);
            break;
        case EPI_BIAS_I_GELU:
            GP_REQUIRE(bias, "gp_gemm_kmajor: bias required");
            launch<EPI_BIAS_I_GELU>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, sk_ws, st);
            break;
        case EPI_BIAS_I_SCALE_RES:
            GP_REQUIRE(bias && scale && res && ldr >= J, "gp_gemm_kmajor: bias/scale/residual required");
            launch<EPI_BIAS_I_SCALE_RES>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, sk_ws, st);
            break;
        case EPI_BIAS_J:
            GP_REQUIRE(bias, "gp_gemm_kmajor: bias required");
            launch<EPI_BIAS_J>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, sk_ws, st);
            break;
        case EPI_BIAS_I_RELU:
            GP_REQUIRE(bias, "gp_gemm_kmajor: bias required");
            launch<EPI_BIAS_I_RELU>(A, lda, B, ldb, D, ldd, I, J, K, bias, scale, res, ldr, sk_ws, st);
            break;
        default: GP_REQUIRE(false, "gp_gemm_kmajor: unknown epilogue %d", epilogue);
    }
    GP_CHECK_LAUNCH("gp_gemm_kmajor");
    return GP_OK;
}

size_t gp_gemm_streamk_bytes() { return kSkHeaderBytes + sizeof(float) * kSkPartialFloats * kMaxSlots; }

int gp_gemm_streamk_reset_launch(float* sk_ws, hipStream_t st)
{
    return hipMemsetAsync(sk_ws, 0, kSkHeaderBytes, st) == hipSuccess ? GP_OK : GP_ELAUNCH;
}

// ---- tile-shape experiments (tools/probe_gemm.py); not part of the product path -------------------
namespace {
__device__ unsigned long long g_clk[4];  // shader-clock / 100 MHz wall-clock stamps of block 0 (tools/probe_clock.py)
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
    unsigned long long c0 = 0, w0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    CFG::run(A + i0, lda, B + j0, ldb, K, smem, acc);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_clk[0] = c0; g_clk[1] = __builtin_readcyclecounter(); g_clk[2] = w0; g_clk[3] = wall_clock64();
    }
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
static int g_probe_occ = 0;  // blocks/CU the runtime reports for the last probe kernel
template <class CFG, int MINW>
int probe_launch(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I, int J, int K, hipStream_t st)
{
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&g_probe_occ, gemm_probe_kernel<CFG, MINW>, CFG::NT, 0);
    if (I % CFG::BM || J % CFG::BN || K % 32) return GP_EINVAL;
    const int ti = I / CFG::BM, tj = J / CFG::BN;
    hipLaunchKernelGGL((gemm_probe_kernel<CFG, MINW>), dim3(xcd_chunked_grid(ti * tj)), dim3(CFG::NT), 0, st, A, lda, B,
                       ldb, D, ldd, ti, tj, K);
    return hipGetLastError() == hipSuccess ? GP_OK : GP_ELAUNCH;
}
}  // namespace

#ifdef GP_PROBES
extern "C" void gp_gemm_set_streamk(int mode) { g_streamk = mode; }
extern "C" int gp_gemm_probe_occupancy(void) { return g_probe_occ; }
extern "C" int gp_gemm_product_occupancy(int streamk)
{
    int n = 0;
    if (streamk) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_streamk_kernel<EPI_BIAS_I_SCALE_RES>, GM::NT, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_kmajor_kernel<EPI_BIAS_I_SCALE_RES>, GM::NT, 0);
    return n;
}
extern "C" int gp_gemm_probe_clock(unsigned long long* host4)
{
    return hipMemcpyFromSymbol(host4, HIP_SYMBOL(g_clk), 4 * sizeof(unsigned long long)) == hipSuccess ? GP_OK : GP_ELAUNCH;
}
extern "C" void gp_gemm_set_group(int g) { g_group = g > 0 ? g : 1; }

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
#endif

extern "C" int gp_gemm_kmajor(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I,
                              int J, int K, int epilogue, const float* bias, const float* scale,
                              const float* residual, int ldr, void* stream)
{
    return gp_gemm_launch(A, lda, B, ldb, D, ldd, I, J, K, epilogue, bias, scale, residual, ldr, nullptr,
                          (hipStream_t)stream);
}

extern "C" size_t gp_gemm_streamk_workspace_bytes(void) { return gp_gemm_streamk_bytes(); }

extern "C" int gp_gemm_streamk_reset(float* scratch, void* stream)
{
    GP_REQUIRE(scratch, "gp_gemm_streamk_reset: null scratch");
    return gp_gemm_streamk_reset_launch(scratch, (hipStream_t)stream);
}

#ifdef GP_PROBES
extern "C" int gp_gemm_streamk_error(const float* scratch, void* stream)
{
    int e = -1;
    if (hipMemcpyAsync(&e, reinterpret_cast<const int*>(scratch) + 1024, sizeof(int), hipMemcpyDeviceToHost,
                       (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return -1;
    return e;
}
#endif

extern "C" int gp_gemm_kmajor_sk(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I,
                                 int J, int K, int epilogue, const float* bias, const float* scale,
                                 const float* residual, int ldr, float* scratch, size_t scratch_bytes, void* stream)
{
    GP_REQUIRE(scratch && scratch_bytes >= gp_gemm_streamk_bytes(), "gp_gemm_kmajor_sk: scratch too small");
    return gp_gemm_launch(A, lda, B, ldb, D, ldd, I, J, K, epilogue, bias, scale, residual, ldr, scratch,
                          (hipStream_t)stream);
}
