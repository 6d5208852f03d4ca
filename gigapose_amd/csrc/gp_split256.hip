// Split-f16 GEMM, second generation: 256 x 256 tiles, ONE accumulator, balanced by stream-K.
//
// What the per-phase probes of gp_split.hip's 128 x 128 kernel showed: a k-step pulls 64 KB per CU
// through the 64 B/clk vector-memory path, reads 0.67 LDS fragments per MFMA, and the chip throttles to ~2.05 GHz under the
// combined matrix + LDS + memory load.  This kernel halves the global and LDS bytes per MFMA (the matcher's structure):
//   * tile 256 (i) x 256 (j) x 32 (k), 8 waves as 4 x 2, wave tile 64 x 128 = acc[2][4]: 12 fragment reads per 24 MFMAs;
//   * ONE accumulator: operands are pre-scaled by powers of two (activations x 8, weights x 64) so that the low halves
//     lo = f16(x - hi) stay in f16's normal range WITHOUT the 2^11 scaling, and all three products
//     hi*hi + hi*lo + lo*hi go into the same f32 accumulator; the tile is rescaled by the exact 2^-9 in the epilogue.
//     Representation error: 2^-22 relative for |8x| >= 2^-3, 2^-25 absolute below (f16 subnormal spacing) -- measured
//     error vs f64 equals the two-accumulator kernel's for activations of typical magnitude >= 0.05 (tests).
//     Range: |activation| < 8190.
//   * 128 KB LDS (2 buffers x 4 planes x 16 KB, 64-byte rows, 16-byte chunks XOR-swizzled by row bits 2-3), one
//     workgroup per CU; grid = 256 slots; each XCD's (tile, k-step) space is cut into equal ranges (stream-K).  A tile
//     split between two workgroups is handed over as an accumulator fragment exactly like gp_gemm.hip's chain-preserving
//     stream-K (the first part is computed first and published; deterministic, no atomics).
//   * step order: write slab s+1 (loaded during the previous step) -> MFMAs of slab s with the 12 loads of slab s+2 one
//     behind each of the first 12 MFMAs -> barrier.
// Activations arrive as f32 k-major [K][n] (converted while staging, balanced over all 512 threads); weights as f16
// planes [n][K] made by gp_split256_weights().
#include "gp_common.h"

typedef _Float16 g16x8 __attribute__((ext_vector_type(8)));
typedef float g32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 g16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr float kActScale = 8.0f, kWScale = 64.0f, kOutScale = 1.0f / (8.0f * 64.0f);
constexpr int TB = 256, TBK = 32, TNT = 512;
constexpr int TROW = 32, TPLANE = TB * TROW;  // halfs
constexpr int TBUF = 4 * TPLANE;              // A hi, A lo, B hi, B lo
constexpr int kSlots = 256;
constexpr int kErrWord = 1025;  // scratch header word: beyond gp_gemm.hip's 1024 slot flags + its own error word (shared scratch)
constexpr size_t kHeaderBytes = 8192;
constexpr size_t kFragFloats = (size_t)TB * TB;
constexpr int kSpin = 400000;

enum { XEPI_NONE = 0, XEPI_BIAS_I = 1, XEPI_BIAS_I_GELU = 2, XEPI_BIAS_I_SCALE_RES = 3, XEPI_BIAS_J = 4, XEPI_BIAS_I_RELU = 5 };

struct Args256 {
    const float* act; int ld_act;
    const _Float16* whi; const _Float16* wlo;
    float* D; int ldd; int K;
    const float* bias; const float* scale; const float* res; int ldr;
    int tiles_i, tiles_j, group;
    int* flags; float* partial; int epoch;
    int* status;
};

__device__ __forceinline__ int toff(int row, int kc) { return row * TROW + ((kc ^ ((row >> 2) & 3)) << 3); }
// GELU: gp_common.h (gp_gelu_scaled), shared with gp_split.hip
__device__ __forceinline__ float gelu_x(float x) { return gp_gelu_scaled(x, 0.5f); }

// W [n][K] f32 (PyTorch [out][in]) -> planes hi = f16(64 w), lo = f16(64 w - hi), same shape
__global__ __launch_bounds__(256) void split256_weights_kernel(const float* __restrict__ W, size_t count, _Float16* __restrict__ hi,
                                                                _Float16* __restrict__ lo)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float v = W[i] * kWScale;
    const _Float16 h = (_Float16)v;
    hi[i] = h;
    lo[i] = (_Float16)(v - (float)h);
}

__device__ unsigned long long g_t256[8];  // probe: per-phase cycle totals of wave 0 of one mid-grid block + step count

template <int EPI, bool ACT_IS_B, bool TIMING = false>
__global__ __launch_bounds__(TNT, 2) void gemm_split256_kernel(const Args256 a)
{
    unsigned long long tc[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
#define X_T(i) do { if (TIMING) { t1 = __builtin_readcyclecounter(); tc[i] += t1 - t0; t0 = t1; } } while (0)
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * TBUF];  // 128 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    constexpr int P_AHI = 0, P_ALO = TPLANE, P_BHI = 2 * TPLANE, P_BLO = 3 * TPLANE;
    const int act_hi = ACT_IS_B ? P_BHI : P_AHI, act_lo = ACT_IS_B ? P_BLO : P_ALO;
    const int w_hi = ACT_IS_B ? P_AHI : P_BHI, w_lo = ACT_IS_B ? P_ALO : P_BLO;

    // ---- this slot's range of (tile, k-step) units inside its XCD's tile chunk
    const int p = blockIdx.x, x = p & 7, n = p >> 3, slots_x = gridDim.x >> 3;
    const int T = a.tiles_i * a.tiles_j;
    const int t_lo = (int)((long long)T * x / 8), n_t = (int)((long long)T * (x + 1) / 8) - t_lo;
    const int nstep = a.K / TBK;
    const long long U = (long long)n_t * nstep;
    const long long u0 = U * n / slots_x, u1 = U * (n + 1) / slots_x;
    const int ta = (int)(u0 / nstep), sa = (int)(u0 % nstep);
    const int tb = (int)(u1 / nstep), sb = (int)(u1 % nstep);
    const int n_head = sb > 0 ? 1 : 0, n_rest = sa > 0 ? 1 : 0;
    const int first_whole = ta + n_rest;
    const int n_seg = n_head + (tb - first_whole) + n_rest;

    // staging roles (all 512 threads): 2 n x 8 k activation micro-block + four 16-byte weight chunks
    const int ng = tid & 127, kg = tid >> 7;
    g32x2 ract[8];
    g16x8 rw[4];
    int bad = 0;  // range guard: some |8 x| this thread converted is beyond f16 or not finite
    // fragment addressing
    const int ar_ = 64 * wr + (lane & 31), br_ = 128 * wc + (lane & 31), kh_ = lane >> 5;
    const int arow = ar_ * TROW, brow = br_ * TROW;
    const int ak0 = ((kh_ ^ ((ar_ >> 2) & 3)) << 3), ak1 = (((kh_ + 2) ^ ((ar_ >> 2) & 3)) << 3);
    const int bk0 = ((kh_ ^ ((br_ >> 2) & 3)) << 3), bk1 = (((kh_ + 2) ^ ((br_ >> 2) & 3)) << 3);

    for (int seg = 0; seg < n_seg; ++seg) {
        const bool is_head = seg < n_head;
        const bool is_rest = n_rest && seg == n_seg - 1;
        const int t = is_head ? tb : (is_rest ? ta : first_whole + seg - n_head);
        const int s0 = is_rest ? sa : 0, s1 = is_head ? sb : nstep;
        // tile order inside the chunk: bands of `group` i-tiles, i fastest
        const int q = t_lo + t;
        const int per_band = a.group * a.tiles_j;
        const int band = q / per_band, rr = q - band * per_band;
        const int first_i = band * a.group;
        const int gsz = min(a.group, a.tiles_i - first_i);
        const int i0 = (first_i + rr % gsz) * TB, j0 = (rr / gsz) * TB;
        const int n_act0 = ACT_IS_B ? j0 : i0, n_w0 = ACT_IS_B ? i0 : j0;

        f32x16 acc[2][4];
        if (is_rest) {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(a.flags + (p - 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
                    __builtin_amdgcn_s_sleep(16);
                    if (++spins > kSpin) {
                        __hip_atomic_store(a.flags + kErrWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gp_raise(a.status, GP_ST_HANDOFF_SPLIT);
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            const f32x4* w = reinterpret_cast<const f32x4*>(a.partial + (size_t)(p - 8) * kFragFloats) + tid * 32;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 v = w[(mi * 4 + ni) * 4 + r4];
                        acc[mi][ni][r4 * 4 + 0] = v[0]; acc[mi][ni][r4 * 4 + 1] = v[1];
                        acc[mi][ni][r4 * 4 + 2] = v[2]; acc[mi][ni][r4 * 4 + 3] = v[3];
                    }
        } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        }

        // ---- k loop over steps [s0, s1)
        const int ns = s1 - s0;
        const float* pa = a.act + (size_t)(s0 * TBK + kg * 8) * a.ld_act + n_act0 + ng * 2;
        const _Float16* pw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = tid + TNT * u;  // plane c >> 10, row (c & 1023) >> 2, k-chunk c & 3
            pw[u] = ((c >> 10) ? a.wlo : a.whi) + (size_t)(n_w0 + ((c & 1023) >> 2)) * a.K + s0 * TBK + (c & 3) * 8;
        }
        const size_t act_step = (size_t)TBK * a.ld_act;
        auto gload_all = [&]() {
#pragma unroll
            for (int r = 0; r < 8; ++r) ract[r] = *reinterpret_cast<const g32x2*>(pa + (size_t)r * a.ld_act);
#pragma unroll
            for (int u = 0; u < 4; ++u) rw[u] = *reinterpret_cast<const g16x8*>(pw[u]);
        };
        auto advance = [&]() {
            pa += act_step;
#pragma unroll
            for (int u = 0; u < 4; ++u) pw[u] += TBK;
        };
        auto stage = [&](int buf) {
            _Float16* L = lds + buf * TBUF;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = tid + TNT * u;
                *reinterpret_cast<g16x8*>(L + ((c >> 10) ? w_lo : w_hi) + toff((c & 1023) >> 2, c & 3)) = rw[u];
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                g16x8 h, l;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float v = ract[r][c] * kActScale;
                    const _Float16 hh = (_Float16)v;
                    h[r] = hh;
                    l[r] = (_Float16)(v - (float)hh);
                    bad |= !(fabsf(v) <= kSplitPlaneLimit);  // !(<=): NaN counts
                }
                const int off = toff(ng * 2 + c, kg);
                *reinterpret_cast<g16x8*>(L + act_hi + off) = h;
                *reinterpret_cast<g16x8*>(L + act_lo + off) = l;
            }
        };
        gload_all();   // slab s0
        stage(0);
        if (ns > 1) advance();
        gload_all();   // slab s0+1 (or s0 again when the segment has a single step: unused)
        if (ns > 2) advance();  // pointers now at the slab loaded during step 0 of the loop
        __syncthreads();

#define X_LD(g)                                                                                        \
    do {                                                                                               \
        if ((g) < 8) ract[(g)] = *reinterpret_cast<const g32x2*>(pa + (size_t)(g) * a.ld_act);         \
        else rw[(g) - 8] = *reinterpret_cast<const g16x8*>(pw[(g) - 8]);                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
#define X_MFMA(A_, B_, mi, ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[mi], B_[ni], acc[mi][ni], 0, 0, 0)
        for (int s = 0; s < ns; ++s) {
            const int buf = s & 1;
            if (TIMING) { t0 = __builtin_readcyclecounter(); tc[5] += 1; }
            if (s + 1 < ns) stage(buf ^ 1);  // slab s+1: loaded one step ago
            X_T(0);
            const _Float16* L = lds + buf * TBUF;
            g16x8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                ah[mi] = *reinterpret_cast<const g16x8*>(L + P_AHI + arow + mi * 32 * TROW + ak0);
                al[mi] = *reinterpret_cast<const g16x8*>(L + P_ALO + arow + mi * 32 * TROW + ak0);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                bh[ni] = *reinterpret_cast<const g16x8*>(L + P_BHI + brow + ni * 32 * TROW + bk0);
                bl[ni] = *reinterpret_cast<const g16x8*>(L + P_BLO + brow + ni * 32 * TROW + bk0);
            }
            // k16 block 0: 24 MFMAs, the 12 loads of slab s+2 one behind each of the first 12
            X_MFMA(ah, bh, 0, 0); X_LD(0);  X_MFMA(ah, bh, 1, 0); X_LD(1);  X_MFMA(ah, bh, 0, 1); X_LD(2);
            X_MFMA(ah, bh, 1, 1); X_LD(3);  X_MFMA(ah, bh, 0, 2); X_LD(4);  X_MFMA(ah, bh, 1, 2); X_LD(5);
            X_MFMA(ah, bh, 0, 3); X_LD(6);  X_MFMA(ah, bh, 1, 3); X_LD(7);  X_MFMA(ah, bl, 0, 0); X_LD(8);
            X_MFMA(ah, bl, 1, 0); X_LD(9);  X_MFMA(ah, bl, 0, 1); X_LD(10); X_MFMA(ah, bl, 1, 1); X_LD(11);
            X_MFMA(ah, bl, 0, 2); X_MFMA(ah, bl, 1, 2); X_MFMA(ah, bl, 0, 3); X_MFMA(ah, bl, 1, 3);
            X_MFMA(al, bh, 0, 0); X_MFMA(al, bh, 1, 0); X_MFMA(al, bh, 0, 1); X_MFMA(al, bh, 1, 1);
            X_MFMA(al, bh, 0, 2); X_MFMA(al, bh, 1, 2); X_MFMA(al, bh, 0, 3); X_MFMA(al, bh, 1, 3);
            X_T(1);
            // k16 block 1
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                ah[mi] = *reinterpret_cast<const g16x8*>(L + P_AHI + arow + mi * 32 * TROW + ak1);
                al[mi] = *reinterpret_cast<const g16x8*>(L + P_ALO + arow + mi * 32 * TROW + ak1);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                bh[ni] = *reinterpret_cast<const g16x8*>(L + P_BHI + brow + ni * 32 * TROW + bk1);
                bl[ni] = *reinterpret_cast<const g16x8*>(L + P_BLO + brow + ni * 32 * TROW + bk1);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) { X_MFMA(ah, bh, 0, ni); X_MFMA(ah, bh, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) { X_MFMA(ah, bl, 0, ni); X_MFMA(ah, bl, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) { X_MFMA(al, bh, 0, ni); X_MFMA(al, bh, 1, ni); }
            if (s + 3 < ns) advance();  // the last steps re-load an in-bounds slab (unused): the step stays branch-free
            X_T(2);
            if (TIMING) asm volatile("" :: "v"(acc[1][3][15]));
            X_T(3);
            __syncthreads();
            X_T(4);
        }
#undef X_MFMA
#undef X_LD

        if (is_head) {  // publish the fragment for slot n+1 (agent-scope release by one lane; guide, Guideline 16)
            f32x4* w = reinterpret_cast<f32x4*>(a.partial + (size_t)p * kFragFloats) + tid * 32;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        f32x4 v;
                        v[0] = acc[mi][ni][r4 * 4 + 0]; v[1] = acc[mi][ni][r4 * 4 + 1];
                        v[2] = acc[mi][ni][r4 * 4 + 2]; v[3] = acc[mi][ni][r4 * 4 + 3];
                        w[(mi * 4 + ni) * 4 + r4] = v;
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.flags + p, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            int tid_ = threadIdx.x;
            asm volatile("" : "+v"(tid_));  // keep the epilogue's address arithmetic inside the segment loop
            const int ln = tid_ & 63, l31 = ln & 31;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int j = j0 + 128 * wc + 32 * ni + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = i0 + 64 * wr + 32 * mi + frag_row(r, ln);
                        float v = acc[mi][ni][r] * kOutScale;
                        if (EPI == XEPI_BIAS_I || EPI == XEPI_BIAS_I_GELU || EPI == XEPI_BIAS_I_SCALE_RES || EPI == XEPI_BIAS_I_RELU)
                            v = v + a.bias[i];
                        if (EPI == XEPI_BIAS_J) v = v + a.bias[j];
                        if (EPI == XEPI_BIAS_I_GELU) v = gelu_x(v);
                        if (EPI == XEPI_BIAS_I_RELU) v = fmaxf(v, 0.f);
                        if (EPI == XEPI_BIAS_I_SCALE_RES) v = a.res[(unsigned)i * (unsigned)a.ldr + (unsigned)j] + a.scale[i] * v;
                        a.D[(unsigned)i * (unsigned)a.ldd + (unsigned)j] = v;
                    }
                }
            __syncthreads();  // LDS buffer 0 is re-staged by the next segment's prologue
        }
    }
    if (bad) gp_raise(a.status, GP_ST_SPLIT_RANGE);
    if (TIMING && blockIdx.x == 100 && tid == 0)
        for (int i = 0; i < 6; ++i) g_t256[i] = tc[i];
}

unsigned g_epoch256 = 0;
#undef X_T


// ---------------------------------------------------------------------------------------------------------------
// Third generation: BOTH operands arrive as pre-split f16 planes [n][K] (weights x 64 as above; activations x 8, written
// token-major by their producers: gp_vit.hip's LayerNorm / attention kernels and this kernel's plane epilogues), so the
// staging of a k-step is eight 16-byte copies per thread -- no conversion, one address register (buffer loads:
// descriptor + per-thread voffset + scalar offset) -- and the two wave groups of the workgroup (waves 0-3 and 4-7, one
// wave of each per SIMD) run half a step apart: while one group issues the 48 MFMAs of slab s (matrix phase C(s)) the
// other writes its share of slab s+1 to LDS and issues its loads of slab s+2 (memory phase M(s+1)), see the loop below.
// The lock-step kernel above idles the matrix pipe during every staging phase (60 % busy in-loop,
// profiles/r01_probe_split256.txt).  Tile, LDS layout, single accumulator, stream-K hand-off and epilogues are those
// of gemm_split256_kernel; the arithmetic (operand values, k order) is identical, so results are bit-identical to it.
// Work distribution: data-parallel rounds of whole tiles first (L2 reuse), stream-K only for the last round + remainder.
enum { PEPI_GELU_PLANES = 6, PEPI_BIAS_I_PLANES = 7 };  // bias along i (6: + GELU), output as activation planes O[j][i] (x 8)
template <int EPI> constexpr bool kEpiPlanesOut = EPI == PEPI_GELU_PLANES || EPI == PEPI_BIAS_I_PLANES;

struct ArgsP {
    const _Float16* ahi; const _Float16* alo;  // A planes [I][K]
    const _Float16* bhi; const _Float16* blo;  // B planes [J][K]
    float* D; int ldd;                         // f32 output D[i][j]
    _Float16* ohi; _Float16* olo; int ldo;     // PEPI_*_PLANES: O[j][i]
    int K;
    const float* bias; const float* scale; const float* res; int ldr;
    int tiles_i, tiles_j, group;
    int* flags; float* partial; int epoch;
    float out_scale;
    int dp;
    unsigned long long* trace;  // TIMING builds: per-block segment time stamps (100 MHz ticks), else null
    int* status;                // guard rails (gp_common.h)
    int strip_j0, strip_fj;     // ragged J: rows [strip_j0, strip_j0 + 32 strip_fj) of B are not tiled, see strip_phase
    int par;                    // fewer tiles than slots: the slots of a tile split its K in PARALLEL (see the kernel)
    float plane_scale;                             // plane epilogues 6 / 7: the output tensor's power-of-two scale (default kActScale = 8)
    float* amax;                                   // calibration launches: max |x| of the planes written (null otherwise)
};

__device__ __forceinline__ float lane_bcast(float v, int lane)  // value held by `lane` (compile-time constant) -> SGPR
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// ---- Epilogues through LDS.  The accumulator fragment of v_mfma_f32_32x32x16_f16 gives a lane ONE column j and rows
// i in groups of four, so direct stores are 4-byte (f32 D[i][j]) or 8-byte (planes O[j][i]) pieces: 512 vector-memory
// instructions per lane and tile for the residual epilogue, each touching 2 (f32) or 32 (planes) cache lines -- the
// per-slot timeline showed the epilogues costing 40-90 us per tile against 73 us for the k loop of a K = 1024 tile.
// After the k loop the 128 KiB of operand buffers are free: each wave owns 16 KiB of them and turns its 64 x 128 tile
// in two rounds, so that every global access is 16 bytes per lane over full 128-byte lines (64 instead of 512
// instructions per lane for the residual epilogue).  Wave-private: no workgroup barrier between the rounds (LDS
// operations of one wave execute in order).  Same arithmetic per element as before: results are bit-identical.
template <int EPI, int NJ = 4, bool PIPE = true>
__device__ __forceinline__ void epilogue_f32_lds(const ArgsP& a, f32x16 (&acc)[2][NJ], float* __restrict__ wl, int i_base, int j_base, int ln)
{
    const int l31 = ln & 31, half = ln >> 5;
    constexpr bool kBiasI = EPI == XEPI_BIAS_I || EPI == XEPI_BIAS_I_GELU || EPI == XEPI_BIAS_I_SCALE_RES || EPI == XEPI_BIAS_I_RELU;
    if constexpr (NJ == 2) {
        // 256 x 128 tiles (launches far below one tile per slot): the wave tile is 64 x 64, a round is wl[32][64], 16 lanes cover a row
        // and a wave four rows per item, eight items per round.  The per-row constants are 4-byte loads of the row a lane finishes
        // (these launches are bound by their latency chain, not by instruction issue).  Per element the arithmetic of the 256-wide form.
        const int l15 = ln & 15, q4 = ln >> 4;
        const unsigned j = (unsigned)(j_base + 4 * l15);
        f32x4 bias_j = {0.f, 0.f, 0.f, 0.f};
        if (EPI == XEPI_BIAS_J) bias_j = *reinterpret_cast<const f32x4*>(a.bias + j);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) wl[frag_row(r, ln) * 64 + 32 * ni + l31] = acc[mi][ni][r];
            f32x4 rs[8];
            float bs[8], ss[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {  // every load of the round first: D may be the residual buffer itself (see below)
                const unsigned i = (unsigned)(i_base + 32 * mi + 4 * u + q4);
                bs[u] = kBiasI ? a.bias[i] : 0.f;
                ss[u] = EPI == XEPI_BIAS_I_SCALE_RES ? a.scale[i] : 0.f;
                if (EPI == XEPI_BIAS_I_SCALE_RES) rs[u] = *reinterpret_cast<const f32x4*>(a.res + i * (unsigned)a.ldr + j);
            }
            if (EPI == XEPI_BIAS_I_SCALE_RES)
                asm volatile("" : "+v"(rs[0]), "+v"(rs[1]), "+v"(rs[2]), "+v"(rs[3]), "+v"(rs[4]), "+v"(rs[5]), "+v"(rs[6]), "+v"(rs[7]));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = 4 * u + q4;
                const f32x4 t = *reinterpret_cast<const f32x4*>(wl + row * 64 + 4 * l15);
                const unsigned i = (unsigned)(i_base + 32 * mi + row);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = t[e] * a.out_scale;
                    if (kBiasI) v = v + bs[u];
                    if (EPI == XEPI_BIAS_J) v = v + bias_j[e];
                    if (EPI == XEPI_BIAS_I_GELU) v = gelu_x(v);
                    if (EPI == XEPI_BIAS_I_RELU) v = fmaxf(v, 0.f);
                    if (EPI == XEPI_BIAS_I_SCALE_RES) v = rs[u][e] + ss[u] * v;
                    o[e] = v;
                }
                *reinterpret_cast<f32x4*>(a.D + i * (unsigned)a.ldd + j) = o;
            }
        }
    } else {
        float bias_l = 0.f, scale_l = 0.f;  // lane l holds the value of row i_base + l
        if (kBiasI) bias_l = a.bias[i_base + ln];
        if (EPI == XEPI_BIAS_I_SCALE_RES) scale_l = a.scale[i_base + ln];
        f32x4 bias_j = {0.f, 0.f, 0.f, 0.f};
        if (EPI == XEPI_BIAS_J) bias_j = *reinterpret_cast<const f32x4*>(a.bias + j_base + 4 * l31);
        // Rows as buffer accesses: one lane offset (column piece + the lane half's row) and a scalar row offset per item -- 64-bit flat
        // addresses cost an address pair per item in flight.
        constexpr bool kRes = EPI == XEPI_BIAS_I_SCALE_RES;
        const __amdgpu_buffer_rsrc_t r_d = __builtin_amdgcn_make_buffer_rsrc((void*)a.D, 0, 0x7ffffff0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(kRes ? a.res : a.D), 0, 0x7ffffff0, 0x00020000);
        const unsigned v_d = ((unsigned)half * (unsigned)a.ldd + (unsigned)(j_base + 4 * l31)) * 4u;
        const unsigned v_res = ((unsigned)half * (unsigned)a.ldr + (unsigned)(j_base + 4 * l31)) * 4u;
        // The tile's 32 items (row pairs) in eight batches of four.  Residual rows (round 6): batch b + 1 is requested BEFORE batch b is
        // computed and stored, into the other half of rs -- the wait for a batch then allows the previous batch's stores to be outstanding.
        // (Round 2's form, eight loads / wait / eight stores / eight loads, made every batch a full round trip INCLUDING the acknowledgment of
        // the stores before it -- s_waitcnt vmcnt(0) four times per tile, 14-16 us per proj / fc2 tile epilogue.)  In place (D == res) this is
        // safe as before: an item reads and writes only its own row piece, and its load precedes its store in program order.
        f32x4 rs[2][4];
        auto res_load = [&](int b) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned row0 = (unsigned)(i_base + 32 * (b >> 2) + 2 * (4 * (b & 3) + u));
                rs[b & 1][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, v_res, row0 * (unsigned)a.ldr * 4u, 0));
            }
        };
        // PIPE = false (the parallel split-K builds, which spill already): round 2's form, eight loads, then their two batches
        if (kRes && PIPE) res_load(0);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int mi = b >> 2;
            if ((b & 3) == 0) {
                // round mi: rows 32 mi .. 32 mi + 31 of the wave tile as wl[32][128] (writes: a lane group covers 32 consecutive
                // words of a row; reads: 16 bytes per lane, a wave covers two whole rows -- both conflict-free without padding)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) wl[frag_row(r, ln) * 128 + 32 * ni + l31] = acc[mi][ni][r];
            }
            if (kRes && PIPE && b + 1 < 8) {
                res_load(b + 1);
                asm volatile("" : "+v"(rs[(b + 1) & 1][0]), "+v"(rs[(b + 1) & 1][1]), "+v"(rs[(b + 1) & 1][2]), "+v"(rs[(b + 1) & 1][3]));
            }
            if (kRes && !PIPE && (b & 1) == 0) {
                res_load(b);
                res_load(b + 1);
                asm volatile("" : "+v"(rs[0][0]), "+v"(rs[0][1]), "+v"(rs[0][2]), "+v"(rs[0][3]), "+v"(rs[1][0]), "+v"(rs[1][1]), "+v"(rs[1][2]), "+v"(rs[1][3]));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = 4 * (b & 3) + u, row = 2 * it + half;
                const f32x4 t = *reinterpret_cast<const f32x4*>(wl + row * 128 + 4 * l31);
                float bb = 0.f, sc = 0.f;
                if (kBiasI) {
                    const float b0 = lane_bcast(bias_l, 32 * mi + 2 * it), b1 = lane_bcast(bias_l, 32 * mi + 2 * it + 1);
                    bb = half ? b1 : b0;
                }
                if (kRes) {
                    const float s0 = lane_bcast(scale_l, 32 * mi + 2 * it), s1 = lane_bcast(scale_l, 32 * mi + 2 * it + 1);
                    sc = half ? s1 : s0;
                }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = t[e] * a.out_scale;
                    if (kBiasI) v = v + bb;
                    if (EPI == XEPI_BIAS_J) v = v + bias_j[e];
                    if (EPI == XEPI_BIAS_I_GELU) v = gelu_x(v);
                    if (EPI == XEPI_BIAS_I_RELU) v = fmaxf(v, 0.f);
                    if (kRes) v = rs[b & 1][u][e] + sc * v;
                    o[e] = v;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r_d, v_d, (unsigned)(i_base + 32 * mi + 2 * it) * (unsigned)a.ldd * 4u, 0);
            }
        }
    }
}

// ---- Thin plane arithmetic (round 4).  The round-3 segment probes put the arithmetic half of a plane epilogue at 8.5 vector
// instructions per element (readlane + select for the bias, multiply, add, x 8, two conversions, a subtraction, a third
// conversion, a compare) and showed that it adds to the matrix work instead of hiding behind it.  Here: the per-row constants come
// from 16-byte loads of exactly the rows a lane holds (no readlane / select), every affine step is ONE fma, a pair of values
// becomes its two f16 planes in three instructions (v_cvt_pk_f16_f32; v_fma_mixlo / mixhi_f16 computing v - hi with the f16
// operand read straight from the packed hi pair), and the range guard is a NaN-propagating v_maximum3_f32 over |v| per pair.
typedef _Float16 g16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_hi_pair(float v0, float v1)
{
    g16x2 h;
    h[0] = (_Float16)v0;
    h[1] = (_Float16)v1;
    return __builtin_bit_cast(unsigned, h);
}
// (f16(v0 - hi.lo), f16(v1 - hi.hi)) in one register.  v - hi is exact in f32 (hi = v rounded to 11 bits), so the single rounding
// of the mixed-precision fma equals the f32 subtraction + conversion the other plane producers use.
__device__ __forceinline__ unsigned lo_pair(float v0, float v1, unsigned hi)
{
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(lo)
        : "v"(v0), "v"(v1), "v"(hi));
    return lo;
}
__device__ __forceinline__ float absmax3(float m, float v0, float v1)  // v_maximum3_f32: a NaN operand makes the result NaN
{
    return __builtin_elementwise_maximum(m, __builtin_elementwise_maximum(__builtin_fabsf(v0), __builtin_fabsf(v1)));
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 8 lanes that share lane >> 3, the same value (bit for bit) on all of them: quads (xor 1, xor 2), then the other quad
__device__ __forceinline__ float sum8_lanes(float v)
{
    v = v + dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v = v + dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + dpp_mov<0x141>(v);  // row_half_mirror: lane i <- lane 7 - i of its group of 8 (every lane of a quad holds the quad's sum)
    return v;
}
// 2 hs * gelu_x(x), bit for bit (the plane scale folded into the exact 0.5 x); hs = half the plane scale (4 for the default x 8)
__device__ __forceinline__ float gelu_fast_x8(float x, float hs = 0.5f * kActScale) { return gp_gelu_scaled(x, hs); }

// The plane epilogues 6 / 7: planes O[j][i] (x 8, hi + lo) of bias_i + out_scale acc [6: through GELU].  Round h = columns 64 h .. 64 h + 63
// of the wave tile, all 64 rows: per plane 64 token rows of 128 bytes in the wave's 16 KiB of LDS, the 8-byte pieces a lane produces
// XOR-placed by the token (2-way on the writes, the 16-byte reads conflict-free), then 16 bytes per lane: 8 lanes cover the 128
// contiguous bytes a token row gets from this wave.  Thin arithmetic (round 4; 0.27 ms per step over the round-3 form, same bits:
// profiles/r04_thin_epilogue_ab.txt): bias rows from 16-byte loads of exactly the rows a lane holds (no readlane + select), ONE fma for
// scale + bias + the planes' x 8 (acc * out_scale is an exact power-of-two scaling, so fma(acc, 8 out_scale, 8 b) rounds where
// (acc * out_scale + b) * 8 did), the pair conversions above, v_maximum3_f32 as the range guard, buffer stores.
// bias_w (round 6): the wave's 64 bias rows in LDS, parked there by the tile's prologue.  Read from global memory inside the rounds each of the
// 16 row quads of a tile was a dependent round trip behind `s_waitcnt vmcnt(0)` -- which in the second round also waits for the first round's 16
// stores to be acknowledged: the ISA showed load, wait, 8 ds_writes, load, wait, ... (the registers for hoisting 32 values do not exist at 235).
template <int EPI, int NJ = 4, int RB = 4>
__device__ __forceinline__ void epilogue_planes_thin(const ArgsP& a, f32x16 (&acc)[2][NJ], char* __restrict__ wl, int i_base, int j_base, int ln,
                                                     const float* __restrict__ bias_w)
{
    const int l31 = ln & 31, half = ln >> 5;
    constexpr bool kGelu = EPI == PEPI_GELU_PLANES;
    const float k8 = kGelu ? 1.0f : a.plane_scale, hs = 0.5f * a.plane_scale;  // the output tensor's power-of-two scale (8 by default)
    const __amdgpu_buffer_rsrc_t r_hi = __builtin_amdgcn_make_buffer_rsrc((void*)a.ohi, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_lo = __builtin_amdgcn_make_buffer_rsrc((void*)a.olo, 0, 0x7ffffff0, 0x00020000);
    const unsigned v_pl = ((unsigned)(j_base + (ln >> 3)) * (unsigned)a.ldo + (unsigned)(i_base + 8 * (ln & 7))) * 2u;
    const float A = a.out_scale * k8;
    float mx = 0.f;
#pragma unroll
    for (int h = 0; h < NJ / 2; ++h) {  // NJ = 2 (256 x 128 tiles): one round
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 bq = RB == 1 ? *reinterpret_cast<const f32x4*>(a.bias + i_base + 32 * mi + 8 * r4 + 4 * half)   // parallel split-K builds: as before
                                   : *reinterpret_cast<const f32x4*>(bias_w + 32 * mi + 8 * r4 + 4 * half);
                if (!kGelu) bq = bq * k8;
                const int c8 = 8 * mi + 2 * r4 + half;  // 8-byte piece of the 128-byte row: rows 4 c8 .. 4 c8 + 3
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) {
                    const int ni = 2 * h + nn, jl = 32 * nn + l31;
                    char* wrow = wl + jl * 128;
                    const int sw = (jl & 7) << 1;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        gp_f32x2 x = {__builtin_fmaf(acc[mi][ni][4 * r4 + e], A, bq[e]), __builtin_fmaf(acc[mi][ni][4 * r4 + e + 1], A, bq[e + 1])};
                        if (kGelu) x = gp_gelu_scaled2(x, hs);   // pairs: the polynomial as v_pk_fma_f32
                        v[e] = x[0];
                        v[e + 1] = x[1];
                    }
                    u32x2 oh, ol;
                    oh[0] = pack_hi_pair(v[0], v[1]);
                    oh[1] = pack_hi_pair(v[2], v[3]);
                    ol[0] = lo_pair(v[0], v[1], oh[0]);
                    ol[1] = lo_pair(v[2], v[3], oh[1]);
                    mx = absmax3(absmax3(mx, v[0], v[1]), v[2], v[3]);
                    *reinterpret_cast<u32x2*>(wrow + ((c8 ^ sw) << 3)) = oh;
                    *reinterpret_cast<u32x2*>(wrow + 8192 + ((c8 ^ sw) << 3)) = ol;
                }
            }
        // RB = 4 row pieces in flight per lane (left to itself hipcc turns this into read, wait, store, read, ... on ONE register quad; RB = 1:
        // that form, for the parallel split-K instantiations, which spill already)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int it0 = 0; it0 < 8; it0 += RB) {
                u32x4 v[RB];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int jl = 8 * (it0 + u) + (ln >> 3), c16 = ln & 7;
                    v[u] = *reinterpret_cast<const u32x4*>(wl + pl * 8192 + jl * 128 + ((c16 ^ (jl & 7)) << 4));
                }
                if constexpr (RB == 4) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
                for (int u = 0; u < RB; ++u)
                    __builtin_amdgcn_raw_buffer_store_b128(v[u], pl ? r_lo : r_hi, v_pl, (unsigned)(64 * h + 8 * (it0 + u)) * (unsigned)a.ldo * 2u, 0);
            }
        }
    }
    if (!(mx <= kSplitPlaneLimit)) gp_raise(a.status, GP_ST_SPLIT_RANGE);  // !(<=): NaN counts (v_maximum3 propagates it)
    if (a.amax) gp_record_amax(a.amax, mx, 1.0f / a.plane_scale);          // calibration launches only (wave-uniform branch)
}

// Strip fragments are handed out on demand: a slot asks for the next fragment when it has finished its tiles.  The per-slot
// time stamps show slots of different XCDs finishing equal work 8-11 % apart (k-step 2.33 us on the fastest XCD, 2.66 on the
// slowest, the same order in every launch of a box), and with the static split (fragment f to slot f) the 64 fragments of a
// ViT-L launch sat on the first 64 slots whatever their speed: a fc2 launch ended with 22 us of strip work on a quarter of
// the chip.  One counter word per scratch, tagged with the launch's epoch (no reset between launches): count in the low 12 bits.
constexpr int kStripCtr = 1030;  // header word (after the slot flags and the two error words)
__device__ __forceinline__ int strip_grab(int* ctr, int tag)
{
    // Once the word carries this launch's tag: one fetch-add per request.  The first arrivals of a launch still see the previous
    // launch's tag: they all compare-and-swap the SAME observed value against tag | 1 -- exactly one wins (and owns fragment 0),
    // the others re-read and join the fetch-adds.  (Measured on the way: a compare-and-swap per request costs a round per
    // contender, 450 us with 256 slots; fetch-add first and repair afterwards never settles while others keep adding.)
    for (;;) {
        int cur = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((cur & ~0xfff) == tag) return __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xfff;
        if (__hip_atomic_compare_exchange_strong(ctr, &cur, tag | 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 0;
    }
}

// ---- The ragged edge of J.  257 tokens per crop: J = 257 B is never a multiple of 256, and at B = 64 the 65th column of
// tiles (64 valid rows of 256) turned a 256-tile problem -- one whole tile per CU, no hand-over at all -- into 260 tiles
// on 256 slots: half the chip split tiles, published and re-read 256 KB accumulator fragments and finished 40 % later
// than the other half (per-slot time stamps, profiles/r02_planes_timeline.txt).  Now the tiles cover
// floor(J_valid / 256) * 256 rows and the remaining rows (< 256) are computed here as 32 x 32 fragments: the eight
// waves of a slot split K of ONE fragment (operands straight from global memory into the MFMA operand registers, up
// to 32 loads in flight per wave -- a single wave walking all of K needs K / 64 dependent memory round trips: 21 us at
// K = 1024, 78 us at K = 4096, measured), the partial accumulators are summed in wave order through LDS (fixed order:
// deterministic) and wave 0 applies the epilogue.  Same three products per k16 block; only the summation order over K
// differs from the tile path (strip outputs agree with tiled ones to f32 round-off, not bit for bit).
template <int EPI>
__device__ __forceinline__ void strip_phase(const ArgsP& a, float* __restrict__ red, int* __restrict__ next_f, int tid)
{
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5, wave = tid >> 6;
    const int nfi = a.tiles_i * (TB / 32), nfrag = nfi * a.strip_fj;
    const int nkb = a.K >> 4;                                        // k16 blocks
    const int kb0 = nkb * wave / 8, kb1 = nkb * (wave + 1) / 8;      // this wave's share of K (direct path)
    // K % 256 == 0 (every ViT shape): operands through LDS in coalesced 256-byte row pieces, two 256-k super-steps (64 KB + padding
    // each) per round with all 16 loads of a thread in flight; wave w then multiplies k16 blocks 2w, 2w + 1 of each super-step.  The
    // direct path below (each lane loading its own operand rows, 32 bytes per row per instruction) moves the same 256 KB per
    // fragment through the CU's vector-memory path at ~12 B/clk: 13 us at K = 1024, 42 us at K = 4096 (r02 timeline).
    constexpr int SP = 264, SPLANE = 32 * SP, SSTEP = 4 * SPLANE;    // halfs: padded row (528 B), plane, super-step (A hi, A lo, B hi, B lo)
    _Float16* sl = reinterpret_cast<_Float16*>(red);
    const bool staged = (a.K & 255) == 0;
    const int tag = (a.epoch & 0x7ffff) << 12;
    for (;;) {                                                       // uniform over the workgroup (barriers inside)
        if (tid == 0) *next_f = strip_grab(a.flags + kStripCtr, tag);
        __syncthreads();
        const int f = *next_f;
        if (f >= nfrag) break;
        const int i0 = (f % nfi) * 32, j0 = a.strip_j0 + (f / nfi) * 32;
        const _Float16* pah = a.ahi + (size_t)(i0 + l31) * a.K + 8 * half;
        const _Float16* pal = a.alo + (size_t)(i0 + l31) * a.K + 8 * half;
        const _Float16* pbh = a.bhi + (size_t)(j0 + l31) * a.K + 8 * half;
        const _Float16* pbl = a.blo + (size_t)(j0 + l31) * a.K + 8 * half;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // epilogue operands of this fragment (used by wave 0 only, requested now: their latency hides behind the K loop)
        constexpr bool kBiasI = EPI == XEPI_BIAS_I || EPI == XEPI_BIAS_I_GELU || EPI == XEPI_BIAS_I_SCALE_RES || EPI == XEPI_BIAS_I_RELU ||
                                EPI == PEPI_GELU_PLANES || EPI == PEPI_BIAS_I_PLANES;
        f32x4 pre_bias[4], pre_scale[4], pre_res[4];
        if (wave == 0) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int i = i0 + frag_row(4 * r4, lane);
                if (kBiasI) pre_bias[r4] = *reinterpret_cast<const f32x4*>(a.bias + i);
                if (EPI == XEPI_BIAS_I_SCALE_RES) {
                    pre_scale[r4] = *reinterpret_cast<const f32x4*>(a.scale + i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pre_res[r4][e] = a.res[(unsigned)(i + e) * (unsigned)a.ldr + (unsigned)(j0 + l31)];
                }
            }
        }
        if (staged) {
            const int srow = tid >> 4, sc = tid & 15;                // this thread's row and 16-byte piece (and piece + 16)
            const _Float16* g[4] = {a.ahi + (size_t)(i0 + srow) * a.K + 8 * sc, a.alo + (size_t)(i0 + srow) * a.K + 8 * sc,
                                    a.bhi + (size_t)(j0 + srow) * a.K + 8 * sc, a.blo + (size_t)(j0 + srow) * a.K + 8 * sc};
            const int nss = a.K >> 8, nr = (nss + 1) >> 1;           // rounds of two super-steps
            u32x4 va[2][4][2], vb[2][4][2];
            auto load = [&](u32x4 (&v)[2][4][2], int r) {
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                    for (int pl = 0; pl < 4; ++pl)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            v[b2][pl][h] = *reinterpret_cast<const u32x4*>(g[pl] + (size_t)min(2 * r + b2, nss - 1) * 256 + 128 * h);
            };
            auto round = [&](const u32x4 (&v)[2][4][2], int r) {
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                    for (int pl = 0; pl < 4; ++pl)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            *reinterpret_cast<u32x4*>(sl + b2 * SSTEP + pl * SPLANE + srow * SP + 8 * sc + 128 * h) = v[b2][pl][h];
                __syncthreads();
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    if (2 * r + b2 < nss) {                          // wave-uniform
                        const _Float16* L = sl + b2 * SSTEP + l31 * SP + 8 * half;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int ko = 16 * (2 * wave + u);
                            const g16x8 ah = *reinterpret_cast<const g16x8*>(L + ko), al = *reinterpret_cast<const g16x8*>(L + SPLANE + ko);
                            const g16x8 bh = *reinterpret_cast<const g16x8*>(L + 2 * SPLANE + ko), bl = *reinterpret_cast<const g16x8*>(L + 3 * SPLANE + ko);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                        }
                    }
                }
                __syncthreads();                                     // the buffers are rewritten by the next round / by `red`
            };
            load(va, 0);
            for (int r = 0; r < nr; r += 2) {                        // the next round's 16 loads fly while this one is written and multiplied
                if (r + 1 < nr) load(vb, r + 1);
                round(va, r);
                if (r + 1 < nr) {
                    if (r + 2 < nr) load(va, r + 2);
                    round(vb, r + 1);
                }
            }
        }
        int kb = staged ? kb1 : kb0;
        for (; kb + 8 <= kb1; kb += 8) {  // 32 unconditional 16-byte loads in flight, then 24 MFMAs (a guard per load would make
            g16x8 ah[8], al[8], bh[8], bl[8];  // hipcc branch around each one and wait for it: 32 dependent round trips)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ah[u] = *reinterpret_cast<const g16x8*>(pah + 16 * (kb + u));
                al[u] = *reinterpret_cast<const g16x8*>(pal + 16 * (kb + u));
                bh[u] = *reinterpret_cast<const g16x8*>(pbh + 16 * (kb + u));
                bl[u] = *reinterpret_cast<const g16x8*>(pbl + 16 * (kb + u));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], bh[u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], bl[u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u], bh[u], acc, 0, 0, 0);
            }
        }
        for (; kb < kb1; ++kb) {  // K / 16 not a multiple of 64: the odd blocks one at a time
            const g16x8 ah = *reinterpret_cast<const g16x8*>(pah + 16 * kb), al = *reinterpret_cast<const g16x8*>(pal + 16 * kb);
            const g16x8 bh = *reinterpret_cast<const g16x8*>(pbh + 16 * kb), bl = *reinterpret_cast<const g16x8*>(pbl + 16 * kb);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            f32x4 v;
            v[0] = acc[4 * r4 + 0]; v[1] = acc[4 * r4 + 1]; v[2] = acc[4 * r4 + 2]; v[3] = acc[4 * r4 + 3];
            *reinterpret_cast<f32x4*>(red + ((wave * 4 + r4) * 64 + lane) * 4) = v;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(red + (r4 * 64 + lane) * 4);
#pragma unroll
                for (int w = 1; w < 8; ++w) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(red + ((w * 4 + r4) * 64 + lane) * 4);
                    t[0] = t[0] + u[0]; t[1] = t[1] + u[1]; t[2] = t[2] + u[2]; t[3] = t[3] + u[3];
                }
                acc[4 * r4 + 0] = t[0]; acc[4 * r4 + 1] = t[1]; acc[4 * r4 + 2] = t[2]; acc[4 * r4 + 3] = t[3];
            }
            // per-element arithmetic of the tile epilogues, direct stores (a few KB per launch)
            const int j = j0 + l31;
            int bad = 0;
            float strip_mx = 0.f;          // 6 / 7: max |scaled value| this lane wrote (calibration launches record it)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int i = i0 + frag_row(4 * r4, lane);
                if (EPI == PEPI_GELU_PLANES || EPI == PEPI_BIAS_I_PLANES) {                    g16x4 oh, ol;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = acc[4 * r4 + e] * a.out_scale + pre_bias[r4][e];
                        const float v = (EPI == PEPI_GELU_PLANES ? gelu_x(x) : x) * a.plane_scale;
                        const _Float16 hh = (_Float16)v;
                        oh[e] = hh;
                        ol[e] = (_Float16)(v - (float)hh);
                        bad |= !(fabsf(v) <= kSplitPlaneLimit);
                        strip_mx = __builtin_elementwise_maximum(strip_mx, __builtin_fabsf(v));
                    }
                    const size_t o = (size_t)(unsigned)j * (unsigned)a.ldo + (unsigned)i;
                    *reinterpret_cast<g16x4*>(a.ohi + o) = oh;
                    *reinterpret_cast<g16x4*>(a.olo + o) = ol;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[4 * r4 + e] * a.out_scale;
                        if (EPI == XEPI_BIAS_I || EPI == XEPI_BIAS_I_GELU || EPI == XEPI_BIAS_I_SCALE_RES || EPI == XEPI_BIAS_I_RELU)
                            v = v + pre_bias[r4][e];
                        if (EPI == XEPI_BIAS_J) v = v + a.bias[j];
                        if (EPI == XEPI_BIAS_I_GELU) v = gelu_x(v);
                        if (EPI == XEPI_BIAS_I_RELU) v = fmaxf(v, 0.f);
                        if (EPI == XEPI_BIAS_I_SCALE_RES) v = pre_res[r4][e] + pre_scale[r4][e] * v;
                        a.D[(unsigned)(i + e) * (unsigned)a.ldd + (unsigned)j] = v;
                    }
                }
            }
            if (bad) gp_raise(a.status, GP_ST_SPLIT_RANGE);
            if ((EPI == PEPI_GELU_PLANES || EPI == PEPI_BIAS_I_PLANES) && a.amax) gp_record_amax(a.amax, strip_mx, 1.0f / a.plane_scale);
        }
        __syncthreads();  // `red` and *next_f are rewritten for the next fragment
    }
}

// NJ: 32-row matrix tiles of B per wave -- 4: the 256 x 256 tile; 2: a 256 x 128 tile (wave tile 64 x 64, half the rows of the B
// planes staged and half the matrix instructions per k-step) for launches whose 256 x 256 tiles fill at most half the slots: twice
// the tiles, so half the slots per tile and half-size partial accumulators, or no split at all (round 4; B <= 16 crops at ViT-L).
// PSPLIT (PAR builds): false = the parallel-split reduction is compiled OUT (every tile whole on its own slot, S = 1).  With the 256 x 128
// tiles taking every launch of at most 128 tiles, the 256 x 256 PAR build only ever runs 129-255 WHOLE tiles (ViT-L: q|k|v at 11-21
// crops, proj / fc2 at 33-63) -- yet it carried the reduction's registers: 256 VGPRs + 150-200 bytes of scratch, and in round 5 an
// unrelated edit moved those spills into the k loop (q|k|v at 16 crops 100 -> 214 us per launch, the 16-crop step 14.1 -> 16.5 ms;
// profiles/r05_b16_regression.txt).  The no-split instantiation has nothing to spill.
template <int EPI, bool TIMING = false, bool PAR = false, int NJ = 4, bool PSPLIT = true>
__global__ __launch_bounds__(TNT, 2) void gemm_planes256_kernel(const ArgsP a)
{
    static_assert(NJ == 4 || NJ == 2, "256 x 256 or 256 x 128 tiles");
    constexpr int TJ = 64 * NJ;  // rows of B (columns j) per tile
    unsigned long long tc[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
    const unsigned long long k_c0 = TIMING ? __builtin_readcyclecounter() : 0, k_w0 = TIMING ? wall_clock64() : 0;
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * TBUF + 2048];  // 128 KiB of operand buffers (+ 4 KiB: the strip's padded rows)
    __shared__ int strip_next;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, grp = wave >> 2;
    constexpr int P_AHI = 0, P_ALO = TPLANE, P_BHI = 2 * TPLANE, P_BLO = 3 * TPLANE;

    // ---- this slot's range of (tile, k-step) units inside its XCD's tile chunk (as gemm_split256_kernel)
    const int p = blockIdx.x, x = p & 7, n = p >> 3, slots_x = gridDim.x >> 3;
    const int T = a.tiles_i * a.tiles_j;
    const int t_lo = (int)((long long)T * x / 8), n_t = (int)((long long)T * (x + 1) / 8) - t_lo;
    const int nstep = a.K / TBK;
    // Data-parallel rounds first: while the chunk holds two or more tiles per slot, round r gives slot n the whole tile
    // r * slots_x + n -- the 32 slots of an XCD then sit on the same k-step of 32 neighbouring tiles (4 i-panels x 8
    // j-panels) and share their operand slabs in that XCD's L2.  Only the last round + remainder is cut stream-K style
    // (its slots run at staggered k offsets and get no L2 reuse: 1.9 GB fetched per fc1 launch when everything was).
    const int rounds_dp = (!PAR && (a.dp & 1) && n_t / slots_x > 1) ? n_t / slots_x - 1 : 0;
    const int n_dp = rounds_dp * slots_x;
    const long long U = (long long)(n_t - n_dp) * nstep;
    long long u0 = U * n / slots_x, u1 = U * (n + 1) / slots_x;
    // PAR (fewer tiles than slots): S = floor(slots / tiles) slots per tile, each an equal share of the tile's k-steps; slot n holds
    // part n % S of tile n / S, slots beyond S * tiles have no tile (they take strip fragments).  S = 1: whole tiles, no exchange.
    // (the GELU build keeps whole tiles, S = 1: with the partial-sum loop next to its epilogue hipcc spills 273 registers, and a
    // launch of T < 256 whole tiles on T slots is within 10 % of the split one at the sizes where it occurs -- fc1 below 16 crops)
    constexpr bool kParSplit = PAR && PSPLIT && EPI != PEPI_GELU_PLANES;
    const int par_S = kParSplit ? max(1, min(a.par, slots_x / max(n_t, 1))) : 1;  // a.par: the host's cap (k-steps per slot, see the launch)
    if (PAR) {
        const int tile = n / par_S, part = n - tile * par_S;
        u0 = u1 = 0;
        if (tile < n_t) {
            u0 = (long long)tile * nstep + nstep * part / par_S;
            u1 = (long long)tile * nstep + nstep * (part + 1) / par_S;
        }
    }
    const int ta = (int)(u0 / nstep), sa = (int)(u0 % nstep);
    const int tb = (int)(u1 / nstep), sb = (int)(u1 % nstep);
    const int n_head = sb > 0 ? 1 : 0, n_rest = sa > 0 ? 1 : 0;
    const int first_whole = ta + n_rest;
    const int n_seg = rounds_dp + n_head + (tb - first_whole) + n_rest;

    // staging: thread = (row tid >> 2 [+128], 16-byte k-chunk tid & 3) of each of the four planes
    const __amdgpu_buffer_rsrc_t r_ahi = __builtin_amdgcn_make_buffer_rsrc((void*)a.ahi, 0, a.tiles_i * TB * a.K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_alo = __builtin_amdgcn_make_buffer_rsrc((void*)a.alo, 0, a.tiles_i * TB * a.K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_bhi = __builtin_amdgcn_make_buffer_rsrc((void*)a.bhi, 0, a.tiles_j * TJ * a.K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_blo = __builtin_amdgcn_make_buffer_rsrc((void*)a.blo, 0, a.tiles_j * TJ * a.K * 2, 0x00020000);
    const unsigned voff = (unsigned)(tid >> 2) * (unsigned)a.K * 2u + (unsigned)(tid & 3) * 16u;
    const unsigned half_rows = 128u * (unsigned)a.K * 2u;  // byte distance of row + 128
    const int wofs = toff(tid >> 2, tid & 3);
    u32x4 rg[8];
    // fragment addressing
    const int ar_ = 64 * wr + (lane & 31), br_ = 32 * NJ * wc + (lane & 31), kh_ = lane >> 5;
    const int arow = ar_ * TROW, brow = br_ * TROW;
    const int ak0 = ((kh_ ^ ((ar_ >> 2) & 3)) << 3), ak1 = (((kh_ + 2) ^ ((ar_ >> 2) & 3)) << 3);
    const int bk0 = ((kh_ ^ ((br_ >> 2) & 3)) << 3), bk1 = (((kh_ + 2) ^ ((br_ >> 2) & 3)) << 3);

    if (TIMING && a.trace && tid == 0) {
        a.trace[(size_t)p * 32 + 0] = wall_clock64();
        a.trace[(size_t)p * 32 + 1] = (unsigned long long)n_seg;
    }
    for (int seg = 0; seg < n_seg; ++seg) {
        const bool is_dp = seg < rounds_dp;
        const bool is_head = !is_dp && seg - rounds_dp < n_head;
        const bool is_rest = !is_dp && n_rest && seg == n_seg - 1;
        const int t = is_dp ? seg * slots_x + n : n_dp + (is_head ? tb : (is_rest ? ta : first_whole + seg - rounds_dp - n_head));
        const int seg_s0 = is_rest ? sa : 0, seg_s1 = is_head ? sb : nstep;
        const int q = t_lo + t;
        const int per_band = a.group * a.tiles_j;
        const int band = q / per_band, rr = q - band * per_band;
        const int first_i = band * a.group;
        const int gsz = min(a.group, a.tiles_i - first_i);
        const int i0 = (first_i + rr % gsz) * TB, j0 = (rr / gsz) * TJ;

        const int s0 = seg_s0, s1 = seg_s1;
        f32x16 acc[2][NJ];
        if (!PAR && is_rest) {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(a.flags + (p - 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
                    __builtin_amdgcn_s_sleep(16);
                    if (++spins > kSpin) {
                        __hip_atomic_store(a.flags + kErrWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gp_raise(a.status, GP_ST_HANDOFF_SPLIT);
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            // piece u of thread tid at [u][tid] (a wave instruction covers 1 KB contiguous); buffer loads: one 32-bit lane offset + a
            // scalar offset per piece (64-bit flat addresses 8 KB apart cost an address pair per piece: 100 spilled registers)
            const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void*)(a.partial + (size_t)(p - 8) * kFragFloats), 0,
                                                                                 (int)(kFragFloats * sizeof(float)), 0x00020000);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NJ; ++ni)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rf, (unsigned)tid * 16u, (unsigned)(((mi * NJ + ni) * 4 + r4) * TNT) * 16u, 0);
                        const f32x4 vf = __builtin_bit_cast(f32x4, v);  // whole-vector cast (element-wise __builtin_bit_cast of vector lanes miscompiles: every r4 got lane group 0)
                        acc[mi][ni][r4 * 4 + 0] = vf[0]; acc[mi][ni][r4 * 4 + 1] = vf[1];
                        acc[mi][ni][r4 * 4 + 2] = vf[2]; acc[mi][ni][r4 * 4 + 3] = vf[3];
                    }
        } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NJ; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        }

        // ---- k loop over steps [s0, s1)
        const int ns = s1 - s0;
        if (TIMING && a.trace && tid == 0 && seg < 7) {  // kind (0 whole tile, 1 head = publishes, 2 rest = continues), steps, start
            a.trace[(size_t)p * 32 + 2 + 4 * seg] = ((unsigned long long)(is_head ? 1 : (is_rest ? 2 : 0)) << 32) | (unsigned)ns;
            a.trace[(size_t)p * 32 + 3 + 4 * seg] = wall_clock64();
        }
        const unsigned sA0 = ((unsigned)i0 * (unsigned)a.K + (unsigned)s0 * TBK) * 2u;  // scalar byte offsets of slab s0
        const unsigned sB0 = ((unsigned)j0 * (unsigned)a.K + (unsigned)s0 * TBK) * 2u;
        auto gload = [&](int slab) {
            const unsigned sA = sA0 + (unsigned)slab * (TBK * 2u), sB = sB0 + (unsigned)slab * (TBK * 2u);
            rg[0] = __builtin_amdgcn_raw_buffer_load_b128(r_ahi, voff, sA, 0);
            rg[1] = __builtin_amdgcn_raw_buffer_load_b128(r_ahi, voff, sA + half_rows, 0);
            rg[2] = __builtin_amdgcn_raw_buffer_load_b128(r_alo, voff, sA, 0);
            rg[3] = __builtin_amdgcn_raw_buffer_load_b128(r_alo, voff, sA + half_rows, 0);
            rg[4] = __builtin_amdgcn_raw_buffer_load_b128(r_bhi, voff, sB, 0);
            if (NJ == 4) rg[5] = __builtin_amdgcn_raw_buffer_load_b128(r_bhi, voff, sB + half_rows, 0);
            rg[6] = __builtin_amdgcn_raw_buffer_load_b128(r_blo, voff, sB, 0);
            if (NJ == 4) rg[7] = __builtin_amdgcn_raw_buffer_load_b128(r_blo, voff, sB + half_rows, 0);
        };
        auto stage = [&](int buf) {
            _Float16* L = lds + buf * TBUF + wofs;
            *reinterpret_cast<u32x4*>(L + P_AHI) = rg[0];
            *reinterpret_cast<u32x4*>(L + P_AHI + 128 * TROW) = rg[1];
            *reinterpret_cast<u32x4*>(L + P_ALO) = rg[2];
            *reinterpret_cast<u32x4*>(L + P_ALO + 128 * TROW) = rg[3];
            *reinterpret_cast<u32x4*>(L + P_BHI) = rg[4];
            if (NJ == 4) *reinterpret_cast<u32x4*>(L + P_BHI + 128 * TROW) = rg[5];
            *reinterpret_cast<u32x4*>(L + P_BLO) = rg[6];
            if (NJ == 4) *reinterpret_cast<u32x4*>(L + P_BLO + 128 * TROW) = rg[7];
        };
        float bias_v = 0.f;   // plane epilogues: the tile's 256 bias rows travel with the first slab and wait in LDS behind the operand buffers
        if constexpr (kEpiPlanesOut<EPI> && !kParSplit)
            if (tid < TB) bias_v = a.bias[i0 + tid];
        gload(0);
        stage(0);
        if constexpr (kEpiPlanesOut<EPI> && !kParSplit)
            if (tid < TB) reinterpret_cast<float*>(lds + 2 * TBUF)[tid] = bias_v;   // (the strip's rows: unused until strip_phase)
        __builtin_amdgcn_sched_barrier(0);
        if (ns > 1) gload(1);
        __syncthreads();

#define X_MFMA(A_, B_, mi, ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[mi], B_[ni], acc[mi][ni], 0, 0, 0)
        auto c_phase = [&](int s) __attribute__((always_inline)) {
            if (TIMING) { t0 = __builtin_readcyclecounter(); tc[5] += 1; }
            const _Float16* L = lds + (s & 1) * TBUF;
            __builtin_amdgcn_s_setprio(1);  // before the fragment reads: they must not queue behind the other group's staging
            g16x8 ah[2], al[2], bh[NJ], bl[NJ], ch[2], cl[2], dh[NJ], dl[NJ];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ah[mi] = *reinterpret_cast<const g16x8*>(L + P_AHI + arow + mi * 32 * TROW + ak0);
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni) bh[ni] = *reinterpret_cast<const g16x8*>(L + P_BHI + brow + ni * 32 * TROW + bk0);
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni) bl[ni] = *reinterpret_cast<const g16x8*>(L + P_BLO + brow + ni * 32 * TROW + bk0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) al[mi] = *reinterpret_cast<const g16x8*>(L + P_ALO + arow + mi * 32 * TROW + ak0);
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni) { X_MFMA(ah, bh, 0, ni); X_MFMA(ah, bh, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni) { X_MFMA(ah, bl, 0, ni); X_MFMA(ah, bl, 1, ni); }
            // second k16 block's fragments replace the dead ones (ah, bl) under the MFMAs in flight
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ch[mi] = *reinterpret_cast<const g16x8*>(L + P_AHI + arow + mi * 32 * TROW + ak1);
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni) dh[ni] = *reinterpret_cast<const g16x8*>(L + P_BHI + brow + ni * 32 * TROW + bk1);
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni) { X_MFMA(al, bh, 0, ni); X_MFMA(al, bh, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni) dl[ni] = *reinterpret_cast<const g16x8*>(L + P_BLO + brow + ni * 32 * TROW + bk1);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) cl[mi] = *reinterpret_cast<const g16x8*>(L + P_ALO + arow + mi * 32 * TROW + ak1);
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni) { X_MFMA(ch, dh, 0, ni); X_MFMA(ch, dh, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni) { X_MFMA(ch, dl, 0, ni); X_MFMA(ch, dl, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni) { X_MFMA(cl, dh, 0, ni); X_MFMA(cl, dh, 1, ni); }
            __builtin_amdgcn_s_setprio(0);
            if (TIMING) { t1 = __builtin_readcyclecounter(); tc[0] += t1 - t0; t0 = t1; }
        };
        auto m_phase = [&](int slab) __attribute__((always_inline)) {  // stages `slab`, loads slab + 1
            if (TIMING) { t0 = __builtin_readcyclecounter(); tc[5] += 1; }
            if (slab < ns) stage(slab & 1);
            __builtin_amdgcn_sched_barrier(0);
            gload(min(slab + 1, ns - 1));  // unconditional (the last ones re-load an L2-hot slab, unused)
            if (TIMING) { t1 = __builtin_readcyclecounter(); tc[2] += t1 - t0; t0 = t1; }
        };
        // One loop body for both groups; waves 4-7 run one memory phase ahead.  ONE barrier per k-step (ns - 1 for every
        // wave); what it orders: C(s+1) after every wave's M(s+1), M(s+2) after every wave's C(s):
        //     waves 0-3:  C0 M1 | C1 M2 | ...          waves 4-7:  M1 C0 | M2 C1 | ...
        // After a barrier waves 0-3 start their MFMAs at once and waves 4-7 stage first, so the offset is kept by the code
        // order; the matrix phases overlap in part (the pipe is shared, its work per step is the same).  Measured on the
        // fc2 shape (profiles/r01_probe_planes256_variants.txt): strict alternation with two barriers per step 4170
        // cycles per step, this 3670 (matrix pipe 84 % busy in-loop), without the matrix-phase priority 3850.
        if (grp) m_phase(1);
        for (int s = 0; s < ns; ++s) {
            c_phase(s);
            if (grp && s + 1 < ns) __syncthreads();
            if (TIMING) { t1 = __builtin_readcyclecounter(); tc[1] += t1 - t0; }
            m_phase(s + 1 + grp);
            if (!grp && s + 1 < ns) __syncthreads();
            if (TIMING) { t1 = __builtin_readcyclecounter(); tc[3] += t1 - t0; }
        }
#undef X_MFMA
        if (TIMING && a.trace && tid == 0 && seg < 7) a.trace[(size_t)p * 32 + 4 + 4 * seg] = wall_clock64();

        if (is_head) {  // publish the fragment for slot n+1 (agent-scope release by one lane)
            const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void*)(a.partial + (size_t)p * kFragFloats), 0,
                                                                                 (int)(kFragFloats * sizeof(float)), 0x00020000);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NJ; ++ni)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        f32x4 vf;
                        vf[0] = acc[mi][ni][r4 * 4 + 0]; vf[1] = acc[mi][ni][r4 * 4 + 1];
                        vf[2] = acc[mi][ni][r4 * 4 + 2]; vf[3] = acc[mi][ni][r4 * 4 + 3];
                        const u32x4 v = __builtin_bit_cast(u32x4, vf);
                        __builtin_amdgcn_raw_buffer_store_b128(v, rf, (unsigned)tid * 16u, (unsigned)(((mi * NJ + ni) * 4 + r4) * TNT) * 16u, 0);
                    }
            // (write-through `sc1` stores through inline asm, which the guide prices at 3.0 us against 8.2 us per 64 KB-per-workgroup
            // publish, measured 17 -> 14 us per 256 KB here: all slots publish at once and the 48 MB are bandwidth -- not kept)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(a.dp & 2))  // test hook (gp_gemm_planes256_set_dp(.. | 2)): a LOST hand-off -> the waiter must time out and flag it
                    __hip_atomic_store(a.flags + p, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (kParSplit && is_rest) {
                // Parallel split-K (fewer tiles than slots: B < 64 crops at ViT-L).  Every slot of this tile started from a zero
                // accumulator and ran its own k range at the same time; the slot holding the LAST range owns the tile: it adds
                // the published partial accumulators of the slots before it -- in slot order, nearest first: a fixed order, the
                // result depends on the shape only -- and runs the epilogue.  (The serial hand-over above keeps a split tile's
                // summation order but makes its slots wait for each other: with T < slots every tile is split and a launch
                // would take as long as one whole tile.)
                const int m_lo = n - (par_S - 1);   // the S - 1 slots before this one hold the earlier k ranges of the tile
                if (tid == 0) {
                    for (int m = n - 1; m >= m_lo; --m) {
                        int spins = 0;
                        while (__hip_atomic_load(a.flags + (x + 8 * m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
                            __builtin_amdgcn_s_sleep(16);
                            if (++spins > kSpin) {
                                __hip_atomic_store(a.flags + kErrWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                gp_raise(a.status, GP_ST_HANDOFF_SPLIT);
                                break;
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                for (int m = n - 1; m >= m_lo; --m) {
                    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void*)(a.partial + (size_t)(x + 8 * m) * kFragFloats), 0,
                                                                                         (int)(kFragFloats * sizeof(float)), 0x00020000);
                    constexpr int RQ = 4;  // 16-byte loads in flight per lane (then their 16 adds); more spills in the GELU build
#pragma unroll
                    for (int q0 = 0; q0 < 8 * NJ; q0 += RQ) {
                        u32x4 v[RQ];
#pragma unroll
                        for (int u = 0; u < RQ; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rf, (unsigned)tid * 16u, (unsigned)((q0 + u) * TNT) * 16u, 0);
#pragma unroll
                        for (int u = 0; u < RQ; ++u) {
                            const int f = q0 + u, mi = f / (4 * NJ), ni = (f >> 2) % NJ, r4 = f & 3;
                            const f32x4 vf = __builtin_bit_cast(f32x4, v[u]);
                            acc[mi][ni][r4 * 4 + 0] += vf[0]; acc[mi][ni][r4 * 4 + 1] += vf[1];
                            acc[mi][ni][r4 * 4 + 2] += vf[2]; acc[mi][ni][r4 * 4 + 3] += vf[3];
                        }
                    }
                }
            }
            int tid_ = threadIdx.x;
            asm volatile("" : "+v"(tid_));  // keep the epilogue's address arithmetic inside the segment loop
            __syncthreads();                // every wave has read its last operand fragments: the buffers are free
            char* wl = reinterpret_cast<char*>(lds) + (tid_ >> 6) * 16384;
            if constexpr (EPI == PEPI_GELU_PLANES || EPI == PEPI_BIAS_I_PLANES)
                epilogue_planes_thin<EPI, NJ, kParSplit ? 1 : 4>(a, acc, wl, i0 + 64 * wr, j0 + 32 * NJ * wc, tid_ & 63, reinterpret_cast<const float*>(lds + 2 * TBUF) + 64 * wr);
            else
                epilogue_f32_lds<EPI, NJ, !kParSplit>(a, acc, reinterpret_cast<float*>(wl), i0 + 64 * wr, j0 + 32 * NJ * wc, tid_ & 63);
            __syncthreads();  // LDS buffer 0 is re-staged by the next segment's prologue
        }
        if (TIMING && a.trace && tid == 0 && seg < 7) a.trace[(size_t)p * 32 + 5 + 4 * seg] = wall_clock64();
    }
    if (a.strip_fj > 0) strip_phase<EPI>(a, reinterpret_cast<float*>(lds), &strip_next, threadIdx.x);
    if (TIMING && a.trace && tid == 0) a.trace[(size_t)p * 32 + 31] = wall_clock64();
    if (TIMING && blockIdx.x == 100 && tid == 0) {
        for (int i = 0; i < 6; ++i) g_t256[i] = tc[i];
        g_t256[6] = __builtin_readcyclecounter() - k_c0;  // whole kernel, shader cycles
        g_t256[7] = wall_clock64() - k_w0;                // whole kernel, 100 MHz ticks
    }
}

// x [count] f32 -> planes hi = f16(scale x), lo = f16(scale x - hi)  (probe / test producer of activation planes)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ X, size_t count, float scale,
                                                            _Float16* __restrict__ hi, _Float16* __restrict__ lo)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float v = X[i] * scale;
    asm volatile("" : "+v"(v));  // no v_fma_mixlo_f16 fold of multiply + conversion: hi and lo must see the same rounded v
    const _Float16 h = (_Float16)v;
    hi[i] = h;
    lo[i] = (_Float16)(v - (float)h);
}

template <int EPI>
void launch256(Args256& a, bool act_is_b, int grid, hipStream_t st)
{
    if (act_is_b) hipLaunchKernelGGL((gemm_split256_kernel<EPI, true>), dim3(grid), dim3(TNT), 0, st, a);
    else hipLaunchKernelGGL((gemm_split256_kernel<EPI, false>), dim3(grid), dim3(TNT), 0, st, a);
}

}  // namespace

bool gp_gemm_split256_usable(int I, int J, int K)
{
    return I % TB == 0 && J % TB == 0 && K % TBK == 0 && (long long)(I / TB) * (J / TB) >= kSlots;
}

size_t gp_gemm_split256_scratch_bytes() { return kHeaderBytes + sizeof(float) * kFragFloats * kSlots; }

// scratch: flags zeroed by the caller once per forward (hipMemsetAsync of the first 8 KiB); one scratch per stream
int gp_gemm_split256_launch(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J,
                            int K, int act_is_b, int epilogue, const float* bias, const float* scale, const float* res, int ldr,
                            float* scratch, hipStream_t st)
{
    GP_REQUIRE(gp_gemm_split256_usable(I, J, K), "gp_gemm_split256: I=%d, J=%d must be multiples of 256 with >= 256 tiles, K=%d of 32", I, J, K);
    GP_REQUIRE(act && whi && wlo && D && scratch && ld_act % 2 == 0 && ((uintptr_t)act % 8 == 0) && ((uintptr_t)whi % 16 == 0) &&
                   ((uintptr_t)wlo % 16 == 0) && ((uintptr_t)scratch % 16 == 0),
               "gp_gemm_split256: null / misaligned operand");
    GP_REQUIRE((long long)I * ldd < (1ll << 31) && (long long)I * (ldr > 0 ? ldr : 1) < (1ll << 31), "gp_gemm_split256: output too large");
    Args256 a{act, ld_act, (const _Float16*)whi, (const _Float16*)wlo, D, ldd, K, bias, scale, res, ldr, I / TB, J / TB, 4,
              reinterpret_cast<int*>(scratch), reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + kHeaderBytes), 0, gp_status_buffer()};
    g_epoch256 = (g_epoch256 + 1) & 0x3fffffff;  // bit 30 marks this kernel's epochs: it may share a scratch (and its
    a.epoch = (int)(0x40000000u | g_epoch256);    // flags) with gp_gemm.hip's stream-K inside one forward
    GpProfScope prof(GP_PROF_GEMM_SPLIT, 2.0 * I * J * K, st);
    switch (epilogue) {
#ifdef GP_PROBES   // plain / ReLU epilogues: tests of the tile machinery only (the ViT's f32-activation fallback uses 1-4)
        case XEPI_NONE: launch256<XEPI_NONE>(a, act_is_b != 0, kSlots, st); break;
        case XEPI_BIAS_I_RELU: launch256<XEPI_BIAS_I_RELU>(a, act_is_b != 0, kSlots, st); break;
#endif
        case XEPI_BIAS_I: launch256<XEPI_BIAS_I>(a, act_is_b != 0, kSlots, st); break;
        case XEPI_BIAS_I_GELU: launch256<XEPI_BIAS_I_GELU>(a, act_is_b != 0, kSlots, st); break;
        case XEPI_BIAS_I_SCALE_RES: launch256<XEPI_BIAS_I_SCALE_RES>(a, act_is_b != 0, kSlots, st); break;
        case XEPI_BIAS_J: launch256<XEPI_BIAS_J>(a, act_is_b != 0, kSlots, st); break;
        default: GP_REQUIRE(false, "gp_gemm_split256: epilogue %d is not built (product: 1-4; 0 and 5 need -DGP_PROBES)", epilogue);
    }
    GP_CHECK_LAUNCH("gp_gemm_split256");
    return GP_OK;
}

static int g_planes_dp = 1;  // 1: data-parallel rounds before the stream-K remainder (0: everything stream-K; A/B hook)

// internal entry (gp_vit.hip): D[i][j] = epi( out_scale * sum_k A[i][k] B[j][k] ), A / B = pre-split planes (see above)
// J_valid <= J: rows of B that carry data (J itself stays a multiple of 256: the buffers are padded).  Tiles cover
// floor(J_valid / 256) * 256 rows, strip_phase the rest (rows beyond round_up(J_valid, 32) are neither read nor written).
static void planes_ragged(int J, int J_valid, int& J_main, int& strip_fj)
{
    J_main = (J_valid >= J || J_valid <= 0) ? J : (J_valid / TB) * TB;
    strip_fj = (J_main == J) ? 0 : (J_valid - J_main + 31) / 32;
}
// Fewer tiles than slots (B < 64 crops at ViT-L): the kernel's parallel split-K mode needs at least one tile per XCD and one
// k-step per slot (every slot's range then lies inside one or two tiles).
static bool planes_par_usable(int I, int J_main, int K)
{
    if (I <= 0 || J_main <= 0 || I % TB || J_main % TB || K % TBK) return false;
    const long long T = (long long)(I / TB) * (J_main / TB);
    return T < kSlots && T >= 8 && (T / 8) * (K / TBK) >= kSlots / 8;
}
static int g_planes_par = 1;  // 0: shapes with fewer tiles than slots are refused (the caller falls back to the 128 x 128 kernel; A/B hook)
static int g_planes_half = 1; // 0: never the 256 x 128 tiles (A/B hook, gp_gemm_planes256_set_half_tiles)
// A tile is split over at most K / 32 / kParMinSteps slots: below 16 k-steps per slot the partial accumulators (256 KB each, published by
// all non-owners at once, then read by the owner) cost more than the shorter k loop saves -- proj (K = 1024) at 16 crops 69 -> 62 us with
// two instead of four slots per tile, at 8 crops 74 -> 55 us with two or four instead of eight (PAR_MIN_STEPS=.. tools/probe_planes_timeline.py).
constexpr int kParMinSteps = 16;
static int g_planes_par_min_steps = kParMinSteps;
bool gp_gemm_planes256_usable(int I, int J, int J_valid, int K)
{
    int J_main, fj;
    planes_ragged(J, J_valid, J_main, fj);
    if (J % TB || J_valid > J) return false;
    if ((long long)(I / 32) * fj + kSlots >= 4096) return false;  // strip fragments must fit the 12-bit counter (strip_grab)
    return gp_gemm_split256_usable(I, J_main, K) || (g_planes_par && planes_par_usable(I, J_main, K));
}

int gp_gemm_planes256_launch(const void* ahi, const void* alo, const void* bhi, const void* blo, float* D, int ldd, void* ohi,
                             void* olo, int ldo, int I, int J, int J_valid, int K, int epilogue, const float* bias, const float* scale,
                             const float* res, int ldr, float out_scale, float* scratch, hipStream_t st, unsigned long long* trace,
                             const GpPlaneOut* po = nullptr)
{
    GP_REQUIRE(gp_gemm_planes256_usable(I, J, J_valid, K),
               "gp_gemm_planes256: I=%d, J=%d must be multiples of 256 with >= 8 tiles below J_valid=%d, K=%d of 32", I, J, J_valid, K);
    int J_main, strip_fj;
    planes_ragged(J, J_valid, J_main, strip_fj);
    GP_REQUIRE((long long)(I / 32) * strip_fj + kSlots < 4096, "gp_gemm_planes256: too many strip fragments for the 12-bit counter");
    GP_REQUIRE(ahi && alo && bhi && blo && scratch && ((uintptr_t)ahi % 16 == 0) && ((uintptr_t)alo % 16 == 0) &&
                   ((uintptr_t)bhi % 16 == 0) && ((uintptr_t)blo % 16 == 0) && ((uintptr_t)scratch % 16 == 0),
               "gp_gemm_planes256: null / misaligned operand");
    GP_REQUIRE((long long)I * K * 2 < (1ll << 31) && (long long)J * K * 2 < (1ll << 31), "gp_gemm_planes256: operand planes too large");
    const bool planes_out = epilogue == PEPI_GELU_PLANES || epilogue == PEPI_BIAS_I_PLANES;
    if (planes_out)
        GP_REQUIRE(ohi && olo && ldo % 4 == 0 && ((uintptr_t)ohi % 8 == 0) && ((uintptr_t)olo % 8 == 0) && bias, "gp_gemm_planes256: bad plane output");
    if (!planes_out)
        GP_REQUIRE(D && (long long)I * ldd < (1ll << 31) && (long long)I * (ldr > 0 ? ldr : 1) < (1ll << 31) && ldd % 4 == 0 && ldr % 4 == 0 &&
                       ((uintptr_t)D % 16 == 0) && ((uintptr_t)res % 16 == 0) && (epilogue != XEPI_BIAS_J || (uintptr_t)bias % 16 == 0),
                   "gp_gemm_planes256: bad f32 output (16-byte rows)");
    if (planes_out)
        GP_REQUIRE(ldo % 8 == 0 && ((uintptr_t)ohi % 16 == 0) && ((uintptr_t)olo % 16 == 0), "gp_gemm_planes256: plane output rows must be 16-byte aligned");
    ArgsP a{(const _Float16*)ahi, (const _Float16*)alo, (const _Float16*)bhi, (const _Float16*)blo, D, ldd, (_Float16*)ohi, (_Float16*)olo, ldo,
            K, bias, scale, res, ldr, I / TB, J_main / TB, 4, reinterpret_cast<int*>(scratch),
            reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + kHeaderBytes), 0, out_scale, g_planes_dp, trace, gp_status_buffer(),
            J_main, strip_fj, (long long)(I / TB) * (J_main / TB) < kSlots ? max(1, (K / TBK) / g_planes_par_min_steps) : 0};
    a.plane_scale = kActScale;
    a.amax = nullptr;
    if (po) {   // per-tensor output scale of the plane epilogues 6 / 7
        GP_REQUIRE(po->scale > 0.f && (planes_out || (po->scale == kActScale && !po->amax)),
                   "gp_gemm_planes256: a plane scale other than 8 needs epilogue 6 or 7 (got %d)", epilogue);
        a.plane_scale = po->scale;
        a.amax = po->amax;
    }
    g_epoch256 = (g_epoch256 + 1) & 0x3fffffff;
    a.epoch = (int)(0x40000000u | g_epoch256);
    GpProfScope prof(GP_PROF_GEMM_SPLIT, 2.0 * I * (J_valid > 0 && J_valid < J ? J_valid : J) * K, st);
    // The product's three epilogues (the ViT plane path: 3 = in-place residual, 6 = GELU -> planes, 7 = bias -> planes); the f32-output
    // epilogues 0, 1, 2, 4, 5 are instantiated in probe builds only (-DGP_PROBES: tests of the tile / strip / split-K machinery on
    // plain GEMMs, the f32-attention A/B path of the ViT)
#ifdef GP_PROBES
#define GP_PLANES_EXTRA(...)                                                                                              \
    case XEPI_NONE: hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_NONE, __VA_ARGS__>), dim3(kSlots), dim3(TNT), 0, st, a); break;           \
    case XEPI_BIAS_I: hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_BIAS_I, __VA_ARGS__>), dim3(kSlots), dim3(TNT), 0, st, a); break;       \
    case XEPI_BIAS_I_GELU: hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_BIAS_I_GELU, __VA_ARGS__>), dim3(kSlots), dim3(TNT), 0, st, a); break; \
    case XEPI_BIAS_J: hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_BIAS_J, __VA_ARGS__>), dim3(kSlots), dim3(TNT), 0, st, a); break;       \
    case XEPI_BIAS_I_RELU: hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_BIAS_I_RELU, __VA_ARGS__>), dim3(kSlots), dim3(TNT), 0, st, a); break;
    if (trace && a.par) {  // probe build of the parallel split-K kernel with per-slot time stamps
        switch (epilogue) {
            case XEPI_BIAS_I_SCALE_RES: hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_BIAS_I_SCALE_RES, true, true>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            case PEPI_GELU_PLANES: hipLaunchKernelGGL((gemm_planes256_kernel<PEPI_GELU_PLANES, true, true>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            case PEPI_BIAS_I_PLANES: hipLaunchKernelGGL((gemm_planes256_kernel<PEPI_BIAS_I_PLANES, true, true>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            default: GP_REQUIRE(false, "gp_gemm_planes256_trace: epilogue %d has no traced parallel build", epilogue);
        }
        GP_CHECK_LAUNCH("gp_gemm_planes256_trace/par");
        return GP_OK;
    }
    if (trace) {  // probe build with per-slot time stamps
        switch (epilogue) {
            case XEPI_NONE: hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_NONE, true>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            case XEPI_BIAS_I_SCALE_RES: hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_BIAS_I_SCALE_RES, true>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            case PEPI_GELU_PLANES: hipLaunchKernelGGL((gemm_planes256_kernel<PEPI_GELU_PLANES, true>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            case PEPI_BIAS_I_PLANES: hipLaunchKernelGGL((gemm_planes256_kernel<PEPI_BIAS_I_PLANES, true>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            default: GP_REQUIRE(false, "gp_gemm_planes256_trace: epilogue %d has no traced build", epilogue);
        }
        GP_CHECK_LAUNCH("gp_gemm_planes256_trace");
        return GP_OK;
    }
#else
#define GP_PLANES_EXTRA(...)
    GP_REQUIRE(!trace, "gp_gemm_planes256: time-stamped launches need the probe build (-DGP_PROBES)");
#endif
    // 256 x 128 tiles where the 256 x 256 ones fill at most half the slots (ViT-L below ~16 crops): the parallel split-K build on twice
    // the tiles -- proj / fc2 split every tile over half as many slots with half-size partial accumulators, q|k|v and fc1 stop
    // splitting / fill the chip with whole tiles.  (Between 128 and 256 whole tiles, twice as many half-width tiles cut stream-K
    // style over all slots by the serial hand-over build was measured too: ViT-L forward at 12 / 16 / 40 / 48 crops 8.55 -> 8.95,
    // 10.39 -> 10.44, 22.95 -> 22.5, 25.19 -> 25.55 ms -- the hand-overs cost what the balance gains; not kept.)
    const long long T256 = (long long)a.tiles_i * a.tiles_j;
    const bool epi_half = epilogue == XEPI_BIAS_I_SCALE_RES || epilogue == PEPI_GELU_PLANES || epilogue == PEPI_BIAS_I_PLANES;
    if (a.par && g_planes_half && epi_half && 2 * T256 <= kSlots) {
        a.tiles_j = J_main / (TB / 2);
        switch (epilogue) {
            case XEPI_BIAS_I_SCALE_RES: hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_BIAS_I_SCALE_RES, false, true, 2>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            case PEPI_GELU_PLANES: hipLaunchKernelGGL((gemm_planes256_kernel<PEPI_GELU_PLANES, false, true, 2>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            default: hipLaunchKernelGGL((gemm_planes256_kernel<PEPI_BIAS_I_PLANES, false, true, 2>), dim3(kSlots), dim3(TNT), 0, st, a); break;
        }
        GP_CHECK_LAUNCH("gp_gemm_planes256/par128");
        return GP_OK;
    }
    if (a.par && 2 * T256 > kSlots && (epilogue == XEPI_BIAS_I_SCALE_RES || epilogue == PEPI_BIAS_I_PLANES)) {
        // 129-255 whole tiles, one per slot (S = floor(256 / tiles) = 1): the PAR build WITHOUT the split-K reduction (see PSPLIT)
        a.par = 1;
        if (epilogue == XEPI_BIAS_I_SCALE_RES)
            hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_BIAS_I_SCALE_RES, false, true, 4, false>), dim3(kSlots), dim3(TNT), 0, st, a);
        else
            hipLaunchKernelGGL((gemm_planes256_kernel<PEPI_BIAS_I_PLANES, false, true, 4, false>), dim3(kSlots), dim3(TNT), 0, st, a);
        GP_CHECK_LAUNCH("gp_gemm_planes256/par-whole");
        return GP_OK;
    }
    if (a.par) {  // fewer tiles than slots: the parallel split-K build (its own instantiation: the serial hand-over kernel keeps its registers)
        switch (epilogue) {
            GP_PLANES_EXTRA(false, true)
            case XEPI_BIAS_I_SCALE_RES: hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_BIAS_I_SCALE_RES, false, true>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            case PEPI_GELU_PLANES: hipLaunchKernelGGL((gemm_planes256_kernel<PEPI_GELU_PLANES, false, true>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            case PEPI_BIAS_I_PLANES: hipLaunchKernelGGL((gemm_planes256_kernel<PEPI_BIAS_I_PLANES, false, true>), dim3(kSlots), dim3(TNT), 0, st, a); break;
            default: GP_REQUIRE(false, "gp_gemm_planes256: epilogue %d is not built (product: 3, 6, 7; the others need -DGP_PROBES)", epilogue);
        }
        GP_CHECK_LAUNCH("gp_gemm_planes256/par");
        return GP_OK;
    }
    switch (epilogue) {
        GP_PLANES_EXTRA(false)
        case XEPI_BIAS_I_SCALE_RES: hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_BIAS_I_SCALE_RES>), dim3(kSlots), dim3(TNT), 0, st, a); break;
        case PEPI_GELU_PLANES: hipLaunchKernelGGL((gemm_planes256_kernel<PEPI_GELU_PLANES>), dim3(kSlots), dim3(TNT), 0, st, a); break;
        case PEPI_BIAS_I_PLANES: hipLaunchKernelGGL((gemm_planes256_kernel<PEPI_BIAS_I_PLANES>), dim3(kSlots), dim3(TNT), 0, st, a); break;
        default: GP_REQUIRE(false, "gp_gemm_planes256: epilogue %d is not built (product: 3, 6, 7; the others need -DGP_PROBES)", epilogue);
    }
#undef GP_PLANES_EXTRA
    GP_CHECK_LAUNCH("gp_gemm_planes256");
    return GP_OK;
}

extern "C" {

size_t gp_gemm_split256_workspace_bytes(void) { return gp_gemm_split256_scratch_bytes(); }

int gp_split256_weights(const float* W, size_t count, void* hi, void* lo, void* stream)
{
    GP_REQUIRE(W && hi && lo && count > 0, "gp_split256_weights: bad arguments");
    hipLaunchKernelGGL(split256_weights_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, count,
                       (_Float16*)hi, (_Float16*)lo);
    GP_CHECK_LAUNCH("gp_split256_weights");
    return GP_OK;
}

int gp_gemm_split256(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J, int K,
                     int act_is_b, int epilogue, const float* bias, const float* scale, const float* residual, int ldr,
                     float* scratch, size_t scratch_bytes, void* stream)
{
    GP_REQUIRE(scratch && scratch_bytes >= gp_gemm_split256_scratch_bytes(), "gp_gemm_split256: scratch too small");
    if (hipMemsetAsync(scratch, 0, kHeaderBytes, (hipStream_t)stream) != hipSuccess) return GP_ELAUNCH;
    return gp_gemm_split256_launch(act, ld_act, whi, wlo, D, ldd, I, J, K, act_is_b, epilogue, bias, scale, residual, ldr, scratch,
                                   (hipStream_t)stream);
}

int gp_split_planes(const float* X, size_t count, float scale, void* hi, void* lo, void* stream)
{
    GP_REQUIRE(X && hi && lo && count > 0, "gp_split_planes: bad arguments");
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, count, scale,
                       (_Float16*)hi, (_Float16*)lo);
    GP_CHECK_LAUNCH("gp_split_planes");
    return GP_OK;
}

/* The plane GEMM as a stage entry: D / planes = epi(out_scale * A B^T) on pre-split planes, J_valid <= J rows of B carrying data (tiles +
 * ragged strip), the output planes of epilogues 6 / 7 carrying the power-of-two plane_scale (8 = the default; the consumer GEMM then runs
 * with out_scale = 1 / (plane_scale * 64)) and, for calibration passes, a device float that receives max |x| of the planes written. */
int gp_gemm_planes256_scaled(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, float* D, int ldd, void* out_hi,
                             void* out_lo, int ldo, int I, int J, int J_valid, int K, int epilogue, const float* bias, const float* scale,
                             const float* residual, int ldr, float out_scale, float plane_scale, float* amax, float* scratch,
                             size_t scratch_bytes, void* stream)
{
    GP_REQUIRE(scratch && scratch_bytes >= gp_gemm_split256_scratch_bytes(), "gp_gemm_planes256_scaled: scratch too small");
    if (hipMemsetAsync(scratch, 0, kHeaderBytes, (hipStream_t)stream) != hipSuccess) return GP_ELAUNCH;
    const GpPlaneOut po{plane_scale, amax};
    return gp_gemm_planes256_launch(a_hi, a_lo, b_hi, b_lo, D, ldd, out_hi, out_lo, ldo, I, J, J_valid, K, epilogue, bias, scale, residual, ldr,
                                    out_scale, scratch, (hipStream_t)stream, nullptr, &po);
}

#ifdef GP_PROBES
/* probe: proj-shaped launch (act_is_b, no epilogue) with per-phase cycle counters; out6 (host): stage, k16-0 issue,
 * k16-1 issue, MFMA drain, barrier, steps */
int gp_gemm_split256_timing(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J, int K,
                            float* scratch, unsigned long long* out6, void* stream)
{
    GP_REQUIRE(gp_gemm_split256_usable(I, J, K) && out6, "gp_gemm_split256_timing: bad arguments");
    if (hipMemsetAsync(scratch, 0, kHeaderBytes, (hipStream_t)stream) != hipSuccess) return GP_ELAUNCH;
    Args256 a{act, ld_act, (const _Float16*)whi, (const _Float16*)wlo, D, ldd, K, nullptr, nullptr, nullptr, 0, I / TB, J / TB, 4,
              reinterpret_cast<int*>(scratch), reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + kHeaderBytes), 0};
    if (++g_epoch256 == 0) ++g_epoch256;
    a.epoch = (int)g_epoch256;
    hipLaunchKernelGGL((gemm_split256_kernel<XEPI_NONE, true, true>), dim3(kSlots), dim3(TNT), 0, (hipStream_t)stream, a);
    GP_CHECK_LAUNCH("gp_gemm_split256_timing");
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return GP_ELAUNCH;
    return hipMemcpyFromSymbol(out6, HIP_SYMBOL(g_t256), 6 * sizeof(unsigned long long)) == hipSuccess ? GP_OK : GP_ELAUNCH;
}

/* probe: no-epilogue launch with per-phase cycle counters of wave 0 of block 100; out6 (host, 8 entries): matrix phase,
 * barrier after it, memory phase, barrier after it, 0, phases, whole-kernel shader cycles, whole-kernel 100 MHz ticks */
int gp_gemm_planes256_timing(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, float* D, int ldd, int I, int J,
                             int K, float* scratch, unsigned long long* out6, void* stream)
{
    GP_REQUIRE(gp_gemm_split256_usable(I, J, K) && out6, "gp_gemm_planes256_timing: bad arguments");
    if (hipMemsetAsync(scratch, 0, kHeaderBytes, (hipStream_t)stream) != hipSuccess) return GP_ELAUNCH;
    ArgsP a{(const _Float16*)a_hi, (const _Float16*)a_lo, (const _Float16*)b_hi, (const _Float16*)b_lo, D, ldd, nullptr, nullptr, 0,
            K, nullptr, nullptr, nullptr, 0, I / TB, J / TB, 4, reinterpret_cast<int*>(scratch),
            reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + kHeaderBytes), 0, kOutScale, g_planes_dp};
    if (++g_epoch256 == 0) ++g_epoch256;
    a.epoch = (int)g_epoch256;
    hipLaunchKernelGGL((gemm_planes256_kernel<XEPI_NONE, true>), dim3(kSlots), dim3(TNT), 0, (hipStream_t)stream, a);
    GP_CHECK_LAUNCH("gp_gemm_planes256_timing");
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return GP_ELAUNCH;
    return hipMemcpyFromSymbol(out6, HIP_SYMBOL(g_t256), 8 * sizeof(unsigned long long)) == hipSuccess ? GP_OK : GP_ELAUNCH;
}

/* probe: the launch of gp_gemm_planes256_scaled (epilogues 0, 3, 6, 7, default plane scale) from a build with time stamps: trace (device,
 * 256 x 32 u64) receives per slot: [0] start, [1] segments, then per segment (first 7): kind << 32 | k-steps, start of its
 * k loop (after the accumulator hand-over wait, if any), end of the k loop, end of its epilogue / publish -- 100 MHz ticks */
int gp_gemm_planes256_trace(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, float* D, int ldd, void* out_hi,
                            void* out_lo, int ldo, int I, int J, int J_valid, int K, int epilogue, const float* bias, const float* scale,
                            const float* residual, int ldr, float out_scale, float* scratch, unsigned long long* trace, void* stream)
{
    GP_REQUIRE(trace && scratch, "gp_gemm_planes256_trace: bad arguments");
    if (hipMemsetAsync(scratch, 0, kHeaderBytes, (hipStream_t)stream) != hipSuccess) return GP_ELAUNCH;
    return gp_gemm_planes256_launch(a_hi, a_lo, b_hi, b_lo, D, ldd, out_hi, out_lo, ldo, I, J, J_valid, K, epilogue, bias, scale, residual, ldr,
                                    out_scale, scratch, (hipStream_t)stream, trace);
}

int gp_gemm_planes256_set_dp(int mode)  // bit 0: data-parallel rounds (default on); bit 1: test hook, head fragments are never published
{
    g_planes_dp = mode & 3;
    return GP_OK;
}

int gp_gemm_planes256_set_half_tiles(int on)  // 0: launches below half a tile per slot keep the 256 x 256 tiles (A/B hook)
{
    g_planes_half = on ? 1 : 0;
    return GP_OK;
}

int gp_gemm_planes256_set_par(int on)  // 0: refuse shapes with fewer tiles than slots (A/B hook: the ViT then takes the 128 x 128 kernels)
{
    g_planes_par = on ? 1 : 0;
    g_planes_par_min_steps = on > 1 ? on : kParMinSteps;   // on >= 2 (A/B hook): at least `on` k-steps per slot of a split tile
    return GP_OK;
}

int gp_gemm_split256_error(const float* scratch, void* stream)
{
    int e = -1;
    if (hipMemcpyAsync(&e, reinterpret_cast<const int*>(scratch) + kErrWord, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) !=
            hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return -1;
    return e;
}
#endif  // GP_PROBES

}  // extern "C"
