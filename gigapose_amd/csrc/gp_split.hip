// "f32-equivalent" dense contraction on the f16 matrix core (gfx950): every f32 operand x is split into two
// halves  x ~= hi + lo * 2^-11,  hi = f16(x),  lo = f16((x - hi) * 2^11)   (22-23 significant bits), and
//     sum_k a_k b_k  ~=  sum_k ahi*bhi  +  2^-11 * sum_k (ahi*blo + alo*bhi)
// runs as THREE v_mfma_f32_32x32x16_f16 per k-block with f32 accumulation (the two sums in separate
// accumulators).  The f16 products are exact in f32, so the only errors are the 2^-22-relative representation
// error per term and the dropped lo*lo term (2^-22) -- the same class as the rounding of an f32 fmaf chain.
// Throughput: 3/16 of the f32-input MFMA's time per flop (f16 MFMA is 16x the f32 MFMA rate).
//
// This is the OPT-IN fast mode of the ViT linear layers (GigaPose numerics contract: DESIGN.md section 2): results
// are NOT bit-identical to the fmaf-chain kernels of gp_gemm.hip (which stay the default and the parity path);
// tests/test_gpu_split.py bounds the difference against an f64 reference next to the exact kernel's own error.
//
// Layouts: same as gp_gemm.hip on the outside -- activations f32 k-major X[K][n] (converted while staging),
// result f32 D[I][ldd] -- plus PRE-SPLIT weights: gp_split_weights() turns W^T [K][n] f32 into two f16 planes
// [n][K] (k contiguous: the MFMA operand is 8 consecutive k per lane).
#include "gp_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * kLoScale);
}

// W^T [K][n] f32 (k-major) -> hi/lo [n][K] f16.  One 32x32 tile per block through LDS (coalesced both ways).
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ Wt, int K, int n, int ldw,
                                                             _Float16* __restrict__ hi, _Float16* __restrict__ lo)
{
    __shared__ float t[32][33];
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) t[r][tx] = (k0 + r < K && n0 + tx < n) ? Wt[(size_t)(k0 + r) * ldw + n0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        if (n0 + r < n && k0 + tx < K) {
            _Float16 h, l;
            split1(t[tx][r], h, l);
            hi[(size_t)(n0 + r) * K + k0 + tx] = h;
            lo[(size_t)(n0 + r) * K + k0 + tx] = l;
        }
    }
}

enum { SEPI_NONE = 0, SEPI_BIAS_I = 1, SEPI_BIAS_I_GELU = 2, SEPI_BIAS_I_SCALE_RES = 3, SEPI_BIAS_J = 4, SEPI_BIAS_I_RELU = 5 };

constexpr int SBM = 128, SBN = 128, SBK = 32, SNT = 256;
constexpr int SROW = 32;                       // halfs per LDS row (64 B, no padding): the 16-byte chunk kc of row r sits at chunk
                                               // position kc ^ ((r >> 2) & 3), so a ds_read_b128 lane group covers all 64 banks once
constexpr int SPLANE = 128 * SROW;             // halfs per (operand, hi|lo) plane
// 2 buffers x 4 planes x 8 KiB = 64 KiB per workgroup: TWO workgroups per CU.  (With 80-byte padded rows the kernel needed
// 80 KiB; 2 x 80 KiB = the whole 160 KiB LDS did not co-reside, occupancy was one 4-wave workgroup per CU -- found with the
// per-phase cycle probe: block time x tile count only matched the kernel time for ONE resident workgroup.)
__device__ __forceinline__ int lds_off(int row, int kc) { return row * SROW + ((kc ^ ((row >> 2) & 3)) << 3); }
constexpr int SBUF = 4 * SPLANE;               // A hi, A lo, B hi, B lo
constexpr int SLDS_BYTES = 2 * SBUF * 2;       // double buffered: 81920 B

struct SplitArgs {
    const float* act;  int ld_act;             // activations f32 k-major [K][ld_act]
    const _Float16* whi; const _Float16* wlo;  // pre-split weights [n][K]
    float* D; int ldd; int K;
    const float* bias; const float* scale; const float* res; int ldr;
    int tiles_i, tiles_j, group;
    const int* j_limit;  // optional device word: tiles whose first column is >= *j_limit have nothing to compute and return (gp_ist.hip: compacted rows)
};

// GELU: gp_common.h (gp_gelu_scaled), shared with gp_split256.hip
__device__ __forceinline__ float gelu_erf_s(float x) { return gp_gelu_scaled(x, 0.5f); }

// probe state (tools/probe_split.py; TIMING instantiation only)
__device__ unsigned long long g_split_clk[4][5];         // [wave][phase] cycle totals of one mid-grid block
__device__ unsigned long long g_split_wall[4];           // that block's shader cycles and 100 MHz wall ticks
__device__ unsigned long long* g_split_trace = nullptr;  // optional per-block trace: [block][start tick, end tick, hw id]

// ACT_IS_B: activations are the j operand (D = W . X: qk / proj / fc1 / fc2); else the i operand (token-major V).

template <int EPI, bool ACT_IS_B, bool TIMING = false>
__global__ __launch_bounds__(SNT, 2) void gemm_split_kernel(const SplitArgs a)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves, 64 x 64 each
    // with a column limit only the first tiles carry work: deal them round-robin over the XCDs (block id % 8) instead of giving each XCD one
    // contiguous chunk of the tile list -- chunked, 40 % live columns kept 3 of 8 XCDs busy and the launch took as long as the full one
    const int q = a.j_limit ? ((int)blockIdx.x < a.tiles_i * a.tiles_j ? (int)blockIdx.x : -1) : xcd_chunked_tile(blockIdx.x, a.tiles_i * a.tiles_j);
    if (q < 0) return;
    const int per_band = a.group * a.tiles_j;
    const int band = q / per_band, rr = q - band * per_band;
    const int first_i = band * a.group;
    const int gsz = min(a.group, a.tiles_i - first_i);
    const int i0 = (first_i + rr % gsz) * SBM, j0 = (rr / gsz) * SBN;
    if (a.j_limit && j0 >= *a.j_limit) return;  // workgroup-uniform
    const int n_act0 = ACT_IS_B ? j0 : i0, n_w0 = ACT_IS_B ? i0 : j0;
    // plane offsets (halfs) inside one buffer
    constexpr int P_AHI = 0, P_ALO = SPLANE, P_BHI = 2 * SPLANE, P_BLO = 3 * SPLANE;
    const int act_hi = ACT_IS_B ? P_BHI : P_AHI, act_lo = ACT_IS_B ? P_BLO : P_ALO;
    const int w_hi = ACT_IS_B ? P_AHI : P_BHI, w_lo = ACT_IS_B ? P_ALO : P_BLO;

    // Staging, balanced over all 256 threads: every thread converts a 2 n x 8 k activation micro-block (8 float2
    // loads, 16 splits, 4 ds_write_b128) and copies four 16-byte chunks of the pre-split weight tile.  (With the
    // conversion on two of the four waves only, those two set the pace of every k-step: 1760 of 3585 cycles, the other
    // two waiting 1400 at the barrier -- measured with gp_gemm_split_timing.)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 ract[8];
    f16x8 rw[4];
    const int ng = tid & 63, kg = tid >> 6;  // 64 column pairs x 4 k-groups of 8
    auto gload = [&](int k0) {
        const float* src = a.act + (size_t)(k0 + kg * 8) * a.ld_act + n_act0 + ng * 2;
#pragma unroll
        for (int r = 0; r < 8; ++r) ract[r] = *reinterpret_cast<const f32x2*>(src + (size_t)r * a.ld_act);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = tid + 256 * u;  // plane c >> 9, row (c & 511) >> 2, k-chunk c & 3
            const _Float16* base = (c >> 9) ? a.wlo : a.whi;
            rw[u] = *reinterpret_cast<const f16x8*>(base + (size_t)(n_w0 + ((c & 511) >> 2)) * a.K + k0 + (c & 3) * 8);
        }
    };
    auto stage = [&](int buf) {
        _Float16* L = lds + buf * SBUF;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = tid + 256 * u;
            *reinterpret_cast<f16x8*>(L + ((c >> 9) ? w_lo : w_hi) + lds_off((c & 511) >> 2, c & 3)) = rw[u];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {  // column n = ng*2 + c: 8 consecutive k
            f16x8 h, l;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                _Float16 hh_, ll_;
                split1(ract[r][c], hh_, ll_);
                h[r] = hh_;
                l[r] = ll_;
            }
            const int off = lds_off(ng * 2 + c, kg);
            *reinterpret_cast<f16x8*>(L + act_hi + off) = h;
            *reinterpret_cast<f16x8*>(L + act_lo + off) = l;
        }
    };

    f32x16 hh[2][2], xx[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) { hh[mi][ni][r] = 0.f; xx[mi][ni][r] = 0.f; }

    const int nstep = a.K / SBK;
    gload(0);
    stage(0);
    if (nstep > 1) gload(SBK);  // slab 1 waits in registers for the first loop iteration
    __syncthreads();
    // fragment addresses: row base + swizzled chunk of k16 block 0 / 1 (the swizzle depends on row bits 2-3 only, so it is
    // the same for the rows +32 of the second MFMA tile)
    const int ar_ = wm * 64 + (lane & 31), br_ = wn * 64 + (lane & 31), kh_ = lane >> 5;
    const int arow = ar_ * SROW, brow = br_ * SROW;
    const int ak0 = ((kh_ ^ ((ar_ >> 2) & 3)) << 3), ak1 = (((kh_ + 2) ^ ((ar_ >> 2) & 3)) << 3);
    const int bk0 = ((kh_ ^ ((br_ >> 2) & 3)) << 3), bk1 = (((kh_ + 2) ^ ((br_ >> 2) & 3)) << 3);
    const bool timed = TIMING && blockIdx.x == gridDim.x / 2 && lane == 0;
    unsigned long long tc[5] = {0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
    const unsigned long long blk_c0 = TIMING ? __builtin_readcyclecounter() : 0, blk_w0 = TIMING ? wall_clock64() : 0;
    if (TIMING && g_split_trace && tid == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_split_trace[3 * (size_t)blockIdx.x + 0] = blk_w0;
        g_split_trace[3 * (size_t)blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
    }
    // running source pointers of this thread's 12 loads per k-step (8 activation float2 rows, 4 weight chunks)
    const int nstep_ = a.K / SBK;
    const int first = nstep_ > 2 ? 2 * SBK : (nstep_ > 1 ? SBK : 0);  // slab prefetched during step 0 (short K: an in-bounds, unused one)
    const float* pa = a.act + (size_t)(first + kg * 8) * a.ld_act + n_act0 + ng * 2;
    const _Float16* pw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = tid + 256 * u;
        pw[u] = ((c >> 9) ? a.wlo : a.whi) + (size_t)(n_w0 + ((c & 511) >> 2)) * a.K + first + (c & 3) * 8;
    }
    const size_t act_step = (size_t)SBK * a.ld_act;
    // One load between every two MFMAs: a global load takes ~85 cycles to ISSUE here (the CU's vector-memory path
    // moves 64 B/clk and a k-step pulls 64 KB per CU) -- time a wave otherwise spends before its first MFMA of the
    // step; behind an MFMA it overlaps with the matrix pipe.  sched_barrier pins the order (the scheduler otherwise
    // clusters all loads).  MFMAs on one accumulator are never issued back to back.  The loads are unconditional: a
    // branch around each one makes the compiler wait vmcnt(0) at every join (12 serialized round trips per step).
#define GP_LD(g)                                                                                              \
    do {                                                                                                      \
        if ((g) < 8) ract[(g)] = *reinterpret_cast<const f32x2*>(pa + (size_t)(g) * a.ld_act);                \
        else rw[(g) - 8] = *reinterpret_cast<const f16x8*>(pw[(g) - 8]);                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
#define GP_GROUP_LD(mi, g0)  /* 6 MFMAs, a load behind each */                                             \
    do {                                                                                                      \
        hh[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[0], hh[mi][0], 0, 0, 0);                \
        GP_LD(g0);                                                                                            \
        xx[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[0], xx[mi][0], 0, 0, 0);                \
        GP_LD(g0 + 1);                                                                                        \
        hh[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[1], hh[mi][1], 0, 0, 0);                \
        GP_LD(g0 + 2);                                                                                        \
        xx[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[1], xx[mi][1], 0, 0, 0);                \
        GP_LD(g0 + 3);                                                                                        \
        xx[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[0], xx[mi][0], 0, 0, 0);                \
        GP_LD(g0 + 4);                                                                                        \
        xx[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[1], xx[mi][1], 0, 0, 0);                \
        GP_LD(g0 + 5);                                                                                        \
    } while (0)
#define GP_GROUP(mi)                                                                                          \
    do {                                                                                                      \
        hh[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[0], hh[mi][0], 0, 0, 0);                \
        xx[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[0], xx[mi][0], 0, 0, 0);                \
        hh[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[1], hh[mi][1], 0, 0, 0);                \
        xx[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[1], xx[mi][1], 0, 0, 0);                \
        xx[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[0], xx[mi][0], 0, 0, 0);                \
        xx[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[1], xx[mi][1], 0, 0, 0);                \
    } while (0)
    // Order inside a step: (1) convert + write slab s+1 (loaded during step s-1: it has had a whole MFMA phase and a
    // barrier to arrive) into the other LDS buffer, (2) the MFMAs of slab s with the loads of slab s+2 behind the first
    // twelve of them (so each has >= half an MFMA phase + the next step's head before it is needed), (3) barrier.
    for (int s = 0; s < nstep; ++s) {
        const int buf = s & 1;
        if (TIMING) t0 = __builtin_readcyclecounter();
        if (s + 1 < nstep) stage(buf ^ 1);
        if (TIMING) { t1 = __builtin_readcyclecounter(); tc[3] += t1 - t0; t0 = t1; }
        const _Float16* L = lds + buf * SBUF;
        f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            ah[mi] = *reinterpret_cast<const f16x8*>(L + P_AHI + arow + mi * 32 * SROW + ak0);
            al[mi] = *reinterpret_cast<const f16x8*>(L + P_ALO + arow + mi * 32 * SROW + ak0);
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            bh[ni] = *reinterpret_cast<const f16x8*>(L + P_BHI + brow + ni * 32 * SROW + bk0);
            bl[ni] = *reinterpret_cast<const f16x8*>(L + P_BLO + brow + ni * 32 * SROW + bk0);
        }
        GP_GROUP_LD(0, 0);
        ah[0] = *reinterpret_cast<const f16x8*>(L + P_AHI + arow + ak1);  // second k16 block of the slab
        al[0] = *reinterpret_cast<const f16x8*>(L + P_ALO + arow + ak1);
        GP_GROUP_LD(1, 6);
        ah[1] = *reinterpret_cast<const f16x8*>(L + P_AHI + arow + 32 * SROW + ak1);
        al[1] = *reinterpret_cast<const f16x8*>(L + P_ALO + arow + 32 * SROW + ak1);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            bh[ni] = *reinterpret_cast<const f16x8*>(L + P_BHI + brow + ni * 32 * SROW + bk1);
            bl[ni] = *reinterpret_cast<const f16x8*>(L + P_BLO + brow + ni * 32 * SROW + bk1);
        }
        GP_GROUP(0);
        GP_GROUP(1);
        if (s + 3 < nstep) {  // the last steps re-load an in-bounds slab (unused): the k-step stays branch-free
            pa += act_step;
#pragma unroll
            for (int u = 0; u < 4; ++u) pw[u] += SBK;
        }
        if (TIMING) {  // issue of the LDS reads + MFMAs + loads, then the MFMA drain (an accumulator element is read)
            t1 = __builtin_readcyclecounter(); tc[1] += t1 - t0; t0 = t1;
            asm volatile("" :: "v"(hh[1][1][15]), "v"(xx[1][1][15]));
            t1 = __builtin_readcyclecounter(); tc[2] += t1 - t0; t0 = t1;
        }
        __syncthreads();
        if (TIMING) { t1 = __builtin_readcyclecounter(); tc[4] += t1 - t0; }
    }
#undef GP_GROUP_LD
#undef GP_GROUP
#undef GP_LD
    if (TIMING && g_split_trace && tid == 0) g_split_trace[3 * (size_t)blockIdx.x + 1] = wall_clock64();
    if (timed) {
        for (int ph = 0; ph < 5; ++ph) g_split_clk[wave][ph] = tc[ph];
        if (wave == 0) { g_split_wall[0] = __builtin_readcyclecounter() - blk_c0; g_split_wall[1] = wall_clock64() - blk_w0; }
    }

    // The 32 bias rows of a lane, loaded ONCE (round 6): inside the element loop each was a 4-byte load + s_waitcnt vmcnt(0) between two stores
    // (D may alias bias as far as hipcc knows) -- 64 dependent round trips per tile, each also waiting for the previous element's store; the
    // tile epilogue cost more than the k loop of the IST heads' K = 256 / 512 GEMMs.
    constexpr bool kBiasI = EPI == SEPI_BIAS_I || EPI == SEPI_BIAS_I_GELU || EPI == SEPI_BIAS_I_SCALE_RES || EPI == SEPI_BIAS_I_RELU;
    f32x4 bias_q[2][4];
    if (kBiasI) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) bias_q[mi][r4] = *reinterpret_cast<const f32x4*>(a.bias + i0 + wm * 64 + mi * 32 + frag_row(4 * r4, lane));
        asm volatile("" : "+v"(bias_q[0][0]), "+v"(bias_q[0][1]), "+v"(bias_q[0][2]), "+v"(bias_q[0][3]), "+v"(bias_q[1][0]), "+v"(bias_q[1][1]),
                          "+v"(bias_q[1][2]), "+v"(bias_q[1][3]));
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int j = j0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wm * 64 + mi * 32 + frag_row(r, lane);
                float v = hh[mi][ni][r] + xx[mi][ni][r] * kLoInv;
                if (kBiasI) v = v + bias_q[mi][r >> 2][r & 3];
                if (EPI == SEPI_BIAS_J) v = v + a.bias[j];
                if (EPI == SEPI_BIAS_I_GELU) v = gelu_erf_s(v);
                if (EPI == SEPI_BIAS_I_RELU) v = fmaxf(v, 0.f);
                if (EPI == SEPI_BIAS_I_SCALE_RES) v = a.res[(unsigned)i * (unsigned)a.ldr + (unsigned)j] + a.scale[i] * v;
                a.D[(unsigned)i * (unsigned)a.ldd + (unsigned)j] = v;
            }
        }
}

template <int EPI>
void launch_split(const SplitArgs& a, bool act_is_b, hipStream_t st)
{
    const int grid = xcd_chunked_grid(a.tiles_i * a.tiles_j);
    if (act_is_b) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<EPI, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SLDS_BYTES);
        hipLaunchKernelGGL((gemm_split_kernel<EPI, true>), dim3(grid), dim3(SNT), SLDS_BYTES, st, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<EPI, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SLDS_BYTES);
        hipLaunchKernelGGL((gemm_split_kernel<EPI, false>), dim3(grid), dim3(SNT), SLDS_BYTES, st, a);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Split-f16 convolution (IST backbone in split numerics): implicit GEMM on channel-LAST activations.
//   Y[pix][co] = epi( sum_k W[co][k] * Xcol[pix][k] ),   k = (dy, dx, ci) with ci fastest
// Activations are f16 planes hi / lo of shape (B, H, W, C): eight consecutive k at a fixed tap are eight consecutive
// channels = one 16-byte chunk, which IS the MFMA operand fragment -- the im2col gather is a plain chunk copy (zero
// for taps outside the image), no conversion.  Weights: planes (CoutPad, K), CoutPad = round_up(Cout, 128), K =
// KH*KW*Cin.  Tile 128 co x 128 pixels, 4 waves (2 x 2 of 64 x 64), main loop identical to gemm_split_kernel.
// Epilogue: folded eval-BatchNorm, residual (from planes), ReLU; output as planes (B, OH, OW, Cout) for the next
// convolution, or f32 NCHW for the last one.
struct ConvSplitArgs {
    const _Float16* xhi; const _Float16* xlo;
    const _Float16* whi; const _Float16* wlo;
    const float* alpha; const float* beta;
    const _Float16* rhi; const _Float16* rlo;
    _Float16* ohi; _Float16* olo; float* of32;
    int B, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, relu, K, tiles_co, tiles_pix;
};

// 8 waves as 4 (co) x 2 (pixels), wave tile 32 x 64 (two MFMA tiles, hh + xx = 64 accumulator registers): <= 128 VGPRs,
// so two 80 KiB workgroups = 16 waves per CU = 4 per SIMD cover each other's LDS / barrier / memory stalls.
__global__ __launch_bounds__(512, 4) void conv_split_kernel(const ConvSplitArgs a)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // wm 0..3, wn 0..1
    const int q = xcd_chunked_tile(blockIdx.x, a.tiles_co * a.tiles_pix);
    if (q < 0) return;
    const int i0 = (q % a.tiles_co) * SBM, j0 = (q / a.tiles_co) * SBN;  // co fastest: neighbours share the pixel tile
    constexpr int P_AHI = 0, P_ALO = SPLANE, P_BHI = 2 * SPLANE, P_BLO = 3 * SPLANE;
    const int OHW = a.OH * a.OW;

    // staging: 4 chunks per thread: weight row r0 and pixel row r0 (both planes), k-chunk kc
    const int r0 = tid >> 2, kc = tid & 3;  // r0 0..127
    int pb, py, px;
    {
        const int pix = j0 + r0;
        const int b = pix / OHW, rem = pix - b * OHW;
        pb = b * a.H;
        py = (rem / a.OW) * a.stride - a.pad;
        px = (rem % a.OW) * a.stride - a.pad;
    }
    f16x8 rw[2], rx[2];
    auto gload = [&](int k0) {
        const size_t wo = (size_t)(i0 + r0) * a.K + k0 + kc * 8;
        rw[0] = *reinterpret_cast<const f16x8*>(a.whi + wo);
        rw[1] = *reinterpret_cast<const f16x8*>(a.wlo + wo);
        const int tap = k0 / a.Cin, ci = k0 - tap * a.Cin + kc * 8;  // Cin % 32 == 0: one tap per 32-k slab
        const int dy = tap / a.KW, dx = tap - dy * a.KW;
        const int iy = py + dy, ix = px + dx;
        f16x8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) { vh[e] = (_Float16)0.f; vl[e] = (_Float16)0.f; }
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
            const size_t xo = ((size_t)(pb + iy) * a.W + ix) * a.Cin + ci;
            vh = *reinterpret_cast<const f16x8*>(a.xhi + xo);
            vl = *reinterpret_cast<const f16x8*>(a.xlo + xo);
        }
        rx[0] = vh;
        rx[1] = vl;
    };
    auto stage = [&](int buf) {
        _Float16* L = lds + buf * SBUF;
        const int off = lds_off(r0, kc);
        *reinterpret_cast<f16x8*>(L + P_AHI + off) = rw[0];
        *reinterpret_cast<f16x8*>(L + P_ALO + off) = rw[1];
        *reinterpret_cast<f16x8*>(L + P_BHI + off) = rx[0];
        *reinterpret_cast<f16x8*>(L + P_BLO + off) = rx[1];
    };

    f32x16 hh[1][2], xx[1][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hh[0][ni][r] = 0.f; xx[0][ni][r] = 0.f; }

    const int nstep = a.K / SBK;
    gload(0);
    stage(0);
    if (nstep > 1) gload(SBK);  // slab 1 waits in registers for the first loop iteration
    __syncthreads();
    const int ar_ = wm * 32 + (lane & 31), br_ = wn * 64 + (lane & 31), kh_ = lane >> 5;
    const int arow = ar_ * SROW, brow = br_ * SROW;
    const int akx[2] = {((kh_ ^ ((ar_ >> 2) & 3)) << 3), (((kh_ + 2) ^ ((ar_ >> 2) & 3)) << 3)};
    const int bkx[2] = {((kh_ ^ ((br_ >> 2) & 3)) << 3), (((kh_ + 2) ^ ((br_ >> 2) & 3)) << 3)};
    for (int s = 0; s < nstep; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstep) gload((s + 1) * SBK);
        const _Float16* L = lds + buf * SBUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[1], al[1], bh[2], bl[2];
            ah[0] = *reinterpret_cast<const f16x8*>(L + P_AHI + arow + akx[ks]);
            al[0] = *reinterpret_cast<const f16x8*>(L + P_ALO + arow + akx[ks]);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                bh[ni] = *reinterpret_cast<const f16x8*>(L + P_BHI + brow + ni * 32 * SROW + bkx[ks]);
                bl[ni] = *reinterpret_cast<const f16x8*>(L + P_BLO + brow + ni * 32 * SROW + bkx[ks]);
            }
#pragma unroll
            for (int mi = 0; mi < 1; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    hh[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], hh[mi][ni], 0, 0, 0);
                    xx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], xx[mi][ni], 0, 0, 0);
                    xx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], xx[mi][ni], 0, 0, 0);
                }
        }
        if (s + 1 < nstep) stage(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane = pixel column, registers = 4 x (4 consecutive output channels)
#pragma unroll
    for (int mi = 0; mi < 1; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int pix = j0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = i0 + wm * 32 + 8 * g + 4 * (lane >> 5);
                if (co >= a.Cout) continue;  // padded weight rows (Cout % 128 == 64)
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = hh[mi][ni][4 * g + e] + xx[mi][ni][4 * g + e] * kLoInv;
                    if (a.alpha) v[e] = v[e] * a.alpha[co + e] + a.beta[co + e];
                }
                if (a.rhi) {
                    const f16x4 rh = *reinterpret_cast<const f16x4*>(a.rhi + (size_t)pix * a.Cout + co);
                    const f16x4 rl = *reinterpret_cast<const f16x4*>(a.rlo + (size_t)pix * a.Cout + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ((float)rh[e] + (float)rl[e] * kLoInv) + v[e];
                }
                if (a.relu)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                if (a.of32) {  // (B, Cout, OH, OW)
                    const int b = pix / OHW, rem = pix - b * OHW;
#pragma unroll
                    for (int e = 0; e < 4; ++e) a.of32[((size_t)b * a.Cout + co + e) * OHW + rem] = v[e];
                } else {
                    f16x4 oh, ol;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 h_, l_;
                        split1(v[e], h_, l_);
                        oh[e] = h_;
                        ol[e] = l_;
                    }
                    *reinterpret_cast<f16x4*>(a.ohi + (size_t)pix * a.Cout + co) = oh;
                    *reinterpret_cast<f16x4*>(a.olo + (size_t)pix * a.Cout + co) = ol;
                }
            }
        }
}

// internal entry (gp_vit.hip).  act: f32 k-major activations [K][ld_act]; whi/wlo: pre-split weights [n_w][K].
// act_is_b: D[i][j] = sum_k W[i][k] X[k][j]  (weights index i); else D[i][j] = sum_k X[k][i] W[j][k].
int gp_gemm_split_launch_limited(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J,
                                 int K, int act_is_b, int epilogue, const float* bias, const float* scale, const float* res, int ldr,
                                 const int* j_limit, hipStream_t st);

int gp_gemm_split_launch(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J,
                         int K, int act_is_b, int epilogue, const float* bias, const float* scale, const float* res, int ldr,
                         hipStream_t st)
{
    return gp_gemm_split_launch_limited(act, ld_act, whi, wlo, D, ldd, I, J, K, act_is_b, epilogue, bias, scale, res, ldr, nullptr, st);
}

// j_limit (device int, may be null): only columns j < *j_limit carry data -- whole tiles beyond it exit at once
int gp_gemm_split_launch_limited(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J,
                                 int K, int act_is_b, int epilogue, const float* bias, const float* scale, const float* res, int ldr,
                                 const int* j_limit, hipStream_t st)
{
    GP_REQUIRE(I > 0 && J > 0 && K > 0 && I % SBM == 0 && J % SBN == 0 && K % SBK == 0,
               "gp_gemm_split: I=%d, J=%d must be multiples of 128 and K=%d of 32", I, J, K);
    GP_REQUIRE(act && whi && wlo && D && ld_act % 4 == 0 && ((uintptr_t)act % 16 == 0) && ((uintptr_t)whi % 16 == 0) &&
                   ((uintptr_t)wlo % 16 == 0),
               "gp_gemm_split: null / misaligned operand");
    GP_REQUIRE((long long)I * ldd < (1ll << 31) && (long long)I * (ldr > 0 ? ldr : 1) < (1ll << 31), "gp_gemm_split: output too large");
    SplitArgs a{act, ld_act, (const _Float16*)whi, (const _Float16*)wlo, D, ldd, K, bias, scale, res, ldr, I / SBM, J / SBN, 8, j_limit};
    // algorithmic flops (the kernel executes 3x as f16 MFMAs).  Column-limited launches (the IST regressor's compacted rows) compute
    // an unknown fraction of their J columns (the count lives on the device): they are timed as "other", with no work figure, so
    // that the TFLOP/s of the gemm_split family -- the roofline's kernel -- is not overstated by work that was skipped.
    GpProfScope prof(j_limit ? GP_PROF_OTHER : GP_PROF_GEMM_SPLIT, j_limit ? 0.0 : 2.0 * I * J * K, st);
    switch (epilogue) {
        case SEPI_NONE: launch_split<SEPI_NONE>(a, act_is_b != 0, st); break;
        case SEPI_BIAS_I: launch_split<SEPI_BIAS_I>(a, act_is_b != 0, st); break;
        case SEPI_BIAS_I_GELU: launch_split<SEPI_BIAS_I_GELU>(a, act_is_b != 0, st); break;
        case SEPI_BIAS_I_SCALE_RES: launch_split<SEPI_BIAS_I_SCALE_RES>(a, act_is_b != 0, st); break;
        case SEPI_BIAS_J: launch_split<SEPI_BIAS_J>(a, act_is_b != 0, st); break;
        case SEPI_BIAS_I_RELU: launch_split<SEPI_BIAS_I_RELU>(a, act_is_b != 0, st); break;
        default: GP_REQUIRE(false, "gp_gemm_split: unknown epilogue %d", epilogue);
    }
    GP_CHECK_LAUNCH("gp_gemm_split");
    return GP_OK;
}

extern "C" {

int gp_split_weights(const float* Wt, int K, int n, int ldw, void* hi, void* lo, void* stream)
{
    GP_REQUIRE(Wt && hi && lo && K > 0 && n > 0 && ldw >= n, "gp_split_weights: bad arguments");
    hipLaunchKernelGGL(split_weights_kernel, dim3((n + 31) / 32, (K + 31) / 32), dim3(256), 0, (hipStream_t)stream, Wt, K, n,
                       ldw, (_Float16*)hi, (_Float16*)lo);
    GP_CHECK_LAUNCH("gp_split_weights");
    return GP_OK;
}

int gp_conv2d_nhwc_split(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* alpha,
                         const float* beta, const void* res_hi, const void* res_lo, int B, int H, int W, int Cin, int Cout,
                         int KH, int KW, int stride, int pad, int relu, void* out_hi, void* out_lo, float* out_f32_nchw,
                         void* stream)
{
    GP_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
               "gp_conv2d_nhwc_split: bad sizes");
    if (B == 0) return GP_OK;
    ConvSplitArgs a;
    a.xhi = (const _Float16*)x_hi; a.xlo = (const _Float16*)x_lo; a.whi = (const _Float16*)w_hi; a.wlo = (const _Float16*)w_lo;
    a.alpha = alpha; a.beta = beta; a.rhi = (const _Float16*)res_hi; a.rlo = (const _Float16*)res_lo;
    a.ohi = (_Float16*)out_hi; a.olo = (_Float16*)out_lo; a.of32 = out_f32_nchw;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.relu = relu;
    a.OH = (H + 2 * pad - KH) / stride + 1;
    a.OW = (W + 2 * pad - KW) / stride + 1;
    a.K = KH * KW * Cin;
    const long long npix = (long long)B * a.OH * a.OW;
    GP_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0, "gp_conv2d_nhwc_split: Cin=%d must be a multiple of 32, Cout=%d of 64", Cin, Cout);
    GP_REQUIRE(npix % SBN == 0 && npix < (1ll << 31), "gp_conv2d_nhwc_split: B*OH*OW=%lld must be a multiple of 128", npix);
    GP_REQUIRE(x_hi && x_lo && w_hi && w_lo && ((out_hi && out_lo) || out_f32_nchw), "gp_conv2d_nhwc_split: null pointer");
    GP_REQUIRE((alpha == nullptr) == (beta == nullptr) && (res_hi == nullptr) == (res_lo == nullptr),
               "gp_conv2d_nhwc_split: alpha/beta and res_hi/res_lo go together");
    a.tiles_co = (Cout + SBM - 1) / SBM;
    a.tiles_pix = (int)(npix / SBN);
    GpProfScope prof(GP_PROF_CONV, 2.0 * Cout * (double)npix * a.K, (hipStream_t)stream);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              SLDS_BYTES);
    hipLaunchKernelGGL(conv_split_kernel, dim3(xcd_chunked_grid(a.tiles_co * a.tiles_pix)), dim3(512), SLDS_BYTES,
                       (hipStream_t)stream, a);
    GP_CHECK_LAUNCH("gp_conv2d_nhwc_split");
    return GP_OK;
}

#ifdef GP_PROBES
/* probe: the fc-shaped GEMM (act_is_b, no epilogue) with per-phase cycle counters of one mid-grid block;
 * out20 (host): [wave 0..3][gload issue, LDS-read+MFMA issue, MFMA drain, convert+LDS-write, barrier] */
int gp_gemm_split_set_trace(unsigned long long* dev_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_split_trace), &dev_buf, sizeof(dev_buf)) == hipSuccess ? GP_OK : GP_ELAUNCH;
}

int gp_gemm_split_timing(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J, int K,
                         unsigned long long* out20 /* 23 entries */, void* stream)
{
    GP_REQUIRE(I % SBM == 0 && J % SBN == 0 && K % SBK == 0 && out20, "gp_gemm_split_timing: bad arguments");
    SplitArgs a{act, ld_act, (const _Float16*)whi, (const _Float16*)wlo, D, ldd, K, nullptr, nullptr, nullptr, 0, I / SBM, J / SBN, 8};
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<SEPI_NONE, true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, SLDS_BYTES);
    hipLaunchKernelGGL((gemm_split_kernel<SEPI_NONE, true, true>), dim3(xcd_chunked_grid(a.tiles_i * a.tiles_j)), dim3(SNT),
                       SLDS_BYTES, (hipStream_t)stream, a);
    GP_CHECK_LAUNCH("gp_gemm_split_timing");
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return GP_ELAUNCH;
    if (hipMemcpyFromSymbol(out20, HIP_SYMBOL(g_split_clk), 20 * sizeof(unsigned long long)) != hipSuccess) return GP_ELAUNCH;
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm_split_kernel<SEPI_NONE, true, true>, SNT, SLDS_BYTES);
    unsigned long long w[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(w, HIP_SYMBOL(g_split_wall), sizeof(w)) != hipSuccess) return GP_ELAUNCH;
    out20[20] = w[0]; out20[21] = w[1]; out20[22] = (unsigned long long)occ;
    return GP_OK;
}
#endif

int gp_gemm_split(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J, int K,
                  int act_is_b, int epilogue, const float* bias, const float* scale, const float* residual, int ldr,
                  void* stream)
{
    return gp_gemm_split_launch(act, ld_act, whi, wlo, D, ldd, I, J, K, act_is_b, epilogue, bias, scale, residual, ldr,
                                (hipStream_t)stream);
}

}  // extern "C"
