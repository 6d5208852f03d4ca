// IST backbone convolutions in split numerics, second generation: the plane x plane GEMM of gp_split256.hip turned into an
// implicit GEMM.  (First generation: gp_split.hip's conv_split_kernel -- 128 x 128 tiles, two accumulators, lock-step
// staging, 8-byte scattered epilogue accesses: 213-223 TF-eq = 26 % of the f16 peak with the plane GEMM at 42 %.)
//
//   Y[pix][co] = epi( 2^-9 * sum_k X8[pix][k] * W64[co][k] ),   k = (dy, dx, ci), ci fastest  (reference resnet.py:26-50, 364-381)
//
//   * activations: channel-LAST f16 planes hi / lo of shape (B, H, W, C) in the single-accumulator convention of
//     gp_split256.hip -- hi = f16(8 x), lo = f16(8 x - hi), |x| < 8190 (guarded) -- weights (Cout, K) planes of 64 w; one
//     16-byte chunk = 8 consecutive channels of one tap = the MFMA operand fragment, so the im2col gather is a chunk copy
//     whose per-thread offset changes once per k-step (one tap and 32 channels per step; taps outside the image read as
//     zeros through the buffer descriptor's range check);
//   * tile 256 pixels (i) x 64 NI output channels (j), NI = 2, 3, 4 for Cout = 128, 192, >= 256: 8 waves as 4 x 2, wave
//     tile 64 x 32 NI, ONE accumulator, k-step 32, two LDS buffers, the two wave groups half a k-step apart with one
//     barrier per step (the loop of gemm_planes256_kernel);
//     (a 512 x 128 tile with the waves as 8 x 1 -- the plane GEMM's 48 matrix instructions per wave and k-step for
//     Cout = 128, all 160 KiB of LDS -- was built and measured: bit-identical, no faster: 2.76 us per k-step against 2 x
//     1.33, the loop is bound by the global -> LDS staging rate per CU, not by LDS reads; DESIGN.md, negative results)
//   * work distribution of gemm_planes256_kernel: data-parallel rounds of whole tiles per XCD chunk, a stream-K remainder
//     with deterministic accumulator hand-overs when the tile count does not fill the slots evenly (layer4 at B = 64:
//     128 tiles on 256 slots -> every tile is cut in two k halves);
//   * epilogue through LDS (per wave, 32 pixel rows at a time): eval-BatchNorm alpha / beta, residual (planes), ReLU,
//     output planes (npix, Cout) with 8 contiguous bytes per lane and 256-byte row segments per 32 lanes -- or f32 NCHW for
//     the last layer.
#include "gp_common.h"

typedef _Float16 c16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 c16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int cu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int cu32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr float kActScale = 8.0f, kOutScale = 1.0f / (8.0f * 64.0f);

// Thin plane arithmetic of the tile epilogues (round 4, as gp_split256.hip's): the same bits with fewer vector instructions.
//   cv_sum_planes: (float)h + (float)l of a residual's two planes -- exact in f32 -- as ONE v_fma_mix_f32 reading both f16 halves in place;
//   cv_hi_pair / cv_lo_pair: a pair's hi plane by v_cvt_pk_f16_f32, its lo plane f16(v - hi) by v_fma_mixlo / mixhi_f16 (v - hi is exact);
//   cv_absmax: v_maximum3_f32 over |v| (propagates NaN) as the range guard.
template <int HALF>
__device__ __forceinline__ float cv_sum_planes(unsigned h, unsigned l)
{
    float r;
    if (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
    return r;
}
__device__ __forceinline__ unsigned cv_hi_pair(float v0, float v1)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 h;
    h[0] = (_Float16)v0;
    h[1] = (_Float16)v1;
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ unsigned cv_lo_pair(float v0, float v1, unsigned hi)
{
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(lo)
        : "v"(v0), "v"(v1), "v"(hi));
    return lo;
}
__device__ __forceinline__ float cv_absmax(float m, float v0, float v1)
{
    return __builtin_elementwise_maximum(m, __builtin_elementwise_maximum(__builtin_fabsf(v0), __builtin_fabsf(v1)));
}
constexpr int CT = 256, CBK = 32, CNT = 512;
constexpr int CROW = 32, CPLANE = CT * CROW;  // halfs
constexpr int CBUF = 4 * CPLANE;              // A hi, A lo, B hi, B lo (B uses 64 NI of its 256 rows)
constexpr int kSlots = 256, kErrWord = 1025;
constexpr size_t kHeaderBytes = 8192;
constexpr size_t kFragFloats = (size_t)CT * CT;
constexpr int kSpin = 400000;
constexpr unsigned kOob = 0x80000000u;        // buffer offset beyond every plane: the load returns zeros

struct ConvPArgs {
    const _Float16* xhi; const _Float16* xlo;   // (B, H, W, Cin) x 8
    const _Float16* whi; const _Float16* wlo;   // (Cout, K) x 64
    const float* alpha; const float* beta;      // folded BatchNorm (Cout) or null
    const _Float16* rhi; const _Float16* rlo;   // residual planes (npix, Cout) x 8 or null
    _Float16* ohi; _Float16* olo; float* of32;  // output planes (npix, Cout) x 8, or f32 (B, Cout, OH, OW)
    int B, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, relu, K;
    int tiles_i, tiles_j;
    int* flags; float* partial; int epoch; int* status;
    int par_cap;  // conv_halo_kernel<.., true>: most slots a tile is split over
    int stem;  // 1: the 7 x 7 / stride 2 stem on a zero-framed 4-channel image, see gp_conv2d_stem_planes
    unsigned long long* trace;  // probe (gp_conv2d_planes_set_trace): per slot 8 words -- segments, k-steps, 100 MHz ticks in prologue / k loop / tail
};

__device__ __forceinline__ int toff(int row, int kc) { return row * CROW + ((kc ^ ((row >> 2) & 3)) << 3); }

template <int NI>
__global__ __launch_bounds__(CNT, 2) void conv_planes_kernel(const ConvPArgs a)
{
    constexpr int MT = CT;                   // pixel rows per tile
    constexpr int BR = CT;                   // weight rows of an LDS buffer (64 NI of them used)
    constexpr int NIW = NI;                  // 32-channel matrix tiles per wave
    constexpr int AH = MT / 128;             // pixel rows each thread stages per k-step (rows srow + 128 h)
    constexpr int BH = NI > 2 ? 2 : 1;       // weight rows each thread stages
    constexpr int P_AHI = 0, P_ALO = MT * CROW, P_BHI = 2 * MT * CROW, P_BLO = P_BHI + BR * CROW;
    constexpr int TBUF = 2 * (MT + BR) * CROW;  // halfs per LDS buffer: 64 KiB
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * TBUF];  // 128 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, grp = wave >> 2;
    constexpr int JT = 64 * NI;  // output channels per tile

    // ---- this slot's range of (tile, k-step) units inside its XCD's tile chunk (gemm_planes256_kernel's scheme)
    const int p = blockIdx.x, x = p & 7, n = p >> 3, slots_x = gridDim.x >> 3;
    const int T = a.tiles_i * a.tiles_j;
    const int t_lo = (int)((long long)T * x / 8), n_t = (int)((long long)T * (x + 1) / 8) - t_lo;
    const int nstep = a.K / CBK;
    const int rounds_dp = (n_t / slots_x > 1) ? n_t / slots_x - 1 : 0;
    const int n_dp = rounds_dp * slots_x;
    const long long U = (long long)(n_t - n_dp) * nstep;
    const long long u0 = U * n / slots_x, u1 = U * (n + 1) / slots_x;
    const int ta = (int)(u0 / nstep), sa = (int)(u0 % nstep);
    const int tb = (int)(u1 / nstep), sb = (int)(u1 % nstep);
    const int n_head = sb > 0 ? 1 : 0, n_rest = sa > 0 ? 1 : 0;
    const int first_whole = ta + n_rest;
    const int n_seg = rounds_dp + n_head + (tb - first_whole) + n_rest;

    const unsigned x_bytes = (unsigned)a.B * a.H * a.W * a.Cin * 2u, w_bytes = (unsigned)a.Cout * a.K * 2u;
    const __amdgpu_buffer_rsrc_t r_xhi = __builtin_amdgcn_make_buffer_rsrc((void*)a.xhi, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_xlo = __builtin_amdgcn_make_buffer_rsrc((void*)a.xlo, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_whi = __builtin_amdgcn_make_buffer_rsrc((void*)a.whi, 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_wlo = __builtin_amdgcn_make_buffer_rsrc((void*)a.wlo, 0, w_bytes, 0x00020000);
    const unsigned r_bytes = a.rhi ? (unsigned)a.B * a.OH * a.OW * a.Cout * 2u : 0u;
    const __amdgpu_buffer_rsrc_t r_rhi = __builtin_amdgcn_make_buffer_rsrc((void*)a.rhi, 0, r_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rlo = __builtin_amdgcn_make_buffer_rsrc((void*)a.rlo, 0, r_bytes, 0x00020000);
    const int srow = tid >> 2, schunk = tid & 3;
    const int wofs = toff(srow, schunk);
    const int OHW = a.OH * a.OW, cps = a.stem ? 1 : a.Cin / CBK;  // k-steps per tap
    cu32x4 ra_h[AH], ra_l[AH], rb_h[BH], rb_l[BH];
    // fragment addressing
    const int ar_ = 64 * wr + (lane & 31), br_ = 32 * NIW * wc + (lane & 31), kh_ = lane >> 5;
    const int arow = ar_ * CROW, brow = br_ * CROW;
    const int ak0 = ((kh_ ^ ((ar_ >> 2) & 3)) << 3), ak1 = (((kh_ + 2) ^ ((ar_ >> 2) & 3)) << 3);
    const int bk0 = ((kh_ ^ ((br_ >> 2) & 3)) << 3), bk1 = (((kh_ + 2) ^ ((br_ >> 2) & 3)) << 3);

    unsigned long long tr_pro = 0, tr_loop = 0, tr_tail = 0, tr_steps = 0, tr_t0 = a.trace ? wall_clock64() : 0;
    const unsigned long long tr_begin = tr_t0;
    for (int seg = 0; seg < n_seg; ++seg) {
        const bool is_dp = seg < rounds_dp;
        const bool is_head = !is_dp && seg - rounds_dp < n_head;
        const bool is_rest = !is_dp && n_rest && seg == n_seg - 1;
        const int t = is_dp ? seg * slots_x + n : n_dp + (is_head ? tb : (is_rest ? ta : first_whole + seg - rounds_dp - n_head));
        const int s0 = is_rest ? sa : 0, s1 = is_head ? sb : nstep;
        const int q = t_lo + t;
        const int i0 = (q / a.tiles_j) * MT, j0 = (q % a.tiles_j) * JT;  // j fastest: the co tiles of one pixel tile are neighbours

        // ---- gather state of this thread's pixel rows (srow + 128 h): byte offset of (b, oy*stride-pad, ox*stride-pad, 0)
        // and one validity bit per tap (KH*KW <= 9)
        int pbase[AH];
        unsigned pvalid[AH];
#pragma unroll
        for (int h = 0; h < AH; ++h) {
            const int pix = i0 + srow + 128 * h;
            const int b = pix / OHW, rem = pix - b * OHW;
            const int oy = rem / a.OW, ox = rem - oy * a.OW;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            pbase[h] = (((b * a.H + iy0) * a.W + ix0) * a.Cin) * 2 + schunk * 16;
            unsigned m = 0;
            if (a.stem) {  // the frame of zeros is part of the image buffer (pad = 0 in its coordinates): every tap row is in range
                m = 0xffffffffu;
            } else {
                for (int dy = 0; dy < a.KH; ++dy)
                    for (int dx = 0; dx < a.KW; ++dx)
                        if (iy0 + dy >= 0 && iy0 + dy < a.H && ix0 + dx >= 0 && ix0 + dx < a.W) m |= 1u << (dy * a.KW + dx);
            }
            pvalid[h] = m;
        }
        // weights: rows j0 + srow (+128) of (Cout, K); rows past Cout (Cout = 192 in a 256-row plane) read as zeros
        unsigned wvoff[BH];
#pragma unroll
        for (int h = 0; h < BH; ++h) {
            const int co = j0 + srow + 128 * h;
            wvoff[h] = (srow + 128 * h < JT && co < a.Cout) ? (unsigned)co * (unsigned)a.K * 2u + (unsigned)schunk * 16u : kOob;
        }

        f32x16 acc[2][NIW];
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
                for (int ni = 0; ni < NIW; ++ni)
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
                for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        }

        // ---- k loop over steps [s0, s1)
        const int ns = s1 - s0;
        auto gload = [&](int slab) {
            const int s = s0 + slab;
            int tap, toffb;  // wave-uniform
            if (a.stem) {    // one k-step = one kernel ROW: 8 taps x 4 channels = 64 contiguous bytes of the framed image
                tap = 0;
                toffb = s * a.W * a.Cin * 2;
            } else {         // one tap, 32 channels per k-step
                tap = s / cps;
                const int c0 = s - tap * cps, dy = tap / a.KW, dx = tap - dy * a.KW;
                toffb = ((dy * a.W + dx) * a.Cin + c0 * CBK) * 2;
            }
            const unsigned sw = (unsigned)s * (CBK * 2u);
#pragma unroll
            for (int h = 0; h < AH; ++h) {
                const unsigned v = ((pvalid[h] >> tap) & 1u) ? (unsigned)(pbase[h] + toffb) : kOob;
                ra_h[h] = __builtin_amdgcn_raw_buffer_load_b128(r_xhi, v, 0, 0);
                ra_l[h] = __builtin_amdgcn_raw_buffer_load_b128(r_xlo, v, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < BH; ++h) {
                rb_h[h] = __builtin_amdgcn_raw_buffer_load_b128(r_whi, wvoff[h], sw, 0);
                rb_l[h] = __builtin_amdgcn_raw_buffer_load_b128(r_wlo, wvoff[h], sw, 0);
            }
        };
        auto stage = [&](int buf) {
            _Float16* L = lds + buf * TBUF + wofs;
#pragma unroll
            for (int h = 0; h < AH; ++h) {
                *reinterpret_cast<cu32x4*>(L + P_AHI + 128 * h * CROW) = ra_h[h];
                *reinterpret_cast<cu32x4*>(L + P_ALO + 128 * h * CROW) = ra_l[h];
            }
#pragma unroll
            for (int h = 0; h < BH; ++h) {
                *reinterpret_cast<cu32x4*>(L + P_BHI + 128 * h * CROW) = rb_h[h];
                *reinterpret_cast<cu32x4*>(L + P_BLO + 128 * h * CROW) = rb_l[h];
            }
        };
        gload(0);
        stage(0);
        __builtin_amdgcn_sched_barrier(0);
        if (ns > 1) gload(1);
        __syncthreads();
        if (a.trace) { const unsigned long long t = wall_clock64(); tr_pro += t - tr_t0; tr_t0 = t; tr_steps += ns; }

#define C_MFMA(A_, B_, mi, ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[mi], B_[ni], acc[mi][ni], 0, 0, 0)
        auto c_phase = [&](int s) __attribute__((always_inline)) {
            const _Float16* L = lds + (s & 1) * TBUF;
            __builtin_amdgcn_s_setprio(1);
            c16x8 ah[2], al[2], bh[NIW], bl[NIW], ch[2], cl[2], dh[NIW], dl[NIW];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ah[mi] = *reinterpret_cast<const c16x8*>(L + P_AHI + arow + mi * 32 * CROW + ak0);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) bh[ni] = *reinterpret_cast<const c16x8*>(L + P_BHI + brow + ni * 32 * CROW + bk0);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) bl[ni] = *reinterpret_cast<const c16x8*>(L + P_BLO + brow + ni * 32 * CROW + bk0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) al[mi] = *reinterpret_cast<const c16x8*>(L + P_ALO + arow + mi * 32 * CROW + ak0);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(ah, bh, 0, ni); C_MFMA(ah, bh, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(ah, bl, 0, ni); C_MFMA(ah, bl, 1, ni); }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ch[mi] = *reinterpret_cast<const c16x8*>(L + P_AHI + arow + mi * 32 * CROW + ak1);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) dh[ni] = *reinterpret_cast<const c16x8*>(L + P_BHI + brow + ni * 32 * CROW + bk1);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(al, bh, 0, ni); C_MFMA(al, bh, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) dl[ni] = *reinterpret_cast<const c16x8*>(L + P_BLO + brow + ni * 32 * CROW + bk1);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) cl[mi] = *reinterpret_cast<const c16x8*>(L + P_ALO + arow + mi * 32 * CROW + ak1);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(ch, dh, 0, ni); C_MFMA(ch, dh, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(ch, dl, 0, ni); C_MFMA(ch, dl, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(cl, dh, 0, ni); C_MFMA(cl, dh, 1, ni); }
            __builtin_amdgcn_s_setprio(0);
        };
        auto m_phase = [&](int slab) __attribute__((always_inline)) {  // stages `slab`, loads slab + 1
            if (slab < ns) stage(slab & 1);
            __builtin_amdgcn_sched_barrier(0);
            gload(min(slab + 1, ns - 1));
        };
        //     waves 0-3:  C0 M1 | C1 M2 | ...          waves 4-7:  M1 C0 | M2 C1 | ...        (| = the one barrier per k-step)
        if (grp) m_phase(1);
        for (int s = 0; s < ns; ++s) {
            c_phase(s);
            if (grp && s + 1 < ns) __syncthreads();
            m_phase(s + 1 + grp);
            if (!grp && s + 1 < ns) __syncthreads();
        }
#undef C_MFMA
        if (a.trace) { const unsigned long long t = wall_clock64(); tr_loop += t - tr_t0; tr_t0 = t; }

        if (is_head) {  // publish the fragment for slot n + 1 (agent-scope release by one lane)
            f32x4* w = reinterpret_cast<f32x4*>(a.partial + (size_t)p * kFragFloats) + tid * 32;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni)
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
            // ---- epilogue through LDS: per wave 32 pixel rows x 32 NI channels of f32 at a time (16 KiB regions, wave-private)
            int tid_ = threadIdx.x;
            asm volatile("" : "+v"(tid_));
            __syncthreads();  // every wave has read its last operand fragments
            const int ln = tid_ & 63, l31 = ln & 31;
            float* wl = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + (tid_ >> 6) * 16384);
            constexpr int WC = 32 * NIW, QPR = 8 * NIW;  // f32 columns / 4-column quads per row of the wave tile
            float* const AB_STASH = wl + 32 * WC;        // NI = 3: 768 bytes behind the 12 KiB of the wave's 16 KiB region that a round uses
            static_assert(NIW != 3 || 32 * WC * 4 + 8 * QPR * 4 <= 16384, "alpha | beta stash must fit the wave's region");
            float mx = 0.f;  // range guard: largest |8 x| written (v_maximum3_f32 propagates NaN)
            // eval-BatchNorm alpha | beta of the channels a lane finishes: item `it` of a lane is quad (ln + 64 it) % QPR of its row -- ONE quad
            // for 16 / 32 quads per row (NI = 2 / 4) -- so they are loaded once per tile.  (Round 6: read inside the item loop they were two
            // 16-byte global loads + s_waitcnt vmcnt(0) PER ITEM -- the stores to the output planes may alias them as far as hipcc knows -- i.e.
            // 16 dependent round trips per tile, each also waiting for the previous item's stores.)
            // (24 quads per row, NI = 3: three quads per lane = 24 registers, which the 192-channel builds do not have -- 7 spilled when tried;
            // there the wave parks its 24 + 24 quads in LDS behind its epilogue region -- AB_STASH -- and an item reads its two from there.)
            constexpr bool kHoist = 64 % QPR == 0;
            f32x4 al_ = {1.f, 1.f, 1.f, 1.f}, be_ = {0.f, 0.f, 0.f, 0.f};
            if constexpr (kHoist) {
                const int co_l = j0 + WC * wc + 4 * (ln % QPR);
                if (a.alpha && co_l < a.Cout) {
                    al_ = *reinterpret_cast<const f32x4*>(a.alpha + co_l);
                    be_ = *reinterpret_cast<const f32x4*>(a.beta + co_l);
                }
            } else if (a.alpha && ln < 2 * QPR) {   // lanes 0 .. 23: alpha quads, 24 .. 47: beta quads
                const int qd_l = ln < QPR ? ln : ln - QPR, co_l = j0 + WC * wc + 4 * qd_l;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (co_l < a.Cout) t = *reinterpret_cast<const f32x4*>((ln < QPR ? a.alpha : a.beta) + co_l);
                *reinterpret_cast<f32x4*>(AB_STASH + 4 * ln) = t;
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) wl[frag_row(r, ln) * WC + 32 * ni + l31] = acc[mi][ni][r];
                // one unconditional use in straight-line code: hipcc then waits for alpha | beta HERE (under the LDS writes above), once -- behind the
                // items' `co < Cout` branches its wait-count pass assumes them pending at every join and emits vmcnt(0) per item, i.e. a wait for
                // the previous item's stores
                if (kHoist && mi == 0) asm volatile("" : "+v"(al_), "+v"(be_));
                // residual planes: the loads of a round of RG items up front (RG x 2 x 8 bytes per lane in flight).  Loaded inside
                // the item loop, one dependent 8-byte load per item, they cost 16 us of a 25 us tile tail (tools/probe_conv_timeline.py).
                constexpr int RG = NIW == 2 ? 8 : (NIW == 3 ? 6 : 4);  // items per round (divides 4 NIW); fewer where the accumulators leave fewer registers
#pragma unroll
                for (int it0 = 0; it0 < 4 * NIW; it0 += RG) {
                    cu32x2 rh[RG], rl[RG];
                    if (a.rhi) {
#pragma unroll
                        for (int u = 0; u < RG; ++u) {
                            const int f = (it0 + u) * 64 + ln, row = f / QPR, qd = f - row * QPR;
                            const int pix = i0 + 64 * wr + 32 * mi + row, co = j0 + WC * wc + 4 * qd;
                            const unsigned ob = co < a.Cout ? ((unsigned)pix * (unsigned)a.Cout + (unsigned)co) * 2u : kOob;  // one offset, both planes
                            rh[u] = __builtin_amdgcn_raw_buffer_load_b64(r_rhi, ob, 0, 0);
                            rl[u] = __builtin_amdgcn_raw_buffer_load_b64(r_rlo, ob, 0, 0);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RG; ++u) {
                        const int f = (it0 + u) * 64 + ln, row = f / QPR, qd = f - row * QPR;
                        const f32x4 tv = *reinterpret_cast<const f32x4*>(wl + row * WC + 4 * qd);
                        const int pix = i0 + 64 * wr + 32 * mi + row, co = j0 + WC * wc + 4 * qd;
                        if (co < a.Cout) {
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = tv[e] * kOutScale;
                            if (a.alpha) {
                                const f32x4 al4 = kHoist ? al_ : *reinterpret_cast<const f32x4*>(AB_STASH + 4 * qd), be4 = kHoist ? be_ : *reinterpret_cast<const f32x4*>(AB_STASH + 4 * (QPR + qd));
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = v[e] * al4[e] + be4[e];
                            }
                            const size_t o = (size_t)pix * a.Cout + co;
                            if (a.rhi) {  // + residual: (h + l) / 8, exact, then ONE rounding -- the bits of ((float)h + (float)l) * (1 / 8) + v
                                v[0] = __builtin_fmaf(cv_sum_planes<0>(rh[u][0], rl[u][0]), 1.0f / kActScale, v[0]);
                                v[1] = __builtin_fmaf(cv_sum_planes<1>(rh[u][0], rl[u][0]), 1.0f / kActScale, v[1]);
                                v[2] = __builtin_fmaf(cv_sum_planes<0>(rh[u][1], rl[u][1]), 1.0f / kActScale, v[2]);
                                v[3] = __builtin_fmaf(cv_sum_planes<1>(rh[u][1], rl[u][1]), 1.0f / kActScale, v[3]);
                            }
                            if (a.relu)
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                            if (a.of32) {  // (B, Cout, OH, OW): the last layer only
                                const int b = pix / OHW, rem = pix - b * OHW;
#pragma unroll
                                for (int e = 0; e < 4; ++e) a.of32[((size_t)b * a.Cout + co + e) * OHW + rem] = v[e];
                            } else {
                                const float s0 = v[0] * kActScale, s1 = v[1] * kActScale, s2 = v[2] * kActScale, s3 = v[3] * kActScale;
                                cu32x2 oh, ol;
                                oh[0] = cv_hi_pair(s0, s1);
                                oh[1] = cv_hi_pair(s2, s3);
                                ol[0] = cv_lo_pair(s0, s1, oh[0]);
                                ol[1] = cv_lo_pair(s2, s3, oh[1]);
                                mx = cv_absmax(cv_absmax(mx, s0, s1), s2, s3);
                                *reinterpret_cast<cu32x2*>(a.ohi + o) = oh;
                                *reinterpret_cast<cu32x2*>(a.olo + o) = ol;
                            }
                        }
                    }
                }
            }
            if (!(mx <= kSplitPlaneLimit)) gp_raise(a.status, GP_ST_SPLIT_RANGE_CONV);  // !(<=): NaN counts
            __syncthreads();  // LDS buffer 0 is re-staged by the next segment's prologue
        }
        if (a.trace) { const unsigned long long t = wall_clock64(); tr_tail += t - tr_t0; tr_t0 = t; }
    }
    if (a.trace && tid == 0) {
        unsigned long long* w = a.trace + (size_t)p * 8;
        w[0] = n_seg; w[1] = tr_steps; w[2] = tr_pro; w[3] = tr_loop; w[4] = tr_tail; w[5] = tr_begin; w[6] = tr_t0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3 x 3 / stride 1 / pad 1 convolutions with the activation HALO resident in LDS (round 3).
//
// conv_planes_kernel gathers its A operand once per k-step (one tap x 32 channels): every input pixel is fetched nine
// times per output tile and its k loop runs at the round trip of loads issued one step ahead -- 1.3-1.5 us per step
// against 0.95 us of matrix work at Cout = 128 (profiles/r02_conv_timeline.txt).  Here a tile is a 16 x 16 BLOCK of
// output pixels and the k order is channel-block major: for each block of 32 input channels the 18 x 18 pixel halo of
// the tile (324 rows of 64 bytes per plane, zeros outside the image through the descriptor's range check) is staged
// ONCE and the nine taps are nine k-steps whose A fragments are read from it at shifted rows; only the weights (16-32 KB
// per step, L2-resident) are staged per step.  The halo of the next channel block arrives during the first taps of the
// current one (two halo buffers).  Global -> LDS bytes per k-step at Cout = 128: 48 KB -> 20.6 KB.
// Same wave layout (4 x 2 waves, wave tile 64 pixels x 32 NI channels), same two-group half-step offset with one barrier per
// step, same epilogue and hand-over scheme as conv_planes_kernel; stream-K units are channel blocks (9 steps each).
// Per output the products are those of conv_planes_kernel summed in the order (channel block, tap) instead of (tap, channel
// block): results agree to f32 round-off, not bit for bit (tests: both against float64).
constexpr int HPX = 18 * 18;                 // halo pixels of a 16 x 16 output block
constexpr int HPLANE = HPX * CROW;           // halfs per halo plane (64-byte rows)
constexpr int HBUF = 2 * HPLANE;             // hi + lo

__device__ __forceinline__ int hoff(int hp, int kc) { return hp * CROW + ((kc ^ ((hp >> 2) & 3)) << 3); }

// PAR (fewer tiles than slots: layer3 / layer4 below 64 crops, layer4 always): S = floor(slots / tiles) slots per tile, each an equal
// share of the tile's channel blocks from a zero accumulator, all at once; all but the last publish their partial accumulator, the
// slot holding the last range adds them in slot order (deterministic) and runs the epilogue -- gemm_planes256_kernel's scheme.  The
// serial hand-over (PAR = false) keeps a split tile's summation order but serialises its slots: layer4's 128 tiles on 256 slots
// took as long as whole tiles.
template <int NI, bool PAR>
__global__ __launch_bounds__(CNT, 2) void conv_halo_kernel(const ConvPArgs a)
{
    constexpr int NIW = NI, JT = 64 * NI;
    constexpr int BH = NI > 2 ? 2 : 1;             // weight rows each thread stages per step (rows srow + 128 h of 64 NI)
    constexpr int WROWS = 64 * NI;
    constexpr int WPLANE = WROWS * CROW, WBUF = 2 * WPLANE;   // halfs
    constexpr int LDS_HALFS = 2 * HBUF + 2 * WBUF;            // 2 halo + 2 weight buffers
    constexpr int EPI_STRIDE = 32 * 32 * NI * 4;              // bytes of a wave's epilogue region (32 rows x 32 NI f32)
    static_assert(LDS_HALFS * 2 >= 8 * EPI_STRIDE, "the epilogue regions must fit the operand buffers");
    __shared__ __attribute__((aligned(16))) _Float16 lds[LDS_HALFS];
    _Float16* const Hb = lds;
    _Float16* const Wb = lds + 2 * HBUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, grp = wave >> 2;

    // ---- this slot's range of (tile, channel-block) units inside its XCD's tile chunk (conv_planes_kernel's scheme)
    const int p = blockIdx.x, x = p & 7, n = p >> 3, slots_x = gridDim.x >> 3;
    const int T = a.tiles_i * a.tiles_j;
    const int t_lo = (int)((long long)T * x / 8), n_t = (int)((long long)T * (x + 1) / 8) - t_lo;
    const int ncb = a.Cin / CBK;
    const int rounds_dp = (!PAR && n_t / slots_x > 1) ? n_t / slots_x - 1 : 0;
    const int n_dp = rounds_dp * slots_x;
    const long long U = (long long)(n_t - n_dp) * ncb;
    long long u0 = U * n / slots_x, u1 = U * (n + 1) / slots_x;
    const int par_S = PAR ? min(min(max(1, slots_x / max(n_t, 1)), ncb), max(1, a.par_cap)) : 1;
    if (PAR) {
        const int tile = n / par_S, part = n - tile * par_S;
        u0 = u1 = 0;
        if (tile < n_t) {
            u0 = (long long)tile * ncb + ncb * part / par_S;
            u1 = (long long)tile * ncb + ncb * (part + 1) / par_S;
        }
    }
    const int ta = (int)(u0 / ncb), sa = (int)(u0 % ncb);
    const int tb = (int)(u1 / ncb), sb = (int)(u1 % ncb);
    const int n_head = sb > 0 ? 1 : 0, n_rest = sa > 0 ? 1 : 0;
    const int first_whole = ta + n_rest;
    const int n_seg = rounds_dp + n_head + (tb - first_whole) + n_rest;

    const unsigned x_bytes = (unsigned)a.B * a.H * a.W * a.Cin * 2u, w_bytes = (unsigned)a.Cout * a.K * 2u;
    const __amdgpu_buffer_rsrc_t r_xhi = __builtin_amdgcn_make_buffer_rsrc((void*)a.xhi, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_xlo = __builtin_amdgcn_make_buffer_rsrc((void*)a.xlo, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_whi = __builtin_amdgcn_make_buffer_rsrc((void*)a.whi, 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_wlo = __builtin_amdgcn_make_buffer_rsrc((void*)a.wlo, 0, w_bytes, 0x00020000);
    const unsigned r_bytes = a.rhi ? (unsigned)a.B * a.OH * a.OW * a.Cout * 2u : 0u;
    const __amdgpu_buffer_rsrc_t r_rhi = __builtin_amdgcn_make_buffer_rsrc((void*)a.rhi, 0, r_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rlo = __builtin_amdgcn_make_buffer_rsrc((void*)a.rlo, 0, r_bytes, 0x00020000);
    const int srow = tid >> 2, schunk = tid & 3;
    const int wofs = toff(srow, schunk);
    const int nbx = a.OW >> 4, nby = a.OH >> 4, OHW = a.OH * a.OW;
    cu32x4 rb_h[BH], rb_l[BH], rh_h, rh_l;
    // fragment addressing: A rows = halo pixels of this lane's two output pixels, B rows as conv_planes_kernel
    const int br_ = 32 * NIW * wc + (lane & 31), kh_ = lane >> 5;
    const int brow = br_ * CROW;
    const int bk0 = ((kh_ ^ ((br_ >> 2) & 3)) << 3), bk1 = (((kh_ + 2) ^ ((br_ >> 2) & 3)) << 3);
    int hbase[2];  // halo index of output pixel (py, px) at tap (0, 0): py * 18 + px
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int pl = 64 * wr + 32 * mi + (lane & 31);
        hbase[mi] = (pl >> 4) * 18 + (pl & 15);
    }

    unsigned long long tr_pro = 0, tr_loop = 0, tr_tail = 0, tr_steps = 0, tr_t0 = a.trace ? wall_clock64() : 0;
    const unsigned long long tr_begin = tr_t0;
    for (int seg = 0; seg < n_seg; ++seg) {
        const bool is_dp = seg < rounds_dp;
        const bool is_head = !is_dp && seg - rounds_dp < n_head;
        const bool is_rest = !is_dp && n_rest && seg == n_seg - 1;
        const int t = is_dp ? seg * slots_x + n : n_dp + (is_head ? tb : (is_rest ? ta : first_whole + seg - rounds_dp - n_head));
        const int c0 = is_rest ? sa : 0, c1 = is_head ? sb : ncb;  // channel blocks [c0, c1)
        const int q = t_lo + t;
        const int ti = q / a.tiles_j, j0 = (q % a.tiles_j) * JT;    // j fastest: the co tiles of one pixel block are neighbours
        const int b = ti / (nby * nbx), rb_ = ti - b * (nby * nbx);
        const int by = rb_ / nbx, bx = rb_ - by * nbx;
        const int pix00 = (b * a.OH + by * 16) * a.OW + bx * 16;    // output pixel (0, 0) of the block; (py, px) adds py * OW + px

        // ---- halo items of this thread: item i = tid + 512 u (u < 3, i < 1296) = (halo pixel i >> 2, 16-byte chunk i & 3)
        unsigned hvoff[3];
        int hlds[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = tid + 512 * u, hp = i >> 2, ck = i & 3;
            const int iy = by * 16 - 1 + hp / 18, ix = bx * 16 - 1 + hp % 18;
            const bool ok = i < 4 * HPX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            hvoff[u] = ok ? (unsigned)(((b * a.H + iy) * a.W + ix) * a.Cin) * 2u + (unsigned)ck * 16u : kOob;
            hlds[u] = i < 4 * HPX ? hoff(hp, ck) : -1;
        }
        // weights: rows j0 + srow (+128) of (Cout, K); rows past Cout read as zeros
        unsigned wvoff[BH];
#pragma unroll
        for (int h = 0; h < BH; ++h) {
            const int co = j0 + srow + 128 * h;
            wvoff[h] = (srow + 128 * h < JT && co < a.Cout) ? (unsigned)co * (unsigned)a.K * 2u + (unsigned)schunk * 16u : kOob;
        }

        f32x16 acc[2][NIW];
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
            // piece u of thread tid at [u][tid]; buffer loads: one lane offset + a scalar offset per piece (no address pair per piece)
            const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void*)(a.partial + (size_t)(p - 8) * kFragFloats), 0,
                                                                                 (int)(kFragFloats * sizeof(float)), 0x00020000);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const cu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rf, (unsigned)tid * 16u, (unsigned)(((mi * 4 + ni) * 4 + r4) * CNT) * 16u, 0);
                        const f32x4 vf = __builtin_bit_cast(f32x4, v);  // whole-vector cast (element-wise __builtin_bit_cast of vector lanes miscompiles: every r4 got lane group 0)
                        acc[mi][ni][r4 * 4 + 0] = vf[0]; acc[mi][ni][r4 * 4 + 1] = vf[1];
                        acc[mi][ni][r4 * 4 + 2] = vf[2]; acc[mi][ni][r4 * 4 + 3] = vf[3];
                    }
        } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        }

        // ---- k loop: steps s = (cb - c0) * 9 + tap
        const int ns = 9 * (c1 - c0);
        auto wload = [&](int slab) {   // weights of step `slab` -> registers
            const int cb = c0 + slab / 9, tap = slab - (slab / 9) * 9;
            const unsigned sw = (unsigned)(tap * a.Cin + cb * CBK) * 2u;
#pragma unroll
            for (int h = 0; h < BH; ++h) {
                rb_h[h] = __builtin_amdgcn_raw_buffer_load_b128(r_whi, wvoff[h], sw, 0);
                rb_l[h] = __builtin_amdgcn_raw_buffer_load_b128(r_wlo, wvoff[h], sw, 0);
            }
        };
        auto wstage = [&](int buf) {
            _Float16* L = Wb + buf * WBUF + wofs;
#pragma unroll
            for (int h = 0; h < BH; ++h) {
                if (BH == 1 || h == 0 || srow + 128 * h < WROWS) {
                    *reinterpret_cast<cu32x4*>(L + 128 * h * CROW) = rb_h[h];
                    *reinterpret_cast<cu32x4*>(L + WPLANE + 128 * h * CROW) = rb_l[h];
                }
            }
        };
        auto hload = [&](int cb, int u) {   // halo item u of channel block cb -> registers
            const unsigned so = (unsigned)cb * (CBK * 2u);
            rh_h = __builtin_amdgcn_raw_buffer_load_b128(r_xhi, hvoff[u], so, 0);
            rh_l = __builtin_amdgcn_raw_buffer_load_b128(r_xlo, hvoff[u], so, 0);
        };
        auto hstage = [&](int buf, int u) {
            if (hlds[u] >= 0) {
                _Float16* L = Hb + buf * HBUF + hlds[u];
                *reinterpret_cast<cu32x4*>(L) = rh_h;
                *reinterpret_cast<cu32x4*>(L + HPLANE) = rh_l;
            }
        };
        // prologue: the whole halo of channel block c0 and the weights of step 0
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            hload(c0, u);
            hstage(c0 & 1, u);
        }
        wload(0);
        wstage(0);
        __builtin_amdgcn_sched_barrier(0);
        if (ns > 1) wload(1);
        __syncthreads();
        if (a.trace) { const unsigned long long t_ = wall_clock64(); tr_pro += t_ - tr_t0; tr_t0 = t_; tr_steps += ns; }

#define C_MFMA(A_, B_, mi, ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[mi], B_[ni], acc[mi][ni], 0, 0, 0)
        auto c_phase = [&](int s) __attribute__((always_inline)) {
            const int cbr = s / 9, tap = s - cbr * 9;                  // wave-uniform
            const int dy = tap / 3, tofs = dy * 18 + (tap - dy * 3);
            const _Float16* LH = Hb + ((c0 + cbr) & 1) * HBUF;
            const _Float16* LW = Wb + (s & 1) * WBUF;
            __builtin_amdgcn_s_setprio(1);
            int a0[2], a1[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int hp = hbase[mi] + tofs;
                a0[mi] = hoff(hp, kh_);
                a1[mi] = hoff(hp, kh_ + 2);
            }
            c16x8 ah[2], al[2], bh[NIW], bl[NIW], ch[2], cl[2], dh[NIW], dl[NIW];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ah[mi] = *reinterpret_cast<const c16x8*>(LH + a0[mi]);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) bh[ni] = *reinterpret_cast<const c16x8*>(LW + brow + ni * 32 * CROW + bk0);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) bl[ni] = *reinterpret_cast<const c16x8*>(LW + WPLANE + brow + ni * 32 * CROW + bk0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) al[mi] = *reinterpret_cast<const c16x8*>(LH + HPLANE + a0[mi]);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(ah, bh, 0, ni); C_MFMA(ah, bh, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(ah, bl, 0, ni); C_MFMA(ah, bl, 1, ni); }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ch[mi] = *reinterpret_cast<const c16x8*>(LH + a1[mi]);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) dh[ni] = *reinterpret_cast<const c16x8*>(LW + brow + ni * 32 * CROW + bk1);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(al, bh, 0, ni); C_MFMA(al, bh, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) dl[ni] = *reinterpret_cast<const c16x8*>(LW + WPLANE + brow + ni * 32 * CROW + bk1);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) cl[mi] = *reinterpret_cast<const c16x8*>(LH + HPLANE + a1[mi]);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(ch, dh, 0, ni); C_MFMA(ch, dh, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(ch, dl, 0, ni); C_MFMA(ch, dl, 1, ni); }
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) { C_MFMA(cl, dh, 0, ni); C_MFMA(cl, dh, 1, ni); }
            __builtin_amdgcn_s_setprio(0);
        };
        // m_phase(slab): stages the weights of `slab` (loaded one phase earlier) and requests those of slab + 1; the halo of the NEXT
        // channel block rides on the phases of taps 1..3 of the current one: item u requested in the phase of tap u + 1 (after the
        // weights: vector-memory results return in order) and written to LDS in the phase of tap u + 2.  The halo buffer it goes
        // to was last read in the previous channel block, whose final barrier every wave has passed.
        auto m_phase = [&](int slab) __attribute__((always_inline)) {
            const int cbr = slab / 9, tap = slab - cbr * 9;            // of the step being staged (wave-uniform)
            const bool more = c0 + cbr + 1 < c1;                       // is there a next channel block to prefetch
            if (slab < ns) {
                if (more && tap >= 2 && tap <= 4) hstage((c0 + cbr + 1) & 1, tap - 2);
                wstage(slab & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            wload(min(slab + 1, ns - 1));
            if (slab < ns && more && tap >= 1 && tap <= 3) hload(c0 + cbr + 1, tap - 1);
        };
        //     waves 0-3:  C0 M1 | C1 M2 | ...          waves 4-7:  M1 C0 | M2 C1 | ...        (| = the one barrier per k-step)
        if (grp) m_phase(1);
        for (int s = 0; s < ns; ++s) {
            c_phase(s);
            if (grp && s + 1 < ns) __syncthreads();
            m_phase(s + 1 + grp);
            if (!grp && s + 1 < ns) __syncthreads();
        }
#undef C_MFMA
        if (a.trace) { const unsigned long long t_ = wall_clock64(); tr_loop += t_ - tr_t0; tr_t0 = t_; }

        if (is_head) {  // publish the fragment for slot n + 1 (write-through stores, then the flag)
            const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void*)(a.partial + (size_t)p * kFragFloats), 0,
                                                                                 (int)(kFragFloats * sizeof(float)), 0x00020000);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        f32x4 vf;
                        vf[0] = acc[mi][ni][r4 * 4 + 0]; vf[1] = acc[mi][ni][r4 * 4 + 1];
                        vf[2] = acc[mi][ni][r4 * 4 + 2]; vf[3] = acc[mi][ni][r4 * 4 + 3];
                        const cu32x4 v = __builtin_bit_cast(cu32x4, vf);
                        __builtin_amdgcn_raw_buffer_store_b128(v, rf, (unsigned)tid * 16u, (unsigned)(((mi * 4 + ni) * 4 + r4) * CNT) * 16u, 0);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.flags + p, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();  // the next segment's prologue rewrites the operand buffers
        } else {
            if (PAR && is_rest) {  // the owner of the tile: add the partial accumulators of the par_S - 1 slots before it, nearest first
                const int m_lo = n - (par_S - 1);
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
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NIW; ++ni) {
                            cu32x4 v[4];
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4)
                                v[r4] = __builtin_amdgcn_raw_buffer_load_b128(rf, (unsigned)tid * 16u, (unsigned)(((mi * 4 + ni) * 4 + r4) * CNT) * 16u, 0);
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4) {
                                const f32x4 vf = __builtin_bit_cast(f32x4, v[r4]);
                                acc[mi][ni][r4 * 4 + 0] += vf[0]; acc[mi][ni][r4 * 4 + 1] += vf[1];
                                acc[mi][ni][r4 * 4 + 2] += vf[2]; acc[mi][ni][r4 * 4 + 3] += vf[3];
                            }
                        }
                }
            }
            // ---- epilogue through LDS (conv_planes_kernel's, with the block's pixel order)
            int tid_ = threadIdx.x;
            asm volatile("" : "+v"(tid_));
            __syncthreads();  // every wave has read its last operand fragments
            const int ln = tid_ & 63, l31 = ln & 31;
            constexpr int EPI_PITCH = EPI_STRIDE + (NIW == 3 ? 1024 : 0);   // NI = 3: + the wave's alpha | beta stash (a constant offset from wl: no register)
            static_assert(LDS_HALFS * 2 >= 8 * EPI_PITCH, "the epilogue regions (+ stashes) must fit the operand buffers");
            float* wl = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + (tid_ >> 6) * EPI_PITCH);
            constexpr int WC = 32 * NIW, QPR = 8 * NIW;
            float* const AB_STASH = wl + EPI_STRIDE / 4;
            float mx = 0.f;  // range guard: largest |8 x| written (v_maximum3_f32 propagates NaN)
            // eval-BatchNorm alpha | beta of the channels a lane finishes: item `it` of a lane is quad (ln + 64 it) % QPR of its row -- ONE quad
            // for 16 / 32 quads per row (NI = 2 / 4) -- so they are loaded once per tile.  (Round 6: read inside the item loop they were two
            // 16-byte global loads + s_waitcnt vmcnt(0) PER ITEM -- the stores to the output planes may alias them as far as hipcc knows -- i.e.
            // 16 dependent round trips per tile, each also waiting for the previous item's stores.)
            // (24 quads per row, NI = 3: three quads per lane = 24 registers, which the 192-channel builds do not have -- 7 spilled when tried;
            // there the wave parks its 24 + 24 quads in LDS behind its epilogue region -- AB_STASH -- and an item reads its two from there.)
            constexpr bool kHoist = 64 % QPR == 0;
            f32x4 al_ = {1.f, 1.f, 1.f, 1.f}, be_ = {0.f, 0.f, 0.f, 0.f};
            if constexpr (kHoist) {
                const int co_l = j0 + WC * wc + 4 * (ln % QPR);
                if (a.alpha && co_l < a.Cout) {
                    al_ = *reinterpret_cast<const f32x4*>(a.alpha + co_l);
                    be_ = *reinterpret_cast<const f32x4*>(a.beta + co_l);
                }
            } else if (a.alpha && ln < 2 * QPR) {   // lanes 0 .. 23: alpha quads, 24 .. 47: beta quads
                const int qd_l = ln < QPR ? ln : ln - QPR, co_l = j0 + WC * wc + 4 * qd_l;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (co_l < a.Cout) t = *reinterpret_cast<const f32x4*>((ln < QPR ? a.alpha : a.beta) + co_l);
                *reinterpret_cast<f32x4*>(AB_STASH + 4 * ln) = t;
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) wl[frag_row(r, ln) * WC + 32 * ni + l31] = acc[mi][ni][r];
                // one unconditional use in straight-line code: hipcc then waits for alpha | beta HERE (under the LDS writes above), once -- behind the
                // items' `co < Cout` branches its wait-count pass assumes them pending at every join and emits vmcnt(0) per item, i.e. a wait for
                // the previous item's stores
                if (kHoist && mi == 0) asm volatile("" : "+v"(al_), "+v"(be_));
                constexpr int RG = NIW == 2 ? 8 : (NIW == 3 ? 6 : 4);
#pragma unroll
                for (int it0 = 0; it0 < 4 * NIW; it0 += RG) {
                    cu32x2 rh[RG], rl[RG];
                    if (a.rhi) {
#pragma unroll
                        for (int u = 0; u < RG; ++u) {
                            const int f = (it0 + u) * 64 + ln, row = f / QPR, qd = f - row * QPR;
                            const int pl = 64 * wr + 32 * mi + row, pix = pix00 + (pl >> 4) * a.OW + (pl & 15), co = j0 + WC * wc + 4 * qd;
                            const unsigned ob = co < a.Cout ? ((unsigned)pix * (unsigned)a.Cout + (unsigned)co) * 2u : kOob;
                            rh[u] = __builtin_amdgcn_raw_buffer_load_b64(r_rhi, ob, 0, 0);
                            rl[u] = __builtin_amdgcn_raw_buffer_load_b64(r_rlo, ob, 0, 0);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RG; ++u) {
                        const int f = (it0 + u) * 64 + ln, row = f / QPR, qd = f - row * QPR;
                        const f32x4 tv = *reinterpret_cast<const f32x4*>(wl + row * WC + 4 * qd);
                        const int pl = 64 * wr + 32 * mi + row, pix = pix00 + (pl >> 4) * a.OW + (pl & 15), co = j0 + WC * wc + 4 * qd;
                        if (co < a.Cout) {
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = tv[e] * kOutScale;
                            if (a.alpha) {
                                const f32x4 al4 = kHoist ? al_ : *reinterpret_cast<const f32x4*>(AB_STASH + 4 * qd), be4 = kHoist ? be_ : *reinterpret_cast<const f32x4*>(AB_STASH + 4 * (QPR + qd));
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = v[e] * al4[e] + be4[e];
                            }
                            const size_t o = (size_t)pix * a.Cout + co;
                            if (a.rhi) {  // + residual: (h + l) / 8, exact, then ONE rounding -- the bits of ((float)h + (float)l) * (1 / 8) + v
                                v[0] = __builtin_fmaf(cv_sum_planes<0>(rh[u][0], rl[u][0]), 1.0f / kActScale, v[0]);
                                v[1] = __builtin_fmaf(cv_sum_planes<1>(rh[u][0], rl[u][0]), 1.0f / kActScale, v[1]);
                                v[2] = __builtin_fmaf(cv_sum_planes<0>(rh[u][1], rl[u][1]), 1.0f / kActScale, v[2]);
                                v[3] = __builtin_fmaf(cv_sum_planes<1>(rh[u][1], rl[u][1]), 1.0f / kActScale, v[3]);
                            }
                            if (a.relu)
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                            if (a.of32) {  // (B, Cout, OH, OW)
                                const int bb = pix / OHW, rem = pix - bb * OHW;
#pragma unroll
                                for (int e = 0; e < 4; ++e) a.of32[((size_t)bb * a.Cout + co + e) * OHW + rem] = v[e];
                            } else {
                                const float s0 = v[0] * kActScale, s1 = v[1] * kActScale, s2 = v[2] * kActScale, s3 = v[3] * kActScale;
                                cu32x2 oh, ol;
                                oh[0] = cv_hi_pair(s0, s1);
                                oh[1] = cv_hi_pair(s2, s3);
                                ol[0] = cv_lo_pair(s0, s1, oh[0]);
                                ol[1] = cv_lo_pair(s2, s3, oh[1]);
                                mx = cv_absmax(cv_absmax(mx, s0, s1), s2, s3);
                                *reinterpret_cast<cu32x2*>(a.ohi + o) = oh;
                                *reinterpret_cast<cu32x2*>(a.olo + o) = ol;
                            }
                        }
                    }
                }
            }
            if (!(mx <= kSplitPlaneLimit)) gp_raise(a.status, GP_ST_SPLIT_RANGE_CONV);  // !(<=): NaN counts
            __syncthreads();  // the operand buffers are re-staged by the next segment's prologue
        }
        if (a.trace) { const unsigned long long t_ = wall_clock64(); tr_tail += t_ - tr_t0; tr_t0 = t_; }
    }
    if (a.trace && tid == 0) {
        unsigned long long* w = a.trace + (size_t)p * 8;
        w[0] = n_seg; w[1] = tr_steps; w[2] = tr_pro; w[3] = tr_loop; w[4] = tr_tail; w[5] = tr_begin; w[6] = tr_t0;
    }
}

// [C][npix] f32 (the f32 stem's channel-major output) -> planes [npix][C] of 8 x (hi + lo, unscaled lo).  32 x 32 tiles through LDS.
__global__ __launch_bounds__(256) void planes_from_cm_kernel(const float* __restrict__ X, int C, int npix, _Float16* __restrict__ hi,
                                                              _Float16* __restrict__ lo, int* __restrict__ status)
{
    __shared__ float t[32][33];
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) t[r][tx] = (c0 + r < C && p0 + tx < npix) ? X[(size_t)(c0 + r) * npix + p0 + tx] : 0.f;
    __syncthreads();
    int bad = 0;
    for (int r = ty; r < 32; r += 8) {
        if (p0 + r < npix && c0 + tx < C) {
            float v = t[tx][r] * kActScale;
            const _Float16 h = (_Float16)v;
            hi[(size_t)(p0 + r) * C + c0 + tx] = h;
            lo[(size_t)(p0 + r) * C + c0 + tx] = (_Float16)(v - (float)h);
            bad |= !(fabsf(v) <= kSplitPlaneLimit);
        }
    }
    if (bad) gp_raise(status, GP_ST_SPLIT_RANGE_CONV);
}

// F.interpolate(x, (S, S), mode="bilinear", align_corners=True) (reference resnet.py:366-368; arithmetic of gp_conv.hip's
// resize_kernel = ATen's upsample_bilinear2d) written straight into what the split stem reads: channel-last planes of 8 x the
// value, 4 channels per pixel (the 4th zero: one pixel = 8 bytes, two pixels = one 16-byte operand chunk) inside a frame of
// zeros -- 3 rows / columns before (the 7 x 7 kernel's padding) and enough after for the 8th, zero-weighted tap column -- so
// that no tap needs a range check and a kernel row of 8 taps is 64 contiguous bytes.  The frame is zeroed once by the host.
__global__ __launch_bounds__(256) void resize_stem_planes_kernel(const float* __restrict__ in, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                                  int B, int IH, int IW, int S, int Hp, int Wp, int* __restrict__ status)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= S) return;
    const float rh = (S > 1) ? (float)(IH - 1) / (float)(S - 1) : 0.f;
    const float rw = (S > 1) ? (float)(IW - 1) / (float)(S - 1) : 0.f;
    const float h1r = rh * y, w1r = rw * x;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = (h1 < IH - 1) ? 1 : 0, w1p = (w1 < IW - 1) ? 1 : 0;
    const float h1l = h1r - h1, h0l = 1.f - h1l, w1l = w1r - w1, w0l = 1.f - w1l;
    c16x4 oh, ol;
    int bad = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* src = in + ((size_t)b * 3 + c) * IH * IW;
        float v = h0l * (w0l * src[h1 * IW + w1] + w1l * src[h1 * IW + w1 + w1p]) +
                  h1l * (w0l * src[(h1 + h1p) * IW + w1] + w1l * src[(h1 + h1p) * IW + w1 + w1p]);
        v = v * kActScale;
        const _Float16 hh = (_Float16)v;
        oh[c] = hh;
        ol[c] = (_Float16)(v - (float)hh);
        bad |= !(fabsf(v) <= kSplitPlaneLimit);
    }
    oh[3] = (_Float16)0.f;
    ol[3] = (_Float16)0.f;
    const size_t o = (((size_t)b * Hp + y + 3) * Wp + x + 3) * 4;
    *reinterpret_cast<c16x4*>(hi + o) = oh;
    *reinterpret_cast<c16x4*>(lo + o) = ol;
    if (bad) gp_raise(status, GP_ST_SPLIT_RANGE_CONV);
}

unsigned g_epoch_conv = 0;
unsigned long long* g_conv_trace = nullptr;

}  // namespace

extern "C" {

#ifdef GP_PROBES
void gp_conv2d_planes_set_trace(unsigned long long* buf) { g_conv_trace = buf; }  // probe: 256 slots x 8 words, see ConvPArgs
#endif

size_t gp_conv2d_planes_workspace_bytes(void) { return kHeaderBytes + sizeof(float) * kFragFloats * kSlots; }

int gp_planes_from_cm(const float* X, int C, int npix, void* hi, void* lo, void* stream)
{
    GP_REQUIRE(X && hi && lo && C > 0 && npix > 0, "gp_planes_from_cm: bad arguments");
    hipLaunchKernelGGL(planes_from_cm_kernel, dim3((npix + 31) / 32, (C + 31) / 32), dim3(256), 0, (hipStream_t)stream, X, C, npix,
                       (_Float16*)hi, (_Float16*)lo, gp_status_buffer());
    GP_CHECK_LAUNCH("gp_planes_from_cm");
    return GP_OK;
}

static int g_conv_halo = 1;  // bit 0: 3 x 3 / stride 1 layers take the halo kernel; bit 1 (with it): its parallel split below 256 tiles (A/B hook)
static int g_conv_par = 1;
static int g_conv_par_min_cb = 1;  // channel blocks (9 k-steps each) per slot of a split tile at least
#ifdef GP_PROBES
int gp_conv2d_planes_set_halo(int on)
{
    g_conv_halo = (on & 1) ? 1 : 0;
    g_conv_par = ((on & 15) == 1 || (on & 2)) ? 1 : 0;   // 1 = both (default), 0 = neither, 5 = halo without the parallel split
    g_conv_par_min_cb = (on >> 4) > 0 ? (on >> 4) : 1;   // probe: on = 1 + 16 n -> at least n channel blocks per slot of a split tile
    return GP_OK;
}
#endif

// 3 x 3 / stride 1 / pad 1 on 16 x 16 pixel blocks with the halo resident in LDS (conv_halo_kernel); same arguments and results
// (to f32 round-off) as gp_conv2d_planes.  Needs H, W multiples of 16.
static bool conv_halo_usable(int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad)
{
    return g_conv_halo && KH == 3 && KW == 3 && stride == 1 && pad == 1 && H % 16 == 0 && W % 16 == 0 && Cin % 32 == 0 && Cout % 64 == 0;
}

static int conv_halo_launch(ConvPArgs& a, float* scratch, hipStream_t st)
{
    const int ni = a.Cout >= 256 ? 4 : a.Cout / 64;
    a.tiles_i = a.B * (a.OH / 16) * (a.OW / 16);
    a.tiles_j = (a.Cout + 64 * ni - 1) / (64 * ni);
    const int ncb = a.Cin / CBK;
    const long long n_tiles = (long long)a.tiles_i * a.tiles_j, units = n_tiles * ncb * 9;
    int slots_x = 32;
    const bool par = g_conv_par && n_tiles < 8 * slots_x && n_tiles >= 8 && ncb >= 2;  // fewer tiles than slots: parallel split over channel blocks
    a.par_cap = max(1, ncb / g_conv_par_min_cb);
    while (!par && slots_x > 1 && n_tiles < 8 * slots_x && units / (8 * slots_x) < 32) slots_x >>= 1;
    a.flags = reinterpret_cast<int*>(scratch);
    a.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + kHeaderBytes);
    g_epoch_conv = (g_epoch_conv + 1) & 0x3fffffff;
    a.epoch = (int)(0x20000000u | g_epoch_conv);
    a.status = gp_status_buffer();
    const long long npix = (long long)a.B * a.OH * a.OW;
    GpProfScope prof(GP_PROF_CONV, 2.0 * a.Cout * (double)npix * a.K, st);
    if (par) {
        if (ni == 2) hipLaunchKernelGGL((conv_halo_kernel<2, true>), dim3(8 * slots_x), dim3(CNT), 0, st, a);
        else if (ni == 3) hipLaunchKernelGGL((conv_halo_kernel<3, true>), dim3(8 * slots_x), dim3(CNT), 0, st, a);
        else hipLaunchKernelGGL((conv_halo_kernel<4, true>), dim3(8 * slots_x), dim3(CNT), 0, st, a);
    } else {
        if (ni == 2) hipLaunchKernelGGL((conv_halo_kernel<2, false>), dim3(8 * slots_x), dim3(CNT), 0, st, a);
        else if (ni == 3) hipLaunchKernelGGL((conv_halo_kernel<3, false>), dim3(8 * slots_x), dim3(CNT), 0, st, a);
        else hipLaunchKernelGGL((conv_halo_kernel<4, false>), dim3(8 * slots_x), dim3(CNT), 0, st, a);
    }
    GP_CHECK_LAUNCH("gp_conv2d_planes/halo");
    return GP_OK;
}

int gp_conv2d_planes(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* alpha, const float* beta,
                     const void* res_hi, const void* res_lo, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                     int relu, void* out_hi, void* out_lo, float* out_f32_nchw, float* scratch, size_t scratch_bytes, void* stream)
{
    GP_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "gp_conv2d_planes: bad sizes");
    if (B == 0) return GP_OK;
    ConvPArgs a;
    a.xhi = (const _Float16*)x_hi; a.xlo = (const _Float16*)x_lo; a.whi = (const _Float16*)w_hi; a.wlo = (const _Float16*)w_lo;
    a.alpha = alpha; a.beta = beta; a.rhi = (const _Float16*)res_hi; a.rlo = (const _Float16*)res_lo;
    a.ohi = (_Float16*)out_hi; a.olo = (_Float16*)out_lo; a.of32 = out_f32_nchw;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.relu = relu;
    a.OH = (H + 2 * pad - KH) / stride + 1;
    a.OW = (W + 2 * pad - KW) / stride + 1;
    a.K = KH * KW * Cin;
    const long long npix = (long long)B * a.OH * a.OW;
    GP_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0 && KH * KW <= 9, "gp_conv2d_planes: Cin=%d must be a multiple of 32, Cout=%d of 64, at most 9 taps", Cin, Cout);
    GP_REQUIRE(npix % CT == 0 && (long long)B * H * W * Cin * 2 < (1ll << 31) && (long long)Cout * a.K * 2 < (1ll << 31) && npix * Cout < (1ll << 31),
               "gp_conv2d_planes: B*OH*OW=%lld must be a multiple of 256 and every plane below 2 GiB", npix);
    GP_REQUIRE(x_hi && x_lo && w_hi && w_lo && ((out_hi && out_lo) || out_f32_nchw), "gp_conv2d_planes: null pointer");
    GP_REQUIRE((alpha == nullptr) == (beta == nullptr) && (res_hi == nullptr) == (res_lo == nullptr), "gp_conv2d_planes: alpha/beta and res_hi/res_lo go together");
    GP_REQUIRE(scratch && scratch_bytes >= gp_conv2d_planes_workspace_bytes() && ((uintptr_t)scratch % 16 == 0), "gp_conv2d_planes: scratch too small");
    GP_REQUIRE(((uintptr_t)x_hi % 16 == 0) && ((uintptr_t)x_lo % 16 == 0) && ((uintptr_t)w_hi % 16 == 0) && ((uintptr_t)w_lo % 16 == 0) &&
                   ((uintptr_t)alpha % 16 == 0) && ((uintptr_t)beta % 16 == 0), "gp_conv2d_planes: misaligned operand");
    a.stem = 0;
    a.par_cap = 1;
    a.trace = g_conv_trace;
    if (conv_halo_usable(H, W, Cin, Cout, KH, KW, stride, pad)) return conv_halo_launch(a, scratch, (hipStream_t)stream);
    const int ni = Cout >= 256 ? 4 : Cout / 64;  // 128 -> 2, 192 -> 3, >= 256 -> 4
    a.tiles_i = (int)(npix / CT);
    a.tiles_j = (Cout + 64 * ni - 1) / (64 * ni);
    hipStream_t st = (hipStream_t)stream;
    // slots: all 256 CUs unless there are fewer tiles than slots AND so little work that cutting tiles would leave a slot with
    // under 32 k-steps per hand-over (the 1 x 1 head: 64 tiles x 16 steps -> 32 slots of two whole tiles)
    const long long n_tiles = (long long)a.tiles_i * a.tiles_j, units = n_tiles * (a.K / CBK);
    int slots_x = 32;
    while (slots_x > 1 && n_tiles < 8 * slots_x && units / (8 * slots_x) < 32) slots_x >>= 1;
    // no per-launch reset of the hand-off flags: a flag is valid only when it holds THIS launch's epoch (below); the
    // scratch must start zeroed once (the caller allocates it with zeros)
    a.flags = reinterpret_cast<int*>(scratch);
    a.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + kHeaderBytes);
    g_epoch_conv = (g_epoch_conv + 1) & 0x3fffffff;
    a.epoch = (int)(0x20000000u | g_epoch_conv);
    a.status = gp_status_buffer();
    GpProfScope prof(GP_PROF_CONV, 2.0 * Cout * (double)npix * a.K, st);
    if (ni == 2) hipLaunchKernelGGL(conv_planes_kernel<2>, dim3(8 * slots_x), dim3(CNT), 0, st, a);
    else if (ni == 3) hipLaunchKernelGGL(conv_planes_kernel<3>, dim3(8 * slots_x), dim3(CNT), 0, st, a);
    else hipLaunchKernelGGL(conv_planes_kernel<4>, dim3(8 * slots_x), dim3(CNT), 0, st, a);
    GP_CHECK_LAUNCH("gp_conv2d_planes");
    return GP_OK;
}

/* The stem of the IST ResNet (reference resnet.py:333-337, 366-370: bilinear resize to S x S, Conv2d(3 -> Cout, 7 x 7, stride 2,
 * padding 3, no bias) + BatchNorm + ReLU) in split numerics.  gp_resize_stem_planes writes the resized crops as framed 4-channel
 * planes (B, S + 6, S + 8, 4) (the caller zeroes the buffers ONCE: the frame is never written); gp_conv2d_stem_planes runs
 * conv_planes_kernel over them: k = dy * 32 + dx * 4 + ci (K = 224; weight planes (Cout, 224) of 64 w with zeros at dx = 7 and
 * ci = 3), output planes (B * (S/2)^2, Cout) x 8 as every later layer reads them. */
int gp_resize_stem_planes(const float* images, void* hi, void* lo, int B, int IH, int IW, int S, void* stream)
{
    GP_REQUIRE(B >= 0 && IH > 0 && IW > 0 && S > 0 && S % 2 == 0, "gp_resize_stem_planes: bad sizes");
    if (B == 0) return GP_OK;
    GP_REQUIRE(images && hi && lo && ((uintptr_t)hi % 16 == 0) && ((uintptr_t)lo % 16 == 0), "gp_resize_stem_planes: null / misaligned pointer");
    GpProfScope prof(GP_PROF_OTHER, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(resize_stem_planes_kernel, dim3((S + 255) / 256, S, B), dim3(256), 0, (hipStream_t)stream, images, (_Float16*)hi, (_Float16*)lo,
                       B, IH, IW, S, S + 6, S + 8, gp_status_buffer());
    GP_CHECK_LAUNCH("gp_resize_stem_planes");
    return GP_OK;
}

int gp_conv2d_stem_planes(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* alpha, const float* beta,
                          int B, int S, int Cout, int relu, void* out_hi, void* out_lo, float* scratch, size_t scratch_bytes, void* stream)
{
    GP_REQUIRE(B >= 0 && S > 0 && S % 2 == 0 && (Cout == 128 || Cout == 192 || Cout % 256 == 0), "gp_conv2d_stem_planes: bad sizes");
    if (B == 0) return GP_OK;
    ConvPArgs a;
    a.xhi = (const _Float16*)x_hi; a.xlo = (const _Float16*)x_lo; a.whi = (const _Float16*)w_hi; a.wlo = (const _Float16*)w_lo;
    a.alpha = alpha; a.beta = beta; a.rhi = nullptr; a.rlo = nullptr; a.ohi = (_Float16*)out_hi; a.olo = (_Float16*)out_lo; a.of32 = nullptr;
    a.B = B; a.H = S + 6; a.W = S + 8; a.Cin = 4; a.OH = S / 2; a.OW = S / 2; a.Cout = Cout; a.KH = 7; a.KW = 8; a.stride = 2; a.pad = 0;
    a.relu = relu; a.K = 7 * 32; a.stem = 1; a.par_cap = 1;
    a.trace = g_conv_trace;
    const long long npix = (long long)B * a.OH * a.OW;
    GP_REQUIRE(npix % CT == 0 && (long long)B * a.H * a.W * 8 < (1ll << 31) && npix * Cout < (1ll << 31), "gp_conv2d_stem_planes: B*OH*OW=%lld must be a multiple of 256", npix);
    GP_REQUIRE(x_hi && x_lo && w_hi && w_lo && out_hi && out_lo && (alpha == nullptr) == (beta == nullptr), "gp_conv2d_stem_planes: null pointer");
    GP_REQUIRE(scratch && scratch_bytes >= gp_conv2d_planes_workspace_bytes() && ((uintptr_t)scratch % 16 == 0), "gp_conv2d_stem_planes: scratch too small");
    const int ni = Cout >= 256 ? 4 : Cout / 64;
    a.tiles_i = (int)(npix / CT);
    a.tiles_j = (Cout + 64 * ni - 1) / (64 * ni);
    hipStream_t st = (hipStream_t)stream;
    // no per-launch reset of the hand-off flags: a flag is valid only when it holds THIS launch's epoch (below); the
    // scratch must start zeroed once (the caller allocates it with zeros)
    a.flags = reinterpret_cast<int*>(scratch);
    a.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + kHeaderBytes);
    g_epoch_conv = (g_epoch_conv + 1) & 0x3fffffff;
    a.epoch = (int)(0x20000000u | g_epoch_conv);
    a.status = gp_status_buffer();
    GpProfScope prof(GP_PROF_CONV, 2.0 * Cout * (double)npix * 147.0, st);  // algorithmic: 3 x 7 x 7 taps
    if (ni == 2) hipLaunchKernelGGL(conv_planes_kernel<2>, dim3(kSlots), dim3(CNT), 0, st, a);
    else if (ni == 3) hipLaunchKernelGGL(conv_planes_kernel<3>, dim3(kSlots), dim3(CNT), 0, st, a);
    else hipLaunchKernelGGL(conv_planes_kernel<4>, dim3(kSlots), dim3(CNT), 0, st, a);
    GP_CHECK_LAUNCH("gp_conv2d_stem_planes");
    return GP_OK;
}

}  // extern "C"
