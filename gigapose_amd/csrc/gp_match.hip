// Fused template matcher for gfx950: LocalSimilarity.test of the reference
// (src/models/matching.py:188-316) without ever materialising the (B,N,256,256) similarity
// tensor in HBM.
//
//   gp_l2norm_cp      F.normalize over C                      (matching.py:224,229; ae_net.py:69)
//   gp_match_tiles    one workgroup per (detection, template): 256x256xC f32 MFMA contraction,
//                     then masks / threshold / row+col argmax / cycle check / score in
//                     registers + LDS                          (matching.py:233-278, :80-113)
//   gp_topk           top-k templates per detection           (matching.py:279)
//   gp_gather_records per-candidate patch records             (matching.py:282-295)
//   gp_format_points  -1-padded (x,y) correspondences         (matching.py:29-61, :63-68)
//
// Arithmetic order is the one fixed in oracle/gp_oracle.c (sequential fmaf over channels), so
// every output -- float scores included -- is bit-identical to the oracle.
#include "gp_common.h"

namespace {

// ------------------------------------------------------------------ F.normalize over C
// x, out: (rows, C, 256).  One workgroup per row, thread p owns patch p (coalesced over p).
__global__ __launch_bounds__(256) void l2norm_cp_kernel(const float* __restrict__ x,
                                                         float* __restrict__ out, int C)
{
    // grid (rows, C / 32): each block recomputes the row's 256 norms (same sequential fma chain) and writes 32 channels
    const size_t base = (size_t)blockIdx.x * C * GP_P + threadIdx.x;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = x[base + (size_t)c * GP_P];
        ss = __builtin_fmaf(v, v, ss);
    }
    const float d = fmaxf(__builtin_sqrtf(ss), 1e-12f);
    const int c0 = blockIdx.y * 32;
    for (int c = c0; c < c0 + 32 && c < C; ++c) out[base + (size_t)c * GP_P] = x[base + (size_t)c * GP_P] / d;
}

// ------------------------------------------------------------------ fused match tile
// 8 waves as 4 (query rows t) x 2 (template cols s); wave tile 64 x 128 = 2 x 4 MFMA tiles.
using MM = KMajor<4, 2, 2, 4, 16>;
static_assert(MM::BM == 256 && MM::BN == 256 && MM::NT == 512, "matcher tile must be 256x256");

struct alignas(16) MatchSmem {
    float stage[MM::LDS_FLOATS];  // 64 KiB operand staging
    float qmask[GP_P];
    float smask[GP_P];
    float rowv[2][GP_P];  // per column-wave partial row maxima
    int rowi[2][GP_P];
    float colv[4][GP_P];  // per row-wave partial column maxima
    int coli[4][GP_P];
    float sc_t2s[GP_P];
    int id_t2s[GP_P];
    float sc_s2t[GP_P];
    int id_s2t[GP_P];
    float contrib[GP_P];
    float maskv[GP_P];
};

// One step of the row-maximum reduce-scatter (match_epilogue): lanes L and L ^ CNT split the 2 CNT rows they both still hold --
// the lane with bit CNT set keeps the upper CNT rows, its partner the lower -- and fold the partner's candidates into theirs.
// CNT is a template parameter so that every rv[] / ri[] index is a compile-time constant (registers, not select chains).
template <int CNT>
__device__ __forceinline__ void rowmax_exchange(float (&rv)[32], int (&ri)[32], int lane)
{
    const bool hi = (lane & CNT) != 0;
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        const float sv = hi ? rv[j] : rv[j + CNT];
        const int si = hi ? ri[j] : ri[j + CNT];
        const float mv = hi ? rv[j + CNT] : rv[j];
        const int mj = hi ? ri[j + CNT] : ri[j];
        const float ov = __shfl_xor(sv, CNT);
        const int oi = __shfl_xor(si, CNT);
        const bool take = (ov > mv) | ((ov == mv) & (oi < mj));
        rv[j] = take ? ov : mv;
        ri[j] = take ? oi : mj;
    }
}

// Everything after the 256x256 similarity tile sits in the accumulators (8 waves as 4 (t) x 2 (s), wave tile
// 64 x 128 = acc[2][4]): masks, threshold, bidirectional argmax, cycle check, template score.  Shared by the
// f32-chain kernel and the split-f16 kernel (gp_match_split below); SM is the kernel's shared-memory struct.
// PERM (split kernel): the tile's rows / columns are the LIVE patches only, compacted and dealt block-cyclically to the waves
// (match_tiles_split_kernel); sm.t_of / sm.s_of map an LDS row back to its patch (kDeadPatch = none), sm.qmask_p / sm.smask_p hold
// the masks in that order.  A masked-out patch contributes exact zeros to every maximum in the reference (sim *= mask,
// matching.py:234-235), and an all-zero row's argmax is index 0 -- so the maxima over the live patches, compared in the
// (value desc, patch index asc) order and reset to index 0 when the maximum is 0, are the reference's, bit for bit.
constexpr int kDeadPatch = 0x7fff;
#include "gp_match_rects.h"
// The rectangle of 32 x 32 similarity blocks a wave holds in its accumulators: row blocks r0 .. r0 + mi_n - 1 (LDS rows 32 r0 ..), column
// blocks c0 .. c0 + ni_n - 1, and the slots its partial row / column maxima go to (rectangles sharing a row block have different row
// slots, rectangles sharing a column block different column slots).  The plain grid (f32-chain kernel, uncompacted tiles): 4 x 2 waves
// of 2 x 4 blocks, row slot = wave column, column slot = wave row.
struct MatchWaveTile {
    int r0, mi_n, c0, ni_n, rslot, cslot, active;
};
__device__ __forceinline__ MatchWaveTile match_wave_tile_grid(int wave)
{
    return MatchWaveTile{2 * (wave >> 1), 2, 4 * (wave & 1), 4, wave & 1, wave >> 1, 1};
}
__device__ __forceinline__ MatchWaveTile match_wave_tile_unpack(unsigned w)
{
    return MatchWaveTile{(int)(w & 15u), (int)((w >> 4) & 3u), (int)((w >> 6) & 15u), (int)((w >> 10) & 7u), (int)((w >> 13) & 3u), (int)((w >> 15) & 3u),
                         (int)((w >> 17) & 1u)};
}
template <bool PERM = false, class SM>
__device__ __forceinline__ void match_epilogue(SM& sm, f32x16 (&acc)[2][4], int b, int n, int N, float thr, float patch_thr,
                                               uint8_t* __restrict__ idx_t2s, float* __restrict__ score_t2s,
                                               float* __restrict__ mask_all, float* __restrict__ sim_avg,
                                               unsigned long long* trace = nullptr,  // probe build only: 8 stamps per tile
                                               int src2tar = 0,   // search_direction == "src2tar" (matching.py:242-244)
                                               int compact = 1,   // PERM: the tile holds the live patches only (see below)
                                               MatchWaveTile wt = MatchWaveTile{0, 0, 0, 0, 0, 0, 0})  // PERM: this wave's rectangle
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: selects per-wave code paths
    if constexpr (!PERM) wt = match_wave_tile_grid(wave);
    // ---- sim *= src_mask; sim *= tar_mask; sim[sim < thr] = 0   (matching.py:234-236)
    const int s_lane = 32 * wt.c0 + (lane & 31);
    const int t_lane = 32 * wt.r0 + 4 * (lane >> 5);
    float sm_s[4];
    int s_idx[4];  // patch index of this lane's column in block ni
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        sm_s[ni] = 0.f;
        s_idx[ni] = kDeadPatch;
        if (!PERM || ni < wt.ni_n) {  // (wave-uniform) blocks beyond the rectangle hold nothing: their accumulators stay zero and are not read
            if constexpr (PERM) {
                sm_s[ni] = sm.smask_p[s_lane + 32 * ni];
                s_idx[ni] = sm.s_of[s_lane + 32 * ni];
            } else {
                sm_s[ni] = sm.smask[s_lane + 32 * ni];
                s_idx[ni] = s_lane + 32 * ni;
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        if (PERM && mi >= wt.mi_n) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float tm;
            if constexpr (PERM) tm = sm.qmask_p[t_lane + 32 * mi + (r & 3) + 8 * (r >> 2)];
            else tm = sm.qmask[t_lane + 32 * mi + (r & 3) + 8 * (r >> 2)];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                float v = acc[mi][ni][r] * sm_s[ni];
                v = v * tm;
                acc[mi][ni][r] = (v < thr) ? 0.f : v;
            }
        }
    }

    // ---- row maxima (over s, first max wins)   torch.max(sim, dim=3)  (matching.py:240)
    // (value desc, index asc) is a total order, so any reduction tree gives the same winner.  Each lane first folds its
    // 4 column blocks, then the 32 lanes of a half-wave run a reduce-scatter butterfly over the 32 rows they share:
    // at offset 16 a lane hands 16 rows to its partner and keeps 16, at 8 it keeps 8, ... -- 31 exchanges per lane
    // instead of 32 x 5, all selects (no short-circuit branches), and lane L ends up owning row L.
    float rv[32];
    int ri[32];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float bv = acc[mi][0][r];
            int bi = s_idx[0];
#pragma unroll
            for (int ni = 1; ni < 4; ++ni) {
                const float x = acc[mi][ni][r];
                // (a column block beyond the rectangle: x = 0 and s_idx = kDeadPatch, the largest index -- it never displaces a candidate)
                const bool take = PERM ? ((x > bv) | ((x == bv) & (s_idx[ni] < bi))) : (x > bv);  // unpermuted: ni ascending == s ascending
                bv = take ? x : bv;
                bi = take ? s_idx[ni] : bi;
            }
            rv[mi * 16 + r] = bv;
            ri[mi * 16 + r] = bi;
        }
    rowmax_exchange<16>(rv, ri, lane);  // exchange offset == rows kept
    rowmax_exchange<8>(rv, ri, lane);
    rowmax_exchange<4>(rv, ri, lane);
    rowmax_exchange<2>(rv, ri, lane);
    rowmax_exchange<1>(rv, ri, lane);
    {
        const int j = lane & 31;  // the row this lane ended up with: mi = j >> 4, r = j & 15
        int t = t_lane + 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2);
        const bool mine = !PERM || (wt.active && (j >> 4) < wt.mi_n);  // a row block of this wave's rectangle
        if constexpr (PERM) t = mine ? sm.t_of[t] : kDeadPatch;
        if (mine && (!PERM || t != kDeadPatch)) {  // rows without a live patch keep the zeros the prologue wrote
            sm.rowv[wt.rslot][t] = rv[0];
            sm.rowi[wt.rslot][t] = ri[0];
        }
    }

    // ---- column maxima (over t, first max wins)   torch.max(sim, dim=2)  (matching.py:241)
    // PERM: LDS rows are not in patch order, so the candidate is tracked by its PATCH index and an exact tie between positive values
    // goes to the lower patch.  Round 6: the patch indices of this lane's 32 rows are read ONCE into registers and every step is a
    // branch-free select (round 5 looked t_of[] up in LDS inside the 128-step loop behind short-circuit branches: the maxima phase of
    // a full tile took 16-20 us against 6.7 us before compaction existed, profiles/r06_match.txt).  Same total order, same bits.
    int tp[32];
    if constexpr (PERM) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {   // rows t_lane + 32 mi + 8 r4 + 0..3: four consecutive shorts = one 8-byte LDS read
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                const s16x4 q4 = *reinterpret_cast<const s16x4*>(&sm.t_of[t_lane + 32 * mi + 8 * r4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) tp[mi * 16 + 4 * r4 + e] = (int)q4[e];
            }
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        if (PERM && (ni >= wt.ni_n || !wt.active)) continue;  // wave-uniform
        float bv = acc[0][ni][0];
        int bi = PERM ? tp[0] : t_lane;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (PERM && mi >= wt.mi_n) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {  // (mi, r) ascending == t ascending for this lane (unpermuted tile)
                const float x = acc[mi][ni][r];
                if constexpr (PERM) {
                    const int tq = tp[mi * 16 + r];
                    const bool take = (x > bv) | ((x == bv) & ((x > 0.f) | (compact == 0)) & (tq < bi));
                    bv = take ? x : bv;
                    bi = take ? tq : bi;
                } else {
                    const int row = t_lane + 32 * mi + (r & 3) + 8 * (r >> 2);
                    if (x > bv) { bv = x; bi = row; }
                }
            }
        }
        const float ov = __shfl_xor(bv, 32);
        const int oi = __shfl_xor(bi, 32);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        if (lane < 32 && (!PERM || s_idx[ni] != kDeadPatch)) {
            sm.colv[wt.cslot][s_idx[ni]] = bv;
            sm.coli[wt.cslot][s_idx[ni]] = bi;
        }
    }
    __syncthreads();
    if (trace && tid == 0) trace[3] = wall_clock64();

    // ---- merge the per-wave partials (lower index range first, strict > keeps first max)
    if (tid < GP_P) {
        float bv = sm.rowv[0][tid];
        int bi = sm.rowi[0][tid];
#pragma unroll
        // compacted tile: up to four rectangles hold pieces of a row; slots never written keep the prologue's zeros, harmless next to
        // values that are all >= 0.  Uncompacted (the plain grid, possibly negative values): exactly two, both written.
        for (int w = 1; w < ((PERM && compact) ? 4 : 2); ++w) {
            const float v1 = sm.rowv[w][tid];
            const int i1 = sm.rowi[w][tid];
            if (v1 > bv || (PERM && v1 == bv && i1 < bi)) { bv = v1; bi = i1; }
        }
        // compacted tile (needs sim_threshold >= 0: every surviving value is >= 0): a maximum of 0 means an all-zero row, whose first
        // index is 0 -- the masked-out patches' zeros are part of the reference's row but not of this tile.  Uncompacted (every
        // patch in the tile, e.g. a negative threshold): the zeros ARE in the tile and the (value, patch index) order already
        // returns the reference's first maximum -- which is the first masked-out patch, not patch 0, when live values are negative.
        if (PERM && compact && bv == 0.f) bi = 0;
        sm.sc_t2s[tid] = bv;
        sm.id_t2s[tid] = bi;
    } else {
        const int s = tid - GP_P;
        float bv = sm.colv[0][s];
        int bi = sm.coli[0][s];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float vw = sm.colv[w][s];
            const int iw = sm.coli[w][s];
            if (vw > bv || (PERM && vw == bv && iw < bi)) { bv = vw; bi = iw; }
        }
        if (PERM && compact && bv == 0.f) bi = 0;
        sm.sc_s2t[s] = bv;
        sm.id_s2t[s] = bi;
    }
    __syncthreads();
    if (trace && tid == 0) trace[4] = wall_clock64();

    // ---- masks + per-patch outputs   (matching.py:247-271, find_consistency_patches :80-113)
    // "A" is what the reference calls tar2src and "B" its src2tar: with search_direction == "tar2src" A = the row maxima (over s,
    // per query patch t) and B = the column maxima; "src2tar" (matching.py:242-244) exchanges them and every later step keeps
    // indexing by POSITION p = 0..255 -- tar_mask at p, src_mask at the matched index (:260-261), whatever p stands for.
    if (tid < GP_P) {
        const int t = tid;
        const float* sc_a = src2tar ? sm.sc_s2t : sm.sc_t2s;
        const int* id_a = src2tar ? sm.id_s2t : sm.id_t2s;
        const float* sc_b = src2tar ? sm.sc_t2s : sm.sc_s2t;
        const int* id_b = src2tar ? sm.id_t2s : sm.id_s2t;
        const int js = id_a[t];
        const float sc = sc_a[t];
        const bool mask_sim = sc >= thr;
        bool mask_cycle = true;  // patch_threshold <= 0: the reference skips the cycle check (matching.py:256-257)
        if (patch_thr > 0.f) {
            const int t2 = id_b[js];
            const float dx = (float)(t2 % GP_G) - (float)(t % GP_G);
            const float dy = (float)(t2 / GP_G) - (float)(t / GP_G);
            const float dist = __builtin_sqrtf(dx * dx + dy * dy);
            const bool mask_dist = dist <= patch_thr;
            const bool mask_sim2 = sc_b[js] >= thr;
            mask_cycle = mask_dist && mask_sim2;
        }
        // reference quirk: (idx_src2tar != 0) is applied at POSITION t, not at the matched s
        float nz = sm.qmask[t] * sm.smask[js];
        nz = nz * (float)(id_b[t] != 0);
        nz = nz * (float)(js != 0);
        const float m = (float)(mask_sim && mask_cycle) * nz;
        const size_t o = ((size_t)b * N + n) * GP_P + t;
        idx_t2s[o] = (uint8_t)js;
        score_t2s[o] = sc;
        mask_all[o] = m;
        sm.contrib[t] = sc * m;
        sm.maskv[t] = m;
    }
    __syncthreads();
    if (trace && tid == 0) trace[5] = wall_clock64();
    if (tid == 0) {  // fixed sequential order (oracle: same loop); loads batched 32 at a time, adds stay in order
        float a = 0.f, cnt = 0.f;
#pragma unroll 1
        for (int t0 = 0; t0 < GP_P; t0 += 32) {
            f32x4 c[8], m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c[u] = *reinterpret_cast<const f32x4*>(&sm.contrib[t0 + 4 * u]);
                m[u] = *reinterpret_cast<const f32x4*>(&sm.maskv[t0 + 4 * u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a = a + c[u][e];
                    cnt = cnt + m[u][e];
                }
        }
        sim_avg[(size_t)b * N + n] = (cnt > 0.f) ? a / 256.0f : 0.f;
        if (trace) trace[6] = wall_clock64();
    }
}

__global__ __launch_bounds__(512, 2) void match_tiles_kernel(
    const float* __restrict__ query,  // (B, C, 256)   matcher-normalised
    const float* __restrict__ bank,   // (O, N, C, 256) matcher-normalised
    const float* __restrict__ qmask,  // (B, 256)
    const float* __restrict__ bmask,  // (O, N, 256)
    const int* __restrict__ labels,   // (B) 0-based object index
    int B, int O, int N, int C, float thr, float patch_thr, int* __restrict__ status,
    uint8_t* __restrict__ idx_t2s,    // (B, N, 256)
    float* __restrict__ score_t2s,    // (B, N, 256)
    float* __restrict__ mask_all,     // (B, N, 256)
    float* __restrict__ sim_avg,      // (B, N)
    int src2tar)
{
    __shared__ MatchSmem sm;
    const int q = xcd_chunked_tile(blockIdx.x, B * N);
    if (q < 0) return;
    // Tile order: bands of 8 detections, b fastest inside a band, then n.  The 32 tiles an XCD runs concurrently
    // (one 84 KB workgroup per CU) are then 8 queries x 4 templates: each k-slab of a query is fetched into that
    // XCD's L2 once per 4 tiles and each template slab once per 8, instead of every query being streamed again
    // for every template (b fastest over all B: 10.4 GB of fabric reads per launch at B=64, N=162).
    const int band = q / (8 * N), r8 = q - band * (8 * N);
    const int gsz = min(8, B - band * 8);
    const int b = band * 8 + r8 % gsz, n = r8 / gsz;
    const int tid = threadIdx.x;
    int lab = labels[b];
    if ((unsigned)lab >= (unsigned)O) {  // the reference raises IndexError at ae_features[label - 1] (gigaPose.py:520)
        if (tid == 0) gp_raise(status, GP_ST_LABEL_RANGE);
        lab = 0;
    }
    const size_t on = (size_t)lab * N + n;
    const float* A = query + (size_t)b * C * GP_P;
    const float* Bm = bank + on * (size_t)C * GP_P;

    if (tid < GP_P) sm.qmask[tid] = qmask[(size_t)b * GP_P + tid];
    else sm.smask[tid - GP_P] = bmask[on * GP_P + (tid - GP_P)];

    f32x16 acc[2][4];
    MM::run(A, GP_P, Bm, GP_P, C, sm.stage, acc);  // ends with __syncthreads(): masks visible

    match_epilogue(sm, acc, b, n, N, thr, patch_thr, idx_t2s, score_t2s, mask_all, sim_avg, nullptr, src2tar);
}

// ------------------------------------------------------------------ split-f16 matcher (opt-in numerics)
// Same tile, same epilogue; the 256x256xC similarity product runs on the f16 matrix core as THREE MFMAs per
// k-block (gp_split.hip explains the split): both operands arrive PRE-SPLIT as f16 planes [patch][C] (channel
// contiguous), scaled by 32 so that the low halves of unit-vector components stay in f16's normal range; the
// three products go into ONE f32 accumulator and the tile is rescaled by the exact factor 2^-10 before the
// epilogue.  LDS: two buffers of 4 planes x 256 rows x 64 B (k-step 32), 16-byte chunks XOR-swizzled by
// (row >> 2) & 3 so that every ds_read_b128 lane group touches all 64 banks once.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr float kFeatScale = 32.0f;
constexpr int MS_BK = 32, MS_PLANE = 256 * MS_BK;  // halfs per plane per buffer

struct alignas(16) MatchSplitSmem {
    _Float16 stage[2 * 4 * MS_PLANE];  // 128 KiB
    float qmask[GP_P];
    float smask[GP_P];
    float qmask_p[GP_P];  // masks by LDS row of the compacted tile (0 where the row holds no live patch)
    float smask_p[GP_P];
    short t_of[GP_P];     // LDS row -> query patch (kDeadPatch: none)
    short s_of[GP_P];     // LDS row of the B planes -> template patch
    int cnt[8];           // live patches per 64-patch slice: [0..3] query, [4..7] template
    float rowv[4][GP_P];  // partial row maxima by slot (MatchWaveTile::rslot)
    int rowi[4][GP_P];
    float colv[4][GP_P];
    int coli[4][GP_P];
    float sc_t2s[GP_P];
    int id_t2s[GP_P];
    float sc_s2t[GP_P];
    int id_s2t[GP_P];
    float contrib[GP_P];
    float maskv[GP_P];
};

// Main loop = gemm_planes256_kernel's (gp_split256.hip): the two wave groups of the workgroup (waves 0-3 / 4-7, one of each
// per SIMD) run half a k-step apart -- one issues its MFMAs from LDS buffer s while the other writes its share of slab
// s + 1 and loads slab s + 2 -- with ONE barrier per step; per accumulator the products keep the order hi*hi, hi*lo, lo*hi
// per k16 block, k ascending.  BANK_LO = false: the bank holds only its f16 hi plane (BASELINE config 5's "fp16 feature bank":
// half the bytes, 2 of the 3 products; the query keeps both planes) -- f16-rounded template features, flip rate in DESIGN.md.
//
// Masked-out patches are not computed (round 3).  The reference multiplies the similarity by both patch masks
// (matching.py:234-235): a row / column of a masked-out patch is exact zeros whatever the features are -- with the disc masks
// of the benchmark 39 % of a 256 x 256 tile is live.  The tile is therefore built from the LIVE patches only: the prologue ranks
// them (ballot + popcount), deals the 32-row blocks round-robin to the four wave rows (query patches) and the two wave columns
// (template patches) so that every wave gets the same share, stages only those rows (everything else reads as zeros through
// the buffer descriptor's range check), and the k loop is instantiated for MI = ceil(live row blocks / 4) in {1, 2} and
// NI = ceil(live column blocks / 2) in {1..4} matrix tiles per wave instead of always 2 x 4.  Each (query patch, template
// patch) dot product is the same instruction sequence as before, the epilogue maps rows back to patches (match_epilogue<PERM>):
// outputs are bit-identical to the uncompacted kernel.
typedef unsigned int mu32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kMatchOob = 0x80000000u;  // beyond every plane: the buffer load returns zeros without touching memory

struct MatchSplitRsrc {
    __amdgpu_buffer_rsrc_t qh, ql, bh, bl;
    unsigned va0, va1, vb0, vb1;  // byte offsets of this thread's two query rows / two template rows (kMatchOob: no live patch)
};

template <bool BANK_LO, int MI, int NI>
__device__ __forceinline__ void match_split_kloop(MatchSplitSmem& sm, f32x16 (&acc)[2][4], const MatchSplitRsrc& rs, int ns, int tid,
                                                  int lane, int r0, int c0, int grp)  // r0 / c0: first row / column block of the wave's rectangle
{
    constexpr int P_AHI = 0, P_ALO = MS_PLANE, P_BHI = 2 * MS_PLANE, P_BLO = 3 * MS_PLANE, MS_BUF = 4 * MS_PLANE;
    const int wofs = (tid >> 2) * MS_BK + (((tid & 3) ^ (((tid >> 2) >> 2) & 3)) << 3);
    mu32x4 rg[8];
    auto gload = [&](int slab) {
        const unsigned so = (unsigned)slab * (MS_BK * 2u);
        rg[0] = __builtin_amdgcn_raw_buffer_load_b128(rs.qh, rs.va0, so, 0);
        rg[2] = __builtin_amdgcn_raw_buffer_load_b128(rs.ql, rs.va0, so, 0);
        rg[1] = __builtin_amdgcn_raw_buffer_load_b128(rs.qh, rs.va1, so, 0);   // rows 128..255: wave rows 2, 3 (rows without a live patch
        rg[3] = __builtin_amdgcn_raw_buffer_load_b128(rs.ql, rs.va1, so, 0);   // carry the out-of-range offset: zeros, no memory traffic)
        rg[4] = __builtin_amdgcn_raw_buffer_load_b128(rs.bh, rs.vb0, so, 0);
        rg[5] = __builtin_amdgcn_raw_buffer_load_b128(rs.bh, rs.vb1, so, 0);
        if (BANK_LO) {
            rg[6] = __builtin_amdgcn_raw_buffer_load_b128(rs.bl, rs.vb0, so, 0);
            rg[7] = __builtin_amdgcn_raw_buffer_load_b128(rs.bl, rs.vb1, so, 0);
        }
    };
    auto stage = [&](int buf) {
        _Float16* L = sm.stage + buf * MS_BUF + wofs;
        *reinterpret_cast<mu32x4*>(L + P_AHI) = rg[0];
        *reinterpret_cast<mu32x4*>(L + P_ALO) = rg[2];
        *reinterpret_cast<mu32x4*>(L + P_AHI + 128 * MS_BK) = rg[1];
        *reinterpret_cast<mu32x4*>(L + P_ALO + 128 * MS_BK) = rg[3];
        *reinterpret_cast<mu32x4*>(L + P_BHI) = rg[4];
        *reinterpret_cast<mu32x4*>(L + P_BHI + 128 * MS_BK) = rg[5];
        if (BANK_LO) {
            *reinterpret_cast<mu32x4*>(L + P_BLO) = rg[6];
            *reinterpret_cast<mu32x4*>(L + P_BLO + 128 * MS_BK) = rg[7];
        }
    };
    // fragment addressing (rows of the wave's rectangle, XOR-swizzled 16-byte chunks)
    const int ar_ = 32 * r0 + (lane & 31), br_ = 32 * c0 + (lane & 31), kh_ = lane >> 5;
    const int arow = ar_ * MS_BK, brow = br_ * MS_BK;
    const int ak0 = ((kh_ ^ ((ar_ >> 2) & 3)) << 3), ak1 = (((kh_ + 2) ^ ((ar_ >> 2) & 3)) << 3);
    const int bk0 = ((kh_ ^ ((br_ >> 2) & 3)) << 3), bk1 = (((kh_ + 2) ^ ((br_ >> 2) & 3)) << 3);
    gload(0);
    stage(0);
    __builtin_amdgcn_sched_barrier(0);
    if (ns > 1) gload(1);
    __syncthreads();

#define M_MFMA(A_, B_, mi, ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[mi], B_[ni], acc[mi][ni], 0, 0, 0)
#define M_ALL(A_, B_)                                       \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)       \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) M_MFMA(A_, B_, mi, ni)
    auto c_phase = [&](int s) __attribute__((always_inline)) {
        const _Float16* L = sm.stage + (s & 1) * MS_BUF;
        __builtin_amdgcn_s_setprio(1);  // before the fragment reads: they must not queue behind the other group's staging
        h16x8 ah[MI], al[MI], bh[NI], bl[NI], ch[MI], cl[MI], dh[NI], dl[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ah[mi] = *reinterpret_cast<const h16x8*>(L + P_AHI + arow + mi * 32 * MS_BK + ak0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bh[ni] = *reinterpret_cast<const h16x8*>(L + P_BHI + brow + ni * 32 * MS_BK + bk0);
        if (BANK_LO) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bl[ni] = *reinterpret_cast<const h16x8*>(L + P_BLO + brow + ni * 32 * MS_BK + bk0);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) al[mi] = *reinterpret_cast<const h16x8*>(L + P_ALO + arow + mi * 32 * MS_BK + ak0);
        M_ALL(ah, bh);
        if (BANK_LO) { M_ALL(ah, bl); }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ch[mi] = *reinterpret_cast<const h16x8*>(L + P_AHI + arow + mi * 32 * MS_BK + ak1);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) dh[ni] = *reinterpret_cast<const h16x8*>(L + P_BHI + brow + ni * 32 * MS_BK + bk1);
        M_ALL(al, bh);
        if (BANK_LO) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) dl[ni] = *reinterpret_cast<const h16x8*>(L + P_BLO + brow + ni * 32 * MS_BK + bk1);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) cl[mi] = *reinterpret_cast<const h16x8*>(L + P_ALO + arow + mi * 32 * MS_BK + ak1);
        M_ALL(ch, dh);
        if (BANK_LO) { M_ALL(ch, dl); }
        M_ALL(cl, dh);
        __builtin_amdgcn_s_setprio(0);
    };
    auto m_phase = [&](int slab) __attribute__((always_inline)) {  // stages `slab`, loads slab + 1
        if (slab < ns) stage(slab & 1);
        __builtin_amdgcn_sched_barrier(0);
        gload(min(slab + 1, ns - 1));  // unconditional (the last ones re-load an L2-hot slab, unused)
    };
    //     waves 0-3:  C0 M1 | C1 M2 | ...          waves 4-7:  M1 C0 | M2 C1 | ...        (| = the one barrier per k-step)
    if (grp) m_phase(1);
    for (int s = 0; s < ns; ++s) {
        c_phase(s);
        if (grp && s + 1 < ns) __syncthreads();
        m_phase(s + 1 + grp);
        if (!grp && s + 1 < ns) __syncthreads();
    }
#undef M_ALL
#undef M_MFMA
}

#ifndef GP_MATCH_BAND
#define GP_MATCH_BAND 8
#endif
constexpr int kSplitBand = GP_MATCH_BAND;
template <bool BANK_LO, bool TRACE = false>
__global__ __launch_bounds__(512, 2) void match_tiles_split_kernel(
    const _Float16* __restrict__ q_hi, const _Float16* __restrict__ q_lo,  // (B, 256, C)
    const _Float16* __restrict__ b_hi, const _Float16* __restrict__ b_lo,  // (O*N, 256, C)
    const float* __restrict__ qmask, const float* __restrict__ bmask, const int* __restrict__ labels, int B, int O, int N, int C,
    float thr, float patch_thr, int* __restrict__ status, uint8_t* __restrict__ idx_t2s, float* __restrict__ score_t2s,
    float* __restrict__ mask_all, float* __restrict__ sim_avg, unsigned long long* __restrict__ trace_all, int compact, int src2tar)
{
    __shared__ MatchSplitSmem sm;
    const unsigned long long w_in = TRACE ? wall_clock64() : 0;
    const int q = xcd_chunked_tile(blockIdx.x, B * N);
    if (q < 0) return;
    unsigned long long* trace = TRACE ? trace_all + (size_t)q * 8 : nullptr;
    if (TRACE && threadIdx.x == 0) {
        unsigned hw;  // HW_ID: cu 8..11, sh 12, se 13..15 (+ XCC_ID register 20 on gfx94x/95x)
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trace[0] = w_in;
        trace[7] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
    }
    // tile order inside an XCD's chunk: bands of kSplitBand crops x all N templates, crops fastest -- the 32 tiles an XCD runs at a time
    // are kSplitBand crops x 32 / kSplitBand templates (GP_MATCH_BAND: compile-time A/B, tools/gpu_r06_match.sh)
    const int band = q / (kSplitBand * N), r8 = q - band * (kSplitBand * N);
    const int gsz = min(kSplitBand, B - band * kSplitBand);
    const int b = band * kSplitBand + r8 % gsz, n = r8 / gsz;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    int lab = labels[b];
    if ((unsigned)lab >= (unsigned)O) {
        if (tid == 0) gp_raise(status, GP_ST_LABEL_RANGE);
        lab = 0;
    }
    const size_t on = (size_t)lab * N + n;

    // ---- the live patches of this tile, ranked: threads 0..255 = query patches, 256..511 = template patches
    const int pidx = tid & 255;
    const float mv = tid < GP_P ? qmask[(size_t)b * GP_P + pidx] : bmask[on * GP_P + pidx];
    const bool live = compact ? (mv != 0.f) : true;  // compact = 0 (A/B hook): every patch counts as live = the full 2 x 4 tile
    const unsigned long long bal = __ballot(live);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) sm.cnt[wave] = __popcll(bal);
    if (tid < GP_P) {
        sm.qmask[pidx] = mv;
        sm.qmask_p[pidx] = 0.f;
        sm.t_of[pidx] = (short)kDeadPatch;
#pragma unroll
        for (int w = 0; w < 4; ++w) { sm.rowv[w][pidx] = 0.f; sm.rowi[w][pidx] = 0; }
    } else {
        sm.smask[pidx] = mv;
        sm.smask_p[pidx] = 0.f;
        sm.s_of[pidx] = (short)kDeadPatch;
#pragma unroll
        for (int w = 0; w < 4; ++w) { sm.colv[w][pidx] = 0.f; sm.coli[w][pidx] = 0; }
    }
    __syncthreads();
    int rank = before;
    for (int w = 4 * grp; w < wave; ++w) rank += sm.cnt[w];
    if (live) {
        // live patch number `rank` -> LDS row `rank`: block rank >> 5 of the compacted tile (which wave multiplies which blocks is the
        // table's business, gp_match_rects.h)
        if (tid < GP_P) {
            sm.t_of[rank] = (short)pidx;
            sm.qmask_p[rank] = mv;
        } else {
            sm.s_of[rank] = (short)pidx;
            sm.smask_p[rank] = mv;
        }
    }
    __syncthreads();
    const int live_t = sm.cnt[0] + sm.cnt[1] + sm.cnt[2] + sm.cnt[3], live_s = sm.cnt[4] + sm.cnt[5] + sm.cnt[6] + sm.cnt[7];
    // Which 32 x 32 blocks this wave multiplies (round 4).  Round 3 dealt the live row blocks to the four wave rows and the live column
    // blocks to the two wave columns: 5 x 5 live blocks (the benchmark's disc masks) cost every wave 2 x 3 = 6 blocks, 48 for 25.  Now
    // the nrb x ncb live blocks are cut into at most eight RECTANGLES (1-2 row blocks x 1-4 column blocks, one per wave), chosen so
    // that the two waves of a SIMD together hold as few blocks as possible: 7 instead of 12 at 5 x 5.  Each dot product is the same
    // instruction sequence wherever it is computed: outputs stay bit-identical.
    // wave-uniform BY CONSTRUCTION (readfirstlane): the k-loop instantiation below is picked per wave and holds barriers, so its
    // selector must live in a scalar register -- a selector the compiler has to treat as divergent (LDS reads + a table load) would
    // put __syncthreads under a divergent switch
    const int nrb = __builtin_amdgcn_readfirstlane((live_t + 31) >> 5), ncb = __builtin_amdgcn_readfirstlane((live_s + 31) >> 5);
    const MatchWaveTile wt = match_wave_tile_unpack((unsigned)__builtin_amdgcn_readfirstlane((int)kMatchRect[nrb][ncb][wave]));

    // staging: thread = (row tid >> 2 [+128], 16-byte k-chunk tid & 3) of each plane; one descriptor per plane of THIS tile
    const unsigned plane_bytes = (unsigned)GP_P * (unsigned)C * 2u;
    MatchSplitRsrc rs;
    rs.qh = __builtin_amdgcn_make_buffer_rsrc((void*)(q_hi + (size_t)b * GP_P * C), 0, plane_bytes, 0x00020000);
    rs.ql = __builtin_amdgcn_make_buffer_rsrc((void*)(q_lo + (size_t)b * GP_P * C), 0, plane_bytes, 0x00020000);
    rs.bh = __builtin_amdgcn_make_buffer_rsrc((void*)(b_hi + on * GP_P * C), 0, plane_bytes, 0x00020000);
    rs.bl = __builtin_amdgcn_make_buffer_rsrc((void*)((BANK_LO ? b_lo : b_hi) + on * GP_P * C), 0, plane_bytes, 0x00020000);
    {
        const int r0 = tid >> 2, r1 = r0 + 128;
        const unsigned ck = (unsigned)(tid & 3) * 16u, rowb = (unsigned)C * 2u;
        const int t0 = sm.t_of[r0], t1 = sm.t_of[r1], s0 = sm.s_of[r0], s1 = sm.s_of[r1];
        rs.va0 = t0 != kDeadPatch ? (unsigned)t0 * rowb + ck : kMatchOob;
        rs.va1 = t1 != kDeadPatch ? (unsigned)t1 * rowb + ck : kMatchOob;
        rs.vb0 = s0 != kDeadPatch ? (unsigned)s0 * rowb + ck : kMatchOob;
        rs.vb1 = s1 != kDeadPatch ? (unsigned)s1 * rowb + ck : kMatchOob;
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int ns = C / MS_BK;
    if (TRACE && tid == 0) trace[1] = wall_clock64();
    if (nrb > 0 && ncb > 0) {  // uniform over the workgroup; inside, every wave runs the instantiation of ITS rectangle (a scalar branch:
        // the barriers of the k loop count waves, not code addresses -- all instantiations execute the same number per step)
        const int sel = wt.mi_n * 8 + wt.ni_n;
#define M_CASE(MI_, NI_) case MI_ * 8 + NI_: match_split_kloop<BANK_LO, MI_, NI_>(sm, acc, rs, ns, tid, lane, wt.r0, wt.c0, grp); break
        switch (sel) {
            M_CASE(1, 1); M_CASE(1, 2); M_CASE(1, 3); M_CASE(1, 4);
            M_CASE(2, 1); M_CASE(2, 2); M_CASE(2, 3);
            default: match_split_kloop<BANK_LO, 2, 4>(sm, acc, rs, ns, tid, lane, wt.r0, wt.c0, grp); break;
        }
#undef M_CASE
    }
    __syncthreads();  // the epilogue reuses nothing of `stage`, but its first LDS writes must follow every wave's mask reads
    if (TRACE && tid == 0) trace[2] = wall_clock64();

    constexpr float inv = 1.0f / (kFeatScale * kFeatScale);  // exact power of two
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= inv;
    match_epilogue<true>(sm, acc, b, n, N, thr, patch_thr, idx_t2s, score_t2s, mask_all, sim_avg, trace, src2tar, compact, wt);
}

// norms of gp_l2norm_cp (same sequential fma over c), then x / d * 32 split into f16 planes [row][patch][C].
// grid (rows, nchunk): a block recomputes the full norm of its 256 patches (the price of keeping the sequential chain) and
// then converts its share of the channels, 32 at a time through an LDS transpose.  nchunk = 4 at C = 1024: with one block
// per 32 channels (the first version) 64 query rows re-read their 1 MB 32 times each -- 0.5-0.66 ms per step for a 67 MB
// conversion (rocprofv3, profiles/r02_kernel_stats.txt); 4 chunks keep 256 CUs busy at an eighth of the traffic.
__global__ __launch_bounds__(256) void l2norm_split_kernel(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                            _Float16* __restrict__ lo, int C, int Cp,
                                                            const float* __restrict__ mask_img = nullptr, int mh = 0, int mw = 0,
                                                            float* __restrict__ patch_mask = nullptr)
{
    __shared__ float t[32][GP_P + 1];
    __shared__ float dn[GP_P];
    const int row = blockIdx.x, p = threadIdx.x;
    // (round 5) the row's 16 x 16 patch mask = F.interpolate(mask, (16, 16)) nearest (matching.py:222, 227) = pixel (i H / 16, j W / 16) of
    // its H x W mask, written by the row's first block: the strided copy was the one ATen launch left between the ViT and the matcher
    if (patch_mask && blockIdx.y == 0)
        patch_mask[(size_t)row * GP_P + p] = mask_img[(size_t)row * mh * mw + (size_t)((p >> 4) * (mh / GP_G)) * mw + (p & 15) * (mw / GP_G)];
    const float* xr = x + (size_t)row * C * GP_P;
    float ss = 0.f;
    int c = 0;
    for (; c + 16 <= C; c += 16) {  // 16 independent loads in flight, then the fmas in order (same chain as gp_l2norm_cp)
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = xr[(size_t)(c + u) * GP_P + p];
#pragma unroll
        for (int u = 0; u < 16; ++u) ss = __builtin_fmaf(v[u], v[u], ss);
    }
    for (; c < C; ++c) {
        const float v = xr[(size_t)c * GP_P + p];
        ss = __builtin_fmaf(v, v, ss);
    }
    dn[p] = fmaxf(__builtin_sqrtf(ss), 1e-12f);
    // this block's channel range: 32-channel groups [g0, g1) of Cp / 32 (Cp = round_up(C, 32): planes are zero-padded along C)
    const int ngrp = Cp / 32, g0 = ngrp * blockIdx.y / gridDim.y, g1 = ngrp * (blockIdx.y + 1) / gridDim.y;
    for (int grp = g0; grp < g1; ++grp) {
        const int c0 = grp * 32;
        __syncthreads();  // dn visible (first round) / previous round's reads of t done
        for (int r = 0; r < 32; ++r) t[r][p] = (c0 + r < C) ? xr[(size_t)(c0 + r) * GP_P + p] : 0.f;  // coalesced along p
        __syncthreads();
        // thread -> (patch pp, 8-channel group g): 256 patches x 4 groups = 1024 items, 4 per thread
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int item = p + 256 * u, pp = item >> 2, g = item & 3;
            h16x8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (t[g * 8 + e][pp] / dn[pp]) * kFeatScale;
                const _Float16 hh = (_Float16)v;
                h[e] = hh;
                l[e] = (_Float16)(v - (float)hh);
            }
            const size_t o = ((size_t)row * GP_P + pp) * Cp + c0 + g * 8;
            *reinterpret_cast<h16x8*>(hi + o) = h;
            *reinterpret_cast<h16x8*>(lo + o) = l;
        }
    }
}

// ------------------------------------------------------------------ top-k per detection
// One wave per detection.  Order: higher score, then lower template index (torch.topk leaves
// ties unspecified; the oracle uses the same rule).
__global__ __launch_bounds__(64) void topk_kernel(const float* __restrict__ sim_avg, int N, int k,
                                                   int* __restrict__ ids, float* __restrict__ scores)
{
    extern __shared__ unsigned char taken[];  // N flags
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int n = lane; n < N; n += 64) taken[n] = 0;
    __syncthreads();
    const float* v = sim_avg + (size_t)b * N;
    for (int j = 0; j < k; ++j) {
        float bv = 0.f;
        int bi = 0x7fffffff;
        for (int n = lane; n < N; n += 64) {
            if (taken[n]) continue;
            const float x = v[n];
            if (bi == 0x7fffffff || x > bv) { bv = x; bi = n; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        if (lane == 0) {
            ids[(size_t)b * k + j] = bi;
            scores[(size_t)b * k + j] = bv;
            taken[bi] = 1;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ gather candidate records
__global__ __launch_bounds__(256) void gather_records_kernel(
    const int* __restrict__ ids, const uint8_t* __restrict__ idx_t2s,
    const float* __restrict__ score_t2s, const float* __restrict__ mask_all, int N, int k,
    uint8_t* __restrict__ rec_idx, float* __restrict__ rec_score, float* __restrict__ rec_mask)
{
    const int bk = blockIdx.x, b = bk / k, t = threadIdx.x;
    const size_t src = ((size_t)b * N + ids[bk]) * GP_P + t;
    const size_t dst = (size_t)bk * GP_P + t;
    rec_idx[dst] = idx_t2s[src];
    rec_score[dst] = score_t2s[src];
    rec_mask[dst] = mask_all[src];
}

// ------------------------------------------------------------------ format_prediction
__global__ __launch_bounds__(256) void format_points_kernel(const uint8_t* __restrict__ rec_idx,
                                                             const float* __restrict__ rec_mask,
                                                             long long* __restrict__ tar_pts,
                                                             long long* __restrict__ src_pts)
{
    const size_t i = (size_t)blockIdx.x * GP_P + threadIdx.x;
    const int t = threadIdx.x;
    const bool valid = rec_mask[i] != 0.f;  // torch.nonzero(mask)  (matching.py:42)
    const int js = rec_idx[i];
    tar_pts[2 * i + 0] = valid ? (t % GP_G) : -1;
    tar_pts[2 * i + 1] = valid ? (t / GP_G) : -1;
    src_pts[2 * i + 0] = valid ? (js % GP_G) : -1;
    src_pts[2 * i + 1] = valid ? (js / GP_G) : -1;
}

// ------------------------------------------------------------------ top-k + gather + format in ONE launch (round 5)
// What LocalSimilarity.test does after the tiles (matching.py:279-316: topk over templates, gather the winners' records,
// format_prediction) took four launches here (topk, gather_records, format_points, a torch int32 -> int64 cast of the ids): per step
// that is ~25 us of queue time for ~20 us of work.  One block per detection: wave 0 selects the k winners with topk_kernel's rule
// (higher score, then lower template index), then the block copies each winner's record and writes the point lists.  Same values
// as the three kernels above (tests compare them).
__global__ __launch_bounds__(256) void select_topk_kernel(const float* __restrict__ sim_avg, const uint8_t* __restrict__ idx_t2s,
                                                           const float* __restrict__ score_t2s, const float* __restrict__ mask_all, int N, int k,
                                                           long long* __restrict__ ids64, float* __restrict__ scores,
                                                           float* __restrict__ rec_score, long long* __restrict__ tar_pts,
                                                           long long* __restrict__ src_pts)
{
    extern __shared__ unsigned char taken[];  // N flags, then (4-byte aligned) k winner ids
    int* win = reinterpret_cast<int*>(taken + ((N + 3) & ~3));
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63;
    for (int n = t; n < N; n += 256) taken[n] = 0;
    __syncthreads();
    const float* v = sim_avg + (size_t)b * N;
    if (t < 64) {   // wave 0 only: its LDS operations are in program order, no barrier inside the selection loop
        for (int j = 0; j < k; ++j) {
            float bv = 0.f;
            int bi = 0x7fffffff;
            for (int n = lane; n < N; n += 64) {
                if (taken[n]) continue;
                const float x = v[n];
                if (bi == 0x7fffffff || x > bv) { bv = x; bi = n; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
            }
            if (lane == 0) {
                ids64[(size_t)b * k + j] = bi;
                scores[(size_t)b * k + j] = bv;
                taken[bi] = 1;
                win[j] = bi;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // lane 0's `taken` write is seen by the wave's next scan
        }
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        const size_t src = ((size_t)b * N + win[j]) * GP_P + t;
        const size_t i = ((size_t)b * k + j) * GP_P + t;
        const bool valid = mask_all[src] != 0.f;  // torch.nonzero(mask)  (matching.py:42)
        const int js = idx_t2s[src];
        rec_score[i] = score_t2s[src];
        tar_pts[2 * i + 0] = valid ? (t % GP_G) : -1;
        tar_pts[2 * i + 1] = valid ? (t / GP_G) : -1;
        src_pts[2 * i + 0] = valid ? (js % GP_G) : -1;
        src_pts[2 * i + 1] = valid ? (js / GP_G) : -1;
    }
}

}  // namespace

extern "C" {

int gp_l2norm_cp(const float* x, float* out, int rows, int C, void* stream)
{
    GP_REQUIRE(rows >= 0 && C > 0, "gp_l2norm_cp: bad arguments (rows=%d C=%d)", rows, C);
    if (rows == 0) return GP_OK;  // empty batch: nothing to do (pointers may be NULL)
    GP_REQUIRE(x && out, "gp_l2norm_cp: null pointer");
    hipLaunchKernelGGL(l2norm_cp_kernel, dim3(rows, (C + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, out, C);
    GP_CHECK_LAUNCH("gp_l2norm_cp");
    return GP_OK;
}

int gp_match_tiles_dir(const float* query, const float* bank, const float* qmask, const float* bmask,
                       const int* labels, int B, int O, int N, int C, float sim_threshold,
                       float patch_threshold, int search_direction, uint8_t* idx_t2s, float* score_t2s, float* mask_all,
                       float* sim_avg, void* stream)
{
    GP_REQUIRE(search_direction == 0 || search_direction == 1, "gp_match_tiles: search_direction must be 0 (tar2src) or 1 (src2tar)");
    GP_REQUIRE(B >= 0 && O > 0 && N > 0, "gp_match_tiles: bad sizes B=%d O=%d N=%d", B, O, N);
    GP_REQUIRE(C > 0 && C % 16 == 0, "gp_match_tiles: C=%d must be a positive multiple of 16", C);
    if (B == 0) return GP_OK;
    GP_REQUIRE(query && bank && qmask && bmask && labels && idx_t2s && score_t2s && mask_all && sim_avg,
               "gp_match_tiles: null pointer");
    GpProfScope prof(GP_PROF_MATCH, 2.0 * B * N * 256.0 * 256.0 * C, (hipStream_t)stream);
    hipLaunchKernelGGL(match_tiles_kernel, dim3(xcd_chunked_grid(B * N)), dim3(512), 0,
                       (hipStream_t)stream, query, bank, qmask, bmask, labels, B, O, N, C, sim_threshold,
                       patch_threshold, gp_status_buffer(), idx_t2s, score_t2s, mask_all, sim_avg, search_direction);
    GP_CHECK_LAUNCH("gp_match_tiles");
    return GP_OK;
}

int gp_l2norm_split_mask(const float* x, void* hi, void* lo, int rows, int C, const float* mask_img, int mask_h, int mask_w,
                         float* patch_mask, void* stream)
{
    GP_REQUIRE(rows >= 0 && C > 0, "gp_l2norm_split: bad arguments (rows=%d C=%d)", rows, C);
    if (rows == 0) return GP_OK;
    GP_REQUIRE(x && hi && lo, "gp_l2norm_split: null pointer");
    GP_REQUIRE(!patch_mask || (mask_img && mask_h > 0 && mask_w > 0 && mask_h % GP_G == 0 && mask_w % GP_G == 0),
               "gp_l2norm_split_mask: the mask image must be (rows, H, W) f32 with H, W multiples of 16 (got %d x %d)", mask_h, mask_w);
    const int ngrp = (C + 31) / 32;
    hipLaunchKernelGGL(l2norm_split_kernel, dim3(rows, ngrp < 4 ? ngrp : 4), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)hi,
                       (_Float16*)lo, C, ngrp * 32, mask_img, mask_h, mask_w, patch_mask);
    GP_CHECK_LAUNCH("gp_l2norm_split");
    return GP_OK;
}

static int g_match_compact = 1;  // 0: every patch treated as live = the full 256 x 256 tile (A/B hook: gp_match_split_set_compact)

static int match_tiles_split_launch(const void* q_hi, const void* q_lo, const void* b_hi, const void* b_lo, const float* qmask,
                                    const float* bmask, const int* labels, int B, int O, int N, int C, float sim_threshold,
                                    float patch_threshold, uint8_t* idx_t2s, float* score_t2s, float* mask_all, float* sim_avg,
                                    unsigned long long* trace, void* stream, int search_direction = 0)
{
    GP_REQUIRE(search_direction == 0 || search_direction == 1, "gp_match_tiles_split: search_direction must be 0 (tar2src) or 1 (src2tar)");
    // Live-patch compaction is bit-identical to the full tile only when every surviving similarity is >= 0 (match_epilogue<PERM>: a
    // row maximum of 0 then means an all-zero row, argmax 0).  With a negative threshold live values in [thr, 0) survive and the
    // masked-out patches' exact zeros win the reference's maximum: such calls run the full 256 x 256 tile.
    const int compact = (g_match_compact && sim_threshold >= 0.f) ? 1 : 0;
    GP_REQUIRE(B >= 0 && O > 0 && N > 0, "gp_match_tiles_split: bad sizes B=%d O=%d N=%d", B, O, N);
    GP_REQUIRE(C > 0 && C % 32 == 0, "gp_match_tiles_split: C=%d must be a positive multiple of 32", C);
    if (B == 0) return GP_OK;
    GP_REQUIRE(q_hi && q_lo && b_hi && qmask && bmask && labels && idx_t2s && score_t2s && mask_all && sim_avg,
               "gp_match_tiles_split: null pointer");
    GP_REQUIRE((long long)GP_P * C * 2 < (1ll << 31), "gp_match_tiles_split: C too large");
    GpProfScope prof(GP_PROF_MATCH_SPLIT, 2.0 * B * N * 256.0 * 256.0 * C, (hipStream_t)stream);
#define GP_MATCH_SPLIT_LAUNCH(LO, TR)                                                                                             \
    hipLaunchKernelGGL((match_tiles_split_kernel<LO, TR>), dim3(xcd_chunked_grid(B * N)), dim3(512), 0, (hipStream_t)stream,      \
                       (const _Float16*)q_hi, (const _Float16*)q_lo, (const _Float16*)b_hi, (const _Float16*)b_lo, qmask, bmask,  \
                       labels, B, O, N, C, sim_threshold, patch_threshold, gp_status_buffer(), idx_t2s, score_t2s, mask_all,      \
                       sim_avg, trace, compact, search_direction)
    if (trace) {
        GP_REQUIRE(b_lo, "gp_match_tiles_split_trace: the probe build takes the two-plane bank");
        GP_MATCH_SPLIT_LAUNCH(true, true);
    } else if (b_lo) {
        GP_MATCH_SPLIT_LAUNCH(true, false);
    } else {  // fp16 bank: hi plane only
        GP_MATCH_SPLIT_LAUNCH(false, false);
    }
#undef GP_MATCH_SPLIT_LAUNCH
    GP_CHECK_LAUNCH("gp_match_tiles_split");
    return GP_OK;
}

int gp_match_tiles_split_dir(const void* q_hi, const void* q_lo, const void* b_hi, const void* b_lo, const float* qmask,
                             const float* bmask, const int* labels, int B, int O, int N, int C, float sim_threshold,
                             float patch_threshold, int search_direction, uint8_t* idx_t2s, float* score_t2s, float* mask_all,
                             float* sim_avg, void* stream)
{
    return match_tiles_split_launch(q_hi, q_lo, b_hi, b_lo, qmask, bmask, labels, B, O, N, C, sim_threshold, patch_threshold,
                                    idx_t2s, score_t2s, mask_all, sim_avg, nullptr, stream, search_direction);
}

#ifdef GP_PROBES
// probe build: trace[(tile q) * 8 + i] = 100 MHz wall-clock stamps (0 entry, 1 first slab staged, 2 k loop done, 3 maxima,
// 4 merge, 5 per-patch outputs, 6 end) and [7] = XCC_ID << 32 | HW_ID (tools/probe_match_fixed.py)
int gp_match_tiles_split_trace(const void* q_hi, const void* q_lo, const void* b_hi, const void* b_lo, const float* qmask,
                               const float* bmask, const int* labels, int B, int O, int N, int C, float sim_threshold,
                               float patch_threshold, uint8_t* idx_t2s, float* score_t2s, float* mask_all, float* sim_avg,
                               unsigned long long* trace, void* stream)
{
    GP_REQUIRE(trace, "gp_match_tiles_split_trace: null trace buffer");
    return match_tiles_split_launch(q_hi, q_lo, b_hi, b_lo, qmask, bmask, labels, B, O, N, C, sim_threshold, patch_threshold,
                                    idx_t2s, score_t2s, mask_all, sim_avg, trace, stream);
}

int gp_match_split_set_compact(int on)
{
    g_match_compact = on ? 1 : 0;
    return GP_OK;
}
#endif

int gp_topk(const float* sim_avg, int B, int N, int k, int* ids, float* scores, void* stream)
{
    // torch.topk raises when k > N (matching.py:279); so do we.
    GP_REQUIRE(k >= 1 && k <= N, "gp_topk: selected index k out of range (k=%d, N=%d)", k, N);
    if (B == 0) return GP_OK;
    GP_REQUIRE(sim_avg && ids && scores, "gp_topk: null pointer");
    hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(64), (size_t)N, (hipStream_t)stream, sim_avg, N, k, ids, scores);
    GP_CHECK_LAUNCH("gp_topk");
    return GP_OK;
}

int gp_select_topk(const float* sim_avg, const uint8_t* idx_t2s, const float* score_t2s, const float* mask_all, int B, int N, int k,
                   long long* ids, float* scores, float* rec_score, long long* tar_pts, long long* src_pts, void* stream)
{
    GP_REQUIRE(k >= 1 && k <= N, "gp_select_topk: selected index k out of range (k=%d, N=%d)", k, N);   // torch.topk raises (matching.py:279)
    if (B == 0) return GP_OK;
    GP_REQUIRE(sim_avg && idx_t2s && score_t2s && mask_all && ids && scores && rec_score && tar_pts && src_pts, "gp_select_topk: null pointer");
    hipLaunchKernelGGL(select_topk_kernel, dim3(B), dim3(256), (size_t)((N + 3) & ~3) + sizeof(int) * (size_t)k, (hipStream_t)stream, sim_avg,
                       idx_t2s, score_t2s, mask_all, N, k, ids, scores, rec_score, tar_pts, src_pts);
    GP_CHECK_LAUNCH("gp_select_topk");
    return GP_OK;
}

int gp_gather_records(const int* ids, const uint8_t* idx_t2s, const float* score_t2s,
                      const float* mask_all, int B, int N, int k, uint8_t* rec_idx, float* rec_score,
                      float* rec_mask, void* stream)
{
    GP_REQUIRE(k >= 1 && N >= 1, "gp_gather_records: bad sizes");
    if (B == 0) return GP_OK;
    GP_REQUIRE(ids && idx_t2s && score_t2s && mask_all && rec_idx && rec_score && rec_mask,
               "gp_gather_records: null pointer");
    hipLaunchKernelGGL(gather_records_kernel, dim3(B * k), dim3(256), 0, (hipStream_t)stream, ids,
                       idx_t2s, score_t2s, mask_all, N, k, rec_idx, rec_score, rec_mask);
    GP_CHECK_LAUNCH("gp_gather_records");
    return GP_OK;
}

int gp_format_points(const uint8_t* rec_idx, const float* rec_mask, int rows, long long* tar_pts,
                     long long* src_pts, void* stream)
{
    if (rows <= 0) return GP_OK;
    GP_REQUIRE(rec_idx && rec_mask && tar_pts && src_pts, "gp_format_points: null pointer");
    hipLaunchKernelGGL(format_points_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, rec_idx,
                       rec_mask, tar_pts, src_pts);
    GP_CHECK_LAUNCH("gp_format_points");
    return GP_OK;
}

}  // extern "C"
