/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the GigaPose coarse-pose hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product path (gigapose_amd/) never links,
 * imports or calls it.
 *
 * Every function cites the reference file:line (nv-nguyen/gigapose) it restates.  Where
 * the reference leaves the floating-point evaluation order to a BLAS / vectorised torch
 * kernel, this file FIXES an order (documented per function) and the HIP kernels are
 * written to the same order, so HIP-vs-oracle comparisons are bit-exact for indices AND
 * floats; oracle-vs-reference is pinned by tests/golden/ (indices exact, floats to 1e-6).
 *
 * Arithmetic conventions shared with gigapose_amd/csrc:
 *   - built with -ffp-contract=off: a*b+c is NEVER fused unless written fmaf();
 *   - dot products over channels: acc = 0; for c ascending: acc = fmaf(a[c], b[c], acc)
 *     (this is bit-for-bit what v_mfma_f32_32x32x2_f32 produces when k-pairs are issued in
 *     ascending order; see MI355X guide "FP32-input MFMA ... k-ordered fmaf chain").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define P 256 /* patches per image, 16x16 (reference matching.py:26, image 224 / patch 14) */
#define G 16  /* patch grid side */

int oracle_abi_version(void) { return 1; }

/* ------------------------------------------------------------------------------------
 * F.normalize(x, dim=C) on a (rows, C, P) channel-major tensor.
 * Reference: ae_net.py:69, matching.py:224,229  (x / max(||x||_2, 1e-12)).
 * Fixed order: ss = fmaf(x_c, x_c, ss) for c ascending; sqrtf; IEEE division.
 * ---------------------------------------------------------------------------------- */
void oracle_l2norm_cp(const float* x, float* out, int rows, int C)
{
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float* xr = x + (size_t)r * C * P;
        float* orow = out + (size_t)r * C * P;
        for (int p = 0; p < P; ++p) {
            float ss = 0.f;
            for (int c = 0; c < C; ++c) { float v = xr[(size_t)c * P + p]; ss = fmaf(v, v, ss); }
            float d = fmaxf(sqrtf(ss), 1e-12f);
            for (int c = 0; c < C; ++c) orow[(size_t)c * P + p] = xr[(size_t)c * P + p] / d;
        }
    }
}

/* ------------------------------------------------------------------------------------
 * One (detection, template) tile of LocalSimilarity.test, steps 3-8 of SURVEY 3.4.
 * Reference: matching.py:233-278 (+ find_consistency_patches :80-113).
 *   q      (C, P)  query features, already matcher-normalised
 *   s      (C, P)  template features, already matcher-normalised
 *   qmask  (P)     query patch mask   (nearest-sampled, matching.py:222)
 *   smask  (P)     template patch mask (matching.py:227)
 * outputs: idx_t2s (P) u8, score_t2s (P), mask_all (P), *sim_avg
 * ---------------------------------------------------------------------------------- */
static void match_tile(const float* q, const float* s, const float* qmask, const float* smask,
                       int C, float thr, float patch_thr, int src2tar,
                       uint8_t* idx_t2s, float* score_t2s, float* mask_all, float* sim_avg,
                       float* sim /* scratch P*P */)
{
    /* sim[t][s] = sum_c q[c][t]*s[c][s]   (matching.py:233), sequential fmaf chain */
    for (int i = 0; i < P * P; ++i) sim[i] = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* qc = q + (size_t)c * P;
        const float* sc = s + (size_t)c * P;
        for (int t = 0; t < P; ++t) {
            float a = qc[t];
            float* row = sim + (size_t)t * P;
            for (int j = 0; j < P; ++j) row[j] = fmaf(a, sc[j], row[j]);
        }
    }
    /* sim *= src_mask; sim *= tar_mask; sim[sim < thr] = 0   (matching.py:234-236) */
    for (int t = 0; t < P; ++t)
        for (int j = 0; j < P; ++j) {
            float v = sim[t * P + j] * smask[j];
            v = v * qmask[t];
            if (v < thr) v = 0.f;
            sim[t * P + j] = v;
        }
    /* torch.max over s and over t: first maximal index (matching.py:239-244).  "A" is what the reference calls tar2src, "B" its
     * src2tar: search_direction == "tar2src" (:239-241): A = maxima over s per query patch t, B = maxima over t per template patch
     * s; "src2tar" (:242-244): the two exchanged -- every later step indexes them by POSITION p = 0..255 whatever p means. */
    float sc_row[P]; int id_row[P]; float sc_col[P]; int id_col[P];
    for (int t = 0; t < P; ++t) {
        float best = sim[t * P]; int bi = 0;
        for (int j = 1; j < P; ++j) if (sim[t * P + j] > best) { best = sim[t * P + j]; bi = j; }
        sc_row[t] = best; id_row[t] = bi;
    }
    for (int j = 0; j < P; ++j) {
        float best = sim[j]; int bi = 0;
        for (int t = 1; t < P; ++t) if (sim[t * P + j] > best) { best = sim[t * P + j]; bi = t; }
        sc_col[j] = best; id_col[j] = bi;
    }
    const float* sc_a = src2tar ? sc_col : sc_row; const int* id_a = src2tar ? id_col : id_row;
    const float* sc_b = src2tar ? sc_row : sc_col; const int* id_b = src2tar ? id_row : id_col;
    /* masks (matching.py:247-271) */
    float acc = 0.f, cnt = 0.f;
    for (int t = 0; t < P; ++t) {
        int js = id_a[t];
        int mask_sim = sc_a[t] >= thr;                                    /* :247 */
        int mask_cycle = 1;                                               /* :257 patch_threshold <= 0: ones */
        if (patch_thr > 0.f) {                                            /* :250-255 find_consistency_patches */
            int t2 = id_b[js];                                            /* :96 gather */
            float dx = (float)(t2 % G) - (float)(t % G);                  /* :98-99 (x=w, y=h) */
            float dy = (float)(t2 / G) - (float)(t / G);
            float dist = sqrtf(dx * dx + dy * dy);                        /* torch.norm :100 */
            int mask_dist = dist <= patch_thr;                            /* :104 */
            int mask_sim2 = sc_b[js] >= thr;                              /* :107-108 */
            mask_cycle = mask_dist && mask_sim2;
        }
        /* mask_non_zero (:263-268); NOTE quirks: (idx_src2tar != 0) is indexed by position t, and in either direction
         * tar_mask is taken at position t and src_mask at the matched index (:260-261) */
        float nz = qmask[t] * smask[js];
        nz = nz * (float)(id_b[t] != 0);
        nz = nz * (float)(id_a[t] != 0);
        float m = (float)(mask_sim && mask_cycle) * nz;                   /* :271 */
        mask_all[t] = m;
        idx_t2s[t] = (uint8_t)js;
        score_t2s[t] = sc_a[t];
        acc = acc + sc_a[t] * m;  /* fixed order: sequential over t (reference: torch.sum) */
        cnt = cnt + m;
    }
    *sim_avg = (cnt > 0.f) ? acc / 256.0f : 0.f;                          /* :274-278 */
}

/* All (b, n) tiles.  labels are 0-based object indices (reference uses label-1,
 * gigaPose.py:520-521).  bank (O,N,C,P), bmask (O,N,P), query (B,C,P), qmask (B,P). */
void oracle_match_dir(const float* query, const float* bank, const float* qmask, const float* bmask,
                      const int32_t* labels, int B, int O, int N, int C, float thr, float patch_thr, int src2tar,
                      uint8_t* idx_t2s, float* score_t2s, float* mask_all, float* sim_avg)
{
    (void)O;
#pragma omp parallel
    {
        float* sim = (float*)malloc(sizeof(float) * P * P);
#pragma omp for schedule(dynamic) collapse(2)
        for (int b = 0; b < B; ++b)
            for (int n = 0; n < N; ++n) {
                size_t o = (size_t)labels[b];
                size_t bn = (size_t)b * N + n;
                match_tile(query + (size_t)b * C * P, bank + (o * N + n) * (size_t)C * P,
                           qmask + (size_t)b * P, bmask + (o * N + n) * P, C, thr, patch_thr, src2tar,
                           idx_t2s + bn * P, score_t2s + bn * P, mask_all + bn * P, sim_avg + bn, sim);
            }
        free(sim);
    }
}

/* search_direction = "tar2src", the reference default (matching.py:17) */
void oracle_match(const float* query, const float* bank, const float* qmask, const float* bmask,
                  const int32_t* labels, int B, int O, int N, int C, float thr, float patch_thr,
                  uint8_t* idx_t2s, float* score_t2s, float* mask_all, float* sim_avg)
{
    oracle_match_dir(query, bank, qmask, bmask, labels, B, O, N, C, thr, patch_thr, 0, idx_t2s, score_t2s, mask_all, sim_avg);
}

/* torch.topk(sim_avg, k, dim=1) (matching.py:279).  Tie order is unspecified in torch;
 * this implementation (and the HIP kernel) define: higher score first, then LOWER index. */
void oracle_topk(const float* sim_avg, int B, int N, int k, int32_t* ids, float* scores)
{
    for (int b = 0; b < B; ++b) {
        const float* v = sim_avg + (size_t)b * N;
        for (int j = 0; j < k; ++j) {
            int bi = -1; float best = 0.f;
            for (int n = 0; n < N; ++n) {
                int taken = 0;
                for (int jj = 0; jj < j; ++jj) if (ids[b * k + jj] == n) taken = 1;
                if (taken) continue;
                if (bi < 0 || v[n] > best) { best = v[n]; bi = n; }
            }
            ids[b * k + j] = bi; scores[b * k + j] = best;
        }
    }
}

/* Gather per-candidate records + format_prediction (matching.py:282-315, :29-61, :63-68).
 * tar_pts/src_pts are (B,k,P,2) int64 (x, y), -1 where mask_all == 0. */
void oracle_gather_format(const int32_t* ids, const uint8_t* idx_t2s, const float* score_t2s,
                          const float* mask_all, int B, int N, int k,
                          float* score_pts, int64_t* tar_pts, int64_t* src_pts)
{
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < k; ++j) {
            size_t bn = (size_t)b * N + ids[b * k + j];
            size_t bk = (size_t)b * k + j;
            for (int t = 0; t < P; ++t) {
                int valid = mask_all[bn * P + t] != 0.f;
                int js = idx_t2s[bn * P + t];
                score_pts[bk * P + t] = score_t2s[bn * P + t];
                tar_pts[(bk * P + t) * 2 + 0] = valid ? (t % G) : -1;
                tar_pts[(bk * P + t) * 2 + 1] = valid ? (t / G) : -1;
                src_pts[(bk * P + t) * 2 + 0] = valid ? (js % G) : -1;
                src_pts[(bk * P + t) * 2 + 1] = valid ? (js / G) : -1;
            }
        }
}

/* ------------------------------------------------------------------------------------
 * k-major GEMM with the epilogues of gigapose_amd/csrc/gp_gemm.hip:
 *   D[i][j] = epi( fmaf-chain_k A[k][i]*B[k][j] ).
 * Restates torch.nn.Linear as used by the DINOv2 block (HF modeling_dinov2.py:199-297) and the
 * IST regressor (reference ist_net.py:140-155) with a FIXED accumulation order (k ascending).
 * epi: 0 none, 1 +bias[i], 2 gelu_erf(+bias[i]), 3 res + scale[i]*(acc+bias[i]), 4 +bias[j],
 *      5 relu(+bias[i]).
 * ---------------------------------------------------------------------------------- */
void oracle_gemm_kmajor(const float* A, int lda, const float* B, int ldb, float* D, int ldd,
                        int I, int J, int K, int epi, const float* bias, const float* scale,
                        const float* res, int ldr)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < I; ++i) {
        float* acc = (float*)calloc((size_t)J, sizeof(float));
        for (int k = 0; k < K; ++k) {
            const float a = A[(size_t)k * lda + i];
            const float* b = B + (size_t)k * ldb;
            for (int j = 0; j < J; ++j) acc[j] = fmaf(a, b[j], acc[j]);
        }
        for (int j = 0; j < J; ++j) {
            float v = acc[j];
            if (epi == 1 || epi == 2 || epi == 3 || epi == 5) v = v + bias[i];
            if (epi == 4) v = v + bias[j];
            if (epi == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            if (epi == 5) v = fmaxf(v, 0.f);
            if (epi == 3) v = res[(size_t)i * ldr + j] + scale[i] * v;
            D[(size_t)i * ldd + j] = v;
        }
        free(acc);
    }
}

/* ------------------------------------------------------------------------------------
 * ISTNet.inference (reference ist_net.py:97-120) for ONE head on gathered features.
 * feats (R, 2D) row-major = cat([tar_feat_, src_feat_], dim=1) of batch.py:46-73 for EVERY row
 * (invalid rows carry anything); valid (R) flags.  MLP 2D -> 2H -> H -> nout, ReLU between,
 * optional tanh (ist_net.py:140-155).  Accumulation fixed as in gp_gemm.hip / gp_ist.hip:
 * sequential fmaf over the input units, then + bias.  Invalid rows -> -1000 (ist_net.py:109-119).
 * weights are torch-layout: W1 (2H,2D), b1, W2 (H,2H), b2, W3 (nout,H), b3.
 * ---------------------------------------------------------------------------------- */
void oracle_ist_head(const float* feats, const uint8_t* valid, int R, int D2, int H2, int H, int nout,
                     const float* W1, const float* b1, const float* W2, const float* b2,
                     const float* W3, const float* b3, int use_tanh, float* out)
{
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
        float* o = out + (size_t)r * nout;
        if (!valid[r]) { for (int j = 0; j < nout; ++j) o[j] = -1000.0f; continue; }
        const float* x = feats + (size_t)r * D2;
        float* h1 = (float*)malloc(sizeof(float) * (H2 + H));
        float* h2 = h1 + H2;
        for (int i = 0; i < H2; ++i) {
            float a = 0.f;
            for (int c = 0; c < D2; ++c) a = fmaf(W1[(size_t)i * D2 + c], x[c], a);
            h1[i] = fmaxf(a + b1[i], 0.f);
        }
        for (int i = 0; i < H; ++i) {
            float a = 0.f;
            for (int c = 0; c < H2; ++c) a = fmaf(W2[(size_t)i * H2 + c], h1[c], a);
            h2[i] = fmaxf(a + b2[i], 0.f);
        }
        for (int j = 0; j < nout; ++j) {
            float a = 0.f;
            for (int c = 0; c < H; ++c) a = fmaf(W3[(size_t)j * H + c], h2[c], a);
            a = a + b3[j];
            o[j] = use_tanh ? tanhf(a) : a;
        }
        free(h1);
    }
}

/* ------------------------------------------------------------------------------------
 * RANSAC.forward / forward_ / _sample (reference ransac.py:108-172, 37-106, 19-35) for R
 * independent problems of P=256 padded correspondences; arithmetic order as gp_pose.hip.
 * ---------------------------------------------------------------------------------- */
typedef struct { float m00, m01, m10, m11, t0, t1; } cand_t;

static cand_t make_cand(const float* sx, const float* sy, const float* tx, const float* ty,
                        const float* sc, const float* cs, const float* sn, int i)
{
    cand_t c;
    c.m00 = cs[i] * sc[i];
    c.m01 = (-sn[i]) * sc[i];
    c.m10 = sn[i] * sc[i];
    c.m11 = cs[i] * sc[i];
    float a0 = c.m00 * sx[i] + c.m01 * sy[i];
    float a1 = c.m10 * sx[i] + c.m11 * sy[i];
    c.t0 = tx[i] - a0;
    c.t1 = ty[i] - a1;
    return c;
}
/* `fused`: the reference's einsum->bmm takes torch's native kernel (plain mul/add) for n <= 45
 * correspondences and MKL sgemm (fma over the 3-term contraction) for n >= 46; errors of exactly
 * 14 px (two query patches on one template patch) make the inlier test depend on it. */
static int cand_inlier(const cand_t* c, float sx, float sy, float tx, float ty, float thr, int fused)
{
    float v0 = fused ? fmaf(c->m01, sy, c->m00 * sx) + c->t0 : (c->m00 * sx + c->m01 * sy) + c->t0;
    float v1 = fused ? fmaf(c->m11, sy, c->m10 * sx) + c->t1 : (c->m10 * sx + c->m11 * sy) + c->t1;
    float d0 = tx - v0, d1 = ty - v1;
    return sqrtf(d0 * d0 + d1 * d1) <= thr;
}

/* `score` (R x P f32, or NULL = ones): the per-correspondence weights of RANSAC.forward(batch, scores=...) (ransac.py:119-120, 98:
 * score_inliers = sum(inliers * val_score)); summed in ascending correspondence order in f32 (torch.sum's order is unspecified:
 * exact for the integer-valued weights the goldens use); the emitted inlier scores are the weights cast to the points' int64
 * (ransac.py:163 assigns into an int64 tensor: truncation). */
void oracle_ransac_scored(const int64_t* src_pts, const int64_t* tar_pts, const float* rel_scale,
                          const float* rel_inplane, const float* score, int R, float patch_size, float thr,
                          float* M, uint8_t* failed, int64_t* inl_src, int64_t* inl_tar, int64_t* inl_score);
void oracle_ransac(const int64_t* src_pts, const int64_t* tar_pts, const float* rel_scale,
                   const float* rel_inplane, int R, float patch_size, float thr,
                   float* M, uint8_t* failed, int64_t* inl_src, int64_t* inl_tar, int64_t* inl_score)
{
    oracle_ransac_scored(src_pts, tar_pts, rel_scale, rel_inplane, NULL, R, patch_size, thr, M, failed, inl_src, inl_tar, inl_score);
}

void oracle_ransac_scored(const int64_t* src_pts, const int64_t* tar_pts, const float* rel_scale,
                          const float* rel_inplane, const float* score, int R, float patch_size, float thr,
                          float* M, uint8_t* failed, int64_t* inl_src, int64_t* inl_tar, int64_t* inl_score)
{
#pragma omp parallel for schedule(dynamic)
    for (int r = 0; r < R; ++r) {
        float sx[P], sy[P], tx[P], ty[P], sc[P], cs[P], sn[P], wt[P]; int orig[P];
        int n = 0;
        for (int p = 0; p < P; ++p) {
            size_t rp = (size_t)r * P + p;
            inl_src[2 * rp] = inl_src[2 * rp + 1] = -1;
            inl_tar[2 * rp] = inl_tar[2 * rp + 1] = -1;
            inl_score[rp] = 0;
            if (src_pts[2 * rp] == -1) continue;                     /* ransac.py:141 */
            sx[n] = (float)src_pts[2 * rp] * patch_size;             /* :57-58 */
            sy[n] = (float)src_pts[2 * rp + 1] * patch_size;
            tx[n] = (float)tar_pts[2 * rp] * patch_size;
            ty[n] = (float)tar_pts[2 * rp + 1] * patch_size;
            sc[n] = rel_scale[rp]; cs[n] = rel_inplane[2 * rp]; sn[n] = rel_inplane[2 * rp + 1];
            wt[n] = score ? score[rp] : 1.0f;
            orig[n] = p; ++n;
        }
        float* Mr = M + (size_t)r * 9;
        if (n == 0) {                                               /* :128-129, :142 */
            for (int i = 0; i < 9; ++i) Mr[i] = (i % 4 == 0) ? 1.f : 0.f;
            failed[r] = 0; continue;
        }
        int best = 0; float bc = 0.f;
        for (int i = 0; i < n; ++i) {
            cand_t c = make_cand(sx, sy, tx, ty, sc, cs, sn, i);
            float cnt = 0.f;                                         /* unit weights: an exact integer count */
            for (int j = 0; j < n; ++j) if (j != i && cand_inlier(&c, sx[j], sy[j], tx[j], ty[j], thr, n >= 46)) cnt = cnt + wt[j];
            if (i == 0 || cnt > bc) { bc = cnt; best = i; }         /* first max (:99) */
        }
        cand_t c = make_cand(sx, sy, tx, ty, sc, cs, sn, best);
        Mr[0] = c.m00; Mr[1] = c.m01; Mr[2] = c.t0; Mr[3] = c.m10; Mr[4] = c.m11; Mr[5] = c.t1;
        Mr[6] = 0.f; Mr[7] = 0.f; Mr[8] = 1.f;
        failed[r] = (bc == 0.f);                                    /* :100 */
        int q = 0;
        for (int j = 0; j < n; ++j) {
            if (j == best || !cand_inlier(&c, sx[j], sy[j], tx[j], ty[j], thr, n >= 46)) continue;
            size_t o = (size_t)r * P + q, s = (size_t)r * P + orig[j];
            inl_src[2 * o] = src_pts[2 * s]; inl_src[2 * o + 1] = src_pts[2 * s + 1];
            inl_tar[2 * o] = tar_pts[2 * s]; inl_tar[2 * o + 1] = tar_pts[2 * s + 1];
            inl_score[o] = (int64_t)wt[j]; ++q;                      /* :160-163 (float -> int64 assignment truncates) */
        }
    }
}

/* ------------------------------------------------------------------------------------
 * ObjectPoseRecovery.forward_recovery / _forward_recovery (reference poses.py:26-122) with
 * lib3d/torch.py:47-65 (inverse_affine) and :150-162 (normalize_affine_transform).
 * ---------------------------------------------------------------------------------- */
static void m3mul(const float* a, const float* b, float* c)
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        c[i * 3 + j] = (a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j]) + a[i * 3 + 2] * b[6 + j];
}
static void m3vec(const float* a, const float* v, float* o)
{
    for (int i = 0; i < 3; ++i) o[i] = (a[i * 3] * v[0] + a[i * 3 + 1] * v[1]) + a[i * 3 + 2] * v[2];
}
static void m3inv(const float* m, float* o)
{
    float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    float det = (m[0] * c00 + m[1] * c01) + m[2] * c02;
    float id = 1.0f / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

void oracle_recover(const int32_t* labels, const float* tar_K, const float* tar_M, const int64_t* id_src,
                    const float* pred_M, const float* tmpl_K, const float* tmpl_M, const float* tmpl_pose,
                    int B, int N, int k, float* out)
{
    for (int bk = 0; bk < B * k; ++bk) {
        int b = bk / k;
        size_t on = (size_t)labels[b] * N + (size_t)id_src[bk];
        const float *qM = tar_M + (size_t)b * 9, *qK = tar_K + (size_t)b * 9, *M = pred_M + (size_t)bk * 9;
        const float *tK = tmpl_K + (size_t)labels[b] * 9, *tM = tmpl_M + on * 9, *tP = tmpl_pose + on * 16;
        float sc = sqrtf(M[0] * M[0] + M[3] * M[3]);
        float Rin[9] = {M[0] / sc, M[1] / sc, 0.f, M[3] / sc, M[4] / sc, 0.f, 0.f, 0.f, 1.f};
        float Rt[9] = {tP[0], tP[1], tP[2], tP[4], tP[5], tP[6], tP[8], tP[9], tP[10]};
        float Rm[9]; m3mul(Rin, Rt, Rm);
        float temp_z = tP[11];
        float tt[3] = {tP[3], tP[7], tP[11]};
        float c2d[3]; m3vec(tK, tt, c2d);
        float cz = c2d[2]; c2d[0] = c2d[0] / cz; c2d[1] = c2d[1] / cz; c2d[2] = c2d[2] / cz;
        float qs = qM[0];
        float inv_qM[9] = {1.0f / qs, 0.f, -qM[2] / qs, 0.f, 1.0f / qs, -qM[5] / qs, 0.f, 0.f, 1.f};
        float tmp[9], aff[9]; m3mul(inv_qM, M, tmp); m3mul(tmp, tM, aff);
        float qc[3]; m3vec(aff, c2d, qc);
        float iK[9]; m3inv(qK, iK);
        float scale2d = sqrtf(aff[0] * aff[0] + aff[3] * aff[3]);
        float focal_ratio = qK[0] / tK[0];
        float qz = (temp_z / scale2d) * focal_ratio;
        float qt[3]; m3vec(iK, qc, qt);
        float w = qt[2]; qt[0] = qt[0] / w; qt[1] = qt[1] / w; qt[2] = qt[2] / w;
        float* o = out + (size_t)bk * 16;
        o[0] = Rm[0]; o[1] = Rm[1]; o[2] = Rm[2]; o[3] = qt[0] * qz;
        o[4] = Rm[3]; o[5] = Rm[4]; o[6] = Rm[5]; o[7] = qt[1] * qz;
        o[8] = Rm[6]; o[9] = Rm[7]; o[10] = Rm[8]; o[11] = qt[2] * qz;
        o[12] = tP[12]; o[13] = tP[13]; o[14] = tP[14]; o[15] = tP[15];
    }
}

/* ------------------------------------------------------------------------------------
 * Conv2d(bias=False) + folded eval-BatchNorm + residual + ReLU on channel-major activations,
 * restating BasicBlock / ResNet stages (reference resnet.py:26-50, 364-381) with the fixed
 * accumulation order of gp_conv.hip: fmaf chain over k = (ci, dy, dx) ascending, zero padding
 * contributes fmaf(w, 0, acc).  X [Cin][B][H][W], W torch layout (Cout,Cin,KH,KW),
 * Y [Cout][B][OH][OW].
 * ---------------------------------------------------------------------------------- */
void oracle_conv2d_cm(const float* X, const float* Wt, float* Y, const float* alpha, const float* beta,
                      const float* res, int Cin, int B, int H, int W, int Cout, int KH, int KW, int stride,
                      int pad, int relu)
{
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    const size_t npix = (size_t)B * OH * OW;
#pragma omp parallel for schedule(static) collapse(2)
    for (int co = 0; co < Cout; ++co)
        for (int b = 0; b < B; ++b)
            for (int oy = 0; oy < OH; ++oy)
                for (int ox = 0; ox < OW; ++ox) {
                    float acc = 0.f;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int dy = 0; dy < KH; ++dy)
                            for (int dx = 0; dx < KW; ++dx) {
                                int iy = oy * stride - pad + dy, ix = ox * stride - pad + dx;
                                float x = (iy >= 0 && iy < H && ix >= 0 && ix < W)
                                              ? X[(((size_t)ci * B + b) * H + iy) * W + ix] : 0.f;
                                acc = fmaf(Wt[(((size_t)co * Cin + ci) * KH + dy) * KW + dx], x, acc);
                            }
                    size_t o = (size_t)co * npix + ((size_t)b * OH + oy) * OW + ox;
                    float v = acc;
                    if (alpha) v = v * alpha[co] + beta[co];
                    if (res) v = res[o] + v;
                    if (relu) v = fmaxf(v, 0.f);
                    Y[o] = v;
                }
}
