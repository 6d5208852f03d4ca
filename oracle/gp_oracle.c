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
                       int C, float thr, float patch_thr,
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
    /* torch.max over s and over t: first maximal index (matching.py:239-241) */
    float sc_s2t[P]; int id_s2t[P]; int id_t2s[P];
    for (int t = 0; t < P; ++t) {
        float best = sim[t * P]; int bi = 0;
        for (int j = 1; j < P; ++j) if (sim[t * P + j] > best) { best = sim[t * P + j]; bi = j; }
        score_t2s[t] = best; id_t2s[t] = bi;
    }
    for (int j = 0; j < P; ++j) {
        float best = sim[j]; int bi = 0;
        for (int t = 1; t < P; ++t) if (sim[t * P + j] > best) { best = sim[t * P + j]; bi = t; }
        sc_s2t[j] = best; id_s2t[j] = bi;
    }
    /* masks (matching.py:247-271) */
    float acc = 0.f, cnt = 0.f;
    for (int t = 0; t < P; ++t) {
        int js = id_t2s[t];
        int mask_sim = score_t2s[t] >= thr;                               /* :247 */
        int t2 = id_s2t[js];                                              /* :96 gather */
        float dx = (float)(t2 % G) - (float)(t % G);                      /* :98-99 (x=w, y=h) */
        float dy = (float)(t2 / G) - (float)(t / G);
        float dist = sqrtf(dx * dx + dy * dy);                            /* torch.norm :100 */
        int mask_dist = dist <= patch_thr;                                /* :104 */
        int mask_sim2 = sc_s2t[js] >= thr;                                /* :107-108 */
        /* mask_non_zero (:263-268); NOTE quirk: (idx_src2tar != 0) is indexed by position t */
        float nz = qmask[t] * smask[js];
        nz = nz * (float)(id_s2t[t] != 0);
        nz = nz * (float)(id_t2s[t] != 0);
        float m = (float)(mask_sim && mask_dist && mask_sim2) * nz;       /* :271 */
        mask_all[t] = m;
        idx_t2s[t] = (uint8_t)js;
        acc = acc + score_t2s[t] * m;  /* fixed order: sequential over t (reference: torch.sum) */
        cnt = cnt + m;
    }
    *sim_avg = (cnt > 0.f) ? acc / 256.0f : 0.f;                          /* :274-278 */
}

/* All (b, n) tiles.  labels are 0-based object indices (reference uses label-1,
 * gigaPose.py:520-521).  bank (O,N,C,P), bmask (O,N,P), query (B,C,P), qmask (B,P). */
void oracle_match(const float* query, const float* bank, const float* qmask, const float* bmask,
                  const int32_t* labels, int B, int O, int N, int C, float thr, float patch_thr,
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
                           qmask + (size_t)b * P, bmask + (o * N + n) * P, C, thr, patch_thr,
                           idx_t2s + bn * P, score_t2s + bn * P, mask_all + bn * P, sim_avg + bn, sim);
            }
        free(sim);
    }
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
