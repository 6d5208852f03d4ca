// Exhaustive 1-correspondence "RANSAC" and 6-D pose recovery on gfx950.
// Reference: RANSAC.forward / forward_ / _sample (src/models/ransac.py:108-172, 37-106, 19-35),
// ObjectPoseRecovery._forward_recovery / forward_recovery (src/models/poses.py:26-122),
// affine_torch / apply_affine / inverse_affine / normalize_affine_transform
// (src/lib3d/torch.py:7-27, 68-89, 47-65, 150-162).
//
// The reference runs a Python double loop (detections x hypotheses) with a host-built (N, N-1) index
// table per call; here every (detection, hypothesis) problem is one workgroup and nothing touches
// the host.  Plain mul/add only (no fused contraction; -ffp-contract=off) in the order written, so
// the CPU oracle reproduces every float bit-for-bit.
#include "gp_common.h"

namespace {

struct RansacSmem {
    float sx[GP_P], sy[GP_P], tx[GP_P], ty[GP_P];  // compacted pixel coordinates
    float sc[GP_P], cs[GP_P], sn[GP_P];            // compacted scale, cos, sin
    float wt[GP_P];                                // compacted weights (RANSAC.forward's `scores`; ones by default)
    short orig[GP_P];                              // compacted -> patch position
    float count[GP_P];                             // weighted inlier score per candidate (unit weights: an exact integer count)
    int wave_n[4];
    int best, n;
    float best_count;
};

// exclusive prefix of `flag` over the 256-thread block (4 waves): returns position, total in *tot
__device__ __forceinline__ int block_compact(bool flag, int* wave_n, int* tot)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_n[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wave_n[w];
    *tot = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
    __syncthreads();
    return base + before;
}

// candidate i -> 2x2 part of M (affine_torch, lib3d/torch.py:19-27) and its translation
// (ransac.py:86-88): t = tar_i - M2 * src_i
__device__ __forceinline__ void candidate(const RansacSmem& s, int i, float& m00, float& m01, float& m10,
                                          float& m11, float& t0, float& t1)
{
    const float c = s.cs[i], sn = s.sn[i], sc = s.sc[i];
    m00 = c * sc;        // R = [[c, -s], [s, c]] (ransac.py:82-83), then *= scale
    m01 = (-sn) * sc;
    m10 = sn * sc;
    m11 = c * sc;
    const float a0 = m00 * s.sx[i] + m01 * s.sy[i];  // apply_affine with zero translation
    const float a1 = m10 * s.sx[i] + m11 * s.sy[i];
    t0 = s.tx[i] - a0;
    t1 = s.ty[i] - a1;
}

// apply_affine on the validation points (lib3d/torch.py:82-88) is an einsum -> torch.bmm over
// (3x3)x(3x(n-1)).  Two query patches matched to the SAME template patch give an error of EXACTLY
// 14 px in exact arithmetic, so `error <= 14` is decided by the rounding of this product and the
// reference's result depends on which bmm path torch takes: its native kernel (plain mul/add) when
// 3*(n-1)*3 < 400, i.e. n <= 45, and MKL sgemm (fused multiply-add over c) for n >= 46.  Both are
// reproduced here bit-for-bit (verified against torch 2.10 CPU, tests/golden/pose.npz, e2e.npz).
__device__ __forceinline__ bool is_inlier(const RansacSmem& s, int j, float m00, float m01, float m10, float m11,
                                          float t0, float t1, float thr, bool fused)
{
    const float v0 = fused ? __builtin_fmaf(m01, s.sy[j], m00 * s.sx[j]) + t0 : (m00 * s.sx[j] + m01 * s.sy[j]) + t0;
    const float v1 = fused ? __builtin_fmaf(m11, s.sy[j], m10 * s.sx[j]) + t1 : (m10 * s.sx[j] + m11 * s.sy[j]) + t1;
    const float d0 = s.tx[j] - v0, d1 = s.ty[j] - v1;
    return __builtin_sqrtf(d0 * d0 + d1 * d1) <= thr;       // torch.norm(dim=2) <= pixel_threshold
}

__global__ __launch_bounds__(256) void ransac_kernel(
    const long long* __restrict__ src_pts, const long long* __restrict__ tar_pts,  // (R,256,2)
    const float* __restrict__ rel_scale, const float* __restrict__ rel_inplane,    // (R,256), (R,256,2)
    const float* __restrict__ score,                                               // (R,256) weights or null = ones (ransac.py:119-120)
    float patch_size, float thr, float* __restrict__ Mout, unsigned char* __restrict__ failed,
    long long* __restrict__ inl_src, long long* __restrict__ inl_tar, long long* __restrict__ inl_score)
{
    __shared__ RansacSmem s;
    const size_t r = blockIdx.x;
    const int p = threadIdx.x;
    const size_t rp = r * GP_P + p;
    const long long sxi = src_pts[2 * rp], syi = src_pts[2 * rp + 1];
    const long long txi = tar_pts[2 * rp], tyi = tar_pts[2 * rp + 1];
    const bool valid = sxi != -1;  // mask = src_keypoint[:, 0] != -1   (ransac.py:141)
    int n;
    const int pos = block_compact(valid, s.wave_n, &n);
    if (valid) {
        s.sx[pos] = (float)sxi * patch_size;  // pts * patch_size (corner coords; ransac.py:57-58)
        s.sy[pos] = (float)syi * patch_size;
        s.tx[pos] = (float)txi * patch_size;
        s.ty[pos] = (float)tyi * patch_size;
        s.sc[pos] = rel_scale[rp];
        s.cs[pos] = rel_inplane[2 * rp];
        s.sn[pos] = rel_inplane[2 * rp + 1];
        s.wt[pos] = score ? score[rp] : 1.0f;
        s.orig[pos] = (short)p;
    }
    // defaults: identity M, not failed, -1 / 0 padding (ransac.py:125-131)
    inl_src[2 * rp] = -1; inl_src[2 * rp + 1] = -1;
    inl_tar[2 * rp] = -1; inl_tar[2 * rp + 1] = -1;
    inl_score[rp] = 0;
    __syncthreads();
    if (n == 0) {
        if (p < 9) Mout[r * 9 + p] = (p % 4 == 0) ? 1.f : 0.f;
        if (p == 0) failed[r] = 0;
        return;
    }
    // every correspondence proposes a similarity; score = sum of the weights of the OTHER correspondences within thr (ransac.py:98;
    // ascending order, f32: with the default unit weights an exact count)
    float cnt = 0.f;
    if (p < n) {
        float m00, m01, m10, m11, t0, t1;
        candidate(s, p, m00, m01, m10, m11, t0, t1);
        for (int j = 0; j < n; ++j)
            if (j != p && is_inlier(s, j, m00, m01, m10, m11, t0, t1, thr, n >= 46)) cnt = cnt + s.wt[j];
        s.count[p] = cnt;
    }
    __syncthreads();
    if (p == 0) {  // torch.max: first maximal candidate (ransac.py:99)
        int best = 0;
        float bc = s.count[0];
        for (int i = 1; i < n; ++i)
            if (s.count[i] > bc) { bc = s.count[i]; best = i; }
        s.best = best;
        s.best_count = bc;
    }
    __syncthreads();
    const int best = s.best;
    float m00, m01, m10, m11, t0, t1;
    candidate(s, best, m00, m01, m10, m11, t0, t1);
    if (p == 0) {
        float* M = Mout + r * 9;
        M[0] = m00; M[1] = m01; M[2] = t0;
        M[3] = m10; M[4] = m11; M[5] = t1;
        M[6] = 0.f; M[7] = 0.f; M[8] = 1.f;
        failed[r] = (s.best_count == 0.f) ? 1 : 0;  // failed = score == 0 (ransac.py:100)
    }
    // inliers of the winner, in ascending order, packed at the front (ransac.py:103-104, 160-163)
    const bool inl = (p < n) && (p != best) && is_inlier(s, p, m00, m01, m10, m11, t0, t1, thr, n >= 46);
    int tot;
    const int q = block_compact(inl, s.wave_n, &tot);
    if (inl) {
        const size_t o = r * GP_P + q;
        const size_t src = r * GP_P + s.orig[p];
        inl_src[2 * o] = src_pts[2 * src]; inl_src[2 * o + 1] = src_pts[2 * src + 1];
        inl_tar[2 * o] = tar_pts[2 * src]; inl_tar[2 * o + 1] = tar_pts[2 * src + 1];
        inl_score[o] = (long long)s.wt[p];  // the weight, cast to the points' int64 as the reference's assignment does (ransac.py:163); default 1
    }
}

// ---------------------------------------------------------------- pose recovery
__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* c)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[i * 3 + j] = (a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j]) + a[i * 3 + 2] * b[6 + j];
}
__device__ __forceinline__ void mat3_vec(const float* a, const float* v, float* o)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = (a[i * 3] * v[0] + a[i * 3 + 1] * v[1]) + a[i * 3 + 2] * v[2];
}
__device__ __forceinline__ void mat3_inv(const float* m, float* o)
{
    const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const float det = (m[0] * c00 + m[1] * c01) + m[2] * c02;
    const float id = 1.0f / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// one thread per (detection, hypothesis)
__global__ __launch_bounds__(64) void recover_kernel(
    const int* __restrict__ labels, const float* __restrict__ tar_K, const float* __restrict__ tar_M,  // (B), (B,3,3) x2
    const long long* __restrict__ id_src, const float* __restrict__ pred_M,                           // (B,k), (B,k,3,3)
    const float* __restrict__ tmpl_K, const float* __restrict__ tmpl_M, const float* __restrict__ tmpl_pose,  // (O,3,3) (O,N,3,3) (O,N,4,4)
    int B, int O, int N, int k, float* __restrict__ out /*(B,k,4,4)*/, int* __restrict__ bad_crop_M, int* __restrict__ status)
{
    const int bk = blockIdx.x * 64 + threadIdx.x;
    if (bk >= B * k) return;
    const int b = bk / k;
    int lab = labels[b];
    long long view = id_src[bk];
    if ((unsigned)lab >= (unsigned)O || (unsigned long long)view >= (unsigned long long)N) {
        gp_raise(status, GP_ST_LABEL_RANGE);
        lab = 0;
        view = 0;
    }
    const size_t on = (size_t)lab * N + (size_t)view;
    const float* qM = tar_M + (size_t)b * 9;
    const float* qK = tar_K + (size_t)b * 9;
    const float* M = pred_M + (size_t)bk * 9;
    const float* tK = tmpl_K + (size_t)lab * 9;
    const float* tM = tmpl_M + on * 9;
    const float* tP = tmpl_pose + on * 16;
    // the reference asserts the crop transform is isotropic scale + translation (lib3d/torch.py:54-55)
    if (qM[3] != 0.f || qM[1] != 0.f || qM[0] != qM[4]) atomicOr(bad_crop_M, 1);

    // Step 1: R = normalised in-plane (2x2 of M / |M[:,0]|, bottom-right 1) x template R (poses.py:64-68)
    const float sc = __builtin_sqrtf(M[0] * M[0] + M[3] * M[3]);
    const float Rin[9] = {M[0] / sc, M[1] / sc, 0.f, M[3] / sc, M[4] / sc, 0.f, 0.f, 0.f, 1.f};
    const float Rt[9] = {tP[0], tP[1], tP[2], tP[4], tP[5], tP[6], tP[8], tP[9], tP[10]};
    float R[9];
    mat3_mul(Rin, Rt, R);
    // Step 2: template centre in its image, through the full template->query 2-D affine (poses.py:71-85)
    const float temp_z = tP[11];
    const float tt[3] = {tP[3], tP[7], tP[11]};
    float c2d[3];
    mat3_vec(tK, tt, c2d);
    const float cz = c2d[2];
    c2d[0] = c2d[0] / cz; c2d[1] = c2d[1] / cz; c2d[2] = c2d[2] / cz;
    const float qs = qM[0];  // inverse_affine (lib3d/torch.py:57-63)
    const float inv_qM[9] = {1.0f / qs, 0.f, -qM[2] / qs, 0.f, 1.0f / qs, -qM[5] / qs, 0.f, 0.f, 1.f};
    float tmp[9], aff[9];
    mat3_mul(inv_qM, M, tmp);
    mat3_mul(tmp, tM, aff);
    float qc[3];
    mat3_vec(aff, c2d, qc);
    float iK[9];
    mat3_inv(qK, iK);  // torch.inverse(query_K) (poses.py:87)
    // Step 3: depth from 2-D scale and focal ratio (poses.py:90-92)
    const float scale2d = __builtin_sqrtf(aff[0] * aff[0] + aff[3] * aff[3]);
    const float focal_ratio = qK[0] / tK[0];
    const float qz = (temp_z / scale2d) * focal_ratio;
    float qt[3];
    mat3_vec(iK, qc, qt);
    const float w = qt[2];
    qt[0] = qt[0] / w; qt[1] = qt[1] / w; qt[2] = qt[2] / w;
    float* o = out + (size_t)bk * 16;
    o[0] = R[0]; o[1] = R[1]; o[2] = R[2];  o[3] = qt[0] * qz;
    o[4] = R[3]; o[5] = R[4]; o[6] = R[5];  o[7] = qt[1] * qz;
    o[8] = R[6]; o[9] = R[7]; o[10] = R[8]; o[11] = qt[2] * qz;
    o[12] = tP[12]; o[13] = tP[13]; o[14] = tP[14]; o[15] = tP[15];
}

// ------------------------------------------------------------------ hypothesis ranking (gigaPose.py:588-594)
// score[b, j] = sum_p inlier_score[b, j, p] / P (an integer sum: exact in any order), then every per-hypothesis tensor
// of the prediction collection is reordered along k by descending score (ties: lower hypothesis index first -- the
// reference's torch.argsort leaves ties unspecified; NaN cannot occur, the sum is an integer).  One launch replaces
// torch.sum + a stable sort + one advanced-indexing kernel per tensor (13 of them).
constexpr int GP_RANK_MAX_TENSORS = 16, GP_RANK_MAX_K = 64;
struct RankTable {
    const unsigned char* src[GP_RANK_MAX_TENSORS];
    unsigned char* dst[GP_RANK_MAX_TENSORS];
    int row_bytes[GP_RANK_MAX_TENSORS];
    int n;
};

__global__ __launch_bounds__(256) void rank_hypotheses_kernel(const long long* __restrict__ inl_score, int k, int P, int sort,
                                                               float* __restrict__ scores, long long* __restrict__ order,
                                                               RankTable tab)
{
    __shared__ long long part[4][GP_RANK_MAX_K];
    __shared__ float sc[GP_RANK_MAX_K];
    __shared__ int rank[GP_RANK_MAX_K];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = 0; j < k; ++j) {
        long long acc = 0;
        for (int p = tid; p < P; p += 256) acc += inl_score[((size_t)b * k + j) * P + p];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) part[wave][j] = acc;
    }
    __syncthreads();
    if (tid < k) {  // torch: int64 sum / int -> both operands converted to float32, true division
        const long long t = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        sc[tid] = (float)t / (float)P;
    }
    __syncthreads();
    if (tid < k) {
        int r = tid;
        if (sort) {
            r = 0;
            for (int i = 0; i < k; ++i) r += (sc[i] > sc[tid]) | ((sc[i] == sc[tid]) & (i < tid));
        }
        rank[tid] = r;
        scores[(size_t)b * k + r] = sc[tid];
        order[(size_t)b * k + r] = tid;
    }
    __syncthreads();
    for (int t = 0; t < tab.n; ++t) {
        const int rb = tab.row_bytes[t];
        const unsigned char* s = tab.src[t] + (size_t)b * k * rb;
        unsigned char* d = tab.dst[t] + (size_t)b * k * rb;
        if ((rb & 15) == 0 && (((size_t)s | (size_t)d) & 15) == 0) {
            const int w = rb >> 4;
            for (int i = tid; i < k * w; i += 256) {
                const int j = i / w, c = i - j * w;
                reinterpret_cast<uint4*>(d)[(size_t)rank[j] * w + c] = reinterpret_cast<const uint4*>(s)[i];
            }
        } else if ((rb & 3) == 0 && (((size_t)s | (size_t)d) & 3) == 0) {
            const int w = rb >> 2;
            for (int i = tid; i < k * w; i += 256) {
                const int j = i / w, c = i - j * w;
                reinterpret_cast<unsigned*>(d)[(size_t)rank[j] * w + c] = reinterpret_cast<const unsigned*>(s)[i];
            }
        } else {
            for (int i = tid; i < k * rb; i += 256) {
                const int j = i / rb, c = i - j * rb;
                d[(size_t)rank[j] * rb + c] = s[i];
            }
        }
    }
}

}  // namespace

extern "C" {

int gp_ransac_scored(const long long* src_pts, const long long* tar_pts, const float* rel_scale,
                     const float* rel_inplane, const float* score, int R, float patch_size, float pixel_threshold, float* M,
                     unsigned char* failed, long long* inl_src, long long* inl_tar, long long* inl_score, void* stream)
{
    GP_REQUIRE(R >= 0, "gp_ransac: bad size");
    if (R == 0) return GP_OK;
    GP_REQUIRE(src_pts && tar_pts && rel_scale && rel_inplane && M && failed && inl_src && inl_tar && inl_score,
               "gp_ransac: null pointer");
    hipLaunchKernelGGL(ransac_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, src_pts, tar_pts, rel_scale,
                       rel_inplane, score, patch_size, pixel_threshold, M, failed, inl_src, inl_tar, inl_score);
    GP_CHECK_LAUNCH("gp_ransac");
    return GP_OK;
}

int gp_recover_poses(const int* labels, const float* tar_K, const float* tar_M, const long long* id_src,
                     const float* pred_M, const float* tmpl_K, const float* tmpl_M, const float* tmpl_pose, int B,
                     int O, int N, int k, float* poses, int* bad_crop_M, void* stream)
{
    GP_REQUIRE(B >= 0 && O > 0 && N > 0 && k > 0, "gp_recover_poses: bad sizes");
    if (B == 0) return GP_OK;
    GP_REQUIRE(labels && tar_K && tar_M && id_src && pred_M && tmpl_K && tmpl_M && tmpl_pose && poses && bad_crop_M,
               "gp_recover_poses: null pointer");
    hipLaunchKernelGGL(recover_kernel, dim3((B * k + 63) / 64), dim3(64), 0, (hipStream_t)stream, labels, tar_K,
                       tar_M, id_src, pred_M, tmpl_K, tmpl_M, tmpl_pose, B, O, N, k, poses, bad_crop_M, gp_status_buffer());
    GP_CHECK_LAUNCH("gp_recover_poses");
    return GP_OK;
}

int gp_rank_hypotheses(const long long* inl_score, int B, int k, int P, int sort, float* scores, long long* order, int n_tensors,
                       const void* const* src, void* const* dst, const int* row_bytes, void* stream)
{
    GP_REQUIRE(B >= 0 && k > 0 && k <= GP_RANK_MAX_K && P > 0, "gp_rank_hypotheses: bad sizes B=%d k=%d P=%d (k <= %d)", B, k, P,
               GP_RANK_MAX_K);
    GP_REQUIRE(n_tensors >= 0 && n_tensors <= GP_RANK_MAX_TENSORS, "gp_rank_hypotheses: at most %d tensors per call, got %d",
               GP_RANK_MAX_TENSORS, n_tensors);
    if (B == 0) return GP_OK;
    GP_REQUIRE(inl_score && scores && order && (n_tensors == 0 || (src && dst && row_bytes)), "gp_rank_hypotheses: null pointer");
    RankTable tab;
    tab.n = n_tensors;
    for (int t = 0; t < n_tensors; ++t) {
        GP_REQUIRE(src[t] && dst[t] && src[t] != dst[t] && row_bytes[t] > 0,
                   "gp_rank_hypotheses: bad tensor %d (null, empty rows, or in place)", t);
        tab.src[t] = static_cast<const unsigned char*>(src[t]);
        tab.dst[t] = static_cast<unsigned char*>(dst[t]);
        tab.row_bytes[t] = row_bytes[t];
    }
    hipLaunchKernelGGL(rank_hypotheses_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, inl_score, k, P, sort, scores, order, tab);
    GP_CHECK_LAUNCH("gp_rank_hypotheses");
    return GP_OK;
}

}  // extern "C"
