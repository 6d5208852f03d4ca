// IST per-correspondence regressor on gfx950: gather + concat + two 3-layer MLP heads.
// Reference: ISTNet.inference (src/models/network/ist_net.py:97-120), gather
// (src/utils/batch.py:46-73), Regressor (ist_net.py:123-162).
//
// The reference compacts valid correspondences, runs the MLPs, and scatters the results into
// (B,P)/(B,P,2) tensors pre-filled with -1000.  So does this file since round 3, without a host sync: the
// gather kernel ranks the valid rows of its (detection, hypothesis) block (ballot + popcount), reserves a
// range of compact rows with one atomic add on a device counter and writes only those; the hidden-layer GEMMs
// are launched over the worst-case row count and tiles beyond the counter return at once (split numerics; the
// f32 stream-K GEMM of the chain mode computes the whole range as before); the head kernel maps a row to its
// compact position.  Rows are independent, so a row's result does not depend on where the atomics put it.
// (Typically 35-60 % of the 256 patches of a hypothesis carry a correspondence: 0.70 -> 0.3x ms per step.)
// The two hidden layers are GEMMs on the transposed feature matrix X^T [2D][rows].
#include "gp_common.h"

int gp_gemm_launch(const float* A, int lda, const float* B, int ldb, float* D, int ldd, int I, int J,
                   int K, int epilogue, const float* bias, const float* scale, const float* res, int ldr,
                   float* sk_ws, hipStream_t st);

int gp_gemm_split_launch_limited(const float* act, int ld_act, const void* whi, const void* wlo, float* D, int ldd, int I, int J,
                                 int K, int act_is_b, int epilogue, const float* bias, const float* scale, const float* res, int ldr,
                                 const int* j_limit, hipStream_t st);

namespace {

// X[c][r]: c < D -> tar_feat[b][c][ti],  c >= D -> src_bank[obj][view][c-D][si]     ("cat([tar, src])",
// ist_net.py:100).  Row r = (b*k + j)*256 + t.  Invalid rows (-1 points) read index 0 (finite, unused).
__global__ __launch_bounds__(256) void ist_gather_kernel(
    const float* __restrict__ tar_feat,  // (B, D, 256)
    const float* __restrict__ src_bank,  // (O, N, D, 256)
    const int* __restrict__ labels, const long long* __restrict__ id_src,  // (B), (B,k)
    const long long* __restrict__ tar_pts, const long long* __restrict__ src_pts,  // (B,k,256,2)
    int O, int N, int k, int D, size_t R, float* __restrict__ X, int* __restrict__ pos, int* __restrict__ count,
    int* __restrict__ status)
{
    __shared__ int wcnt[4], base;
    const int bk = blockIdx.x, b = bk / k, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int lab = labels[b];
    long long view = id_src[bk];
    if ((unsigned)lab >= (unsigned)O || (unsigned long long)view >= (unsigned long long)N) {  // the reference's index would raise
        if (t == 0) gp_raise(status, GP_ST_LABEL_RANGE);
        lab = 0;
        view = 0;
    }
    const size_t r = (size_t)bk * GP_P + t;
    const long long tx = tar_pts[2 * r], ty = tar_pts[2 * r + 1];
    const long long sx = src_pts[2 * r], sy = src_pts[2 * r + 1];
    const bool valid = (tx != -1) && (ty != -1) && (sx != -1) && (sy != -1);
    // compact row of this correspondence: rank inside the block + a range reserved with ONE atomic add per block
    const unsigned long long bal = __ballot(valid);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    if (t == 0) base = atomicAdd(count, wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]);
    __syncthreads();
    int rank = before;
    for (int w = 0; w < wave; ++w) rank += wcnt[w];
    const int rc = valid ? base + rank : -1;
    pos[r] = rc;
    if (!valid) return;
    const int ti = (int)(ty * GP_G + tx);  // index = y * W + x   (batch.py:63)
    const int si = (int)(sy * GP_G + sx);
    const float* tf = tar_feat + (size_t)b * D * GP_P + ti;
    const float* sf = src_bank + (((size_t)lab * N + (size_t)view) * D) * GP_P + si;
    // sixteen gathers in flight per lane (one load per dependent store left the kernel at the latency of 512 round trips: 186 us for
    // 67 MB of live rows); D is a multiple of 16 (the launcher requires 2 D % 32 == 0 in split numerics, 2 D % 16 == 0 otherwise)
    for (int c0 = 0; c0 < D; c0 += 16) {
        float a[16], bq[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = min(c0 + u, D - 1);
            a[u] = tf[(size_t)c * GP_P];
            bq[u] = sf[(size_t)c * GP_P];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (c0 + u < D) {
                X[(size_t)(c0 + u) * R + rc] = a[u];
                X[(size_t)(D + c0 + u) * R + rc] = bq[u];
            }
        }
    }
}

// Final Linear(H -> nout) (+ tanh) and the -1000 fill of invalid rows (ist_net.py:109-119).
// out layout (R, nout).  Accumulation: sequential fmaf over the H hidden units, then + bias.
template <int NOUT>
__global__ __launch_bounds__(256) void ist_head_kernel(const float* __restrict__ Hid /*[H][R]*/,
                                                        const float* __restrict__ W3 /*[NOUT][H]*/,
                                                        const float* __restrict__ b3,
                                                        const long long* __restrict__ tar_pts,
                                                        const long long* __restrict__ src_pts, const int* __restrict__ pos, int H, size_t R,
                                                        int use_tanh, float* __restrict__ out)
{
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] = 0.f;
    const int rc = pos[r];  // compact row (-1: no correspondence -> -1000 below; row 0 is read, finite or not, and discarded)
    const size_t rr = rc >= 0 ? (size_t)rc : 0;
    for (int h0 = 0; h0 < H; h0 += 16) {  // sixteen loads in flight, then the fmas IN ORDER (the chain is unchanged); H % 128 == 0
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = Hid[(size_t)(h0 + u) * R + rr];
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) acc[o] = __builtin_fmaf(W3[o * H + h0 + u], x[u], acc[o]);
    }
    // validity as the reference computes it (ist_net.py:114-115): both coordinates != -1
    const bool sv = (src_pts[2 * r] != -1) && (src_pts[2 * r + 1] != -1);
    const bool tv = (tar_pts[2 * r] != -1) && (tar_pts[2 * r + 1] != -1);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float v = acc[o] + b3[o];
        if (use_tanh) v = tanhf(v);
        out[r * NOUT + o] = (sv && tv && rc >= 0) ? v : -1000.0f;
    }
}

}  // namespace

extern "C" {

size_t gp_ist_workspace_bytes(int B, int k, int D, int H)
{
    if (B <= 0 || k <= 0) return 0;
    const size_t R = (size_t)B * k * GP_P;
    return sizeof(float) * R * ((size_t)2 * D + 2 * H + H) + sizeof(int) * (R + 64);  // X, H1, H2, compact positions, row counter
}

int gp_ist_regress(const float* tar_feat, const float* src_bank, const int* labels, const long long* id_src,
                   const long long* tar_pts, const long long* src_pts, int B, int O, int N, int k, int D,
                   int H, const float* const* weights, int n_weights, int use_tanh, float* workspace,
                   size_t workspace_bytes, float* scales, float* cos_sin, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    GP_REQUIRE(B >= 0 && O > 0 && N > 0 && k > 0, "gp_ist_regress: bad sizes");
    GP_REQUIRE(D > 0 && (2 * D) % 16 == 0 && H % 128 == 0 && H > 0,
               "gp_ist_regress: descriptor %d / hidden %d not supported (2D %% 16, H %% 128)", D, H);
    GP_REQUIRE(n_weights != 20 || (2 * D) % 32 == 0, "gp_ist_regress: split numerics need 2D %% 32 == 0");
    GP_REQUIRE(n_weights == 12 || n_weights == 20, "gp_ist_regress: expected 12 (or 20: + split planes) weight pointers, got %d", n_weights);
    if (B == 0) return GP_OK;
    GP_REQUIRE(tar_feat && src_bank && labels && id_src && tar_pts && src_pts && weights && workspace && scales &&
                   cos_sin, "gp_ist_regress: null pointer");
    GP_REQUIRE(workspace_bytes >= gp_ist_workspace_bytes(B, k, D, H), "gp_ist_regress: workspace too small");
    for (int i = 0; i < n_weights; ++i) GP_REQUIRE(weights[i], "gp_ist_regress: weight pointer %d is null", i);
    const bool split = n_weights == 20;  // entries 12..19: per head W1 hi, lo ([2H][2D]) and W2 hi, lo ([H][2H]) f16 planes, w ~= hi + lo 2^-11
    const size_t R = (size_t)B * k * GP_P;
    GP_REQUIRE(R < (size_t)1 << 31, "gp_ist_regress: too many rows");
    float* X = workspace;
    float* H1 = X + (size_t)2 * D * R;
    float* H2 = H1 + (size_t)2 * H * R;
    int* pos = reinterpret_cast<int*>(H2 + (size_t)H * R);
    int* count = pos + R;
    if (hipMemsetAsync(count, 0, sizeof(int), st) != hipSuccess) return GP_ELAUNCH;
    hipLaunchKernelGGL(ist_gather_kernel, dim3(B * k), dim3(256), 0, st, tar_feat, src_bank, labels, id_src,
                       tar_pts, src_pts, O, N, k, D, R, X, pos, count, gp_status_buffer());
    GP_CHECK_LAUNCH("gp_ist_regress/gather");
    int rc;
    for (int head = 0; head < 2; ++head) {  // 0: scale_predictor, 1: inplane_predictor (ist_net.py:140-155)
        const float* const* w = weights + head * 6;  // W1^T [2D][2H], b1, W2^T [2H][H], b2, W3 [nout][H], b3
        if (split) {  // split numerics (3 x f16 MFMA, gp_split.hip): the two hidden layers = 99 % of the head's flops
            const float* const* sp = weights + 12 + head * 4;
            if ((rc = gp_gemm_split_launch_limited(X, (int)R, sp[0], sp[1], H1, (int)R, 2 * H, (int)R, 2 * D, 1, 5, w[1], nullptr, nullptr, 0, count, st)))
                return rc;
            if ((rc = gp_gemm_split_launch_limited(H1, (int)R, sp[2], sp[3], H2, (int)R, H, (int)R, 2 * H, 1, 5, w[3], nullptr, nullptr, 0, count, st)))
                return rc;
        } else {
            if ((rc = gp_gemm_launch(w[0], 2 * H, X, (int)R, H1, (int)R, 2 * H, (int)R, 2 * D, 5, w[1], nullptr, nullptr,
                                     0, nullptr, st)))
                return rc;
            if ((rc = gp_gemm_launch(w[2], H, H1, (int)R, H2, (int)R, H, (int)R, 2 * H, 5, w[3], nullptr, nullptr, 0,
                                     nullptr, st)))
                return rc;
        }
        if (head == 0)
            hipLaunchKernelGGL(ist_head_kernel<1>, dim3((unsigned)(R / 256)), dim3(256), 0, st, H2, w[4], w[5],
                               tar_pts, src_pts, pos, H, R, 0, scales);
        else
            hipLaunchKernelGGL(ist_head_kernel<2>, dim3((unsigned)(R / 256)), dim3(256), 0, st, H2, w[4], w[5],
                               tar_pts, src_pts, pos, H, R, use_tanh, cos_sin);
        GP_CHECK_LAUNCH("gp_ist_regress/head");
    }
    return GP_OK;
}

}  // extern "C"
