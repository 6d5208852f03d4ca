// IST backbone convolutions on gfx950: implicit-GEMM Conv2d + folded eval-BatchNorm + residual + ReLU.
// Reference: ResNet.forward / BasicBlock.forward (src/models/network/resnet.py:364-381, 26-50) as
// called by ISTNet.forward_by_chunk (src/models/network/ist_net.py:45-51).
//
// Activations are channel-major over the whole batch, X[c][b][y][x] ("CNHW"), so a convolution is the
// same k-major contraction as every other dense op of the path:
//     Y[co][pix] = sum_k Wt[k][co] * Xcol[k][pix],   k = (ci, dy, dx),  pix = (b, oy, ox)
// with Xcol gathered on the fly while staging the B operand into LDS (never materialised), f32-input
// MFMA 32x32x2, accumulation = sequential fmaf chain over k.  One thread owns one output pixel column
// of the 256-pixel block tile: (b, oy, ox) is decoded once, k -> (ci, dy, dx) is wave-uniform (SALU).
#include "gp_common.h"

namespace {

constexpr int CBM = 64, CBN = 256, CKS = 16, CNT = 256;  // block tile: 64 out-channels x 256 pixels

struct ConvArgs {
    const float* X;      // [Cin][B][H][W]
    const float* Wt;     // [Kpad][Cout]   k = ci*KH*KW + dy*KW + dx ; rows >= Kreal are zero
    float* Y;            // [Cout][B][OH][OW]  (or NCHW when nchw_out)
    const float* alpha;  // [Cout] folded BN scale  = gamma / sqrt(var + eps)      (nullable)
    const float* beta;   // [Cout] folded BN shift  = beta - mean * alpha
    const float* res;    // residual [Cout][B][OH][OW]                              (nullable)
    int Cin, B, H, W, OH, OW, Cout, KH, KW, stride, pad, Kreal, Kpad, relu, nchw_out;
};

__global__ __launch_bounds__(CNT, 2) void conv_kernel(ConvArgs p)
{
    __shared__ float smem[2 * CKS * (CBM + CBN)];
    float* sA = smem;                  // [2][16][64]
    float* sB = smem + 2 * CKS * CBM;  // [2][16][256]
    const int tiles_co = p.Cout / CBM;
    const int OHW = p.OH * p.OW;
    const int npix = p.B * OHW;
    const int q = xcd_chunked_tile(blockIdx.x, tiles_co * (npix / CBN));
    if (q < 0) return;
    const int co0 = (q % tiles_co) * CBM, pix0 = (q / tiles_co) * CBN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;

    // this thread's gather column
    const int pix = pix0 + tid;
    const int b = pix / OHW, prem = pix % OHW;
    const int iy0 = (prem / p.OW) * p.stride - p.pad, ix0 = (prem % p.OW) * p.stride - p.pad;
    const size_t chan_stride = (size_t)p.B * p.H * p.W;
    const float* Xb = p.X + (size_t)b * p.H * p.W;
    const int KHW = p.KH * p.KW;
    // this thread's weight float4: row r = tid / 16, columns (tid % 16) * 4
    const float* Wp = p.Wt + (size_t)(tid >> 4) * p.Cout + co0 + (tid & 15) * 4;

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    f32x4 ra;
    float rb[CKS];
    auto gather = [&](int k0) {
        ra = *reinterpret_cast<const f32x4*>(Wp + (size_t)k0 * p.Cout);
#pragma unroll
        for (int r = 0; r < CKS; ++r) {
            const int k = k0 + r;  // wave-uniform
            const int ci = k / KHW, rem = k % KHW;
            const int iy = iy0 + rem / p.KW, ix = ix0 + rem % p.KW;
            const bool ok = (k < p.Kreal) && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W);
            rb[r] = ok ? Xb[(size_t)ci * chan_stride + (size_t)iy * p.W + ix] : 0.f;
        }
    };
    auto stage = [&](int buf) {
        *reinterpret_cast<f32x4*>(sA + buf * CKS * CBM + tid * 4) = ra;
#pragma unroll
        for (int r = 0; r < CKS; ++r) sB[buf * CKS * CBN + r * CBN + tid] = rb[r];
    };

    const int nslab = p.Kpad / CKS;
    gather(0);
    stage(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gather((s + 1) * CKS);
        const float* cA = sA + buf * CKS * CBM;
        const float* cB = sB + buf * CKS * CBN;
#pragma unroll
        for (int kk = 0; kk < CKS / 2; ++kk) {
            const int k = 2 * kk + khalf;
            const float a0 = cA[k * CBM + l31], a1 = cA[k * CBM + 32 + l31];
            const float b0 = cB[k * CBN + wave * 64 + l31], b1 = cB[k * CBN + wave * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (s + 1 < nslab) stage(buf ^ 1);
        __syncthreads();
    }

    // epilogue: y = conv * alpha + beta (eval BatchNorm as torch folds it), + residual, ReLU
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int opix = pix0 + wave * 64 + ni * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + mi * 32 + frag_row(r, lane);
                float v = acc[mi][ni][r];
                if (p.alpha) v = v * p.alpha[co] + p.beta[co];
                if (p.res) v = p.res[(size_t)co * npix + opix] + v;
                if (p.relu) v = fmaxf(v, 0.f);
                const size_t o = p.nchw_out ? ((size_t)(opix / OHW) * p.Cout + co) * OHW + (opix % OHW)
                                            : (size_t)co * npix + opix;
                p.Y[o] = v;
            }
        }
}


// ---------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 (16 of the 21 convolutions, 90 % of the backbone's FLOPs): direct convolution.
// The block's input window (TH+2 x TW+2 halo tile of 8 channels) is staged in LDS ONCE and every
// (ci,dy,dx) B-operand row is a shifted ds_read of it -- 6.5x fewer global loads than gathering
// im2col rows.  Block tile: 64 out-channels x 256 pixels (TW x 256/TW window of one image); k stays in
// ascending (ci,dy,dx) order, so the result is bit-identical to conv_kernel / the oracle.
template <int TW>
__global__ __launch_bounds__(CNT, 2) void conv3x3_kernel(ConvArgs p)
{
    constexpr int TH = 256 / TW, HWID = TW + 2, HHGT = TH + 2, CC = 8, KC = CC * 9;
    constexpr int XN = CC * HHGT * HWID;           // halo-tile floats per chunk
    constexpr int WN4 = KC * CBM / 4;               // weight float4 per chunk (1152)
    constexpr int XU = (XN + CNT - 1) / CNT, WU = (WN4 + CNT - 1) / CNT;
    __shared__ float smem[2 * (KC * CBM + XN)];
    float* sW = smem;                 // [2][72][64]
    float* sX = smem + 2 * KC * CBM;  // [2][8][TH+2][TW+2]
    const int tiles_co = p.Cout / CBM, tx_n = p.W / TW, ty_n = p.H / TH;
    const int q = xcd_chunked_tile(blockIdx.x, tiles_co * tx_n * ty_n * p.B);
    if (q < 0) return;
    const int co0 = (q % tiles_co) * CBM;
    int rest = q / tiles_co;
    const int x0 = (rest % tx_n) * TW; rest /= tx_n;
    const int y0 = (rest % ty_n) * TH;
    const int b = rest / ty_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;
    const size_t chan_stride = (size_t)p.B * p.H * p.W;
    const float* Xb = p.X + (size_t)b * p.H * p.W;

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    f32x4 rw[WU];
    float rx[XU];
    auto gload = [&](int chunk) {
        const float* wsrc = p.Wt + (size_t)chunk * KC * p.Cout + co0;
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int f = tid + u * CNT;
            if (f < WN4) rw[u] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)(f >> 4) * p.Cout + (f & 15) * 4);
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int e = tid + u * CNT;
            float v = 0.f;
            if (e < XN) {
                const int ci = e / (HHGT * HWID), rem = e % (HHGT * HWID);
                const int iy = y0 - 1 + rem / HWID, ix = x0 - 1 + rem % HWID;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    v = Xb[(size_t)(chunk * CC + ci) * chan_stride + (size_t)iy * p.W + ix];
            }
            rx[u] = v;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int f = tid + u * CNT;
            if (f < WN4) *reinterpret_cast<f32x4*>(sW + buf * KC * CBM + f * 4) = rw[u];
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int e = tid + u * CNT;
            if (e < XN) sX[buf * XN + e] = rx[u];
        }
    };
    // this lane's pixel inside the window, for its two 32-pixel MFMA column tiles (nt = 2*wave + ni)
    int pbase[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int nt = 2 * wave + ni;
        const int row = (TW == 32) ? nt : 2 * nt + (l31 >> 4), col = (TW == 32) ? l31 : (l31 & 15);
        pbase[ni] = row * HWID + col;
    }

    const int nchunk = p.Cin / CC;
    gload(0);
    stage(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload(c + 1);
        const float* cW = sW + buf * KC * CBM;
        const float* cX = sX + buf * XN;
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            constexpr int dummy = 0;
            (void)dummy;
            const int k0 = 2 * kk, k1 = 2 * kk + 1;  // compile-time after unrolling
            const int off0 = (k0 / 9) * (HHGT * HWID) + ((k0 % 9) / 3) * HWID + (k0 % 3);
            const int off1 = (k1 / 9) * (HHGT * HWID) + ((k1 % 9) / 3) * HWID + (k1 % 3);
            const int xoff = khalf ? off1 : off0;
            const int k = 2 * kk + khalf;
            const float a0 = cW[k * CBM + l31], a1 = cW[k * CBM + 32 + l31];
            const float b0 = cX[xoff + pbase[0]], b1 = cX[xoff + pbase[1]];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (c + 1 < nchunk) stage(buf ^ 1);
        __syncthreads();
    }

    const int HWo = p.H * p.W;
    const size_t npix = (size_t)p.B * HWo;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int nt = 2 * wave + ni;
            const int row = (TW == 32) ? nt : 2 * nt + (l31 >> 4), col = (TW == 32) ? l31 : (l31 & 15);
            const size_t opix = (size_t)b * HWo + (size_t)(y0 + row) * p.W + (x0 + col);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + mi * 32 + frag_row(r, lane);
                float v = acc[mi][ni][r];
                if (p.alpha) v = v * p.alpha[co] + p.beta[co];
                if (p.res) v = p.res[(size_t)co * npix + opix] + v;
                if (p.relu) v = fmaxf(v, 0.f);
                const size_t o = p.nchw_out ? ((size_t)b * p.Cout + co) * HWo + (opix - (size_t)b * HWo)
                                            : (size_t)co * npix + opix;
                p.Y[o] = v;
            }
        }
}

// F.interpolate(x, (S,S), mode="bilinear", align_corners=True) (resnet.py:366-368), NCHW in ->
// channel-major out [C][B][S][S].  Same arithmetic as ATen's upsample_bilinear2d (float scales).
__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                      int C, int IH, int IW, int S)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, cb = blockIdx.z, c = cb / B, b = cb % B;
    if (x >= S) return;
    const float rh = (S > 1) ? (float)(IH - 1) / (float)(S - 1) : 0.f;
    const float rw = (S > 1) ? (float)(IW - 1) / (float)(S - 1) : 0.f;
    const float h1r = rh * y, w1r = rw * x;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = (h1 < IH - 1) ? 1 : 0, w1p = (w1 < IW - 1) ? 1 : 0;
    const float h1l = h1r - h1, h0l = 1.f - h1l, w1l = w1r - w1, w0l = 1.f - w1l;
    const float* src = in + ((size_t)b * C + c) * IH * IW;
    const float v = h0l * (w0l * src[h1 * IW + w1] + w1l * src[h1 * IW + w1 + w1p]) +
                    h1l * (w0l * src[(h1 + h1p) * IW + w1] + w1l * src[(h1 + h1p) * IW + w1 + w1p]);
    out[(((size_t)c * B + b) * S + y) * S + x] = v;
}

}  // namespace

static bool g_conv_direct = true;

extern "C" {

#ifdef GP_PROBES
/* test hook: 0 forces the generic gather kernel for every shape (both must agree bit-for-bit) */
void gp_conv_set_direct(int on) { g_conv_direct = on != 0; }
#endif

int gp_resize_bilinear_cm(const float* images, float* out, int B, int C, int IH, int IW, int S, void* stream)
{
    GP_REQUIRE(B >= 0 && C > 0 && IH > 0 && IW > 0 && S > 0, "gp_resize_bilinear_cm: bad sizes");
    if (B == 0) return GP_OK;
    GP_REQUIRE(images && out, "gp_resize_bilinear_cm: null pointer");
    GpProfScope prof(GP_PROF_OTHER, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(resize_kernel, dim3((S + 255) / 256, S, C * B), dim3(256), 0, (hipStream_t)stream, images, out,
                       B, C, IH, IW, S);
    GP_CHECK_LAUNCH("gp_resize_bilinear_cm");
    return GP_OK;
}

int gp_conv2d_cm(const float* X, const float* Wt, float* Y, const float* alpha, const float* beta,
                 const float* residual, int Cin, int B, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                 int relu, int nchw_out, void* stream)
{
    GP_REQUIRE(Cin > 0 && B >= 0 && H > 0 && W > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
               "gp_conv2d_cm: bad sizes");
    if (B == 0) return GP_OK;
    GP_REQUIRE(X && Wt && Y, "gp_conv2d_cm: null pointer");
    GP_REQUIRE((alpha == nullptr) == (beta == nullptr), "gp_conv2d_cm: alpha and beta go together");
    ConvArgs a;
    a.X = X; a.Wt = Wt; a.Y = Y; a.alpha = alpha; a.beta = beta; a.res = residual;
    a.Cin = Cin; a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.OH = (H + 2 * pad - KH) / stride + 1;
    a.OW = (W + 2 * pad - KW) / stride + 1;
    a.Kreal = Cin * KH * KW;
    a.Kpad = (a.Kreal + CKS - 1) / CKS * CKS;
    a.relu = relu; a.nchw_out = nchw_out;
    const long long npix = (long long)B * a.OH * a.OW;
    GP_REQUIRE(Cout % CBM == 0, "gp_conv2d_cm: Cout=%d must be a multiple of 64", Cout);
    GP_REQUIRE(npix % CBN == 0 && npix < (1ll << 31), "gp_conv2d_cm: B*OH*OW=%lld must be a multiple of 256", npix);
    GP_REQUIRE((uintptr_t)Wt % 16 == 0, "gp_conv2d_cm: weights must be 16-byte aligned");
    const int tiles = (Cout / CBM) * (int)(npix / CBN);
    GpProfScope prof(GP_PROF_CONV, 2.0 * Cout * (double)npix * a.Kreal, (hipStream_t)stream);
    const bool direct = (KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin % 8 == 0 && g_conv_direct);
    if (direct && W % 32 == 0 && H % 8 == 0)
        hipLaunchKernelGGL(conv3x3_kernel<32>, dim3(xcd_chunked_grid(tiles)), dim3(CNT), 0, (hipStream_t)stream, a);
    else if (direct && W % 16 == 0 && H % 16 == 0)
        hipLaunchKernelGGL(conv3x3_kernel<16>, dim3(xcd_chunked_grid(tiles)), dim3(CNT), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(conv_kernel, dim3(xcd_chunked_grid(tiles)), dim3(CNT), 0, (hipStream_t)stream, a);
    GP_CHECK_LAUNCH("gp_conv2d_cm");
    return GP_OK;
}

}  // extern "C"
