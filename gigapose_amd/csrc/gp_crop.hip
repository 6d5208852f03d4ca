// Detection pre-processing on the GPU: the step immediately before the hot path.
//   reference: CropResizePad.__call__            src/utils/crop.py:11-61
//              process_real (rgb/255 * mask)     src/dataloader/train.py:80-123
//              collate_fn normalize(real_data.rgb) src/dataloader/test.py:295-315, configs/data/transform.yaml:1-12
// The reference runs this per detection in Python on the CPU (DataLoader workers): slice, nearest
// F.interpolate(scale_factor), F.pad, nearest F.interpolate(size) -- four passes and three temporaries per
// detection.  Both resizes are nearest-neighbour, so the whole chain is ONE gather: output pixel (y, x) ->
// index in the padded image -> index in the scaled crop -> source pixel.  The kernels below compose the index
// maps with exactly the integer/float arithmetic of ATen's nearest kernels (restated in oracle/crop_numpy.py,
// which is pinned bit-exactly to the reference golden) and write each output pixel once: HBM-bound, coalesced
// stores, one block per output row.
#include <math.h>

#include "gp_common.h"

namespace {

struct CropGeom {
    int x0, y0, cw, ch;      // crop window after the border clamp
    int h1, w1;              // size after the first resize
    int pad_t, pad_l, hp, wp;
    float inv1;              // (float)(1.0 / scale): source-index scale of the first resize
    int mode1y, mode1x;      // 0 identity, 1 dst >> 1, 2 floorf(dst * inv1)
    float s2y, s2x;          // in/out scales of the final resize
    int mode2y, mode2x;
    float scale32;
    int bad;
};

__device__ __forceinline__ int resize_mode(int out, int in) { return out == in ? 0 : (out == 2 * in ? 1 : 2); }

// ATen nearest_idx (UpSample.h): identity / >>1 shortcuts, else min(floorf(dst * scale), in - 1)
__device__ __forceinline__ int nearest_src(int dst, int in, int mode, float scale)
{
    if (mode == 0) return dst;
    if (mode == 1) return dst >> 1;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

__device__ void make_geom(const long long* box, int H, int W, int target, CropGeom& g)
{
    const long long bx0 = box[0], by0 = box[1], bx1 = box[2], by1 = box[3];
    g.bad = !(0 <= bx0 && bx0 < bx1 && 0 <= by0 && by0 < by1 && bx0 < W && by0 < H && bx1 - bx0 < (1 << 24) &&
              by1 - by0 < (1 << 24));
    if (g.bad) return;
    g.x0 = (int)bx0;
    g.y0 = (int)by0;
    const int bw = (int)(bx1 - bx0), bh = (int)(by1 - by0);
    g.scale32 = (float)target / (float)(bw > bh ? bw : bh);            // crop.py:20 (float32 tensor division)
    const double scale = (double)g.scale32;                            // .item()
    g.cw = (int)((bx1 < W ? bx1 : W) - bx0);                            // slicing clamps (crop.py:31)
    g.ch = (int)((by1 < H ? by1 : H) - by0);
    g.h1 = (int)floor((double)g.ch * scale);                           // F.interpolate(scale_factor): floor(in * s)
    g.w1 = (int)floor((double)g.cw * scale);
    if (g.h1 <= 0 || g.w1 <= 0) { g.bad = 1; return; }
    g.inv1 = (float)(1.0 / scale);
    g.mode1y = resize_mode(g.h1, g.ch);
    g.mode1x = resize_mode(g.w1, g.cw);
    g.pad_t = g.pad_l = 0;
    g.hp = g.h1;
    g.wp = g.w1;
    if (g.w1 != g.h1) {                                                // crop.py:37-47
        g.pad_t = (target - g.h1) >= 0 ? (target - g.h1) / 2 : -((g.h1 - target + 1) / 2);  // Python floor division
        int pad_b = target - g.h1 - g.pad_t;
        if (pad_b < 0) pad_b = 0;
        g.pad_l = (target - g.w1) >= 0 ? (target - g.w1) / 2 : -((g.w1 - target + 1) / 2);
        if (g.pad_l < 0) g.pad_l = 0;
        const int pad_r = target - g.w1 - g.pad_l;
        g.hp = g.h1 + g.pad_t + pad_b;
        g.wp = g.w1 + g.pad_l + pad_r;
    }
    if (g.hp <= 0 || g.wp <= 0) { g.bad = 1; return; }
    g.mode2y = resize_mode(target, g.hp);
    g.mode2x = resize_mode(target, g.wp);
    g.s2y = (float)g.hp / (float)target;                               // scales not given: in / out
    g.s2x = (float)g.wp / (float)target;
}

// source pixel of output (y, x) inside the frame, or -1 when it falls in the zero padding
__device__ __forceinline__ int source_y(const CropGeom& g, int y)
{
    const int yp = nearest_src(y, g.hp, g.mode2y, g.s2y) - g.pad_t;
    if (yp < 0 || yp >= g.h1) return -1;
    return g.y0 + nearest_src(yp, g.ch, g.mode1y, g.inv1);
}
__device__ __forceinline__ int source_x(const CropGeom& g, int x)
{
    const int xp = nearest_src(x, g.wp, g.mode2x, g.s2x) - g.pad_l;
    if (xp < 0 || xp >= g.w1) return -1;
    return g.x0 + nearest_src(xp, g.cw, g.mode1x, g.inv1);
}

__device__ __forceinline__ void write_M(const CropGeom& g, float* M)
{
    // M = M_resize_pad @ M_crop (crop.py:26-49): [[s, 0, s*(-x0) + pad_l], [0, s, s*(-y0) + pad_t], [0, 0, 1]]
    const float s = g.scale32;
    const float pl = g.w1 != g.h1 ? (float)g.pad_l : 0.f, pt = g.w1 != g.h1 ? (float)g.pad_t : 0.f;
    M[0] = s; M[1] = 0.f; M[2] = s * (-(float)g.x0) + pl;
    M[3] = 0.f; M[4] = s; M[5] = s * (-(float)g.y0) + pt;
    M[6] = 0.f; M[7] = 0.f; M[8] = 1.f;
}

// grid (target rows, D); block = 256 threads, thread = output column
__global__ __launch_bounds__(256) void crop_resize_pad_kernel(const float* __restrict__ images,
                                                               const long long* __restrict__ boxes, int C, int H, int W,
                                                               int target, float* __restrict__ out, float* __restrict__ M,
                                                               int* __restrict__ err)
{
    __shared__ CropGeom g;
    const int d = blockIdx.y, y = blockIdx.x;
    if (threadIdx.x == 0) {
        make_geom(boxes + 4 * d, H, W, target, g);
        if (y == 0) {
            if (g.bad) atomicExch(err, d + 1);
            else write_M(g, M + 9 * d);
        }
    }
    __syncthreads();
    if (g.bad) return;
    const int sy = source_y(g, y);
    for (int x = threadIdx.x; x < target; x += blockDim.x) {
        const int sx = sy < 0 ? -1 : source_x(g, x);
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
            if (sx >= 0) v = images[(((size_t)d * C + c) * H + sy) * W + sx];
            out[(((size_t)d * C + c) * target + y) * target + x] = v;
        }
    }
}

__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ rgb, const float* __restrict__ masks,
                                                          const long long* __restrict__ boxes, const int* __restrict__ im_id,
                                                          int n_img, int H, int W, int target, float m0, float m1, float m2,
                                                          float s0, float s1, float s2, float* __restrict__ tar_img,
                                                          float* __restrict__ tar_mask, float* __restrict__ M,
                                                          int* __restrict__ err)
{
    __shared__ CropGeom g;
    __shared__ int img;
    const int d = blockIdx.y, y = blockIdx.x;
    if (threadIdx.x == 0) {
        make_geom(boxes + 4 * d, H, W, target, g);
        img = im_id[d];
        if (img < 0 || img >= n_img) g.bad = 1;
        if (y == 0) {
            if (g.bad) atomicExch(err, d + 1);
            else write_M(g, M + 9 * d);
        }
    }
    __syncthreads();
    if (g.bad) return;
    const int sy = source_y(g, y);
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    const size_t plane = (size_t)H * W;
    for (int x = threadIdx.x; x < target; x += blockDim.x) {
        const int sx = sy < 0 ? -1 : source_x(g, x);
        float m = 0.f;
        float v[3] = {0.f, 0.f, 0.f};
        if (sx >= 0) {
            const size_t o = (size_t)sy * W + sx;
            m = masks[(size_t)d * plane + o];
#pragma unroll
            for (int c = 0; c < 3; ++c)  // rgb / 255.0 * mask (train.py:83,107)
                v[c] = ((float)rgb[((size_t)img * 3 + c) * plane + o] / 255.0f) * m;
        }
        const size_t po = (size_t)y * target + x, tt = (size_t)target * target;
#pragma unroll
        for (int c = 0; c < 3; ++c)  // torchvision Normalize: (x - mean) / std
            tar_img[((size_t)d * 3 + c) * tt + po] = (v[c] - mean[c]) / stdv[c];
        tar_mask[(size_t)d * tt + po] = m;
    }
}

}  // namespace

extern "C" {

int gp_crop_resize_pad(const float* images, const long long* boxes, int D, int C, int H, int W, int target, float* out,
                       float* M, int* err_flag, void* stream)
{
    GP_REQUIRE(D >= 0 && C > 0 && H > 0 && W > 0 && target > 0 && target <= 4096, "gp_crop_resize_pad: bad sizes");
    if (D == 0) return GP_OK;
    GP_REQUIRE(images && boxes && out && M && err_flag, "gp_crop_resize_pad: null pointer");
    GpProfScope prof(GP_PROF_OTHER, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(crop_resize_pad_kernel, dim3(target, D), dim3(256), 0, (hipStream_t)stream, images, boxes, C, H, W,
                       target, out, M, err_flag);
    GP_CHECK_LAUNCH("gp_crop_resize_pad");
    return GP_OK;
}

int gp_preprocess_detections(const uint8_t* rgb, const float* masks, const long long* boxes, const int* im_id, int n_img,
                             int D, int H, int W, int target, const float* mean3_host, const float* std3_host,
                             float* tar_img, float* tar_mask, float* M, int* err_flag, void* stream)
{
    GP_REQUIRE(D >= 0 && n_img > 0 && H > 0 && W > 0 && target > 0 && target <= 4096, "gp_preprocess_detections: bad sizes");
    if (D == 0) return GP_OK;
    GP_REQUIRE(rgb && masks && boxes && im_id && mean3_host && std3_host && tar_img && tar_mask && M && err_flag,
               "gp_preprocess_detections: null pointer");
    GpProfScope prof(GP_PROF_OTHER, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(preprocess_kernel, dim3(target, D), dim3(256), 0, (hipStream_t)stream, rgb, masks, boxes, im_id,
                       n_img, H, W, target, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1],
                       std3_host[2], tar_img, tar_mask, M, err_flag);
    GP_CHECK_LAUNCH("gp_preprocess_detections");
    return GP_OK;
}

}  // extern "C"
