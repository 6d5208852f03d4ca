// Micro-benchmark (gfx950): do matrix instructions of one wave and vector-ALU instructions of ANOTHER wave on the same SIMD overlap,
// and do they overlap inside one wave when interleaved in program order?  Answers the question behind attention_split_kernel's timing
// (DESIGN.md section 4): its launch time is close to (matrix cycles + vector-ALU cycles), not to their maximum.
//   mode 0: every wave runs NM independent v_mfma_f32_32x32x16_f16 per iteration (4 accumulators)
//   mode 1: every wave runs NV independent v_fma_f32 per iteration (16 chains)
//   mode 2: waves 0..3 of the workgroup (SIMD 0..3) run mode 0's work, waves 4..7 (the second wave of every SIMD) mode 1's
//   mode 3: every wave runs both, interleaved in program order (1 matrix instruction : NV / NM vector instructions)
//   mode 4: every wave runs both, NOT interleaved (all matrix instructions of the iteration, then all vector instructions)
//   mode 5 / 6: as 1 / 3 with v_exp_f32 (quarter rate) for a quarter of the vector instructions
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_mfma_valu_overlap tools/ubench/mfma_valu_overlap.hip ; run: no arguments.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 v16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NM = 16, NV = 128;  // per iteration: 16 matrix instructions = 512 matrix cycles; 128 vector instructions = 512 issue cycles

template <bool EXP>
__device__ __forceinline__ void valu_block(float (&c)[16], float a, float b)
{
#pragma unroll
    for (int r = 0; r < NV / 16; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (EXP && (i & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(c[i]));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c[i]) : "v"(a), "v"(b));
        }
}

__device__ __forceinline__ void mfma1(f32x16& acc, v16x8 a, v16x8 b)
{
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float c[16];
    for (int i = 0; i < 16; ++i) c[i] = 0.5f + 0.01f * i;
    const float fa = 0.999f, fb = 0.0005f;
    constexpr bool EXP = MODE >= 5;
    constexpr int M = MODE == 5 ? 1 : (MODE == 6 ? 3 : MODE);
    const bool do_m = M == 0 || M == 3 || M == 4 || (M == 2 && wave < 4), do_v = M == 1 || M == 3 || M == 4 || (M == 2 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (M == 3) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                mfma1(acc[m & 3], a, b);
#pragma unroll
                for (int i = 0; i < NV / NM; ++i) {
                    const int j = (m * (NV / NM) + i) & 15;
                    if (EXP && (j & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(c[j]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c[j]) : "v"(fa), "v"(fb));
                }
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int m = 0; m < NM; ++m) mfma1(acc[m & 3], a, b);
            }
            if (do_v) valu_block<EXP>(c, fa, fb);
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    for (int i = 0; i < 16; ++i) s += c[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
static float run(float* out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.f * 1e3f;
}

int main()
{
    const int blocks = 256, iters = 400;
    float* out;
    if (hipMalloc(&out, sizeof(float) * 512 * blocks) != hipSuccess) { printf("no device\n"); return 1; }
    const double mcyc = (double)iters * NM * 32, vcyc = (double)iters * NV * 4;
    printf("# 256 workgroups x 8 waves (2 per SIMD), %d iterations; per wave and launch: matrix %.0f cycles, vector issue %.0f cycles\n", iters, mcyc, vcyc);
    const float t0 = run<0>(out, blocks, iters), t1 = run<1>(out, blocks, iters), t2 = run<2>(out, blocks, iters), t3 = run<3>(out, blocks, iters),
                t4 = run<4>(out, blocks, iters), t5 = run<5>(out, blocks, iters), t6 = run<6>(out, blocks, iters);
    printf("mode 0  matrix only, 2 waves per SIMD                     %8.1f us  (-> %.2f GHz if the pipe never idles)\n", t0, 2 * mcyc / t0 * 1e-3);
    printf("mode 1  vector only, 2 waves per SIMD                     %8.1f us  (-> %.2f GHz at 4 cycles per instruction)\n", t1, 2 * vcyc / t1 * 1e-3);
    printf("mode 2  one matrix wave + one vector wave per SIMD        %8.1f us  (overlap: %.1f, additive: %.1f)\n", t2, (t0 > t1 ? t0 : t1) / 2, (t0 + t1) / 2);
    printf("mode 3  both in every wave, interleaved 1 : %d            %8.1f us  (overlap: %.1f, additive: %.1f)\n", NV / NM, t3, t0 > t1 ? t0 : t1, t0 + t1);
    printf("mode 4  both in every wave, matrix block then vector block %7.1f us\n", t4);
    printf("mode 5  vector only with 1/4 v_exp_f32                    %8.1f us\n", t5);
    printf("mode 6  mode 3 with 1/4 v_exp_f32                         %8.1f us  (overlap: %.1f, additive: %.1f)\n", t6, t0 > t5 ? t0 : t5, t0 + t5);
    return 0;
}
