// Micro-benchmark (gfx950), second part: WHICH vector-ALU instructions overlap with matrix instructions, and is the missing overlap a
// power effect?  Every wave runs, per iteration, 16 independent v_mfma_f32_32x32x16_f16 (512 matrix cycles) and / or 128 vector
// instructions of one kind, interleaved 1 : 8 in program order.  Run on the whole chip (256 workgroups of 8 waves) and on an eighth of
// it (32 workgroups: far below the power limit).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_mfma_valu_kinds tools/ubench/mfma_valu_kinds.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 v16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NM = 16, NV = 128;
enum { K_FMA, K_MUL, K_ADD, K_MAX, K_AND, K_ADDU, K_PKFMA, K_PKADD, K_CVTPK, K_EXP, K_MOV, NKIND };
static const char* kind_name[NKIND] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_and_b32", "v_add_u32", "v_pk_fma_f32", "v_pk_add_f32",
                                       "v_cvt_pk_f16_f32", "v_exp_f32", "v_mov_b32"};

template <int KIND>
__device__ __forceinline__ void vop(float& c, f32x2& p, float a, float b)
{
    if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
    if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c) : "v"(a));
    if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(c) : "v"(b));
    if (KIND == K_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(c) : "v"(b));
    if (KIND == K_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(c) : "v"(a));
    if (KIND == K_ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(a));
    if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
    if (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(p));
    if (KIND == K_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(c) : "v"(a));
    if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(c));
    if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(c) : "v"(a));
}

// WHAT: 1 = vector only, 2 = matrix only, 3 = both interleaved
template <int KIND, int WHAT>
__global__ __launch_bounds__(512) void k(float* out, int iters)
{
    const int lane = threadIdx.x & 63;
    v16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float c[16];
    f32x2 p[16];
    for (int i = 0; i < 16; ++i) { c[i] = 0.5f + 0.01f * i; p[i] = f32x2{0.5f, 0.25f + 0.01f * i}; }
    const float fa = 0.999f, fb = 0.0005f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (WHAT & 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
            if (WHAT & 1) {
#pragma unroll
                for (int i = 0; i < NV / NM; ++i) { const int j = (m * (NV / NM) + i) & 15; vop<KIND>(c[j], p[j], fa, fb); }
            }
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    for (int i = 0; i < 16; ++i) s += c[i] + p[i].x + p[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int WHAT>
static float run(float* out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, WHAT>), dim3(blocks), dim3(512), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<KIND, WHAT>), dim3(blocks), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.f * 1e3f;
}

template <int KIND>
static void kind(float* out, int blocks, int iters, float tm)
{
    const float tv = run<KIND, 1>(out, blocks, iters), tb = run<KIND, 3>(out, blocks, iters);
    const float hidden = (tm + tv - tb) / tv;  // 1 = the vector work is free next to the matrix work, 0 = additive
    printf("  %-18s vector only %7.1f us (%5.2f matrix-cycle units per instruction) | interleaved %7.1f us | max %7.1f, sum %7.1f | hidden fraction %5.2f\n",
           kind_name[KIND], tv, tv / tm * (NM * 32.0) / NV, tb, tm > tv ? tm : tv, tm + tv, hidden);
}

int main()
{
    const int iters = 400;
    float* out;
    if (hipMalloc(&out, sizeof(float) * 512 * 256) != hipSuccess) { printf("no device\n"); return 1; }
    for (int blocks : {256, 32}) {
        const float tm = run<K_FMA, 2>(out, blocks, iters);
        printf("# %d workgroups x 8 waves (2 per SIMD): matrix only %7.1f us = %.2f GHz if the pipe never idles\n", blocks, tm, 2.0 * iters * NM * 32 / tm * 1e-3);
        kind<K_FMA>(out, blocks, iters, tm); kind<K_MUL>(out, blocks, iters, tm); kind<K_ADD>(out, blocks, iters, tm); kind<K_MAX>(out, blocks, iters, tm);
        kind<K_AND>(out, blocks, iters, tm); kind<K_ADDU>(out, blocks, iters, tm); kind<K_PKFMA>(out, blocks, iters, tm); kind<K_PKADD>(out, blocks, iters, tm);
        kind<K_CVTPK>(out, blocks, iters, tm); kind<K_EXP>(out, blocks, iters, tm); kind<K_MOV>(out, blocks, iters, tm);
    }
    return 0;
}
