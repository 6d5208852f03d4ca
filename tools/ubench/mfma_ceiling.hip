// Micro-benchmark (gfx950): what does the matrix pipe of ONE MI355X socket sustain on v_mfma_f32_32x32x16_f16, and what decides it?
// The guide (MI355X_MICROARCH.md: "Peak BF16/FP16 MFMA") reports 2495 TFLOP/s measured; round 3's MFMA-only loop here reached 1.7
// PFLOP/s on the whole chip (profiles/r03_ubench_mfma_valu.txt).  This file separates the candidates the round-5 review named:
// operand DATA (zeros / a small ramp / random normal), operand VARIETY (the same two registers every instruction vs different
// registers every instruction, as in a GEMM), BURST LENGTH (every configuration runs >= 1 s; socket power and shader clock are sampled
// through rocm_smi in the second half of the run), waves per SIMD, and the LDS fragment traffic of the 3-product split pattern.
//   R  register loop: NOPS distinct A and B fragments per wave, rotated so that consecutive instructions see different operands;
//      4 independent accumulators; no memory access inside the loop.
//   L  split-pattern loop: a wave tile of RB x CB 32 x 32 blocks; per k16 step it reads 2 RB + 2 CB operand fragments (hi and lo
//      planes) from LDS with ds_read_b128 and issues 3 RB CB MFMAs (hi hi, hi lo, lo hi) into RB CB accumulators:
//      RB x CB = 2 x 4 is gemm_planes256_kernel's wave tile (12 reads per 24 MFMAs, 2 waves per SIMD),
//      4 x 4 the one-wave-per-SIMD variant the review asks for (16 reads per 48 MFMAs).
// TFLOP/s = 2 * 32 * 32 * 16 flops per instruction / event time; "clk" = the shader clock measured INSIDE the kernel (s_memtime
// ticks against the 100 MHz s_memrealtime counter, wave 0 of workgroup 0, last launch); "pipe" = matrix cycles issued per SIMD /
// (kernel time x clk) = how much of the pipe's issue capacity at that clock was used.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_mfma_ceiling tools/ubench/mfma_ceiling.hip -lrocm_smi64 ; run: no arguments.
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

typedef _Float16 v16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma1(f32x16& acc, v16x8 a, v16x8 b)
{
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

__device__ __forceinline__ f32x16 mfmab(f32x16 acc, v16x8 a, v16x8 b)   // compiler-allocated: accumulators may sit in AGPRs
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}

struct Clk { unsigned long long cyc, wall; };

// ---- R: operands stay in registers ------------------------------------------------------------------------------------------------
template <int NOPS, int THREADS>
__global__ __launch_bounds__(THREADS) void k_regs(const v16x8* __restrict__ ops, float* out, int iters, Clk* clk)
{
    const int lane = threadIdx.x & 63;
    v16x8 a[NOPS], b[NOPS];
#pragma unroll
    for (int i = 0; i < NOPS; ++i) { a[i] = ops[(2 * i) * 64 + lane]; b[i] = ops[(2 * i + 1) * 64 + lane]; }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) mfma1(acc[m & 3], a[m % NOPS], b[(m + m / NOPS) % NOPS]);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk->cyc = c1 - c0; clk->wall = w1 - w0; }
}

// ---- L: the split pattern with its LDS fragment reads -----------------------------------------------------------------------------
constexpr int LDS_FRAGS = 128;   // 128 fragments x 1 KB (64 lanes x 16 bytes) = 128 KB of random operand data per workgroup
template <int RB, int CB, int THREADS>
__global__ __launch_bounds__(THREADS) void k_lds(const v16x8* __restrict__ ops, float* out, int iters, Clk* clk)
{
    extern __shared__ v16x8 frag[];   // [LDS_FRAGS][64]
    for (int i = threadIdx.x; i < LDS_FRAGS * 64; i += blockDim.x) frag[i] = ops[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[RB * CB];
    for (int q = 0; q < RB * CB; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    constexpr int NF = 2 * RB + 2 * CB;
    int base = (wave * 7) % LDS_FRAGS;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        v16x8 ah[RB], al[RB], bh[CB], bl[CB];
#pragma unroll
        for (int r = 0; r < RB; ++r) { ah[r] = frag[((base + 2 * r) % LDS_FRAGS) * 64 + lane]; al[r] = frag[((base + 2 * r + 1) % LDS_FRAGS) * 64 + lane]; }
#pragma unroll
        for (int c = 0; c < CB; ++c) { bh[c] = frag[((base + 2 * RB + 2 * c) % LDS_FRAGS) * 64 + lane]; bl[c] = frag[((base + 2 * RB + 2 * c + 1) % LDS_FRAGS) * 64 + lane]; }
        base = (base + NF) % LDS_FRAGS;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                mfma1(acc[r * CB + c], ah[r], bh[c]);
                mfma1(acc[r * CB + c], ah[r], bl[c]);
                mfma1(acc[r * CB + c], al[r], bh[c]);
            }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int q = 0; q < RB * CB; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk->cyc = c1 - c0; clk->wall = w1 - w0; }
}

// ---- LP: the same pattern, software-pipelined in registers (the fragments of step i + 1 are read while step i multiplies) -----------
template <int RB, int CB, int THREADS>
__global__ __launch_bounds__(THREADS) void k_lds_pipe(const v16x8* __restrict__ ops, float* out, int iters, Clk* clk)
{
    extern __shared__ v16x8 frag[];
    for (int i = threadIdx.x; i < LDS_FRAGS * 64; i += blockDim.x) frag[i] = ops[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[RB * CB];
    for (int q = 0; q < RB * CB; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    constexpr int NF = 2 * RB + 2 * CB;
    int base = (wave * 7) % LDS_FRAGS;
    v16x8 cur[NF], nxt[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) cur[f] = frag[((base + f) % LDS_FRAGS) * 64 + lane];
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        base = (base + NF) % LDS_FRAGS;
#pragma unroll
        for (int f = 0; f < NF; ++f) nxt[f] = frag[((base + f) % LDS_FRAGS) * 64 + lane];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                mfma1(acc[r * CB + c], cur[2 * r], cur[2 * RB + 2 * c]);
                mfma1(acc[r * CB + c], cur[2 * r], cur[2 * RB + 2 * c + 1]);
                mfma1(acc[r * CB + c], cur[2 * r + 1], cur[2 * RB + 2 * c]);
            }
#pragma unroll
        for (int f = 0; f < NF; ++f) cur[f] = nxt[f];
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int q = 0; q < RB * CB; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk->cyc = c1 - c0; clk->wall = w1 - w0; }
}

// ---- LPP: as LP with two register sets used in turn (loop unrolled by two k-steps: no register copies) ----------------------------------
template <int RB, int CB, int THREADS>
__global__ __launch_bounds__(THREADS) void k_lds_pp(const v16x8* __restrict__ ops, float* out, int iters, Clk* clk)
{
    extern __shared__ v16x8 frag[];
    for (int i = threadIdx.x; i < LDS_FRAGS * 64; i += blockDim.x) frag[i] = ops[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[RB * CB];
    for (int q = 0; q < RB * CB; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    constexpr int NF = 2 * RB + 2 * CB;
    int base = (wave * 7) % LDS_FRAGS;
    v16x8 s0[NF], s1[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) s0[f] = frag[((base + f) % LDS_FRAGS) * 64 + lane];
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it += 2) {
        base = (base + NF) % LDS_FRAGS;
#pragma unroll
        for (int f = 0; f < NF; ++f) s1[f] = frag[((base + f) % LDS_FRAGS) * 64 + lane];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                acc[r * CB + c] = mfmab(acc[r * CB + c], s0[2 * r], s0[2 * RB + 2 * c]);
                acc[r * CB + c] = mfmab(acc[r * CB + c], s0[2 * r], s0[2 * RB + 2 * c + 1]);
                acc[r * CB + c] = mfmab(acc[r * CB + c], s0[2 * r + 1], s0[2 * RB + 2 * c]);
            }
        base = (base + NF) % LDS_FRAGS;
#pragma unroll
        for (int f = 0; f < NF; ++f) s0[f] = frag[((base + f) % LDS_FRAGS) * 64 + lane];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                acc[r * CB + c] = mfmab(acc[r * CB + c], s1[2 * r], s1[2 * RB + 2 * c]);
                acc[r * CB + c] = mfmab(acc[r * CB + c], s1[2 * r], s1[2 * RB + 2 * c + 1]);
                acc[r * CB + c] = mfmab(acc[r * CB + c], s1[2 * r + 1], s1[2 * RB + 2 * c]);
            }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int q = 0; q < RB * CB; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    for (int f = 0; f < NF; ++f) s += (float)s0[f][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk->cyc = c1 - c0; clk->wall = w1 - w0; }
}

// ---- host -------------------------------------------------------------------------------------------------------------------------
static bool g_smi = false;
struct Sample { double watts = 0, mhz = 0; int n = 0; };

static void sample_smi(Sample& s)
{
    if (!g_smi) return;
    uint64_t p = 0;
    RSMI_POWER_TYPE type;
    rsmi_frequencies_t f;
    if (rsmi_dev_power_get(0, &p, &type) == RSMI_STATUS_SUCCESS) s.watts += p * 1e-6;
    if (rsmi_dev_gpu_clk_freq_get(0, RSMI_CLK_TYPE_SYS, &f) == RSMI_STATUS_SUCCESS && f.current < RSMI_MAX_NUM_FREQUENCIES) s.mhz += f.frequency[f.current] * 1e-6;
    s.n += 1;
}

enum Data { ZERO, RAMP, NORMAL };
static const char* data_name[] = {"zeros", "ramp", "normal"};

static void fill(std::vector<_Float16>& h, Data d)
{
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (size_t i = 0; i < h.size(); ++i) {
        const int lane = (i / 8) % 64, e = i % 8;
        h[i] = d == ZERO ? (_Float16)0.f : d == RAMP ? (_Float16)(0.001f * (lane + e)) : (_Float16)nd(rng);   // ramp = round 3's operands
    }
}

template <typename Launch>
static void run(const char* name, Data d, int waves, double mfma_per_wave_iter, int iters, v16x8* ops, Clk* clk, Launch launch, double seconds = 1.2)
{
    std::vector<_Float16> h((size_t)LDS_FRAGS * 64 * 8);
    fill(h, d);
    hipMemcpy(ops, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(iters);
    hipDeviceSynchronize();
    // calibrate the number of launches for ~`seconds` of back-to-back work
    hipEventRecord(e0, 0);
    for (int r = 0; r < 4; ++r) launch(iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms4 = 0.f;
    hipEventElapsedTime(&ms4, e0, e1);
    const int reps = std::max(8, (int)(seconds * 1e3 / (ms4 / 4)));
    Sample smp;
    std::atomic<bool> stop{false};
    const auto t0 = std::chrono::steady_clock::now();
    std::thread sampler([&] {
        while (!stop.load()) {
            const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (t > 0.5 * seconds) sample_smi(smp);
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
    });
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch(iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    stop.store(true);
    sampler.join();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    Clk c;
    hipMemcpy(&c, clk, sizeof(Clk), hipMemcpyDeviceToHost);
    const double per_launch_s = ms * 1e-3 / reps;
    const double flops = 256.0 * waves * iters * mfma_per_wave_iter * 2.0 * 32 * 32 * 16;
    const double tf = flops / per_launch_s * 1e-12;
    const double mhz = c.wall ? (double)c.cyc / ((double)c.wall * 10e-9) * 1e-6 : 0.0;   // s_memrealtime: 100 MHz
    // matrix cycles per SIMD: waves / 4 waves share a SIMD, 32 cycles per instruction (guide: 32 cyc/SIMD back to back)
    const double pipe = (waves / 4.0) * iters * mfma_per_wave_iter * 32.0 / (per_launch_s * mhz * 1e6);
    printf("%-44s %-7s %2d waves/SIMD  %7.1f us/launch x %5d  %7.1f TFLOP/s  clk %6.0f MHz  pipe %.3f  smi: %6.0f W %5.0f MHz (%d samples, %.2f s)\n", name,
           data_name[d], waves / 4, per_launch_s * 1e6, reps, tf, mhz, pipe, smp.n ? smp.watts / smp.n : 0.0, smp.n ? smp.mhz / smp.n : 0.0, smp.n, ms * 1e-3);
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv)
{
    // `brief [seconds]`: the four lines bench.py quotes (zeros / random data in registers, the split pattern fed from LDS), 0.4 s each
    const bool brief = argc > 1 && std::string(argv[1]) == "brief";
    const double seconds = brief ? (argc > 2 ? atof(argv[2]) : 0.4) : (argc > 1 ? atof(argv[1]) : 1.2);
    v16x8* ops; float* out; Clk* clk;
    if (hipMalloc(&ops, (size_t)LDS_FRAGS * 64 * sizeof(v16x8)) != hipSuccess) { printf("no device\n"); return 1; }
    hipMalloc(&out, sizeof(float) * 512 * 256);
    hipMalloc(&clk, sizeof(Clk));
    g_smi = rsmi_init(0) == RSMI_STATUS_SUCCESS;
    uint64_t cap = 0;
    if (g_smi && rsmi_dev_power_cap_get(0, 0, &cap) == RSMI_STATUS_SUCCESS) printf("# rocm_smi: power cap %.0f W\n", cap * 1e-6);
    else printf("# rocm_smi: %s\n", g_smi ? "no power cap reported" : "not available (power / clock columns are zero)");
    printf("# v_mfma_f32_32x32x16_f16, 256 workgroups (one per CU), every configuration >= %.1f s back to back; dense peak 2500 TFLOP/s = 2.4 GHz x 1024 SIMDs x 1024 flops/cycle\n", seconds);
    const int blocks = 256;
    const size_t lds = (size_t)LDS_FRAGS * 64 * sizeof(v16x8);
    hipFuncSetAttribute((const void*)k_lds<2, 4, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_lds<2, 4, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_lds<4, 4, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_lds_pp<2, 4, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_lds_pp<2, 4, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_lds_pp<4, 4, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_lds_pp<4, 2, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_lds_pipe<2, 4, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_lds_pipe<2, 4, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_lds_pipe<4, 4, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (brief) {
        run("R  same 2 operand registers every MFMA", ZERO, 8, 16, 4000, ops, clk, [&](int it) { hipLaunchKernelGGL((k_regs<1, 512>), dim3(blocks), dim3(512), 0, 0, ops, out, it, clk); }, seconds);
        run("R  same 2 operand registers every MFMA", NORMAL, 8, 16, 4000, ops, clk, [&](int it) { hipLaunchKernelGGL((k_regs<1, 512>), dim3(blocks), dim3(512), 0, 0, ops, out, it, clk); }, seconds);
        run("R  8 + 8 operand registers, rotated", NORMAL, 4, 16, 4000, ops, clk, [&](int it) { hipLaunchKernelGGL((k_regs<8, 256>), dim3(blocks), dim3(256), 0, 0, ops, out, it, clk); }, seconds);
        run("LPP 2x4 wave tile, two register sets", NORMAL, 8, 24, 2600, ops, clk, [&](int it) { hipLaunchKernelGGL((k_lds_pp<2, 4, 512>), dim3(blocks), dim3(512), lds, 0, ops, out, it, clk); }, seconds);
        if (g_smi) rsmi_shut_down();
        return 0;
    }
    for (Data d : {ZERO, NORMAL}) {   // (a small ramp, round 3's operands, behaves like normal data: profiles/r06_ubench_mfma_ceiling.txt, first run)
        for (int waves : {8, 4}) {
            run("R  same 2 operand registers every MFMA", d, waves, 16, 4000, ops, clk, [&](int it) {
                if (waves == 8) hipLaunchKernelGGL((k_regs<1, 512>), dim3(blocks), dim3(512), 0, 0, ops, out, it, clk);
                else hipLaunchKernelGGL((k_regs<1, 256>), dim3(blocks), dim3(256), 0, 0, ops, out, it, clk); }, seconds);
            run("R  8 + 8 operand registers, rotated", d, waves, 16, 4000, ops, clk, [&](int it) {
                if (waves == 8) hipLaunchKernelGGL((k_regs<8, 512>), dim3(blocks), dim3(512), 0, 0, ops, out, it, clk);
                else hipLaunchKernelGGL((k_regs<8, 256>), dim3(blocks), dim3(256), 0, 0, ops, out, it, clk); }, seconds);
        }
        run("L  2x4 wave tile: 12 ds_read_b128 / 24 MFMA", d, 8, 24, 2600, ops, clk, [&](int it) { hipLaunchKernelGGL((k_lds<2, 4, 512>), dim3(blocks), dim3(512), lds, 0, ops, out, it, clk); }, seconds);
        run("L  2x4 wave tile: 12 ds_read_b128 / 24 MFMA", d, 4, 24, 2600, ops, clk, [&](int it) { hipLaunchKernelGGL((k_lds<2, 4, 256>), dim3(blocks), dim3(256), lds, 0, ops, out, it, clk); }, seconds);
        run("L  4x4 wave tile: 16 ds_read_b128 / 48 MFMA", d, 4, 48, 1300, ops, clk, [&](int it) { hipLaunchKernelGGL((k_lds<4, 4, 256>), dim3(blocks), dim3(256), lds, 0, ops, out, it, clk); }, seconds);
        run("LP 2x4 wave tile, reads one step ahead", d, 8, 24, 2600, ops, clk, [&](int it) { hipLaunchKernelGGL((k_lds_pipe<2, 4, 512>), dim3(blocks), dim3(512), lds, 0, ops, out, it, clk); }, seconds);
        run("LP 2x4 wave tile, reads one step ahead", d, 4, 24, 2600, ops, clk, [&](int it) { hipLaunchKernelGGL((k_lds_pipe<2, 4, 256>), dim3(blocks), dim3(256), lds, 0, ops, out, it, clk); }, seconds);
        run("LP 4x4 wave tile, reads one step ahead", d, 4, 48, 1300, ops, clk, [&](int it) { hipLaunchKernelGGL((k_lds_pipe<4, 4, 256>), dim3(blocks), dim3(256), lds, 0, ops, out, it, clk); }, seconds);
        run("LPP 2x4 wave tile, two register sets", d, 8, 24, 2600, ops, clk, [&](int it) { hipLaunchKernelGGL((k_lds_pp<2, 4, 512>), dim3(blocks), dim3(512), lds, 0, ops, out, it, clk); }, seconds);
        run("LPP 2x4 wave tile, two register sets", d, 4, 24, 2600, ops, clk, [&](int it) { hipLaunchKernelGGL((k_lds_pp<2, 4, 256>), dim3(blocks), dim3(256), lds, 0, ops, out, it, clk); }, seconds);
        run("LPP 4x2 wave tile, two register sets", d, 4, 24, 2600, ops, clk, [&](int it) { hipLaunchKernelGGL((k_lds_pp<4, 2, 256>), dim3(blocks), dim3(256), lds, 0, ops, out, it, clk); }, seconds);
        run("LPP 4x4 wave tile, two register sets", d, 4, 48, 1300, ops, clk, [&](int it) { hipLaunchKernelGGL((k_lds_pp<4, 4, 256>), dim3(blocks), dim3(256), lds, 0, ops, out, it, clk); }, seconds);
        run("R  2 + 2 operand registers, rotated", d, 8, 16, 4000, ops, clk, [&](int it) { hipLaunchKernelGGL((k_regs<2, 512>), dim3(blocks), dim3(512), 0, 0, ops, out, it, clk); }, seconds);
        run("R  4 + 4 operand registers, rotated", d, 8, 16, 4000, ops, clk, [&](int it) { hipLaunchKernelGGL((k_regs<4, 512>), dim3(blocks), dim3(512), 0, 0, ops, out, it, clk); }, seconds);
    }
    // burst length: the same random-data register loop as ONE short launch after an idle second (what a 3 x 250 us measurement sees)
    {
        std::vector<_Float16> h((size_t)LDS_FRAGS * 64 * 8);
        fill(h, NORMAL);
        hipMemcpy(ops, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int it : {250, 1000, 4000, 16000}) {
            hipDeviceSynchronize();
            std::this_thread::sleep_for(std::chrono::milliseconds(1000));
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k_regs<8, 512>), dim3(blocks), dim3(512), 0, 0, ops, out, it, clk);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            Clk c;
            hipMemcpy(&c, clk, sizeof(Clk), hipMemcpyDeviceToHost);
            printf("burst: ONE launch of %5d iterations after 1 s idle (normal data, 8 + 8 registers, 2 waves/SIMD): %8.1f us  %7.1f TFLOP/s  clk %6.0f MHz\n", it, ms * 1e3,
                   256.0 * 8 * it * 16 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) * 1e-12, c.wall ? (double)c.cyc / ((double)c.wall * 10e-9) * 1e-6 : 0.0);
        }
    }
    if (g_smi) rsmi_shut_down();
    return 0;
}
