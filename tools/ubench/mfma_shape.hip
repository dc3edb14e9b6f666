// Micro-benchmark: does the bf16 MFMA shape change the FLOP/s a bf16x3-style loop sustains on RANDOM data (the chip
// lowers its clock under load; MI355X_MICROARCH.md, DVFS give-back item 7)?  Two waves per SIMD; every wave runs groups of
// six dependent MFMAs per accumulator block with fragment reads from LDS and a VALU filler behind every MFMA, as the
// library's k-loop does.  32x32x16: one block = 32x32 outputs, 16 k.  16x16x32: four 16x16 blocks x 32 k = the same
// MACs per group of 6 x 4.  Prints TFLOP/s of bf16 MFMA work and the in-kernel clock.  (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE, int FILL>
__global__ __launch_bounds__(512, 2) void k(int iters, const float* __restrict__ src, float* out, long long* clk) {
    __shared__ __attribute__((aligned(16))) char lds[98304];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 98304 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = src[(blockIdx.x * 977 + i) % (1 << 20)];
    __syncthreads();
    float x[4] = {src[tid], src[tid + 512], src[tid + 1024], src[tid + 1536]};
    const long long t0 = clock64(), w0 = wall_clock64();
    float fsum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[5];
        for (int i = 0; i < 5; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int blk = 0; blk < 5; ++blk) {
                bf16x8 f[6];
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    f[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(lds + ((it * 7 + blk * 6 + j) * 1024 + lane * 16) % 98304));
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[j], f[(j + 3) % 6], acc[blk], 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < FILL; ++v) {
                        const unsigned u = __float_as_uint(x[v & 3]) & 0xffff0000u;
                        x[v & 3] = x[(v + 1) & 3] - __uint_as_float(u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        for (int i = 0; i < 5; ++i)
            for (int e = 0; e < 16; ++e) fsum += acc[i][e];
    } else {
        f32x4 acc[20];
        for (int i = 0; i < 20; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int blk = 0; blk < 5; ++blk) {  // the same MACs as one 32x32x16 group of 6 (x2 k): 4 blocks x 6 x (16x16x32)... per 2 k-steps
                bf16x8 f[6];
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    f[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(lds + ((it * 7 + blk * 6 + j) * 1024 + lane * 16) % 98304));
#pragma unroll
                for (int j = 0; j < 6; ++j) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {  // 2 x (16x16x32) = the MACs of one 32x32x16 ... (16*16*32*2 = 32*32*16)
                        acc[blk * 4 + q + 2 * (j & 1)] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[j], f[(j + 3) % 6], acc[blk * 4 + q + 2 * (j & 1)], 0, 0, 0);
                    }
#pragma unroll
                    for (int v = 0; v < FILL; ++v) {
                        const unsigned u = __float_as_uint(x[v & 3]) & 0xffff0000u;
                        x[v & 3] = x[(v + 1) & 3] - __uint_as_float(u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        for (int i = 0; i < 20; ++i)
            for (int e = 0; e < 4; ++e) fsum += acc[i][e];
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    if (tid == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
    if (fsum + x[0] + x[1] + x[2] + x[3] == 123.456f) out[tid] = fsum;
}

template <int SHAPE, int FILL>
void run(const char* name, const float* src, float* out, long long* clk) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, FILL>), dim3(blocks), dim3(512), 0, 0, iters, src, out, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SHAPE, FILL>), dim3(blocks), dim3(512), 0, 0, iters, src, out, clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double macs_per_wave_iter = 5.0 * 6 * 32 * 32 * 16;  // both shapes
    const double flops = 2.0 * macs_per_wave_iter * iters * 8 * blocks * reps;
    printf("%-28s %8.3f ms/launch  %8.1f TFLOP/s bf16   in-kernel clock %.2f GHz  cycles/iter %.0f\n", name, ms / reps,
           flops / (ms * 1e-3) / 1e12, (double)h[0] / ((double)h[1] * 10.0) , (double)h[0] / iters);
}

int main() {
    float *src, *out;
    long long* clk;
    hipMalloc(&src, (1 << 20) * 4 + 8192);
    hipMalloc(&out, 4096);
    hipMalloc(&clk, 256 * 16);
    float* h = (float*)malloc((1 << 20) * 4 + 8192);
    srand(1);
    for (int i = 0; i < (1 << 20) + 2048; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(src, h, (1 << 20) * 4 + 8192, hipMemcpyHostToDevice);
    run<32, 0>("32x32x16, no filler", src, out, clk);
    run<16, 0>("16x16x32, no filler", src, out, clk);
    run<32, 3>("32x32x16, 3 VALU per MFMA", src, out, clk);
    run<16, 3>("16x16x32, 3 VALU per 2 MFMA", src, out, clk);
    run<32, 0>("32x32x16, no filler (again)", src, out, clk);
    run<16, 0>("16x16x32, no filler (again)", src, out, clk);
    return 0;
}
