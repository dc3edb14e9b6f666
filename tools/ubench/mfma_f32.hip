// Micro-benchmark: issue rate of the exact-f32 MFMAs (v_mfma_f32_16x16x4_f32 / 32x32x2) per SIMD, 1 or 2 waves per SIMD,
// 1..4 independent accumulator chains.  hipcc --offload-arch=gfx950 -O3 mfma_f32.hip -o mfma_f32 && ./mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ void k16(float* out, int iters, long long* cyc) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
__global__ void k32(float* out, int iters, long long* cyc) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int j = 0; j < 16; ++j) acc[c][j] = 0;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][15];
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename K>
void run(const char* name, K kern, int ch, int threads, double flop_per_mfma) {
    float* out;
    long long* cyc;
    hipMalloc(&out, 1024 * 512 * 4);
    hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<256, threads>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[1];
    hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * ch;  // MFMAs per wave
    const double waves_per_simd = threads / 256.0;
    printf("%-10s chains %d, %g waves/SIMD: %.1f clock64 ticks / MFMA / wave; %.3f ms -> %.1f ns per MFMA per SIMD, %.1f TFLOP/s chip\n",
           name, ch, waves_per_simd, h[0] / n, ms, ms * 1e6 / (n * waves_per_simd), 256.0 * threads / 64 * n * flop_per_mfma / (ms * 1e-3) / 1e12);
    hipFree(out);
    hipFree(cyc);
}
int main() {
    for (int th : {256, 512}) {
        run("16x16x4", k16<1>, 1, th, 2048);
        run("16x16x4", k16<2>, 2, th, 2048);
        run("16x16x4", k16<4>, 4, th, 2048);
        run("32x32x2", k32<1>, 1, th, 4096);
        run("32x32x2", k32<2>, 2, th, 4096);
    }
    return 0;
}
