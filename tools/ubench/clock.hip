// What shader clock does gfx950 sustain under MFMA load?  s_memtime (shader cycles) vs s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void k(int mode, int iters, long long* out) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)1.0f; }
    float x = lane;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 64; ++j) x = x * 1.0001f + 0.5f;
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = x;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[8] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

int main() {
    long long *d, h[2];
    hipMalloc(&d, 128);
    for (int mode = 0; mode < 2; ++mode)
        for (int blocks : {256, 512, 1024}) {
            k<<<blocks, 256>>>(mode, 20000, d);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("%s blocks %4d: %lld shader cycles in %lld ticks of 100 MHz -> %.0f MHz; %.1f cycles per %s\n",
                   mode == 0 ? "mfma" : "valu", blocks, h[0], h[1], h[0] / (h[1] / 100.0),
                   (double)h[0] / 20000 / (mode == 0 ? 16 : 64), mode == 0 ? "MFMA" : "FMA");
        }
    return 0;
}
