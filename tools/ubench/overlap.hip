// Micro-benchmark: do MFMA work in some waves and VALU / LDS-write / LDS-read work in other waves of the same
// workgroup overlap on gfx950?  Prints time for each alone and together.  (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// mode bits: 1 = consumer waves (0-3) issue MFMAs, 2 = producer waves do VALU, 4 = producers do ds_write,
// 8 = consumers do ds_read before MFMAs, 16 = barrier per iteration
template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k(int mode, int iters, float* out) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool consumer = wave < 4;
    if (consumer) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)1.0f; }
        for (int it = 0; it < iters; ++it) {
            if (mode & 8) {
                f32x4 v[12];
                for (int j = 0; j < 12; ++j)
                    v[j] = *reinterpret_cast<const f32x4*>(lds + ((wave * 12 + j) * 1024 + lane * 16) % 49152);
                for (int j = 0; j < 12; ++j) asm volatile("" ::"v"(v[j]));
            }
            if (mode & 1) {
                if (mode & 32) {  // four dependent chains of 6, back to back
#pragma unroll
                    for (int j = 0; j < 24; ++j)
                        acc[j / 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j / 6], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < 24; ++j)
                        acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
                }
            }
            if (mode & 16) __syncthreads();
        }
        float s = 0;
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) s += acc[i][e];
        if (s == 123.456f) out[tid] = s;
    } else {
        float x[8];
        for (int e = 0; e < 8; ++e) x[e] = lane * 0.37f + e;
        const int ptid = tid - 256;
        for (int it = 0; it < iters; ++it) {
            unsigned p[6] = {0, 0, 0, 0, 0, 0};
            if (mode & 2) {
#pragma unroll
                for (int r = 0; r < (NWAVES == 12 ? 1 : 2); ++r)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {  // ~10 VALU per element, like the split
                        unsigned u = __float_as_uint(x[e]);
                        unsigned h0 = u & 0xffff0000u;
                        float r1 = x[e] - __uint_as_float(h0);
                        unsigned h1 = __float_as_uint(r1) & 0xffff0000u;
                        float r2 = r1 - __uint_as_float(h1);
                        unsigned h2 = __float_as_uint(r2) & 0xffff0000u;
                        p[(e >> 1) % 2] ^= (e & 1) ? h0 : (h0 >> 16);
                        p[2 + (e >> 1) % 2] ^= (e & 1) ? h1 : (h1 >> 16);
                        p[4 + (e >> 1) % 2] ^= (e & 1) ? h2 : (h2 >> 16);
                        x[e] = x[e] * 1.0001f + 0.5f;
                    }
            }
            if (mode & 4) {
#pragma unroll
                for (int r = 0; r < (NWAVES == 12 ? 3 : 6); ++r)
                    *reinterpret_cast<uint2*>(lds + ((ptid * 8 + r * 8192 + it * 64) % 49152)) =
                        make_uint2(p[(2 * r) % 6], p[(2 * r + 1) % 6]);
            } else {
                asm volatile("" ::"v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]));
            }
            if (mode & 16) __syncthreads();
        }
        float s = 0;
        for (int e = 0; e < 8; ++e) s += x[e];
        if (s == 123.456f) out[tid] = s;
    }
}

template <int NW>
float run(int mode, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NW><<<256, NW * 64>>>(mode, iters, out);
    hipEventRecord(e0);
    k<NW><<<256, NW * 64>>>(mode, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 2000;
    const char* names[] = {"mfma", "valu", "mfma+valu", "ldsw", "mfma+ldsw", "valu+ldsw", "mfma+valu+ldsw",
                           "ldsr", "mfma+ldsr", "all", "all+barrier", "mfma dep-chains", "dep+ldsr+ldsw"};
    int modes[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 31, 33, 45};
    for (int i = 0; i < 13; ++i) {
        float t8 = run<8>(modes[i], iters, out), t12 = run<12>(modes[i], iters, out);
        printf("%-18s 8 waves: %8.1f us (%6.0f cyc/iter @2.2GHz)   12 waves: %8.1f us (%6.0f)\n", names[i], t8,
               t8 * 2200 / iters, t12, t12 * 2200 / iters);
    }
    return 0;
}
