// Micro-benchmark / exactness probe: the exact three-way bf16 split of fp32 pairs
//   (A) as the GEMM stagers do it today: AND, SUB, AND, SUB per element + 3 v_perm per pair (11 VALU per pair)
//   (B) with v_dot2c_f32_bf16: pack the high halves first (v_perm), then residual = x - piece as a dot product of the
//       PACKED pair with (-1, 0) / (0, -1) accumulated onto x: 7 VALU per pair, the same truncated pieces.
// Checks that (B) writes bit-identical planes for ordinary, huge, tiny and denormal inputs, and times both (register-only
// loops on all CUs).      hipcc --offload-arch=gfx950 -O3 -o split_dot2 split_dot2.hip && ./split_dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_a(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
    const float x0 = __uint_as_float(ux & 0xFFFF0000u), y0 = __uint_as_float(uy & 0xFFFF0000u);
    const float rx = x - x0, ry = y - y0;
    const unsigned urx = __float_as_uint(rx), ury = __float_as_uint(ry);
    const float x1 = __uint_as_float(urx & 0xFFFF0000u), y1 = __uint_as_float(ury & 0xFFFF0000u);
    const float sx = rx - x1, sy = ry - y1;
    p0 = __builtin_amdgcn_perm(uy, ux, 0x07060302u);
    p1 = __builtin_amdgcn_perm(ury, urx, 0x07060302u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
}
__device__ __forceinline__ void split_b(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
    const bf16x2 mlo = __builtin_bit_cast(bf16x2, 0x0000BF80u), mhi = __builtin_bit_cast(bf16x2, 0xBF800000u);
    p0 = __builtin_amdgcn_perm(__float_as_uint(y), __float_as_uint(x), 0x07060302u);
    const float rx = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p0), mlo, x, false);
    const float ry = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p0), mhi, y, false);
    p1 = __builtin_amdgcn_perm(__float_as_uint(ry), __float_as_uint(rx), 0x07060302u);
    const float sx = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p1), mlo, rx, false);
    const float sy = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p1), mhi, ry, false);
    p2 = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
}

template <int WHICH>
__global__ void check(const float* in, unsigned* out, int npairs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    unsigned p0, p1, p2;
    if (WHICH == 0) split_a(in[2 * i], in[2 * i + 1], p0, p1, p2);
    else split_b(in[2 * i], in[2 * i + 1], p0, p1, p2);
    out[3 * i] = p0; out[3 * i + 1] = p1; out[3 * i + 2] = p2;
}

template <int WHICH>
__global__ __launch_bounds__(256) void rate(int iters, const float* in, unsigned* out) {
    float x[8], y[8];
    for (int j = 0; j < 8; ++j) { x[j] = in[threadIdx.x + 64 * j]; y[j] = in[threadIdx.x + 64 * j + 7]; }
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned p0, p1, p2;
            if (WHICH == 0) split_a(x[j], y[j], p0, p1, p2);
            else split_b(x[j], y[j], p0, p1, p2);
            acc ^= p0 + p1 + p2;
            x[j] = __uint_as_float(__float_as_uint(x[j]) + (p2 & 1u) + 2u);  // keep the loop honest
            y[j] = __uint_as_float(__float_as_uint(y[j]) + 3u);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int npairs = 1 << 20;
    std::vector<float> h(2 * npairs);
    srand(1);
    for (int i = 0; i < 2 * npairs; ++i) {
        const int cls = i % 8;
        const float u = (float)rand() / RAND_MAX * 2.f - 1.f;
        int e = 0;
        if (cls < 4) e = rand() % 40 - 20;            // ordinary
        else if (cls == 4) e = 100 + rand() % 27;     // huge
        else if (cls == 5) e = -(90 + rand() % 36);   // tiny normal: residuals may be denormal
        else if (cls == 6) e = -(126 + rand() % 23);  // denormal inputs
        else e = rand() % 3 - 1;
        h[i] = ldexpf(u, e);
        if (cls == 7 && (rand() % 16) == 0) h[i] = (rand() & 1) ? 0.f : -0.f;
    }
    float* d_in; unsigned *d_a, *d_b;
    hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_a, 3 * npairs * 4); hipMalloc(&d_b, 3 * npairs * 4);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check<0>, dim3(npairs / 256), dim3(256), 0, 0, d_in, d_a, npairs);
    hipLaunchKernelGGL(check<1>, dim3(npairs / 256), dim3(256), 0, 0, d_in, d_b, npairs);
    std::vector<unsigned> a(3 * npairs), b(3 * npairs);
    hipMemcpy(a.data(), d_a, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d_b, b.size() * 4, hipMemcpyDeviceToHost);
    long diff[8] = {0}, tot[8] = {0};
    int shown = 0;
    for (int i = 0; i < npairs; ++i) {
        for (int half = 0; half < 2; ++half) {
            const int cls = (2 * i + half) % 8;
            tot[cls]++;
            bool bad = false;
            for (int p = 0; p < 3; ++p) {
                const unsigned va = (a[3 * i + p] >> (16 * half)) & 0xFFFFu, vb = (b[3 * i + p] >> (16 * half)) & 0xFFFFu;
                if (va != vb) bad = true;
            }
            if (bad) {
                diff[cls]++;
                if (shown < 6) {
                    ++shown;
                    printf("  differs: x = %a  A = %04x %04x %04x  B = %04x %04x %04x\n", h[2 * i + half],
                           (a[3 * i] >> (16 * half)) & 0xFFFF, (a[3 * i + 1] >> (16 * half)) & 0xFFFF, (a[3 * i + 2] >> (16 * half)) & 0xFFFF,
                           (b[3 * i] >> (16 * half)) & 0xFFFF, (b[3 * i + 1] >> (16 * half)) & 0xFFFF, (b[3 * i + 2] >> (16 * half)) & 0xFFFF);
                }
            }
        }
    }
    const char* names[8] = {"ordinary", "ordinary", "ordinary", "ordinary", "huge 2^100..2^126", "tiny 2^-90..2^-125", "denormal inputs", "near 1 and zeros"};
    for (int c = 4; c < 8; ++c) printf("class %-20s: %ld of %ld elements differ\n", names[c], diff[c], tot[c]);
    printf("class %-20s: %ld of %ld elements differ\n", "ordinary", diff[0] + diff[1] + diff[2] + diff[3], tot[0] + tot[1] + tot[2] + tot[3]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(rate<0>, dim3(256 * 8), dim3(256), 0, 0, iters, d_in, d_a);
            else hipLaunchKernelGGL(rate<1>, dim3(256 * 8), dim3(256), 0, 0, iters, d_in, d_a);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double pairs = (double)256 * 8 * 256 * iters * 8;
        printf("%s: %.3f ms, %.2f G pairs/s\n", which == 0 ? "A and/sub/perm (11 VALU per pair + loop overhead)" : "B perm/dot2c    ( 7 VALU per pair + loop overhead)", ms, pairs / ms * 1e-6);
    }
    return 0;
}
