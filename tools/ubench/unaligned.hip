// Are 16-byte global loads / stores at 4-byte-aligned (not 16-byte-aligned) addresses legal on gfx950, and what do they
// cost?  Reads rows of a [rows, ld] fp32 matrix with float4 accesses starting at element offset `shift` of each row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void copy4(const float* __restrict__ in, float* __restrict__ out, int rows, int ld, int n4, int shift) {
    const int r = blockIdx.x;
    const float* src = in + (size_t)r * ld + shift;
    float* dst = out + (size_t)r * ld + shift;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
        v += 1.0f;
        *reinterpret_cast<f32x4*>(dst + 4 * i) = v;
    }
}
int main() {
    const int rows = 4096, ld = 20003, n4 = 5000;
    size_t n = (size_t)rows * ld;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 1000);
    float *in, *out;
    hipMalloc(&in, n * 4);
    hipMalloc(&out, n * 4);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(out, 0, n * 4);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        copy4<<<rows, 256>>>(in, out, rows, ld, n4, shift);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int it = 0; it < 10; ++it) copy4<<<rows, 256>>>(in, out, rows, ld, n4, shift);
        hipEventRecord(b);
        hipError_t e = hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, a, b);
        std::vector<float> o(n);
        hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int r = 0; r < rows; r += 97)
            for (int c = 0; c < 4 * n4; ++c) {
                size_t i = (size_t)r * ld + shift + c;
                if (o[i] != h[i] + 1.0f) ++bad;
            }
        printf("shift %d (row start %% 4 varies: ld = %d): %s, mismatches %zu, %.1f GB/s\n", shift, ld, hipGetErrorString(e),
               bad, 10.0 * rows * n4 * 16 * 2 / ms / 1e6);
    }
    return 0;
}
