// Micro-benchmark: cost of ds_read_b64_tr_b16 (and ds_read_b128) under the lane -> address patterns the GEMM kernel's
// fragment reads produce, to find which patterns the LDS serves without bank conflicts.  One workgroup per CU, W waves;
// every wave issues the same read N times (8 independent reads in flight) and reports cycles per read instruction.
// Patterns are lane -> byte offset tables built on the host.  (diagnostic, not part of the library)
//   hipcc --offload-arch=gfx950 -O3 -o tr_banks tr_banks.hip && ./tr_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>  // 0: ds_read_b64_tr_b16, 1: ds_read_b128, 2: ds_read_b64
__global__ __launch_bounds__(512) void k(int iters, const int* __restrict__ offs, long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    const int off = offs[lane];
    float acc = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const char* a = lds + off + (u & 1) * 16384;  // (two copies of the pattern 16 KiB apart: the same banks)
            if (KIND == 0) {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a));
                acc += (float)v[0] + (float)v[3];
            } else if (KIND == 1) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(a);
                acc += v[0] + v[3];
            } else {
                const float2 v = *reinterpret_cast<const float2*>(a);
                acc += v.x + v.y;
            }
        }
    }
    const long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
    if (acc == 123.456f) sink[tid] = acc;
}

struct Pattern {
    std::string name;
    int kind;
    std::vector<int> off;
};

int main() {
    std::vector<Pattern> ps;
    auto add = [&](const char* name, int kind, auto f) {
        Pattern p{name, kind, std::vector<int>(64)};
        for (int l = 0; l < 64; ++l) p.off[l] = f(l);
        ps.push_back(p);
    };
    // lane decomposition used below: g = l >> 4 (16-lane group), q = (l & 15) >> 2, p = l & 3
    const int R = 256;  // k-row stride 2 R = 512 bytes (160: 320 bytes)
    add("b64 linear (8 B per lane, 512 B)", 2, [](int l) { return l * 8; });
    add("b128 linear (16 B per lane, 1 KiB)", 1, [](int l) { return l * 16; });
    add("b128 KC 32x32: row l&31 (64 B rows), chunk (l>>5)^swz", 1,
        [](int l) { int r = l & 31, c = (l >> 5) ^ ((r >> 2) & 3); return r * 64 + c * 16; });
    add("b128 KC 16x16: row l&15, chunk (l>>4)^swz", 1,
        [](int l) { int r = l & 15, c = (l >> 4) ^ ((r >> 2) & 3); return r * 64 + c * 16; });
    add("tr linear (8 B per lane)", 0, [](int l) { return l * 8; });
    add("tr P32  R=256: k-row 8(l>>5)+q, slot q, half (l>>4)&1", 0, [=](int l) {
        int q = (l & 15) >> 2, p = l & 3; return (8 * (l >> 5) + q) * 2 * R + q * 64 + 32 * ((l >> 4) & 1) + 8 * p; });
    add("tr P16a R=256: k-row 8g+q, slot q, half 0", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (8 * g + q) * 2 * R + q * 64 + 8 * p; });
    add("tr P16b R=256: k-row 8g+q, slot q, half g&1", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (8 * g + q) * 2 * R + q * 64 + 32 * (g & 1) + 8 * p; });
    add("tr P16c R=256: k-row 8g+q, slot q, half g>>1", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (8 * g + q) * 2 * R + q * 64 + 32 * (g >> 1) + 8 * p; });
    add("tr P16d R=256: k-row 8g+q, slot q^g, half 0", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (8 * g + q) * 2 * R + (q ^ g) * 64 + 8 * p; });
    add("tr P16e R=256: k-row 8g+q, slot q, +128 g (all 4 groups distinct 128-B quarters? no: slot q + 2..)", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (8 * g + q) * 2 * R + ((q + g) & 3) * 64 + 8 * p; });
    add("tr P16f R=256: groups 16 B apart inside a 64-B slot (g*16 + 8*(p&1)...) rows of 4 lanes", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (8 * g + q) * 2 * R + q * 64 + 8 * p + 0 * g; });
    add("tr 4 groups same k-rows, 4 consecutive 32-B chunks (k-row q, chunk g)", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return q * 2 * R + (q ^ 0) * 128 % 512 + 32 * g + 8 * p; });
    add("tr groups (0,2) half 0 / (1,3) half 1, k-row 8g+q", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (8 * g + q) * 2 * R + q * 64 + 32 * (g & 1) + 8 * p; });
    add("tr groups (0,1) half 0 / (2,3) half 1 with slot q^(g&1)*2", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (8 * g + q) * 2 * R + (q ^ (2 * (g & 1))) * 64 + 32 * (g >> 1) + 8 * p; });
    add("tr k-row stride 528 (padded rows), 8g+q, slot 0", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (8 * g + q) * 528 + 8 * p; });
    add("tr k-row stride 544 (padded rows), 8g+q, slot 0", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (8 * g + q) * 544 + 8 * p; });
    add("tr k-row stride 520, 8g+q", 0, [=](int l) {
        int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (8 * g + q) * 520 + 8 * p; });
    add("tr P32-like with stride 544", 0, [=](int l) {
        int q = (l & 15) >> 2, p = l & 3; return (8 * (l >> 5) + q) * 544 + 32 * ((l >> 4) & 1) + 8 * p; });
    add("tr one k-row per lane quad: k-row (l>>2), 8 p (16 k-rows x 32 B, stride 512)", 0, [=](int l) { return (l >> 2) * 512 + 8 * (l & 3); });
    add("tr k-row (l>>2), stride 512, slot (l>>2)&3 and half (l>>4)&1 ... (k-row*512 + ((l>>2)&3)*64 + ((l>>4)&1)*32)", 0,
        [=](int l) { int kr = l >> 2; return kr * 512 + (kr & 3) * 64 + ((kr >> 2) & 1) * 32 + 8 * (l & 3); });
    add("tr k-row (l>>2): + (kr&3)*64 + ((kr>>2)&3)*... 16 distinct 32-B chunks mod 512", 0,
        [=](int l) { int kr = l >> 2; return kr * 512 + (kr & 15) * 32 + 8 * (l & 3); });

    int* d_off;
    long long* d_out;
    float* d_sink;
    hipMalloc(&d_off, 64 * sizeof(int));
    hipMalloc(&d_out, 256 * 8 * sizeof(long long));
    hipMalloc(&d_sink, 4096);
    const int iters = 2000;
    for (int waves : {1, 4}) {
        printf("---- %d wave(s) per workgroup, one workgroup per CU; cycles per read instruction per wave\n", waves);
        for (auto& p : ps) {
            hipMemcpy(d_off, p.off.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
            for (int rep = 0; rep < 2; ++rep) {
                if (p.kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, iters, d_off, d_out, d_sink);
                if (p.kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, iters, d_off, d_out, d_sink);
                if (p.kind == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, iters, d_off, d_out, d_sink);
                hipDeviceSynchronize();
            }
            std::vector<long long> h(256 * 8);
            hipMemcpy(h.data(), d_out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            double s = 0;
            for (int b = 0; b < 256; ++b) s += (double)h[b * 8];
            printf("%7.2f   %s\n", s / 256 / (iters * 8.0), p.name.c_str());
        }
    }
    return 0;
}
