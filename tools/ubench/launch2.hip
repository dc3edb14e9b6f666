// Boundary cost when consecutive dispatches are DIFFERENT kernels with different resource footprints.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void ka(float* p) { if (blockIdx.x == 0 && threadIdx.x == 0) p[0] += 1.f; }
__global__ void kb(float* p) { __shared__ float s[8192]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) p[1] += s[5]; }
__global__ __launch_bounds__(256) void kc(float* p) {  // many VGPRs
    float a[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) a[i] = p[i + threadIdx.x];
    float s = 0;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += a[i] * a[95 - i];
    if (s == 12345.f) p[2] = s;
}
__global__ void kd(float* p, long long a0, long long a1, long long a2, long long a3, long long a4, long long a5, long long a6, long long a7) {
    if (blockIdx.x == 0 && threadIdx.x == 0) p[3] += (float)(a0 + a7);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* d;
    (void)hipMalloc(&d, 1 << 24);
    (void)hipMemset(d, 0, 1 << 24);
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 400;
    for (int variant = 0; variant < 3; ++variant) {
        auto run = [&]() {
            for (int i = 0; i < N; ++i) {
                const int w = variant == 0 ? 0 : (variant == 1 ? i % 2 : i % 4);
                if (w == 0) ka<<<64, 256, 0, s>>>(d);
                else if (w == 1) kb<<<64, 256, 0, s>>>(d);
                else if (w == 2) kc<<<64, 256, 0, s>>>(d);
                else kd<<<64, 256, 0, s>>>(d, 1, 2, 3, 4, 5, 6, 7, 8);
            }
        };
        run(); (void)hipStreamSynchronize(s);
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        run();
        (void)hipStreamEndCapture(s, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
        double t2 = now();
        for (int r = 0; r < 5; ++r) (void)hipGraphLaunch(ge, s);
        (void)hipStreamSynchronize(s);
        double t3 = now();
        printf("%s: graph %.2f us/kernel\n", variant == 0 ? "same kernel" : variant == 1 ? "2 kernels alternating (LDS 0 / 32 KB)" : "4 kernels cycling (LDS, VGPR, kernarg differ)",
               (t3 - t2) / (5 * N) * 1e6);
    }
    return 0;
}
