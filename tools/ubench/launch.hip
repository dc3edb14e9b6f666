// What does one DEPENDENT kernel boundary cost on this box?  N trivial kernels back to back on one stream, eager and
// replayed from a captured hipGraph; trivial = 1 workgroup or 256 workgroups; with a small and a 256-byte kernarg.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { long long v[32]; };
__global__ void k_small(float* p, int n) { if (blockIdx.x == 0 && threadIdx.x == 0) p[0] += 1.f; }
__global__ void k_big(Big b, float* p) { if (blockIdx.x == 0 && threadIdx.x == 0) p[0] += (float)b.v[3]; }
__global__ void k_touch(float* p, int n) {  // every workgroup reads and writes a little (dirty lines at the boundary)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* d;
    hipMalloc(&d, 1 << 24);
    hipMemset(d, 0, 1 << 24);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 400;
    Big b = {};
    auto run = [&](int variant) {
        for (int i = 0; i < N; ++i) {
            if (variant == 0) k_small<<<1, 64, 0, s>>>(d, 1);
            else if (variant == 1) k_small<<<256, 256, 0, s>>>(d, 1);
            else if (variant == 2) k_big<<<256, 256, 0, s>>>(b, d);
            else k_touch<<<1024, 256, 0, s>>>(d, 1024 * 256);
        }
    };
    const char* names[] = {"1 WG x 64 thr", "256 WG x 256", "256 WG, 256-B kernarg", "1024 WG touching 1 MiB"};
    for (int v = 0; v < 4; ++v) {
        run(v); hipStreamSynchronize(s);
        double t0 = now(); run(v); hipStreamSynchronize(s); double t1 = now();
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        run(v);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        double t2 = now();
        for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        double t3 = now();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s eager %.2f us/kernel   graph %.2f us/kernel (host clock over 5 replays)   %.2f us/kernel (events, 1 replay)\n",
               names[v], (t1 - t0) / N * 1e6, (t3 - t2) / (5 * N) * 1e6, ms * 1e3 / N);
    }
    return 0;
}
