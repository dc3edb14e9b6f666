// Micro-benchmark: what does a kernel boundary cost BEHIND a kernel that has just written tens of megabytes?  Kernel A
// (256 workgroups, one per CU) writes `mb` MB with ordinary / non-temporal stores and stamps its exit (100 MHz wall clock)
// per workgroup; kernel B (dependent, same stream) stamps its entry.  Prints B.entry - max(A.exit) and A's own span.
// (diagnostic, not part of the library)    hipcc --offload-arch=gfx950 -O3 -o drain drain.hip && ./drain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(512) void writer(f32x4* out, long n4_per_wg, long long* stamps) {
    // (the GEMM kernel's footprint: 160 KB of LDS per workgroup, touched)
    __shared__ __attribute__((aligned(16))) char lds[160 * 1024];
    for (int i = threadIdx.x; i < 160 * 1024 / 16; i += 512) reinterpret_cast<f32x4*>(lds)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    if (reinterpret_cast<float*>(lds)[threadIdx.x * 7 % 4096] == 123.f) out[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long base = (long)blockIdx.x * n4_per_wg;
    if (threadIdx.x == 0) stamps[blockIdx.x * 2] = wall_clock64();
    for (long i = threadIdx.x; i < n4_per_wg; i += 512) {
        f32x4 v = {(float)i, 1.f, 2.f, 3.f};
        if (NT)
            __builtin_nontemporal_store(v, out + base + i);
        else
            out[base + i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        stamps[blockIdx.x * 2 + 1] = wall_clock64();
    }
}

__global__ void reader(const f32x4* in, long n4, long long* stamp, float* sink) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *stamp = wall_clock64();
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) s += in[i][0];
    if (s == 123.f) *sink = s;
}

int main() {
    const int WG = 256;
    f32x4* buf;
    long long *st, *sb;
    float* sink;
    hipMalloc(&buf, 256L << 20);
    hipMalloc(&st, WG * 2 * sizeof(long long));
    hipMalloc(&sb, sizeof(long long));
    hipMalloc(&sink, 4);
    for (int graph = 0; graph < 2; ++graph)
    for (int nt = 0; nt < 2; ++nt)
        for (int mb : {0, 2, 8, 32, 64, 128}) {
            const long n4 = ((long)mb << 20) / 16 / WG;
            std::vector<double> gaps, spans;
            hipGraphExec_t exec = nullptr;
            if (graph) {  // the same pair as a captured graph
                hipStream_t cs;
                hipStreamCreate(&cs);
                hipGraph_t gr;
                hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
                if (nt)
                    hipLaunchKernelGGL(writer<1>, dim3(WG), dim3(512), 0, cs, buf, n4, st);
                else
                    hipLaunchKernelGGL(writer<0>, dim3(WG), dim3(512), 0, cs, buf, n4, st);
                hipLaunchKernelGGL(reader, dim3(64), dim3(256), 0, cs, buf, (long)(1 << 16), sb, sink);
                hipStreamEndCapture(cs, &gr);
                hipGraphInstantiate(&exec, gr, nullptr, nullptr, 0);
            }
            for (int rep = 0; rep < 12; ++rep) {
                if (graph) {
                    hipGraphLaunch(exec, 0);
                } else {
                    if (nt)
                        hipLaunchKernelGGL(writer<1>, dim3(WG), dim3(512), 0, 0, buf, n4, st);
                    else
                        hipLaunchKernelGGL(writer<0>, dim3(WG), dim3(512), 0, 0, buf, n4, st);
                    hipLaunchKernelGGL(reader, dim3(64), dim3(256), 0, 0, buf, (long)(1 << 16), sb, sink);
                }
                hipDeviceSynchronize();
                std::vector<long long> h(WG * 2);
                long long b;
                hipMemcpy(h.data(), st, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
                hipMemcpy(&b, sb, sizeof(long long), hipMemcpyDeviceToHost);
                long long first = h[0], last = h[1];
                for (int i = 0; i < WG; ++i) {
                    first = std::min(first, h[2 * i]);
                    last = std::max(last, h[2 * i + 1]);
                }
                if (rep >= 2) {
                    gaps.push_back((b - last) / 100.0);
                    spans.push_back((last - first) / 100.0);
                }
            }
            std::sort(gaps.begin(), gaps.end());
            std::sort(spans.begin(), spans.end());
            printf("%s %s stores, %3d MB written: writer span %7.1f us   boundary (last exit -> next kernel's entry) %6.2f us (min %5.2f)\n",
                   graph ? "graph " : "stream", nt ? "non-temporal" : "ordinary    ", mb, spans[spans.size() / 2], gaps[gaps.size() / 2], gaps[0]);
        }
    return 0;
}
