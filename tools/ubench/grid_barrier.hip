// Price of a grid-wide barrier inside one persistent kernel on gfx950, against the ~4.7 us floor of a kernel boundary in
// a captured graph.  Every workgroup writes a line, all meet at the barrier, every workgroup reads the line of a
// workgroup on another XCD (checks visibility).  Bounded spin: a barrier that does not complete sets an error flag and
// the kernel still drains.   build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip ; ./grid_barrier [wgs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* err) {
    __syncthreads();
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        int ok = 1;
        __threadfence();  // release: this workgroup's stores are visible device-wide before its arrival is
        atomicAdd(counter, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {  // seconds: give up, flag, drain
                atomicExch(err, 1u);
                ok = 0;
                break;
            }
        }
        __threadfence();  // acquire
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

__global__ __launch_bounds__(256) void barrier_loop(int rounds, unsigned* counter, unsigned* err, float* data, float* out) {
    const int wg = blockIdx.x, n = gridDim.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x < 64) data[(size_t)wg * 64 + threadIdx.x] = (float)(r + wg);
        if (!grid_barrier(counter, (unsigned)(r + 1) * n, err)) return;
        const int nb = (wg + n / 2 + 1) % n;  // a workgroup on another XCD (round-robin placement)
        if (threadIdx.x < 64) {
            const float v = data[(size_t)nb * 64 + threadIdx.x];
            if (v != (float)(r + nb)) atomicExch(err, 2u);
            acc += v;
        }
        // the neighbour's next write must not pass this read
        if (!grid_barrier(counter + 32, (unsigned)(r + 1) * n, err)) return;
    }
    if (threadIdx.x < 64) out[(size_t)wg * 64 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const int rounds = 200;
    unsigned* counter;
    unsigned* err;
    float *data, *out;
    hipMalloc(&counter, 256);
    hipMalloc(&err, 4);
    hipMalloc(&data, (size_t)wgs * 256);
    hipMalloc(&out, (size_t)wgs * 256);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(counter, 0, 256);
        hipMemset(err, 0, 4);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(barrier_loop, dim3(wgs), dim3(256), 0, 0, rounds, counter, err, data, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        unsigned e;
        hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        printf("wgs %d: %d rounds x 2 barriers: %.1f us total, %.2f us per barrier, err %u\n", wgs, rounds, ms * 1e3,
               ms * 1e3 / (2 * rounds), e);
    }
    return 0;
}
