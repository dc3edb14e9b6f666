// Shared device/host helpers for libmmvae_hip.so (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmvae_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Launch with the runtime's sticky last-error word cleared first: MMVAE_LAUNCH_CHECK must report THIS launch, not an
// error some earlier, unrelated HIP call of the process left behind (e.g. a probe made by the host framework while it
// initialises the device after this library was already loaded).
#define MMVAE_LAUNCH(...)                  \
    do {                                   \
        (void)hipGetLastError();           \
        hipLaunchKernelGGL(__VA_ARGS__);   \
    } while (0)

#define MMVAE_LAUNCH_CHECK()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return MMVAE_ERR_LAUNCH;        \
    } while (0)

static inline int ceil_div_i(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div_l(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Sum across the 64 lanes of a wavefront; every lane gets the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
// Sum across each 32-lane half of a wavefront (lanes 0-31 and 32-63 reduce independently).
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
