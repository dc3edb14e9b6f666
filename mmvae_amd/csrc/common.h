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

// Thread-0 tail of the clip + Adam preparation (mmvae_adam_prepare; also the last workgroup of mmvae_adv_dw_f32): the
// global gradient norm from the fp64 sum of squares, the clip coefficient, the step count and the bias corrections.
__device__ __forceinline__ void adam_state_finish(float* __restrict__ state, double sumsq, unsigned flags, float max_norm,
                                                  float grad_scale, float beta1, float beta2) {
    float norm = state[1];
    if (flags & MMVAE_PREPARE_NORM) norm = (float)(sqrt(sumsq) * (double)fabsf(grad_scale));
    float step = state[0];
    if (flags & MMVAE_PREPARE_ADVANCE) step += 1.f;
    state[0] = step;
    state[1] = norm;
    float clip = 1.f;
    if (max_norm > 0.f) {
        clip = max_norm / (norm + 1e-6f);
        if (clip > 1.f) clip = 1.f;
    }
    state[2] = clip;
    state[3] = 1.f - powf(beta1, step);
    state[4] = 1.f - powf(beta2, step);
}

// Exact three-way bf16 split of an fp32 value by truncation: a = p0 + p1 + p2, each piece an fp32 bit pattern whose low
// 16 bits are zero (p0 = top 16 bits of a, p1 = top 16 bits of a - p0, p2 = a - p0 - p1: 3 x 8 = 24 significant bits).
// The bf16x3 GEMMs (gemm_dev.h) multiply such pieces; producers that write pre-split planes use the same function.
__device__ __forceinline__ void x3_split(float a, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned u = __float_as_uint(a);
    p0 = u & 0xFFFF0000u;
    const float r1 = a - __uint_as_float(p0);
    p1 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(p1);
    p2 = __float_as_uint(r2);  // <= 8 significant bits left: already a bf16 value
}

// Column-per-lane kernels (lane = column c, c even on even lanes): the three bf16 pieces of v, packed with the right
// neighbour's into one dword per plane and stored by the even lane at planes[p][r][c .. c + 1].  Both lanes of a pair
// must be active (N even).
__device__ __forceinline__ void store_planes_lanepair(unsigned short* __restrict__ P, int64_t ldp, int64_t pstride, int r,
                                                      int c, float v, int lane) {
    unsigned q[3];
    x3_split(v, q[0], q[1], q[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const unsigned mine = q[p] >> 16;
        const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
        if (!(lane & 1)) *reinterpret_cast<unsigned*>(P + p * pstride + (int64_t)r * ldp + c) = mine | (other << 16);
    }
}
