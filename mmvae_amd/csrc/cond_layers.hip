// Conditional layers of the CLVAE for gfx950: every cell goes through the Linear of its OWN condition.
//
// Replaces (reference): ConditionalLayer.forward (modules/base/components.py:365-413) -- a Python loop over the
// conditions present in the batch, each doing index_select -> nn.Linear -> index_copy_ -- and its autograd, for the
// configuration the reference's model YAMLs use (conditional_config: one Linear(Z, Z), LayerNorm without affine handled
// by mmvae_layernorm_fwd/bwd, no activation / BatchNorm / dropout).
//
// The condition blocks are thousands of small parameter tensors (4 644 donor blocks of 128 x 128 in the reference's
// human config) that live in ONE optimiser arena; the kernels address them through per-condition element offsets into
// that arena, so "gather the rows of a condition, run a GEMM, scatter back" needs neither gathers nor per-condition
// launches:
//   forward   one workgroup per cell: y[b] = W[c_b] x[b] + bias[c_b]   (a 128 x 128 block is 64 KB: cells of one
//             condition hit it in L2; a wave reduces one output row per 512-B coalesced read of W)
//   backward  dx[b] = W[c_b]^T dy[b] (lanes along the input axis: no reduction);
//             dW[c] = sum over the cells of c of dy[b] (x) x[b], db[c] = sum dy[b]: one workgroup per PRESENT condition
//             walks its cells in batch order (host-sorted segments -> bitwise reproducible, no atomics) and writes
//             the block's gradient once.  Absent conditions are not touched: their parameters keep "no gradient",
//             which the optimiser honours (torch.optim.Adam skips them).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void cond_linear_fwd_kernel(int n_in, int n_out, const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ params,
                                                              const int64_t* __restrict__ w_off,
                                                              const int64_t* __restrict__ b_off,
                                                              const int32_t* __restrict__ cond, float* __restrict__ y,
                                                              int64_t ldy) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // the cell's input row
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = cond[b];
    const float* W = params + w_off[c];
    const float* bias = params + b_off[c];
    for (int k = tid; k < n_in; k += 256) xs[k] = x[(int64_t)b * ldx + k];
    __syncthreads();
    for (int o = w; o < n_out; o += 4) {  // one wave per output row of W: coalesced reads, fixed-order wave reduction
        const float* wr = W + (int64_t)o * n_in;
        float s = 0.f;
        for (int k = lane; k < n_in; k += 64) s += wr[k] * xs[k];
        s = wave_sum(s);
        if (lane == 0) y[(int64_t)b * ldy + o] = s + bias[o];
    }
}

__global__ __launch_bounds__(256) void cond_linear_bwd_dx_kernel(int n_in, int n_out, const float* __restrict__ dy,
                                                                 int64_t lddy, const float* __restrict__ params,
                                                                 const int64_t* __restrict__ w_off,
                                                                 const int32_t* __restrict__ cond, float* __restrict__ dx,
                                                                 int64_t lddx) {
    extern __shared__ __attribute__((aligned(16))) float dys[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* W = params + w_off[cond[b]];
    for (int o = tid; o < n_out; o += 256) dys[o] = dy[(int64_t)b * lddy + o];
    __syncthreads();
    for (int k = tid; k < n_in; k += 256) {  // lanes along the input axis: row o of W is read coalesced, no reduction
        float s = 0.f;
        for (int o = 0; o < n_out; ++o) s += W[(int64_t)o * n_in + k] * dys[o];
        dx[(int64_t)b * lddx + k] = s;
    }
}

// One workgroup per present condition (group g): rows[start[g] .. start[g+1]) are its cells in batch order.
__global__ __launch_bounds__(256) void cond_linear_bwd_dw_kernel(int n_in, int n_out, const int32_t* __restrict__ group_cond,
                                                                 const int32_t* __restrict__ group_start,
                                                                 const int32_t* __restrict__ rows,
                                                                 const float* __restrict__ dy, int64_t lddy,
                                                                 const float* __restrict__ x, int64_t ldx,
                                                                 float* __restrict__ grads,
                                                                 const int64_t* __restrict__ w_off,
                                                                 const int64_t* __restrict__ b_off) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // dy row [n_out] | x row [n_in]
    float* dys = sh;
    float* xs = sh + n_out;
    const int g = blockIdx.x, tid = threadIdx.x;
    const int c = group_cond[g];
    float* dW = grads + w_off[c];
    float* db = grads + b_off[c];
    const int n = n_in * n_out;
    const int beg = group_start[g], end = group_start[g + 1];
    // each thread owns the entries e = tid, tid + 256, ... of dW (and of db): accumulate over the group's cells
    for (int e0 = 0; e0 < n; e0 += 256 * 16) {
        float acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = 0.f;
        float accb = 0.f;
        for (int r = beg; r < end; ++r) {
            const int b = rows[r];
            __syncthreads();
            for (int o = tid; o < n_out; o += 256) dys[o] = dy[(int64_t)b * lddy + o];
            for (int k = tid; k < n_in; k += 256) xs[k] = x[(int64_t)b * ldx + k];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = e0 + tid + 256 * u;
                if (e < n) acc[u] += dys[e / n_in] * xs[e % n_in];
            }
            if (e0 == 0 && tid < n_out) accb += dys[tid];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = e0 + tid + 256 * u;
            if (e < n) dW[e] = acc[u];
        }
        if (e0 == 0) {
            for (int o = tid; o < n_out; o += 256) {  // n_out > 256: the rest of the bias gradient, row by row
                if (o < 256) {
                    db[o] = accb;
                } else {
                    float s = 0.f;
                    for (int r = beg; r < end; ++r) s += dy[(int64_t)rows[r] * lddy + o];
                    db[o] = s;
                }
            }
        }
    }
}

}  // namespace

extern "C" int mmvae_cond_linear_fwd(int B, int n_in, int n_out, const float* x, int64_t ldx, const float* params,
                                     const int64_t* w_off, const int64_t* b_off, const int32_t* cond, float* y,
                                     int64_t ldy, mmvae_stream_t stream) {
    if (B <= 0 || n_in <= 0 || n_out <= 0 || n_in > 8192 || !x || !params || !w_off || !b_off || !cond || !y ||
        ldx < n_in || ldy < n_out)
        return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(cond_linear_fwd_kernel, dim3(B), dim3(256), n_in * sizeof(float), (hipStream_t)stream, n_in, n_out, x,
                 ldx, params, w_off, b_off, cond, y, ldy);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_cond_linear_bwd_dx(int B, int n_in, int n_out, const float* dy, int64_t lddy, const float* params,
                                        const int64_t* w_off, const int32_t* cond, float* dx, int64_t lddx,
                                        mmvae_stream_t stream) {
    if (B <= 0 || n_in <= 0 || n_out <= 0 || n_out > 8192 || !dy || !params || !w_off || !cond || !dx ||
        lddy < n_out || lddx < n_in)
        return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(cond_linear_bwd_dx_kernel, dim3(B), dim3(256), n_out * sizeof(float), (hipStream_t)stream, n_in, n_out,
                 dy, lddy, params, w_off, cond, dx, lddx);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_cond_linear_bwd_dw(int n_groups, const int32_t* group_cond, const int32_t* group_start,
                                        const int32_t* rows, int n_in, int n_out, const float* dy, int64_t lddy,
                                        const float* x, int64_t ldx, float* grads, const int64_t* w_off,
                                        const int64_t* b_off, mmvae_stream_t stream) {
    if (n_groups <= 0 || n_in <= 0 || n_out <= 0 || n_in + n_out > 12288 || !group_cond || !group_start || !rows || !dy ||
        !x || !grads || !w_off || !b_off || lddy < n_out || ldx < n_in)
        return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(cond_linear_bwd_dw_kernel, dim3(n_groups), dim3(256), (n_in + n_out) * sizeof(float), (hipStream_t)stream,
                 n_in, n_out, group_cond, group_start, rows, dy, lddy, x, ldx, grads, w_off, b_off);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}
