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

// Forward over cells taken in condition-sorted order (`rows`): workgroup (i, slice) owns FW_CB consecutive sorted cells
// and FW_RO output rows (FW_RW per wave).  For the layers with few conditions (species, sex, assay) the cells share
// their block: a wave keeps the FW_RW rows of W it reduces in registers while the condition stays the same, so a
// block is read once per workgroup instead of once per cell; when every cell has its own block (donor_id) the FW_RW
// row loads of a cell are independent and in flight together.  Per output the arithmetic (lane-strided partial sums,
// then the wave reduction) is that of cond_linear_fwd_kernel: same bits.
constexpr int FW_CB = 8;
constexpr int FW_RW = 4;           // output rows per wave
constexpr int FW_RO = 4 * FW_RW;   // output rows per workgroup
template <int T>  // T = ceil(n_in / 64) <= 4
__global__ __launch_bounds__(256) void cond_linear_fwd_sorted_kernel(int B, int n_in, int n_out, const float* __restrict__ x,
                                                                     int64_t ldx, const float* __restrict__ params,
                                                                     const int64_t* __restrict__ w_off,
                                                                     const int64_t* __restrict__ b_off,
                                                                     const int32_t* __restrict__ cond,
                                                                     const int32_t* __restrict__ rows,
                                                                     float* __restrict__ y, int64_t ldy) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // FW_CB input rows
    __shared__ int cell[FW_CB], cnd[FW_CB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r0 = blockIdx.x * FW_CB, nb = min(FW_CB, B - r0);
    if (tid < nb) {
        const int b = rows[r0 + tid];
        cell[tid] = b;
        cnd[tid] = cond[b];
    }
    __syncthreads();
    for (int i = tid; i < nb * n_in; i += 256) {
        const int j = i / n_in, k = i - j * n_in;
        xs[i] = x[(int64_t)cell[j] * ldx + k];
    }
    __syncthreads();
    const int o0 = blockIdx.y * FW_RO + w * FW_RW;
    float wv[FW_RW][T], bias[FW_RW];
    int prev = -1;
    for (int j = 0; j < nb; ++j) {
        const int c = cnd[j];
        if (c != prev) {  // wave-uniform
            const float* W = params + w_off[c];
            const float* bs = params + b_off[c];
#pragma unroll
            for (int r = 0; r < FW_RW; ++r) {
                const int o = o0 + r;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int k = lane + 64 * t;
                    wv[r][t] = (o < n_out && k < n_in) ? W[(int64_t)o * n_in + k] : 0.f;
                }
                bias[r] = o < n_out ? bs[o] : 0.f;
            }
            prev = c;
        }
        const float* xr = xs + j * n_in;
        float s[FW_RW];
#pragma unroll
        for (int r = 0; r < FW_RW; ++r) {
            s[r] = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int k = lane + 64 * t;
                if (k < n_in) s[r] += wv[r][t] * xr[k];
            }
        }
#pragma unroll
        for (int r = 0; r < FW_RW; ++r) s[r] = wave_sum(s[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < FW_RW; ++r)
                if (o0 + r < n_out) y[(int64_t)cell[j] * ldy + o0 + r] = s[r] + bias[r];
        }
    }
}

__global__ __launch_bounds__(256) void cond_linear_bwd_dx_kernel(int n_in, int n_out, const float* __restrict__ dy,
                                                                 int64_t lddy, const float* __restrict__ params,
                                                                 const int64_t* __restrict__ w_off,
                                                                 const int32_t* __restrict__ cond, float* __restrict__ dx,
                                                                 int64_t lddx, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float dys[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* W = params + w_off[cond[b]];
    for (int o = tid; o < n_out; o += 256) dys[o] = dy[(int64_t)b * lddy + o];
    __syncthreads();
    for (int k = tid; k < n_in; k += 256) {  // lanes along the input axis: row o of W is read coalesced, no reduction
        float s = 0.f;
        for (int o = 0; o < n_out; ++o) s += W[(int64_t)o * n_in + k] * dys[o];
        float* out = dx + (int64_t)b * lddx + k;
        *out = accumulate ? *out + s : s;
    }
}

// Workgroup (g, slice): group g = one PRESENT condition, rows[start[g] .. start[g+1]) are its cells in batch order; the
// slice is DW_U * 256 consecutive entries of the block's [n_out, n_in] gradient (slice 0 also owns the bias gradient).
// A condition shared by many cells (the species block: all of them) is a real [n_out x cells] . [cells x n_in] product
// walked by ONE workgroup per slice: the block is sliced finely, cells are staged in LDS DW_STAGE floats at a time and
// the next stage is fetched into registers while the current one is consumed, so the walk is bound by its FMAs, not
// by a global-load round trip per cell.  Every entry is accumulated over the group's cells in batch order by one
// thread: bitwise reproducible, no atomics.
constexpr int DW_U = 4;          // entries per thread
constexpr int DW_STAGE = 4096;   // floats of (dy | x) rows staged per barrier pair
constexpr int DW_NL = DW_STAGE / 256;
constexpr int DW_ROWS = 1024;
__global__ __launch_bounds__(256) void cond_linear_bwd_dw_kernel(int n_in, int n_out, const int32_t* __restrict__ group_cond,
                                                                 const int32_t* __restrict__ group_start,
                                                                 const int32_t* __restrict__ rows,
                                                                 const float* __restrict__ dy, int64_t lddy,
                                                                 const float* __restrict__ x, int64_t ldx,
                                                                 float* __restrict__ grads,
                                                                 const int64_t* __restrict__ w_off,
                                                                 const int64_t* __restrict__ b_off) {
    __shared__ __attribute__((aligned(16))) float sh[DW_STAGE];  // cb x (dy row [n_out] | x row [n_in])
    __shared__ int cells[DW_ROWS];  // the group's first DW_ROWS cell indices (one round trip less per stage)
    const int g = blockIdx.x, tid = threadIdx.x;
    const int c = group_cond[g];
    if (c < 0) return;  // padding of a fixed-size launch (captured programs launch one group per cell)
    float* dW = grads + w_off[c];
    float* db = grads + b_off[c];
    const int n = n_in * n_out;
    const int beg = group_start[g], end = group_start[g + 1];
    const int e_base = blockIdx.y * (256 * DW_U);
    const int stride = n_in + n_out;
    const int CB = DW_STAGE / stride;  // cells per stage (>= 2: n_in + n_out <= 2048)
    int ro[DW_U], co[DW_U];
    float acc[DW_U];
#pragma unroll
    for (int u = 0; u < DW_U; ++u) {
        int e = e_base + tid + 256 * u;
        if (e >= n) e = n - 1;  // clamped: computed, never stored
        ro[u] = e / n_in;
        co[u] = n_out + e % n_in;
        acc[u] = 0.f;
    }
    // element q of this thread's share of a stage: cell qc[q] of the stage, column qj[q] of its (dy | x) row
    int qc[DW_NL], qj[DW_NL];
#pragma unroll
    for (int q = 0; q < DW_NL; ++q) {
        const int i = tid + 256 * q;
        qc[q] = i / stride;
        qj[q] = i - qc[q] * stride;
    }
    // cells[] is a window of the group's cell indices starting at row `win`; loads below are branch-free (clamped
    // addresses + selects): a conditional load per element would serialise DW_NL round trips per stage
    int win = beg;
    auto window = [&](int r0) {
        __syncthreads();
        for (int i = tid; i < min(end - r0, DW_ROWS); i += 256) cells[i] = rows[r0 + i];
        win = r0;
        __syncthreads();
    };
    float pre[DW_NL];
    auto fetch = [&](int r0) {
        if (r0 + CB > win + DW_ROWS) window(r0);  // uniform over the workgroup
        const int nb = min(CB, end - r0);
#pragma unroll
        for (int q = 0; q < DW_NL; ++q) {
            const bool valid = qc[q] < nb;
            const int b = cells[r0 - win + (valid ? qc[q] : 0)];
            const bool from_dy = qj[q] < n_out;
            const float* src = from_dy ? dy + ((int64_t)b * lddy + qj[q]) : x + ((int64_t)b * ldx + (qj[q] - n_out));
            const float v = *src;
            pre[q] = valid ? v : 0.f;
        }
    };
    const bool bias_here = blockIdx.y == 0;
    float accb = 0.f;
    if (beg >= end) return;  // (never: a listed group has cells)
    window(beg);
    fetch(beg);
    for (int r0 = beg; r0 < end; r0 += CB) {
        const int nb = min(CB, end - r0);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < DW_NL; ++q) sh[tid + 256 * q] = pre[q];
        __syncthreads();
        if (r0 + CB < end) fetch(r0 + CB);
#pragma unroll 4
        for (int cb = 0; cb < nb; ++cb) {
            const float* row = sh + cb * stride;
#pragma unroll
            for (int u = 0; u < DW_U; ++u) acc[u] += row[ro[u]] * row[co[u]];
            if (bias_here && tid < n_out) accb += row[tid];
        }
    }
#pragma unroll
    for (int u = 0; u < DW_U; ++u) {
        const int e = e_base + tid + 256 * u;
        if (e < n) dW[e] = acc[u];
    }
    if (bias_here) {
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

}  // namespace

extern "C" int mmvae_cond_linear_fwd(int B, int n_in, int n_out, const float* x, int64_t ldx, const float* params,
                                     const int64_t* w_off, const int64_t* b_off, const int32_t* cond,
                                     const int32_t* rows, float* y, int64_t ldy, mmvae_stream_t stream) {
    if (B <= 0 || n_in <= 0 || n_out <= 0 || n_in > 8192 || !x || !params || !w_off || !b_off || !cond || !y ||
        ldx < n_in || ldy < n_out)
        return MMVAE_ERR_ARG;
    if (rows && n_in <= 256) {
        const dim3 grid((B + FW_CB - 1) / FW_CB, (n_out + FW_RO - 1) / FW_RO);
        const size_t lds = (size_t)FW_CB * n_in * sizeof(float);
#define MMVAE_FWD_SORTED(T)                                                                                            \
    MMVAE_LAUNCH(cond_linear_fwd_sorted_kernel<T>, grid, dim3(256), lds, (hipStream_t)stream, B, n_in, n_out, x, ldx,  \
                 params, w_off, b_off, cond, rows, y, ldy)
        switch ((n_in + 63) / 64) {
            case 1: MMVAE_FWD_SORTED(1); break;
            case 2: MMVAE_FWD_SORTED(2); break;
            case 3: MMVAE_FWD_SORTED(3); break;
            default: MMVAE_FWD_SORTED(4); break;
        }
#undef MMVAE_FWD_SORTED
        MMVAE_LAUNCH_CHECK();
        return MMVAE_OK;
    }
    MMVAE_LAUNCH(cond_linear_fwd_kernel, dim3(B), dim3(256), n_in * sizeof(float), (hipStream_t)stream, n_in, n_out, x,
                 ldx, params, w_off, b_off, cond, y, ldy);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_cond_linear_bwd_dx(int B, int n_in, int n_out, const float* dy, int64_t lddy, const float* params,
                                        const int64_t* w_off, const int32_t* cond, float* dx, int64_t lddx,
                                        int accumulate, mmvae_stream_t stream) {
    if (B <= 0 || n_in <= 0 || n_out <= 0 || n_out > 8192 || !dy || !params || !w_off || !cond || !dx ||
        lddy < n_out || lddx < n_in)
        return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(cond_linear_bwd_dx_kernel, dim3(B), dim3(256), n_out * sizeof(float), (hipStream_t)stream, n_in, n_out,
                 dy, lddy, params, w_off, cond, dx, lddx, accumulate);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_cond_linear_bwd_dw(int n_groups, const int32_t* group_cond, const int32_t* group_start,
                                        const int32_t* rows, int n_in, int n_out, const float* dy, int64_t lddy,
                                        const float* x, int64_t ldx, float* grads, const int64_t* w_off,
                                        const int64_t* b_off, mmvae_stream_t stream) {
    if (n_groups <= 0 || n_in <= 0 || n_out <= 0 || n_in + n_out > 2048 || !group_cond || !group_start || !rows || !dy ||
        !x || !grads || !w_off || !b_off || lddy < n_out || ldx < n_in)
        return MMVAE_ERR_ARG;
    const int64_t n = (int64_t)n_in * n_out;
    const int64_t slices = (n + 256 * DW_U - 1) / (256 * DW_U);
    if (slices > 65535) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(cond_linear_bwd_dw_kernel, dim3(n_groups, (unsigned)slices), dim3(256), 0, (hipStream_t)stream, n_in,
                 n_out, group_cond, group_start, rows, dy, lddy, x, ldx, grads, w_off, b_off);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}
