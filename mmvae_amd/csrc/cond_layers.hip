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
#include <type_traits>

namespace {

// (r5) Several POSITIONS of a "parallel" selection order in one launch (ConditionalLayers.forward with
// selection_order = ["parallel"], components.py:586-631: every conditional layer reads the SAME input and their outputs
// are concatenated -- the layers are independent of each other).  blockIdx.z is the position; each position has its own
// index tables (`tbl` int32 elements apart), its own column block of the output / output gradient (`y` floats apart),
// its own input (`x` floats apart: 0 when all positions read one input) and its own partial slots.
struct PosStride {
    int64_t tbl, x, y, part;
};

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
                                                                     float* __restrict__ y, int64_t ldy, PosStride ps) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // FW_CB input rows
    __shared__ int cell[FW_CB], cnd[FW_CB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    cond += blockIdx.z * ps.tbl;
    rows += blockIdx.z * ps.tbl;
    x += blockIdx.z * ps.x;
    y += blockIdx.z * ps.y;
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
    float wv[FW_RW][T], bias[FW_RW], wn[FW_RW][T], bn[FW_RW];
    auto fetch = [&](int c, float (&wd)[FW_RW][T], float (&bd)[FW_RW]) {
        const float* W = params + w_off[c];
        const float* bs = params + b_off[c];
#pragma unroll
        for (int r = 0; r < FW_RW; ++r) {
            const int o = o0 + r;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int k = lane + 64 * t;
                wd[r][t] = (o < n_out && k < n_in) ? W[(int64_t)o * n_in + k] : 0.f;
            }
            bd[r] = o < n_out ? bs[o] : 0.f;
        }
    };
    fetch(cnd[0], wv, bias);
    for (int j = 0; j < nb; ++j) {
        // the next cell's block rows (when it has another block) are requested before this cell's reduction: one HBM
        // round trip per cell is hidden behind the previous cell's arithmetic (donor_id: every cell has its own block)
        const bool change = j + 1 < nb && cnd[j + 1] != cnd[j];  // wave-uniform
        if (change) fetch(cnd[j + 1], wn, bn);
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
        if (change) {
#pragma unroll
            for (int r = 0; r < FW_RW; ++r) {
#pragma unroll
                for (int t = 0; t < T; ++t) wv[r][t] = wn[r][t];
                bias[r] = bn[r];
            }
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

// dx over cells taken in condition-sorted order.  Workgroups come in sets of DX_CB: set g owns the DX_CB consecutive
// sorted cells rows[DX_CB g ..].  When they all share their block (the layers with few conditions), workgroup 0 of the set
// computes all of them -- every element of W is loaded once and used DX_CB times -- and the others return at once;
// otherwise workgroup s computes cell s alone (one block per cell: nothing to share).  Per cell the arithmetic is the
// same on both paths: thread (h, k) sums W[o][k] dy[o] over its half h of the output rows in ascending order, the two
// halves are added through LDS -- the result does not depend on how the cells happen to be grouped.
constexpr int DX_CB = 8;
constexpr int DX_K = 128;  // input columns per pass
constexpr int DX_U = 8;    // output rows whose W elements are in flight together
__global__ __launch_bounds__(256) void cond_linear_bwd_dx_sorted_kernel(int B, int n_in, int n_out,
                                                                        const float* __restrict__ dy, int64_t lddy,
                                                                        const float* __restrict__ params,
                                                                        const int64_t* __restrict__ w_off,
                                                                        const int32_t* __restrict__ cond,
                                                                        const int32_t* __restrict__ rows,
                                                                        float* __restrict__ dx, int64_t lddx,
                                                                        int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // dy rows [DX_CB][n_out] | half-sums [DX_CB][DX_K]
    __shared__ int cell[DX_CB], cnd[DX_CB];
    const int tid = threadIdx.x, set = blockIdx.x / DX_CB, s = blockIdx.x % DX_CB;
    const int r0 = set * DX_CB, nb = min(DX_CB, B - r0);
    if (tid < nb) {
        const int b = rows[r0 + tid];
        cell[tid] = b;
        cnd[tid] = cond[b];
    }
    __syncthreads();
    bool uniform = true;
    for (int j = 1; j < nb; ++j) uniform = uniform && (cnd[j] == cnd[0]);
    if (uniform ? (s != 0) : (s >= nb)) return;
    const int j0 = uniform ? 0 : s, nc = uniform ? nb : 1;  // this workgroup's cells: j0 .. j0 + nc - 1
    float* dys = sh;
    float* part = sh + DX_CB * n_out;
    for (int i = tid; i < nc * n_out; i += 256) {
        const int j = i / n_out, o = i - j * n_out;
        dys[j * n_out + o] = dy[(int64_t)cell[j0 + j] * lddy + o];
    }
    __syncthreads();
    const float* W = params + w_off[cnd[j0]];
    const int h = tid >> 7, kk = tid & (DX_K - 1);
    const int o_mid = (n_out + 1) / 2, o_beg = h ? o_mid : 0, o_end = h ? n_out : o_mid;
    for (int k0 = 0; k0 < n_in; k0 += DX_K) {
        const int k = k0 + kk;
        float acc[DX_CB];
#pragma unroll
        for (int j = 0; j < DX_CB; ++j) acc[j] = 0.f;
        if (k < n_in) {
            // batches of DX_U output rows: the loads of a batch are issued together, then consumed in ascending order
            const float* wp = W + k;
            auto run = [&](auto full) {  // full: all DX_CB cells (no per-cell predicate in the inner loop)
                int o = o_beg;
                for (; o + DX_U <= o_end; o += DX_U) {
                    float w[DX_U];
#pragma unroll
                    for (int u = 0; u < DX_U; ++u) w[u] = wp[(int64_t)(o + u) * n_in];
#pragma unroll
                    for (int u = 0; u < DX_U; ++u) {
#pragma unroll
                        for (int j = 0; j < DX_CB; ++j)
                            if (decltype(full)::value || j < nc) acc[j] += w[u] * dys[j * n_out + o + u];
                    }
                }
                for (; o < o_end; ++o) {
                    const float w1 = wp[(int64_t)o * n_in];
#pragma unroll
                    for (int j = 0; j < DX_CB; ++j)
                        if (decltype(full)::value || j < nc) acc[j] += w1 * dys[j * n_out + o];
                }
            };
            if (nc == DX_CB)
                run(std::true_type{});
            else
                run(std::false_type{});
        }
        __syncthreads();  // (the previous pass has finished reading `part`)
        if (h == 1) {
#pragma unroll
            for (int j = 0; j < DX_CB; ++j)
                if (j < nc) part[j * DX_K + kk] = acc[j];
        }
        __syncthreads();
        if (h == 0 && k < n_in) {
#pragma unroll
            for (int j = 0; j < DX_CB; ++j)
                if (j < nc) {
                    float* out = dx + (int64_t)cell[j0 + j] * lddx + k;
                    const float v = acc[j] + part[j * DX_K + kk];
                    *out = accumulate ? *out + v : v;
                }
        }
    }
}

// Weight / bias gradient of the blocks PRESENT in the batch.  The host cuts every block's cells (sorted order `rows`,
// cells of a block in batch order) into chunks of at most DW_CHUNK cells; workgroup (chunk, tile) owns a 128 x 128 tile
// of dW = sum_cells dy (x) x over its chunk: the chunk's dy / x rows are staged in LDS by one branch-free pass (one
// float per thread and cell), every thread accumulates an 8 x 8 register tile (16 LDS floats per 64 FMAs).  A block with
// one chunk (dst >= 0) is written straight into the gradient arena; the chunks of a larger block (the species block:
// every cell) leave partials in scratch slots (dst = -2 - slot) that cond_dw_reduce_kernel sums in chunk order.  Fixed
// summation tree, no atomics: bitwise reproducible; a block shared by many cells is spread over workgroups.
constexpr int DW_CHUNK = MMVAE_COND_DW_CHUNK;
constexpr int DW_T = 128;  // tile edge
__global__ __launch_bounds__(256) void cond_dw_chunk_kernel(int n_in, int n_out, const int32_t* __restrict__ chunk_dst,
                                                            const int32_t* __restrict__ chunk_beg,
                                                            const int32_t* __restrict__ chunk_end,
                                                            const int32_t* __restrict__ rows,
                                                            const float* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ x, int64_t ldx,
                                                            float* __restrict__ grads, const int64_t* __restrict__ w_off,
                                                            const int64_t* __restrict__ b_off, float* __restrict__ partials,
                                                            PosStride ps) {
    __shared__ __attribute__((aligned(16))) float sh[DW_CHUNK * 2 * DW_T];  // per cell: dy tile rows | x tile columns
    __shared__ int cells[DW_CHUNK];
    chunk_dst += blockIdx.z * ps.tbl;
    chunk_beg += blockIdx.z * ps.tbl;
    chunk_end += blockIdx.z * ps.tbl;
    rows += blockIdx.z * ps.tbl;
    dy += blockIdx.z * ps.y;
    x += blockIdx.z * ps.x;
    if (partials) partials += blockIdx.z * ps.part;
    const int dst = chunk_dst[blockIdx.x];
    if (dst == -1) return;  // padding of a fixed-size launch
    const int tid = threadIdx.x;
    const int beg = chunk_beg[blockIdx.x], nb = chunk_end[blockIdx.x] - beg;
    const int tiles_k = (n_in + DW_T - 1) / DW_T;
    const int o_base = (blockIdx.y / tiles_k) * DW_T, k_base = (blockIdx.y % tiles_k) * DW_T;
    if (tid < DW_CHUNK) cells[tid] = rows[beg + min(tid, nb - 1)];
    __syncthreads();
    {
        // thread t < 128 stages dy[cell][o_base + t], thread t >= 128 stages x[cell][k_base + t - 128]: unconditional
        // loads from clamped addresses + a select (a conditional load per cell would serialise the round trips)
        const bool is_dy = tid < DW_T;
        const int j = is_dy ? o_base + tid : k_base + tid - DW_T;
        const bool inside = is_dy ? j < n_out : j < n_in;
        const int jc = inside ? j : 0;
        const float* src = is_dy ? dy + jc : x + jc;
        const int64_t ld = is_dy ? lddy : ldx;
        float v[DW_CHUNK];
#pragma unroll
        for (int q = 0; q < DW_CHUNK; ++q) v[q] = src[(int64_t)cells[q] * ld];
#pragma unroll
        for (int q = 0; q < DW_CHUNK; ++q) sh[q * 2 * DW_T + tid] = (inside && q < nb) ? v[q] : 0.f;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;  // rows o_base + 8 ty .. + 7, columns k_base + 8 tx .. + 7
    float acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    float accb = 0.f;
    const bool bias_here = (blockIdx.y % tiles_k) == 0 && tid < DW_T;
    for (int q = 0; q < nb; ++q) {
        const float* cell = sh + q * 2 * DW_T;
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(cell + 8 * ty);
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(cell + 8 * ty + 4);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(cell + DW_T + 8 * tx);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(cell + DW_T + 8 * tx + 4);
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] += dv[i] * xv[j];
        if (bias_here) accb += cell[tid];
    }
    float* dW;
    float* db;
    if (dst >= 0) {
        dW = grads + w_off[dst];
        db = grads + b_off[dst];
    } else {
        dW = partials + (int64_t)(-2 - dst) * ((int64_t)n_in * n_out + n_out);
        db = dW + (int64_t)n_in * n_out;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int o = o_base + 8 * ty + i;
        if (o >= n_out) break;
        float* out = dW + (int64_t)o * n_in + k_base + 8 * tx;
        if (k_base + 8 * tx + 8 <= n_in && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
            *reinterpret_cast<f32x4*>(out) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
            *reinterpret_cast<f32x4*>(out + 4) = f32x4{acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k_base + 8 * tx + j < n_in) out[j] = acc[i][j];
        }
    }
    if (bias_here && o_base + tid < n_out) db[o_base + tid] = accb;
}

// Blocks cut into several chunks: grads[block] = sum of their partial slots, in chunk order.
__global__ __launch_bounds__(256) void cond_dw_reduce_kernel(int n_in, int n_out, const int32_t* __restrict__ red_cond,
                                                             const int32_t* __restrict__ red_slot,
                                                             const int32_t* __restrict__ red_n,
                                                             const float* __restrict__ partials, float* __restrict__ grads,
                                                             const int64_t* __restrict__ w_off,
                                                             const int64_t* __restrict__ b_off, PosStride ps) {
    red_cond += blockIdx.z * ps.tbl;
    red_slot += blockIdx.z * ps.tbl;
    red_n += blockIdx.z * ps.tbl;
    partials += blockIdx.z * ps.part;
    const int c = red_cond[blockIdx.x];
    if (c < 0) return;
    const int64_t n = (int64_t)n_in * n_out, tot = n + n_out;
    const float* src = partials + (int64_t)red_slot[blockIdx.x] * tot;
    const int pieces = red_n[blockIdx.x];
    float* dW = grads + w_off[c];
    float* db = grads + b_off[c];
    // The species block is cut into 16 pieces: summed one dependent load at a time that was 64 round trips per thread (23 us
    // for the batched launch).  Four pieces x four elements are requested together (unconditional loads from clamped
    // addresses) and added in piece order: the same sums, bit for bit.
    int64_t e[4];
    bool ok[4];
    float s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        e[u] = (int64_t)blockIdx.y * 1024 + threadIdx.x + 256 * u;
        ok[u] = e[u] < tot;
        if (!ok[u]) e[u] = 0;
        s[u] = src[e[u]];
    }
    for (int p = 1; p < pieces; p += 4) {
        float v[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t piece = min(p + q, pieces - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[q][u] = src[piece * tot + e[u]];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (p + q < pieces) {
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] += v[q][u];
            }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
        if (e[u] < n)
            dW[e[u]] = s[u];
        else
            db[e[u] - n] = s[u];
    }
}

// dx[b] = sum over the positions j of W[c_j(b)]^T dy_j[b] (parallel selection order: one input, n_pos outputs).  One
// workgroup per cell; per position thread (h, k) sums W[o][k] dy[o] over its half h of the output rows in ascending order
// (DX_U loads in flight together), the halves are added through LDS and the positions' values in DESCENDING position
// order -- the order the per-position launches accumulated in.  No sorting needed: every cell is its own workgroup.
__global__ __launch_bounds__(256) void cond_linear_bwd_dx_multi_kernel(int n_pos, int n_in, int n_out,
                                                                       const float* __restrict__ dy, int64_t lddy,
                                                                       const float* __restrict__ params,
                                                                       const int64_t* __restrict__ w_off,
                                                                       const int32_t* __restrict__ cond,
                                                                       float* __restrict__ dx, int64_t lddx,
                                                                       int accumulate, PosStride ps) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // dy rows of every position [n_pos][n_out] | half-sums [DX_K]
    const int b = blockIdx.x, tid = threadIdx.x;
    float* dys = sh;
    float* part = sh + n_pos * n_out;
    for (int i = tid; i < n_pos * n_out; i += 256) {
        const int j = i / n_out, o = i - j * n_out;
        dys[i] = dy[(int64_t)b * lddy + j * ps.y + o];
    }
    __syncthreads();
    const int h = tid >> 7, kk = tid & (DX_K - 1);
    const int o_mid = (n_out + 1) / 2, o_beg = h ? o_mid : 0, o_end = h ? n_out : o_mid;
    for (int k0 = 0; k0 < n_in; k0 += DX_K) {
        const int k = k0 + kk;
        float total = 0.f;
        for (int j = n_pos - 1; j >= 0; --j) {
            float acc = 0.f;
            if (k < n_in) {
                const float* wp = params + w_off[cond[j * ps.tbl + b]] + k;
                const float* d = dys + j * n_out;
                int o = o_beg;
                for (; o + DX_U <= o_end; o += DX_U) {
                    float w[DX_U];
#pragma unroll
                    for (int u = 0; u < DX_U; ++u) w[u] = wp[(int64_t)(o + u) * n_in];
#pragma unroll
                    for (int u = 0; u < DX_U; ++u) acc += w[u] * d[o + u];
                }
                for (; o < o_end; ++o) acc += wp[(int64_t)o * n_in] * d[o];
            }
            __syncthreads();  // (the previous position has finished reading `part`)
            if (h == 1) part[kk] = acc;
            __syncthreads();
            if (h == 0) {
                const float v = acc + part[kk];
                total = (j == n_pos - 1) ? v : total + v;
            }
        }
        if (h == 0 && k < n_in) {
            float* out = dx + (int64_t)b * lddx + k;
            *out = accumulate ? *out + total : total;
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
                 params, w_off, b_off, cond, rows, y, ldy, PosStride{0, 0, 0, 0})
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
                                        const int64_t* w_off, const int32_t* cond, const int32_t* rows, float* dx,
                                        int64_t lddx, int accumulate, mmvae_stream_t stream) {
    if (B <= 0 || n_in <= 0 || n_out <= 0 || n_out > 8192 || !dy || !params || !w_off || !cond || !dx ||
        lddy < n_out || lddx < n_in)
        return MMVAE_ERR_ARG;
    if (rows && n_out <= 1024) {
        const int sets = (B + DX_CB - 1) / DX_CB;
        MMVAE_LAUNCH(cond_linear_bwd_dx_sorted_kernel, dim3(sets * DX_CB), dim3(256),
                     (size_t)DX_CB * (n_out + DX_K) * sizeof(float), (hipStream_t)stream, B, n_in, n_out, dy, lddy, params,
                     w_off, cond, rows, dx, lddx, accumulate);
        MMVAE_LAUNCH_CHECK();
        return MMVAE_OK;
    }
    MMVAE_LAUNCH(cond_linear_bwd_dx_kernel, dim3(B), dim3(256), n_out * sizeof(float), (hipStream_t)stream, n_in, n_out,
                 dy, lddy, params, w_off, cond, dx, lddx, accumulate);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_cond_linear_bwd_dw(int n_chunks, const int32_t* chunk_dst, const int32_t* chunk_beg,
                                        const int32_t* chunk_end, const int32_t* rows, int n_in, int n_out,
                                        const float* dy, int64_t lddy, const float* x, int64_t ldx, float* grads,
                                        const int64_t* w_off, const int64_t* b_off, int n_red, const int32_t* red_cond,
                                        const int32_t* red_slot, const int32_t* red_n, float* partials,
                                        mmvae_stream_t stream) {
    if (n_chunks <= 0 || n_in <= 0 || n_out <= 0 || !chunk_dst || !chunk_beg || !chunk_end || !rows || !dy || !x ||
        !grads || !w_off || !b_off || lddy < n_out || ldx < n_in || n_red < 0)
        return MMVAE_ERR_ARG;
    if (n_red > 0 && (!red_cond || !red_slot || !red_n || !partials)) return MMVAE_ERR_ARG;
    const int64_t tiles = (int64_t)((n_out + DW_T - 1) / DW_T) * ((n_in + DW_T - 1) / DW_T);
    const int64_t red_y = ((int64_t)n_in * n_out + n_out + 1023) / 1024;
    if (tiles > 65535 || red_y > 65535) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(cond_dw_chunk_kernel, dim3(n_chunks, (unsigned)tiles), dim3(256), 0, (hipStream_t)stream, n_in, n_out,
                 chunk_dst, chunk_beg, chunk_end, rows, dy, lddy, x, ldx, grads, w_off, b_off, partials, PosStride{0, 0, 0, 0});
    MMVAE_LAUNCH_CHECK();
    if (n_red > 0) {
        MMVAE_LAUNCH(cond_dw_reduce_kernel, dim3(n_red, (unsigned)red_y), dim3(256), 0, (hipStream_t)stream, n_in, n_out,
                     red_cond, red_slot, red_n, partials, grads, w_off, b_off, PosStride{0, 0, 0, 0});
        MMVAE_LAUNCH_CHECK();
    }
    return MMVAE_OK;
}

// ---- (r5) n_pos positions of a "parallel" selection order per launch (PosStride above).  Tables of position j start
// tbl_stride int32 elements behind those of position j - 1; outputs / output gradients are column blocks y_pos_stride
// floats apart inside rows of ldy floats; x_pos_stride = 0 when every position reads the same input.
extern "C" int mmvae_cond_linear_fwd_multi(int n_pos, int B, int n_in, int n_out, const float* x, int64_t ldx,
                                           int64_t x_pos_stride, const float* params, const int64_t* w_off,
                                           const int64_t* b_off, const int32_t* cond, const int32_t* rows,
                                           int64_t tbl_stride, float* y, int64_t ldy, int64_t y_pos_stride,
                                           mmvae_stream_t stream) {
    if (n_pos <= 0 || n_pos > 64 || B <= 0 || n_in <= 0 || n_out <= 0 || n_in > 256 || !x || !params || !w_off || !b_off ||
        !cond || !rows || !y || ldx < n_in || ldy < n_out || tbl_stride < B)
        return MMVAE_ERR_ARG;
    const dim3 grid((B + FW_CB - 1) / FW_CB, (n_out + FW_RO - 1) / FW_RO, n_pos);
    const size_t lds = (size_t)FW_CB * n_in * sizeof(float);
    const PosStride ps{tbl_stride, x_pos_stride, y_pos_stride, 0};
#define MMVAE_FWD_SORTED(T)                                                                                            \
    MMVAE_LAUNCH(cond_linear_fwd_sorted_kernel<T>, grid, dim3(256), lds, (hipStream_t)stream, B, n_in, n_out, x, ldx,  \
                 params, w_off, b_off, cond, rows, y, ldy, ps)
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

extern "C" int mmvae_cond_linear_bwd_dx_multi(int n_pos, int B, int n_in, int n_out, const float* dy, int64_t lddy,
                                              int64_t dy_pos_stride, const float* params, const int64_t* w_off,
                                              const int32_t* cond, int64_t tbl_stride, float* dx, int64_t lddx,
                                              int accumulate, mmvae_stream_t stream) {
    if (n_pos <= 0 || n_pos > 64 || B <= 0 || n_in <= 0 || n_out <= 0 || n_out > 1024 || !dy || !params || !w_off || !cond ||
        !dx || lddy < (int64_t)(n_pos - 1) * dy_pos_stride + n_out || lddx < n_in || tbl_stride < B)
        return MMVAE_ERR_ARG;
    const size_t lds = ((size_t)n_pos * n_out + DX_K) * sizeof(float);
    if (lds > 64 * 1024) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(cond_linear_bwd_dx_multi_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, n_pos, n_in, n_out, dy, lddy,
                 params, w_off, cond, dx, lddx, accumulate, PosStride{tbl_stride, 0, dy_pos_stride, 0});
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_cond_linear_bwd_dw_multi(int n_pos, int n_chunks, const int32_t* chunk_dst, const int32_t* chunk_beg,
                                              const int32_t* chunk_end, const int32_t* rows, int64_t tbl_stride, int n_in,
                                              int n_out, const float* dy, int64_t lddy, int64_t dy_pos_stride,
                                              const float* x, int64_t ldx, int64_t x_pos_stride, float* grads,
                                              const int64_t* w_off, const int64_t* b_off, int n_red,
                                              const int32_t* red_cond, const int32_t* red_slot, const int32_t* red_n,
                                              float* partials, int64_t part_pos_stride, mmvae_stream_t stream) {
    if (n_pos <= 0 || n_pos > 64 || n_chunks <= 0 || n_in <= 0 || n_out <= 0 || !chunk_dst || !chunk_beg || !chunk_end ||
        !rows || !dy || !x || !grads || !w_off || !b_off || lddy < n_out || ldx < n_in || n_red < 0)
        return MMVAE_ERR_ARG;
    if (n_red > 0 && (!red_cond || !red_slot || !red_n || !partials)) return MMVAE_ERR_ARG;
    const int64_t tiles = (int64_t)((n_out + DW_T - 1) / DW_T) * ((n_in + DW_T - 1) / DW_T);
    const int64_t red_y = ((int64_t)n_in * n_out + n_out + 1023) / 1024;
    if (tiles > 65535 || red_y > 65535) return MMVAE_ERR_ARG;
    const PosStride ps{tbl_stride, x_pos_stride, dy_pos_stride, part_pos_stride};
    MMVAE_LAUNCH(cond_dw_chunk_kernel, dim3(n_chunks, (unsigned)tiles, n_pos), dim3(256), 0, (hipStream_t)stream, n_in, n_out,
                 chunk_dst, chunk_beg, chunk_end, rows, dy, lddy, x, ldx, grads, w_off, b_off, partials, ps);
    MMVAE_LAUNCH_CHECK();
    if (n_red > 0) {
        MMVAE_LAUNCH(cond_dw_reduce_kernel, dim3(n_red, (unsigned)red_y, n_pos), dim3(256), 0, (hipStream_t)stream, n_in,
                     n_out, red_cond, red_slot, red_n, partials, grads, w_off, b_off, ps);
        MMVAE_LAUNCH_CHECK();
    }
    return MMVAE_OK;
}
