// FCBlock layer tails of the MMVAE step for gfx950 (k6, k7 of SURVEY 2b): split-K slab reduction + bias +
// BatchNorm1d + ReLU + Dropout, forward and backward, and LayerNorm(no affine).
//
// Replaces (reference): nn.BatchNorm1d(momentum=0.01, eps=0.001) components.py:279, activation :282-286,
// nn.Dropout :287-288, nn.LayerNorm(elementwise_affine=False) :281, and their autograd.
//
// These are HBM/L2-bound column reductions over the batch axis.  Layout: a 2-D grid of workgroups, each owning a
// strip of 64 feature columns (lane = column, so every row access is one coalesced 256-B wave access) x a chunk of
// 32 batch rows (4 wavefronts x 8 rows) -> (N/64) x (B/32) workgroups fill the chip even for the 256..1024-wide core
// layers.  Column statistics are two-pass without atomics: pass 1 leaves per-chunk partials (mean + centred M2 for
// BatchNorm, combined with Chan's formula; plain sums in backward) in a small workspace, pass 2 merges them in chunk
// order (bitwise reproducible) and normalises.  The split-K partial slabs of the producing GEMM are summed on the
// fly in pass 1: no separate reduce pass exists.
#include "common.h"

namespace {

constexpr int CW = 64;   // columns per workgroup (one per lane: every row access is one coalesced 256-B wave load)
constexpr int RPC = 32;  // rows per workgroup ("row chunk"): 4 wavefronts x 8 rows
constexpr int RPW = 8;
constexpr int CT = 256;

struct FwdArgs {
    const float* in;
    int64_t ld_in, slab_stride;
    int n_slabs;
    const float* bias;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    int64_t* nbt;
    float momentum, eps;
    int training, relu;
    const uint8_t* mask;
    float keep_scale;
    float* z_out;
    float* a_out;
    float* d_out;
    int64_t ld_out;
    float* save_mean;
    float* save_invstd;
    float* ws;  // [2][RC][N] per-chunk (mean, M2)
    int B, N, RC;
    unsigned short* d_planes;  // optional: the three bf16 planes of d_out (pre-split operand of a bf16x3 GEMM)
    int64_t ldp, pstride;
    // optional piggy-backed pass (fc_fwd_apply): the workgroup rows beyond the layer's own grid (blockIdx.y >= RC) split
    // an unrelated fp32 matrix [sp_rows, 8 * sp_groups] into its bf16 planes -- the engine's input batch, beside the
    // first layer's tail, where a launch of its own would sit on the critical path of the step
    const float* sp_src;
    unsigned short* sp_planes;
    int64_t sp_ld_src, sp_ld, sp_pstride;
    int sp_rows, sp_groups, sp_cols;
};

__device__ __forceinline__ void split_rows_job(const FwdArgs& a, int wg, int nwg) {
    const int64_t total = (int64_t)a.sp_rows * a.sp_groups;
    for (int64_t idx = (int64_t)wg * CT + threadIdx.x; idx < total; idx += (int64_t)nwg * CT) {
        const int row = (int)(idx / a.sp_groups), gq = (int)(idx - (int64_t)row * a.sp_groups);
        const float* sp = a.sp_src + (int64_t)row * a.sp_ld_src + 8 * gq;
        f32x4 v0, v1;
        if (8 * gq + 8 <= a.sp_cols) {
            v0 = *reinterpret_cast<const f32x4*>(sp);
            v1 = *reinterpret_cast<const f32x4*>(sp + 4);
        } else {  // a column count off a multiple of 8: the group's tail is zeros (mmvae_split_planes_f32)
            const int left = a.sp_cols - 8 * gq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v0[j] = j < left ? sp[j] : 0.f;
                v1[j] = 4 + j < left ? sp[4 + j] : 0.f;
            }
        }
        unsigned q[8][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x3_split(v0[j], q[j][0], q[j][1], q[j][2]);
            x3_split(v1[j], q[4 + j][0], q[4 + j][1], q[4 + j][2]);
        }
        unsigned short* dp = a.sp_planes + (int64_t)row * a.sp_ld + 8 * gq;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint4*>(dp + p * a.sp_pstride) =
                // (the odd element keeps its top half only: the last piece of a value in the fp32 denormal range has low bits)
                make_uint4((q[0][p] >> 16) | (q[1][p] & 0xFFFF0000u), (q[2][p] >> 16) | (q[3][p] & 0xFFFF0000u),
                           (q[4][p] >> 16) | (q[5][p] & 0xFFFF0000u), (q[6][p] >> 16) | (q[7][p] & 0xFFFF0000u));
    }
}

// Sum of `v` over the 4 waves for each of the 64 columns; result valid in every thread.  Fixed order.
__device__ __forceinline__ float block_colsum(float v, float (*red)[CW], int w, int lane) {
    __syncthreads();
    red[w][lane] = v;
    __syncthreads();
    return (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// acc[i] += sum over the split-K slabs of element (row r0 + i, column c), accumulated in slab order (bitwise the plain
// loop) but with the loads of 4 slabs x RPW rows issued together: at 4 wavefronts per CU a dependent load per slab is
// pure latency (16 slabs of the G-wide GEMMs took 34 us for a 2 MB matrix).  Rows >= B are clamped (callers mask).
__device__ __forceinline__ void slab_sum_rows(float (&acc)[RPW], const float* __restrict__ in, int64_t ld,
                                              int64_t slab_stride, int n_slabs, int r0, int B, int c) {
    const float* p[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) p[i] = in + (int64_t)min(r0 + i, B - 1) * ld + c;
    int s = 0;
    for (; s + 4 <= n_slabs; s += 4) {
        float t[4][RPW];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < RPW; ++i) t[u][i] = p[i][(int64_t)(s + u) * slab_stride];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < RPW; ++i) acc[i] += t[u][i];
    }
    for (; s < n_slabs; ++s) {
        float t[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) t[i] = p[i][(int64_t)s * slab_stride];
#pragma unroll
        for (int i = 0; i < RPW; ++i) acc[i] += t[i];
    }
}

// Pass 1 of training BatchNorm: z = bias + sum of slabs (stored), per-chunk column mean and centred M2.
__global__ __launch_bounds__(CT) void fc_fwd_stats_kernel(const FwdArgs a) {
    __shared__ float red[4][CW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * CW + lane;
    const int chunk = blockIdx.y;
    const bool cv = c < a.N;
    const float bias = (cv && a.bias) ? a.bias[c] : 0.f;
    const int r0 = chunk * RPC + w * RPW;
    float zv[RPW];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RPW; ++i) zv[i] = bias;
    slab_sum_rows(zv, a.in, a.ld_in, a.slab_stride, a.n_slabs, r0, a.B, cv ? c : a.N - 1);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = r0 + i;
        if (cv && r < a.B) {
            a.z_out[(int64_t)r * a.ld_out + c] = zv[i];
            s += zv[i];
        } else {
            zv[i] = 0.f;
        }
    }
    const int nb = min(RPC, a.B - chunk * RPC);
    const float mean = block_colsum(s, red, w, lane) / (float)nb;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < RPW; ++i)
        if (r0 + i < a.B) {
            const float d = zv[i] - mean;
            m2 += d * d;
        }
    m2 = block_colsum(m2, red, w, lane);
    if (w == 0 && cv) {
        a.ws[(int64_t)chunk * a.N + c] = mean;
        a.ws[(int64_t)(a.RC + chunk) * a.N + c] = m2;
    }
}

// Pass 2 (or the only pass without training BN): normalise / activate / drop this chunk's rows.
template <bool HAS_BN>
__global__ __launch_bounds__(CT) void fc_fwd_apply_kernel(const FwdArgs a) {
    if ((int)blockIdx.y >= a.RC) {  // (block-uniform) the piggy-backed split job
        split_rows_job(a, ((int)blockIdx.y - a.RC) * gridDim.x + blockIdx.x, ((int)gridDim.y - a.RC) * gridDim.x);
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * CW + lane;
    const int chunk = blockIdx.y;
    if (c >= a.N) return;
    const float bias = a.bias ? a.bias[c] : 0.f;
    float mean = 0.f, invstd = 1.f, gam = 1.f, bet = 0.f;
    const bool stats = HAS_BN && a.training;
    const int r0 = chunk * RPC + w * RPW;
    float zs[RPW];
    unsigned char mk[RPW];
    if (stats || a.mask) {  // this thread's 8 rows of z and keep-mask bytes: in flight beside the statistics below
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int rc = min(r0 + i, a.B - 1);
            if (stats) zs[i] = a.z_out[(int64_t)rc * a.ld_out + c];
            mk[i] = a.mask ? a.mask[(int64_t)rc * a.N + c] : (unsigned char)1;
        }
    }
    if (HAS_BN) {
        gam = a.gamma ? a.gamma[c] : 1.f;
        bet = a.beta ? a.beta[c] : 0.f;
        if (stats) {
            // Chan et al. pairwise merge of the per-chunk (n, mean, M2), in chunk order (bitwise reproducible).  The
            // partials are fetched 8 chunks at a time AHEAD of the merge: the recurrence (two divisions per chunk) is a
            // dependent chain, and with a load inside every link the 16 chunks of a 512-row batch were 16 L2 round trips
            // (r3: 13.5 us for a 2 MB layer)
            float n = 0.f, M2 = 0.f;
            for (int ch0 = 0; ch0 < a.RC; ch0 += 8) {
                float mb[8], m2b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ch = min(ch0 + u, a.RC - 1);
                    mb[u] = a.ws[(int64_t)ch * a.N + c];
                    m2b[u] = a.ws[(int64_t)(a.RC + ch) * a.N + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ch = ch0 + u;
                    if (ch < a.RC) {
                        const float nb = (float)min(RPC, a.B - ch * RPC);
                        const float delta = mb[u] - mean;
                        const float nn = n + nb;
                        mean += delta * (nb / nn);
                        M2 += m2b[u] + delta * delta * (n * nb / nn);
                        n = nn;
                    }
                }
            }
            const float var = M2 / (float)a.B;
            invstd = 1.0f / sqrtf(var + a.eps);
            if (chunk == 0 && w == 0) {
                if (a.save_mean) a.save_mean[c] = mean;
                if (a.save_invstd) a.save_invstd[c] = invstd;
                if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mean;
                if (a.running_var) {
                    const float unb = a.B > 1 ? var * ((float)a.B / (float)(a.B - 1)) : var;
                    a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * unb;
                }
                if (blockIdx.x == 0 && lane == 0 && a.nbt) *a.nbt += 1;
            }
        } else {
            mean = a.running_mean[c];
            invstd = 1.0f / sqrtf(a.running_var[c] + a.eps);
        }
    }
    if (!stats) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) zs[i] = bias;
        slab_sum_rows(zs, a.in, a.ld_in, a.slab_stride, a.n_slabs, r0, a.B, c);
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = r0 + i;
        if (r >= a.B) break;
        const int64_t o = (int64_t)r * a.ld_out + c;
        const float z = zs[i];
        if (!stats && a.z_out) a.z_out[o] = z;
        float y = z;
        if (HAS_BN) y = (z - mean) * invstd * gam + bet;
        if (a.relu) y = fmaxf(y, 0.f);
        if (a.a_out) a.a_out[o] = y;
        if (a.d_out) {
            float d = y;
            if (a.mask) d = mk[i] ? y * a.keep_scale : 0.f;
            a.d_out[o] = d;
            if (a.d_planes) store_planes_lanepair(a.d_planes, a.ldp, a.pstride, r, c, d, lane);
        }
    }
}

struct BwdArgs {
    const float* din;
    int64_t ld_in, slab_stride;
    int n_slabs;
    const float* addend;
    const float* addend_a;  // gradient on the pre-dropout activation: added behind the keep mask
    const float* row_scale;
    const uint8_t* mask;
    float keep_scale;
    int relu;
    const float* a;
    const float* z;
    const float* gamma;
    const float* save_mean;
    const float* save_invstd;
    float* dz_out;
    int64_t ld_out;
    float* dbias;
    float* dgamma;
    float* dbeta;
    float* ws;  // [3][RC][N] per-chunk column sums: dy, dy*xhat, xhat
    int B, N, RC;
    unsigned short* dz_planes;  // optional: the three bf16 planes of the final dz_out
    int64_t ldp, pstride;
};

// Operands of bwd_dy for this thread's RPW rows, fetched together ahead of the arithmetic (inside the per-row branch
// they were one memory round trip per row).  Rows >= B are clamped; the caller masks them.
struct BwdRowOps {
    float addend[RPW], row_scale[RPW], addend_a[RPW], act[RPW], z[RPW];
    unsigned char mask[RPW];
};
template <bool HAS_BN>
__device__ __forceinline__ void bwd_load_rows(const BwdArgs& a, int r0, int c, BwdRowOps& o) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = min(r0 + i, a.B - 1);
        const int64_t oo = (int64_t)r * a.ld_out + c;
        o.addend[i] = a.addend ? a.addend[oo] : 0.f;
        o.row_scale[i] = a.row_scale ? a.row_scale[r] : 1.f;
        o.mask[i] = a.mask ? a.mask[(int64_t)r * a.N + c] : (unsigned char)1;
        o.addend_a[i] = a.addend_a ? a.addend_a[oo] : 0.f;
        o.act[i] = a.relu ? a.a[oo] : 1.f;
        o.z[i] = HAS_BN ? a.z[oo] : 0.f;
    }
}
__device__ __forceinline__ float bwd_dy(const BwdArgs& a, const BwdRowOps& o, int i, float g) {
    // addend / a / z / dz_out share the [B, ld_out] geometry of the layer's own activations
    if (a.addend) g += o.addend[i];
    if (a.row_scale) g *= o.row_scale[i];
    if (a.mask) g = o.mask[i] ? g * a.keep_scale : 0.f;
    if (a.addend_a) g += o.addend_a[i];
    if (a.relu) g = (o.act[i] > 0.f) ? g : 0.f;
    return g;
}

// Pass 1: dy (through dropout / ReLU) stored to dz_out, per-chunk column sums to the workspace.
template <bool HAS_BN>
__global__ __launch_bounds__(CT) void fc_bwd_stats_kernel(const BwdArgs a) {
    __shared__ float red[4][CW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * CW + lane;
    const int chunk = blockIdx.y;
    const bool cv = c < a.N;
    float mean = 0.f, invstd = 1.f;
    if (HAS_BN && cv) {
        mean = a.save_mean[c];
        invstd = a.save_invstd[c];
    }
    const int r0 = chunk * RPC + w * RPW;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    float gs[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) gs[i] = 0.f;
    BwdRowOps ops;
    bwd_load_rows<HAS_BN>(a, r0, cv ? c : a.N - 1, ops);
    slab_sum_rows(gs, a.din, a.ld_in, a.slab_stride, a.n_slabs, r0, a.B, cv ? c : a.N - 1);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = r0 + i;
        if (cv && r < a.B) {
            const float dy = bwd_dy(a, ops, i, gs[i]);
            const int64_t oo = (int64_t)r * a.ld_out + c;
            if (a.dz_out) a.dz_out[oo] = dy;
            if (!HAS_BN && a.dz_planes) store_planes_lanepair(a.dz_planes, a.ldp, a.pstride, r, c, dy, lane);
            s1 += dy;
            if (HAS_BN) {
                const float xh = (ops.z[i] - mean) * invstd;
                s2 += dy * xh;
                s3 += xh;
            }
        }
    }
    if (!a.ws) return;  // block-uniform: no column sums requested
    s1 = block_colsum(s1, red, w, lane);
    if (HAS_BN) {
        s2 = block_colsum(s2, red, w, lane);
        s3 = block_colsum(s3, red, w, lane);
    }
    if (w == 0 && cv) {
        a.ws[(int64_t)chunk * a.N + c] = s1;
        if (HAS_BN) {
            a.ws[(int64_t)(a.RC + chunk) * a.N + c] = s2;
            a.ws[(int64_t)(2 * a.RC + chunk) * a.N + c] = s3;
        }
    }
}

// Pass 2 with BN: dz = gamma * invstd * (dy - mean(dy) - xhat * mean(dy * xhat)); dbeta, dgamma; the gradient of
// the Linear bias ahead of a BatchNorm is zero in exact arithmetic: sum_b dz = -gamma*invstd*mean(dy*xhat)*sum_b xhat.
__global__ __launch_bounds__(CT) void fc_bwd_apply_kernel(const BwdArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * CW + lane;
    const int chunk = blockIdx.y;
    if (c >= a.N) return;
    const int r0 = chunk * RPC + w * RPW;
    float dys[RPW], zsv[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {  // 16 loads in flight together (rows >= B clamped)
        const int64_t oc = (int64_t)min(r0 + i, a.B - 1) * a.ld_out + c;
        dys[i] = a.dz_out[oc];
        zsv[i] = a.z[oc];
    }
    float S1 = 0.f, S2 = 0.f, S3 = 0.f;
    for (int ch0 = 0; ch0 < a.RC; ch0 += 8) {  // 8 chunks' partials in flight together, summed in chunk order
        float t1[8], t2[8], t3[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ch = min(ch0 + u, a.RC - 1);
            t1[u] = a.ws[(int64_t)ch * a.N + c];
            t2[u] = a.ws[(int64_t)(a.RC + ch) * a.N + c];
            t3[u] = a.ws[(int64_t)(2 * a.RC + ch) * a.N + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (ch0 + u < a.RC) {
                S1 += t1[u];
                S2 += t2[u];
                S3 += t3[u];
            }
    }
    const float mean = a.save_mean[c], invstd = a.save_invstd[c];
    const float gam = a.gamma ? a.gamma[c] : 1.f;
    const float invB = 1.f / (float)a.B;
    const float m1 = S1 * invB, m2 = S2 * invB;
    if (chunk == 0 && w == 0) {
        if (a.dbeta) a.dbeta[c] = S1;
        if (a.dgamma) a.dgamma[c] = S2;
        if (a.dbias) a.dbias[c] = -gam * invstd * m2 * S3;
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = r0 + i;
        if (r >= a.B) break;
        const int64_t oo = (int64_t)r * a.ld_out + c;
        const float dy = dys[i];
        const float xh = (zsv[i] - mean) * invstd;
        const float dz = gam * invstd * (dy - m1 - xh * m2);
        a.dz_out[oo] = dz;
        if (a.dz_planes) store_planes_lanepair(a.dz_planes, a.ldp, a.pstride, r, c, dz, lane);
    }
}

// Pass 2 without BN: dbias[c] = sum over chunks of the partial column sums (chunk order: reproducible).
__global__ __launch_bounds__(256) void fc_colsum_finish_kernel(const float* __restrict__ ws, int RC, int N,
                                                               float* __restrict__ dbias) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int ch0 = 0; ch0 < RC; ch0 += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ws[(int64_t)min(ch0 + u, RC - 1) * N + c];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (ch0 + u < RC) s += t[u];
    }
    dbias[c] = s;
}

// LayerNorm without affine: one wavefront per row.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(int B, int N, const float* __restrict__ x, int64_t ldx,
                                                            float eps, float* __restrict__ y, int64_t ldy,
                                                            float* save_mean, float* save_invstd) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* xr = x + (int64_t)row * ldx;
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += xr[j];
    const float mean = wave_sum(s) / (float)N;
    float s2 = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float d = xr[j] - mean;
        s2 += d * d;
    }
    const float invstd = 1.0f / sqrtf(wave_sum(s2) / (float)N + eps);
    float* yr = y + (int64_t)row * ldy;
    for (int j = lane; j < N; j += 64) yr[j] = (xr[j] - mean) * invstd;
    if (lane == 0) {
        if (save_mean) save_mean[row] = mean;
        if (save_invstd) save_invstd[row] = invstd;
    }
}

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(int B, int N, const float* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ y, int64_t ldy,
                                                            const float* __restrict__ save_invstd,
                                                            float* __restrict__ dx, int64_t lddx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* gr = dy + (int64_t)row * lddy;
    const float* yr = y + (int64_t)row * ldy;
    float s1 = 0.f, s2 = 0.f;
    for (int j = lane; j < N; j += 64) {
        s1 += gr[j];
        s2 += gr[j] * yr[j];
    }
    const float m1 = wave_sum(s1) / (float)N, m2 = wave_sum(s2) / (float)N;
    const float invstd = save_invstd[row];
    float* dr = dx + (int64_t)row * lddx;
    for (int j = lane; j < N; j += 64) dr[j] = invstd * (gr[j] - m1 - yr[j] * m2);
}

}  // namespace

extern "C" size_t mmvae_fc_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return (size_t)3 * (size_t)ceil_div_i(B, RPC) * (size_t)N * sizeof(float);
}

static int fc_fwd_impl(int B, int N, const float* in, int64_t ld_in, int n_slabs, const float* bias,
                       const mmvae_bn_params* bn, int training, int relu, const uint8_t* keep_mask, float dropout_p,
                       float* z_out, float* a_out, float* d_out, int64_t ld_out, float* save_mean, float* save_invstd,
                       float* workspace, size_t workspace_bytes, uint16_t* d_planes, int64_t ldp, int64_t pstride,
                       mmvae_stream_t stream, int sp_rows = 0, int sp_cols = 0, const float* sp_src = nullptr,
                       int64_t sp_ld_src = 0, uint16_t* sp_planes = nullptr, int64_t sp_ld = 0, int64_t sp_pstride = 0) {
    if (d_planes && (!d_out || N % 2 != 0 || ldp < N || ldp % 2 != 0 || pstride < (int64_t)B * ldp)) return MMVAE_ERR_ARG;
    if (sp_planes && (sp_rows <= 0 || sp_cols <= 0 || !sp_src || sp_ld_src < sp_cols || sp_ld < (sp_cols + 7) / 8 * 8 ||
                      sp_ld % 8 != 0 || sp_pstride % 8 != 0 || sp_pstride < (int64_t)sp_rows * sp_ld ||
                      (reinterpret_cast<uintptr_t>(sp_planes) & 15u)))
        return MMVAE_ERR_ARG;
    if (B <= 0 || N <= 0 || !in || n_slabs < 1 || ld_in < N || ld_out < N) return MMVAE_ERR_ARG;
    if (!a_out && !d_out) return MMVAE_ERR_ARG;
    if (keep_mask && (dropout_p < 0.f || dropout_p >= 1.f || !d_out)) return MMVAE_ERR_ARG;
    if (bn && training && (!z_out || !save_mean || !save_invstd)) return MMVAE_ERR_ARG;
    if (bn && !training && (!bn->running_mean || !bn->running_var)) return MMVAE_ERR_ARG;
    const bool stats = bn && training;
    if (stats && (!workspace || workspace_bytes < mmvae_fc_workspace_bytes(B, N))) return MMVAE_ERR_WORKSPACE;
    FwdArgs a = {};
    a.in = in;
    a.ld_in = ld_in;
    a.slab_stride = (int64_t)B * ld_in;
    a.n_slabs = n_slabs;
    a.bias = bias;
    if (bn) {
        a.gamma = bn->gamma;
        a.beta = bn->beta;
        a.running_mean = bn->running_mean;
        a.running_var = bn->running_var;
        a.nbt = bn->num_batches_tracked;
        a.momentum = bn->momentum;
        a.eps = bn->eps;
    }
    a.training = training;
    a.relu = relu;
    a.mask = keep_mask;
    a.keep_scale = keep_mask ? 1.0f / (1.0f - dropout_p) : 1.f;
    a.z_out = z_out;
    a.a_out = a_out;
    a.d_out = d_out;
    a.ld_out = ld_out;
    a.save_mean = save_mean;
    a.save_invstd = save_invstd;
    a.ws = workspace;
    a.B = B;
    a.N = N;
    a.RC = ceil_div_i(B, RPC);
    a.d_planes = reinterpret_cast<unsigned short*>(d_planes);
    a.ldp = ldp;
    a.pstride = pstride;
    const dim3 grid(ceil_div_i(N, CW), a.RC);
    dim3 grid_apply = grid;
    if (sp_planes) {
        a.sp_src = sp_src;
        a.sp_planes = reinterpret_cast<unsigned short*>(sp_planes);
        a.sp_ld_src = sp_ld_src;
        a.sp_ld = sp_ld;
        a.sp_pstride = sp_pstride;
        a.sp_rows = sp_rows;
        a.sp_groups = (sp_cols + 7) / 8;
        a.sp_cols = sp_cols;
        // ~3 workgroups per CU for the split job, as rows of the layer's own grid width
        grid_apply.y = a.RC + ceil_div_i(768, (int)grid.x);
    }
    hipStream_t s = (hipStream_t)stream;
    if (stats) {
        MMVAE_LAUNCH(fc_fwd_stats_kernel, grid, dim3(CT), 0, s, a);
        MMVAE_LAUNCH_CHECK();
    }
    if (bn)
        MMVAE_LAUNCH(fc_fwd_apply_kernel<true>, grid_apply, dim3(CT), 0, s, a);
    else
        MMVAE_LAUNCH(fc_fwd_apply_kernel<false>, grid_apply, dim3(CT), 0, s, a);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_fc_epilogue_fwd(int B, int N, const float* in, int64_t ld_in, int n_slabs, const float* bias,
                                     const mmvae_bn_params* bn, int training, int relu, const uint8_t* keep_mask,
                                     float dropout_p, float* z_out, float* a_out, float* d_out, int64_t ld_out,
                                     float* save_mean, float* save_invstd, float* workspace, size_t workspace_bytes,
                                     mmvae_stream_t stream) {
    return fc_fwd_impl(B, N, in, ld_in, n_slabs, bias, bn, training, relu, keep_mask, dropout_p, z_out, a_out, d_out, ld_out,
                       save_mean, save_invstd, workspace, workspace_bytes, nullptr, 0, 0, stream);
}

extern "C" int mmvae_fc_epilogue_fwd_planes(int B, int N, const float* in, int64_t ld_in, int n_slabs, const float* bias,
                                            const mmvae_bn_params* bn, int training, int relu, const uint8_t* keep_mask,
                                            float dropout_p, float* z_out, float* a_out, float* d_out, int64_t ld_out,
                                            float* save_mean, float* save_invstd, float* workspace,
                                            size_t workspace_bytes, uint16_t* d_planes, int64_t ldp, int64_t plane_stride,
                                            mmvae_stream_t stream) {
    return fc_fwd_impl(B, N, in, ld_in, n_slabs, bias, bn, training, relu, keep_mask, dropout_p, z_out, a_out, d_out, ld_out,
                       save_mean, save_invstd, workspace, workspace_bytes, d_planes, ldp, plane_stride, stream);
}

extern "C" int mmvae_fc_epilogue_fwd_split(int B, int N, const float* in, int64_t ld_in, int n_slabs, const float* bias,
                                           const mmvae_bn_params* bn, int training, int relu, const uint8_t* keep_mask,
                                           float dropout_p, float* z_out, float* a_out, float* d_out, int64_t ld_out,
                                           float* save_mean, float* save_invstd, float* workspace, size_t workspace_bytes,
                                           int sp_rows, int sp_cols, const float* sp_src, int64_t sp_ld_src,
                                           uint16_t* sp_planes, int64_t sp_ld, int64_t sp_plane_stride,
                                           mmvae_stream_t stream) {
    if (!sp_planes) return MMVAE_ERR_ARG;
    return fc_fwd_impl(B, N, in, ld_in, n_slabs, bias, bn, training, relu, keep_mask, dropout_p, z_out, a_out, d_out, ld_out,
                       save_mean, save_invstd, workspace, workspace_bytes, nullptr, 0, 0, stream, sp_rows, sp_cols, sp_src,
                       sp_ld_src, sp_planes, sp_ld, sp_plane_stride);
}

static int fc_bwd_impl(int B, int N, const float* din, int64_t ld_in, int n_slabs, const float* addend,
                       const float* addend_a, const float* row_scale, const uint8_t* keep_mask, float dropout_p, int relu,
                       const float* a_act, const float* z, const float* gamma, const float* save_mean,
                       const float* save_invstd, int has_bn, float* dz_out, int64_t ld_out, float* dbias, float* dgamma,
                       float* dbeta, float* workspace, size_t workspace_bytes, uint16_t* dz_planes, int64_t ldp,
                       int64_t pstride, mmvae_stream_t stream) {
    if (dz_planes && (!dz_out || N % 2 != 0 || ldp < N || ldp % 2 != 0 || pstride < (int64_t)B * ldp)) return MMVAE_ERR_ARG;
    if (B <= 0 || N <= 0 || !din || n_slabs < 1 || ld_in < N || ld_out < N) return MMVAE_ERR_ARG;
    if (relu && !a_act) return MMVAE_ERR_ARG;
    if (keep_mask && (dropout_p < 0.f || dropout_p >= 1.f)) return MMVAE_ERR_ARG;
    if (has_bn && (!z || !save_mean || !save_invstd || !dz_out)) return MMVAE_ERR_ARG;
    if (!dz_out && !dbias && !workspace) return MMVAE_ERR_ARG;
    const bool need_ws = has_bn || dbias || workspace;  // without BN a given workspace always receives the partials
    if (need_ws && (!workspace || workspace_bytes < mmvae_fc_workspace_bytes(B, N))) return MMVAE_ERR_WORKSPACE;
    BwdArgs a = {};
    a.din = din;
    a.ld_in = ld_in;
    a.slab_stride = (int64_t)B * ld_in;
    a.n_slabs = n_slabs;
    a.addend = addend;
    a.addend_a = addend_a;
    a.row_scale = row_scale;
    a.mask = keep_mask;
    a.keep_scale = keep_mask ? 1.0f / (1.0f - dropout_p) : 1.f;
    a.relu = relu;
    a.a = a_act;
    a.z = z;
    a.gamma = gamma;
    a.save_mean = save_mean;
    a.save_invstd = save_invstd;
    a.dz_out = dz_out;
    a.ld_out = ld_out;
    a.dbias = dbias;
    a.dgamma = dgamma;
    a.dbeta = dbeta;
    a.ws = need_ws ? workspace : nullptr;
    a.B = B;
    a.N = N;
    a.RC = ceil_div_i(B, RPC);
    a.dz_planes = reinterpret_cast<unsigned short*>(dz_planes);
    a.ldp = ldp;
    a.pstride = pstride;
    const dim3 grid(ceil_div_i(N, CW), a.RC);
    hipStream_t s = (hipStream_t)stream;
    if (has_bn) {
        MMVAE_LAUNCH(fc_bwd_stats_kernel<true>, grid, dim3(CT), 0, s, a);
        MMVAE_LAUNCH_CHECK();
        MMVAE_LAUNCH(fc_bwd_apply_kernel, grid, dim3(CT), 0, s, a);
    } else {
        MMVAE_LAUNCH(fc_bwd_stats_kernel<false>, grid, dim3(CT), 0, s, a);
        MMVAE_LAUNCH_CHECK();
        if (dbias)
            MMVAE_LAUNCH(fc_colsum_finish_kernel, dim3(ceil_div_i(N, 256)), dim3(256), 0, s, workspace, a.RC, N,
                               dbias);
    }
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_fc_epilogue_bwd(int B, int N, const float* din, int64_t ld_in, int n_slabs, const float* addend,
                                     const float* addend_a, const float* row_scale, const uint8_t* keep_mask, float dropout_p, int relu,
                                     const float* a_act, const float* z, const float* gamma, const float* save_mean,
                                     const float* save_invstd, int has_bn, float* dz_out, int64_t ld_out, float* dbias,
                                     float* dgamma, float* dbeta, float* workspace, size_t workspace_bytes,
                                     mmvae_stream_t stream) {
    return fc_bwd_impl(B, N, din, ld_in, n_slabs, addend, addend_a, row_scale, keep_mask, dropout_p, relu, a_act, z, gamma,
                       save_mean, save_invstd, has_bn, dz_out, ld_out, dbias, dgamma, dbeta, workspace, workspace_bytes,
                       nullptr, 0, 0, stream);
}

extern "C" int mmvae_fc_epilogue_bwd_planes(int B, int N, const float* din, int64_t ld_in, int n_slabs,
                                            const float* addend, const float* addend_a, const float* row_scale,
                                            const uint8_t* keep_mask, float dropout_p, int relu, const float* a_act,
                                            const float* z, const float* gamma, const float* save_mean,
                                            const float* save_invstd, int has_bn, float* dz_out, int64_t ld_out,
                                            float* dbias, float* dgamma, float* dbeta, float* workspace,
                                            size_t workspace_bytes, uint16_t* dz_planes, int64_t ldp,
                                            int64_t plane_stride, mmvae_stream_t stream) {
    return fc_bwd_impl(B, N, din, ld_in, n_slabs, addend, addend_a, row_scale, keep_mask, dropout_p, relu, a_act, z, gamma,
                       save_mean, save_invstd, has_bn, dz_out, ld_out, dbias, dgamma, dbeta, workspace, workspace_bytes,
                       dz_planes, ldp, plane_stride, stream);
}

extern "C" int mmvae_layernorm_fwd(int B, int N, const float* x, int64_t ldx, float eps, float* y, int64_t ldy,
                                   float* save_mean, float* save_invstd, mmvae_stream_t stream) {
    if (B <= 0 || N <= 0 || !x || !y || ldx < N || ldy < N) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(layernorm_fwd_kernel, dim3(ceil_div_i(B, 4)), dim3(256), 0, (hipStream_t)stream, B, N, x, ldx,
                       eps, y, ldy, save_mean, save_invstd);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_layernorm_bwd(int B, int N, const float* dy, int64_t lddy, const float* y, int64_t ldy,
                                   const float* save_invstd, float* dx, int64_t lddx, mmvae_stream_t stream) {
    if (B <= 0 || N <= 0 || !dy || !y || !save_invstd || !dx || lddy < N || ldy < N || lddx < N) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(layernorm_bwd_kernel, dim3(ceil_div_i(B, 4)), dim3(256), 0, (hipStream_t)stream, B, N, dy,
                       lddy, y, ldy, save_invstd, dx, lddx);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}
