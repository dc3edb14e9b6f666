// Row-owner adversary passes for gfx950 (include/mmvae_hip.h, "Row-owner adversary passes").
//
// Replaces (reference): one adversarial phase of CMMVAEModel.grf, models/cmmvae_model.py:59-101 -- Adversarial.encoder
// (modules/base/components.py:638-674: an FCBlock of Linear -> ReLU -> Dropout layers without BatchNorm), every head,
// CrossEntropyLoss(reduction="sum") (cmmvae_model.py:54,85), its autograd, GradientReversalFunction
// (components.py:879-899) -- and the clip_gradients / grad-norm bookkeeping of :118-131 that follows the backward.
//
// Nothing in such a stack couples cells except the weight gradients: a workgroup that owns 16 cells runs the forward,
// the cross-entropy (log-sum-exp is row-local) and the backward down to the hidden representation without a grid
// barrier.  The step had 12-17 latency-bound launches per adversary and phase; this file makes it three per phase for all
// adversaries together (adv_fwd_kernel + adv_bwd_kernel, adv_dw_kernel).  Exact-f32 MFMA (v_mfma_f32_16x16x4_f32): the work is small
// (< 0.5 % of the step's FLOPs) and bound by launch latency and L2 -> CU operand traffic, not by the matrix cores.
#include "common.h"

#ifndef MMVAE_ADV_STAMPS
#define MMVAE_ADV_STAMPS 0  // diagnostics build (tools/adv_stamps.py): wall-clock stamps per workgroup and phase
#endif
#if MMVAE_ADV_STAMPS
__device__ long long g_adv_trace[3][1024 * 8];
#define ADV_STAMP(k, bid, i)                                                                         \
    do {                                                                                             \
        if (threadIdx.x == 0 && (bid) < 1024) g_adv_trace[k][(bid)*8 + (i)] = wall_clock64();       \
    } while (0)
extern "C" int mmvae_debug_adv_trace(int which, long long* out, int n_blocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_adv_trace), sizeof(long long) * 8 * n_blocks,
                               sizeof(long long) * 8 * 1024 * which) == hipSuccess ? 0 : 1;
}
#else
#define ADV_STAMP(k, bid, i) \
    do {                     \
    } while (0)
#endif

namespace {

constexpr int AR = 16;       // cells per workgroup
constexpr int AW = 8;        // waves per workgroup
constexpr int AT = AW * 64;  // threads

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 sel4(bool ok, f32x4 v) {
    return f32x4{ok ? v[0] : 0.f, ok ? v[1] : 0.f, ok ? v[2] : 0.f, ok ? v[3] : 0.f};
}

// M[r][k .. k + 3] of a row-major matrix [R, K] (k % 4 == 0), zeros outside.  FAST (K % 4 == 0, 16-byte aligned base):
// ONE unconditional 16-byte load from a clamped address + a select -- no branch, so that a batch of these loads is in
// flight together (these kernels are bound by load latency).  Otherwise guarded scalar loads (odd toy shapes).
template <bool FAST>
__device__ __forceinline__ f32x4 ldm4(const float* __restrict__ M, int r, int R, int K, int k) {
    if (FAST) {
        const int rr = r < R ? r : R - 1, kk = k < K ? k : K - 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(M + (int64_t)rr * K + kk);
        return sel4(r < R && k < K, v);
    }
    f32x4 v = zero4();
    if (r < R) {
        const float* row = M + (int64_t)r * K;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j < K) v[j] = row[k + j];
    }
    return v;
}
// M[r][c] or 0 outside [0, R) x [0, C)
template <bool FAST>
__device__ __forceinline__ float ldm1(const float* __restrict__ M, int r, int R, int C, int c) {
    if (FAST) {
        const int rr = r < R ? r : R - 1, cc = c < C ? c : C - 1;
        const float v = M[(int64_t)rr * C + cc];
        return (r < R && c < C) ? v : 0.f;
    }
    return (r < R && c < C) ? M[(int64_t)r * C + c] : 0.f;
}

__host__ __device__ inline int pad16(int v) { return (v + 15) & ~15; }
__host__ __device__ inline int pad4(int v) { return (v + 3) & ~3; }
__host__ __device__ inline int head_tiles(int classes) { return (pad4(classes) + 15) / 16; }

// LDS plan of the pass kernels (float offsets).  Activations [16][stride] with 4 floats of row padding (conflict-free
// 16-byte fragment reads), two gradient buffers, one weight tile per wave -- reused as that wave's merge record.
struct AdvLds {
    int act_off[MMVAE_ADV_MAX_LAYERS + 1], act_stride[MMVAE_ADV_MAX_LAYERS + 1];
    int da_off, dz_off, d_stride;
    int wt_off, wt_stride;  // per wave: 16 rows x wt_stride
    int misc_off;
    int total;
};
__host__ __device__ inline void adv_lds_layout(AdvLds& o, const int32_t* width, int L, int net, int H, int splits) {
    int p = 0, maxw = 16;
    for (int l = 0; l <= L; ++l) {
        int wp = pad16(width[l]);
        if (l == L && wp < 16 * net) wp = 16 * net;
        o.act_stride[l] = wp + 4;
        o.act_off[l] = p;
        p += AR * (wp + 4);
        if (l >= 1 && wp > maxw) maxw = wp;
    }
    for (int l = L + 1; l <= MMVAE_ADV_MAX_LAYERS; ++l) o.act_off[l] = o.act_stride[l] = 0;
    o.d_stride = maxw + 4;
    o.da_off = p;
    p += AR * o.d_stride;
    o.dz_off = p;
    p += AR * o.d_stride;
    o.wt_stride = 16 * net + 4;
    o.wt_off = p;
    p += AW * 16 * o.wt_stride > AW * AR * 64 ? AW * 16 * o.wt_stride : AW * AR * 64;  // (backward: [wave][16][64] sums)
    o.misc_off = p;
    p += H * AR + splits * H * AR + H * AR + 64;  // lse, merge weights, labels
    o.total = p;
}

__device__ __forceinline__ void load_job(mmvae_adv_job* dst_lds, const mmvae_adv_job* __restrict__ src) {
    const int* s = reinterpret_cast<const int*>(src);
    int* d = reinterpret_cast<int*>(dst_lds);
    for (int i = threadIdx.x; i < (int)(sizeof(mmvae_adv_job) / 4); i += AT) d[i] = s[i];
    __syncthreads();
}

// rows row0 .. row0 + 15 of a row-major [B, K] matrix (leading dimension ld) -> LDS [16][stride], zero padded to Kp columns
template <bool FAST>
__device__ __forceinline__ void rows_to_lds(const float* __restrict__ src, int64_t ld, int row0, int B, int K, int Kp,
                                            float* dst, int stride) {
    if (FAST) {  // K % 4 == 0, ld % 4 == 0, aligned
        const int q4 = Kp / 4;
        for (int e = threadIdx.x; e < AR * q4; e += AT) {
            const int r = e / q4, k = 4 * (e - r * q4), row = row0 + r;
            const int rr = row < B ? row : B - 1, kk = k < K ? k : K - 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)rr * ld + kk);
            *reinterpret_cast<f32x4*>(dst + r * stride + k) = sel4(row < B && k < K, v);
        }
    } else {
        for (int e = threadIdx.x; e < AR * Kp; e += AT) {
            const int r = e / Kp, k = e - r * Kp, row = row0 + r;
            dst[r * stride + k] = (row < B && k < K) ? src[(int64_t)row * ld + k] : 0.f;
        }
    }
}

// Forward half of a pass: encoder layers (every class split repeats them: cheaper than a launch), then this split's share
// of the heads' class tiles -> logits + the partial (max, sum, softmax . W) of every head.  blockIdx.y = adversary,
// blockIdx.x = cell tile * splits + class split.
template <int NET, bool FAST>
__global__ __launch_bounds__(AT) void adv_fwd_kernel(const mmvae_adv_job* __restrict__ jobs, int B, int splits) {
    extern __shared__ float lds[];
    __shared__ mmvae_adv_job J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    load_job(&J, jobs + blockIdx.y);
    const int L = J.n_layers, H = J.H, NE = J.width[L], Ct = J.Ct;
    const int sbid = blockIdx.y * gridDim.x + blockIdx.x;
    (void)sbid;
    ADV_STAMP(0, sbid, 0);
    __shared__ AdvLds Y;  // (indexed by layer: as a local it would live in scratch memory)
    if (tid == 0) adv_lds_layout(Y, J.width, L, NET, H, splits);  // (written in place: no private copy)
    __syncthreads();
    const int rt = blockIdx.x / splits, sp = blockIdx.x - rt * splits;
    const int row0 = rt * AR;
    constexpr int NEP = 16 * NET;
    constexpr int PST = 32 + AR * NEP;  // floats of one (cell tile, split, head) partial: max[16], sum[16], product[16][NEP]

    rows_to_lds<FAST>(J.x, J.ldx, row0, B, J.width[0], Y.act_stride[0] - 4, lds + Y.act_off[0], Y.act_stride[0]);
    __syncthreads();
    // ---- encoder layers: out = dropout(relu(in . W^T + b)); tile t of 16 outputs per wave
    for (int l = 0; l < L; ++l) {
        const int K = J.width[l], N = J.width[l + 1], Kp = pad16(K);
        const float* in = lds + Y.act_off[l];
        const int ist = Y.act_stride[l];
        float* out = lds + Y.act_off[l + 1];
        const int ost = Y.act_stride[l + 1], Npad = ost - 4;
        const float* __restrict__ W = J.W[l];
        const float* __restrict__ bias = J.b[l];
        const uint8_t* __restrict__ mask = J.mask[l];
        float* __restrict__ act_out = (sp == 0) ? J.act[l] : nullptr;
        const float scale = mask ? 1.f / (1.f - J.p_drop[l]) : 1.f;
        const bool relu = J.relu[l] != 0;
        for (int t = wave; t * 16 < Npad; t += AW) {
            const int n = 16 * t + l15;
            f32x4 acc0 = zero4(), acc1 = zero4();
            // the weight fragments of 16 k-steps are requested together, then multiplied (two accumulator chains)
            for (int kb = 0; kb < Kp; kb += 256) {
                f32x4 w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = ldm4<FAST>(W, n, N, K, kb + 16 * u + 4 * g);
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    if (kb + 16 * u < Kp) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(in + l15 * ist + kb + 16 * u + 4 * g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc0 = mfma4(a[j], w[u][j], acc0);
                    }
                    if (kb + 16 * u + 16 < Kp) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(in + l15 * ist + kb + 16 * u + 16 + 4 * g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc1 = mfma4(a[j], w[u + 1][j], acc1);
                    }
                }
            }
            const f32x4 acc = acc0 + acc1;
            const float bv = (n < N && bias) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // accumulator register r: cell 4g + r, output n
                const int cell = 4 * g + r, row = row0 + cell;
                float v = acc[r] + bv;
                if (relu) v = fmaxf(v, 0.f);
                const bool live = n < N && row < B;
                if (mask && live) v = mask[(int64_t)row * N + n] ? v * scale : 0.f;
                if (!live) v = 0.f;
                out[cell * ost + n] = v;
                if (act_out && live) act_out[(int64_t)row * N + n] = v;
            }
        }
        __syncthreads();
    }
    ADV_STAMP(0, sbid, 1);

    // ---- heads, flash-style.  Product 1 (transposed): S^T[class][cell] = W_tile . e^T, so that a lane holds 4 classes of
    // ONE cell -- exactly the A fragment of product 2, acc[cell][k] += P[cell][class] . W_tile[class][k].
    float ef[NET][4];  // e[cell = l15][16 i + 4 g + j]: the B fragments of product 1
    {
        const float* eL = lds + Y.act_off[L];
        const int est = Y.act_stride[L];
#pragma unroll
        for (int i = 0; i < NET; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(eL + l15 * est + 16 * i + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) ef[i][j] = v[j];
        }
    }
    const int WS = Y.wt_stride;
    float* wt = lds + Y.wt_off + wave * 16 * WS;
    const float* __restrict__ Wh = J.Wh;
    const float* __restrict__ bh = J.bh;
    float* part = J.partials + (int64_t)(rt * splits + sp) * H * PST;
    float* trash = J.partials + (int64_t)((B + AR - 1) / AR) * splits * H * PST;  // 256 floats nobody reads
#if MMVAE_ADV_STAMPS
    long long qsum[4] = {0, 0, 0, 0};
#endif
    for (int h = 0; h < H; ++h) {
        const int C = J.classes[h], col = J.col[h], C4 = pad4(C);
        const int lo = J.seg_lo[sp][h], hi = J.seg_hi[sp][h];  // this split's tiles of head h (mmvae_adv_pass_plan)
        float m_run = -INFINITY, s_run = 0.f;
        f32x4 acc[NET];
#pragma unroll
        for (int t = 0; t < NET; ++t) acc[t] = zero4();
        // weight tile tt as the A fragments of product 1 (lane: class row 16 tt + l15) + the biases of this lane's 4 classes
        // (a tile index past the range is clamped: a redundant load instead of a branch around the prefetch)
        auto load_tile = [&](int tt, f32x4(&w)[NET], f32x4& bv) {
            tt = tt < hi ? tt : hi - 1;
            const int cls = 16 * tt + l15;
#pragma unroll
            for (int i = 0; i < NET; ++i) w[i] = ldm4<FAST>(Wh + (int64_t)col * NE, cls, C, NE, 16 * i + 4 * g);
            const int cb = 16 * tt + 4 * g;
            if (FAST) {  // col, cb multiples of 4; col + C4 <= Ct: the 16-byte group is inside the vector
                // (no bias: the load goes to the weights instead and is dropped -- a select, not a branch)
                const f32x4 v = *reinterpret_cast<const f32x4*>((bh ? bh : Wh) + col + (cb < C4 ? cb : C4 - 4));
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = (bh && cb + r < C) ? v[r] : 0.f;
            } else {
                bv = zero4();
                if (bh)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (cb + r < C) bv[r] = bh[col + cb + r];
            }
        };
        // One class tile: logits, running softmax statistics, running product.  `w`, `bv`: the tile's register set,
        // requested two tiles ago; it is re-loaded (tile tt + 2 AW) as soon as product 1 has read it.  The loop below is
        // unrolled by two over two FIXED register sets: rotating one set into the other would make every iteration wait
        // for the loads it has just issued.
        auto tile = [&](int tt, f32x4(&w)[NET], f32x4& bv) {
#if MMVAE_ADV_STAMPS
            const long long q0 = wall_clock64();
#endif
            const int c0 = 16 * tt;
            const f32x4 bias4 = bv;
#pragma unroll
            for (int i = 0; i < NET; ++i) *reinterpret_cast<f32x4*>(wt + l15 * WS + 16 * i + 4 * g) = w[i];
            f32x4 s = zero4(), s2 = zero4();
#pragma unroll
            for (int i = 0; i < NET; ++i) {
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    s = mfma4(w[i][j], ef[i][j], s);
                    s2 = mfma4(w[i][j + 1], ef[i][j + 1], s2);
                }
            }
            load_tile(tt + 2 * AW, w, bv);
            s += s2;
            // s[r]: cell l15, class c0 + 4g + r
            const int cb = c0 + 4 * g;
            float lg[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) lg[r] = (cb + r < C) ? s[r] + bias4[r] : -INFINITY;
            const int row = row0 + l15;
            if (FAST) {
                // unconditional store: lanes outside the matrix write to the scratch words behind the partials
                float* lp = (row < B && cb < C4) ? J.logits + (int64_t)row * Ct + col + cb : trash + 4 * lane;
                *reinterpret_cast<f32x4*>(lp) = f32x4{lg[0], lg[1], lg[2], lg[3]};
            } else if (row < B && cb < C4) {
                float* lp = J.logits + (int64_t)row * Ct + col + cb;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + cb + r < Ct) lp[r] = lg[r];
            }
#if MMVAE_ADV_STAMPS
            const long long q1 = wall_clock64() + (lg[0] == 12345.678f ? 1 : 0);  // (behind product 1's results)
#endif
            float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);  // finite: every tile holds at least one real class
            const float alpha = expf(m_run - m_new);
            float p[4], rs = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = (cb + r < C) ? expf(lg[r] - m_new) : 0.f;
                rs += p[r];
            }
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            s_run = s_run * alpha + rs;
            m_run = m_new;
            // the accumulators hold cells 4g + r: their rescale factors live in the lanes whose l15 is that cell
            float ar[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * g + r, 64);
#pragma unroll
            for (int t = 0; t < NET; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] *= ar[r];
#if MMVAE_ADV_STAMPS
            const long long q2 = wall_clock64() + (acc[0][0] == 12345.678f ? 1 : 0);  // (behind the rescale)
#endif
            // (the tile was written and is read by this wave only: LDS operations of a wave execute in order)
            __builtin_amdgcn_wave_barrier();
            float wb[4][NET];  // (all fragments requested, then multiplied: one LDS latency per tile, not sixteen)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NET; ++t) wb[j][t] = wt[(4 * g + j) * WS + 16 * t + l15];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NET; ++t) acc[t] = mfma4(p[j], wb[j][t], acc[t]);
            __builtin_amdgcn_wave_barrier();
#if MMVAE_ADV_STAMPS
            const long long q3 = wall_clock64() + (acc[0][0] == 12345.678f ? 1 : 0);  // (behind product 2's results)
            qsum[0] += q1 - q0, qsum[1] += q2 - q1, qsum[2] += q3 - q2, qsum[3] += 1;
#endif
        };
        if (lo < hi) {
            f32x4 w0[NET], w1[NET], b0, b1;
            load_tile(lo + wave, w0, b0);
            load_tile(lo + wave + AW, w1, b1);
            for (int tt = lo + wave; tt < hi; tt += 2 * AW) {
                tile(tt, w0, b0);
                if (tt + AW < hi) tile(tt + AW, w1, b1);
            }
        }
        if (lo >= hi) {  // no tile of this head in this split: an empty partial (weight 0 in the merge)
            for (int e = tid; e < PST; e += AT) part[h * PST + e] = e < AR ? -INFINITY : 0.f;
            continue;
        }
        // this wave's record (over its weight tile): max[16], sum[16], product[16][NEP]
        if (g == 0) {
            wt[l15] = m_run;
            wt[16 + l15] = s_run;
        }
#pragma unroll
        for (int t = 0; t < NET; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) wt[32 + (4 * g + r) * NEP + 16 * t + l15] = acc[t][r];
        __syncthreads();
        // merge the waves' records in wave order -> this split's partial of head h
        {
            const float* R = lds + Y.wt_off;
            const int RS = 16 * WS;
            for (int e = tid; e < AR * NEP; e += AT) {
                const int cell = e / NEP;
                float M = -INFINITY;
#pragma unroll
                for (int w = 0; w < AW; ++w) M = fmaxf(M, R[w * RS + cell]);
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < AW; ++w) {
                    const float mw = R[w * RS + cell];
                    if (mw > -INFINITY) v += R[w * RS + 32 + e] * expf(mw - M);
                }
                part[h * PST + 32 + e] = v;
            }
            if (tid < AR) {
                float M = -INFINITY, S = 0.f;
#pragma unroll
                for (int w = 0; w < AW; ++w) M = fmaxf(M, R[w * RS + tid]);
#pragma unroll
                for (int w = 0; w < AW; ++w) {
                    const float mw = R[w * RS + tid];
                    if (mw > -INFINITY) S += R[w * RS + 16 + tid] * expf(mw - M);
                }
                part[h * PST + tid] = M;
                part[h * PST + 16 + tid] = S;
            }
        }
        __syncthreads();
    }
    ADV_STAMP(0, sbid, 2);
#if MMVAE_ADV_STAMPS
    if (tid == 0 && sbid < 1024)
        for (int i = 0; i < 4; ++i) g_adv_trace[0][sbid * 8 + 3 + i] = qsum[i];
#endif
}

// Backward half of a pass (a launch of its own: the kernel boundary is what makes the splits' partials visible across the
// chip's L2 slices -- a ticket inside one launch needs an L2 write-back + invalidate per workgroup, ~35 us here):
// merge the splits' partials in split order, log-sum-exp and per-cell losses, d(loss)/d(encoded), then back through
// the encoder layers.  blockIdx.x = cell tile, blockIdx.y = adversary.
template <int NET, bool FAST>
__global__ __launch_bounds__(AT) void adv_bwd_kernel(const mmvae_adv_job* __restrict__ jobs, int B, int splits) {
    extern __shared__ float lds[];
    __shared__ mmvae_adv_job J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    load_job(&J, jobs + blockIdx.y);
    const int L = J.n_layers, H = J.H, NE = J.width[L], Ct = J.Ct;
    const int sbid = blockIdx.y * gridDim.x + blockIdx.x;
    (void)sbid;
    ADV_STAMP(1, sbid, 0);
    __shared__ AdvLds Y;
    if (tid == 0) adv_lds_layout(Y, J.width, L, NET, H, splits);  // (written in place: no private copy)
    __syncthreads();
    const int rt = blockIdx.x, row0 = rt * AR;
    constexpr int NEP = 16 * NET;
    constexpr int PST = 32 + AR * NEP;
    const float* __restrict__ P = J.partials + (int64_t)rt * splits * H * PST;
    const float* __restrict__ Wh = J.Wh;
    // the layer outputs of these cells (written by the forward launch) -> LDS: the ReLU / dropout pattern of the backward
    for (int l = 1; l <= L; ++l)
        rows_to_lds<FAST>(J.act[l - 1], J.width[l], row0, B, J.width[l], Y.act_stride[l] - 4, lds + Y.act_off[l],
                          Y.act_stride[l]);
    float* lseb = lds + Y.misc_off;  // [H][16]
    float* wgt = lseb + H * AR;      // [splits][H][16]: exp(max_s - max) / sum
    if (tid < AR * H) {
        const int h = tid / AR, cell = tid - h * AR;
        float ms[8], ss[8];  // (splits <= 8)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int sc = s < splits ? s : splits - 1;
            ms[s] = P[(int64_t)(sc * H + h) * PST + cell];
            ss[s] = P[(int64_t)(sc * H + h) * PST + 16 + cell];
            if (s >= splits) ms[s] = -INFINITY;
        }
        const int row = row0 + cell;
        const int64_t y = J.labels[(int64_t)h * B + (row < B ? row : B - 1)];
        const bool valid = row < B && y >= 0 && y < J.classes[h];
        const float ly = J.logits[(int64_t)(row < B ? row : B - 1) * Ct + J.col[h] + (valid ? y : 0)];
        float M = -INFINITY, S = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) M = fmaxf(M, ms[s]);
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (ms[s] > -INFINITY) S += ss[s] * expf(ms[s] - M);
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (s < splits) wgt[(s * H + h) * AR + cell] = (ms[s] > -INFINITY) ? expf(ms[s] - M) / S : 0.f;
        const float lse = M + logf(S);
        lseb[h * AR + cell] = lse;
        reinterpret_cast<int*>(wgt + splits * H * AR)[h * AR + cell] = valid ? (int)y : -1;
        if (row < B) {
            J.lse[(int64_t)h * B + row] = lse;
            J.loss_rows[(int64_t)h * B + row] = valid ? lse - ly : 0.f;
        }
    }
    __syncthreads();
    ADV_STAMP(1, sbid, 1);
    // d(loss)/d(encoded) = gscale * sum over heads (softmax . W - W[label])
    float* dA = lds + Y.da_off;
    float* dZ = lds + Y.dz_off;
    const int ds = Y.d_stride;
    int* ylab = reinterpret_cast<int*>(wgt + splits * H * AR);  // [H][16]: label or -1 (written above)
    for (int e = tid; e < AR * NEP; e += AT) {
        const int cell = e / NEP, k = e - cell * NEP, row = row0 + cell;
        const bool live = row < B && k < NE;
        const int kc = k < NE ? k : NE - 1;
        // every load of this element is requested before the first is used: H label rows of the head weights and
        // H x splits partial products (indices past H / splits are clamped and weighted 0)
        float v = 0.f, vy = 0.f;
        for (int hb = 0; hb < H; hb += 4) {  // (four heads per batch: 36 loads in flight)
            float wy[4], pv[4][8];
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                const int h = hb + hh, hc = h < H ? h : H - 1;
                const int y = ylab[hc * AR + cell];
                wy[hh] = Wh[(int64_t)(J.col[hc] + (y >= 0 ? y : 0)) * NE + kc];
                if (h >= H || y < 0) wy[hh] = 0.f;
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    pv[hh][s] = P[(int64_t)((s < splits ? s : splits - 1) * H + hc) * PST + 32 + e];
            }
#pragma unroll
            for (int hh = 0; hh < 4; ++hh)
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (hb + hh < H && s < splits) {
                        const float w = wgt[(s * H + hb + hh) * AR + cell];
                        if (w != 0.f) v += pv[hh][s] * w;
                    }
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) vy += wy[hh];
        }
        v -= vy;
        dA[cell * ds + k] = live ? J.gscale * v : 0.f;
    }
    __syncthreads();
    ADV_STAMP(1, sbid, 2);
    // backward through the encoder layers
    for (int l = L - 1; l >= 0; --l) {
        const int N = J.width[l + 1], K = J.width[l], Np = pad16(N), Kp = pad16(K);
        const float* a1 = lds + Y.act_off[l + 1];
        const int ast = Y.act_stride[l + 1];
        const uint8_t* __restrict__ mask = J.mask[l];
        const float scale = mask ? 1.f / (1.f - J.p_drop[l]) : 1.f;
        const bool relu = J.relu[l] != 0;
        float* __restrict__ dz_out = J.dz[l];
        for (int e = tid; e < AR * Np; e += AT) {
            const int cell = e / Np, n = e - cell * Np, row = row0 + cell;
            float v = 0.f;
            if (n < N && row < B) {
                // the stored activation is the one after dropout: > 0 <=> kept and past the ReLU
                float f = 1.f;
                if (mask) f = mask[(int64_t)row * N + n] ? scale : 0.f;
                if (relu && !(a1[cell * ast + n] > 0.f)) f = 0.f;
                v = dA[cell * ds + n] * f;
                if (dz_out) dz_out[(int64_t)row * N + n] = v;
            }
            dZ[cell * ds + n] = v;
        }
        __syncthreads();
        if (l > 0 || J.gx) {
            const float* __restrict__ W = J.W[l];
            const int nblk = (K + 63) / 64, kh = Np >= 32 ? 2 : 1;
            if (FAST && nblk * kh <= AW) {
                // dact[16, K] = dZ[16, N] . W[N, K].  A lane reads W[o][64 b + 4 l15 .. + 3] (a wave: 256 contiguous bytes
                // per row o) and uses value c as the B fragment of column tile c = columns {64 b + 4 i + c}: 4 loads feed 16
                // MFMAs.  One wave per 64-column block and half of the k range (the halves are summed through LDS).
                float* red = lds + Y.wt_off;  // [wave][16 cells][64]: the weight tiles' space is free in this kernel
                const int khb = kh == 2 ? Np / 32 * 16 : Np;  // first half: [0, khb), second: [khb, Np)
                if (wave < nblk * kh) {
                    const int blk = wave / kh, half = wave - blk * kh;
                    const int kb0 = half == 0 ? 0 : khb, kb1 = half == 0 ? khb : Np;
                    const int cbase = 64 * blk + 4 * l15;
                    f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
                    for (int kb = kb0; kb < kb1; kb += 64) {
                        f32x4 b[4][4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int o = kb + 16 * u + 4 * g + j;
                                b[u][j] = ldm4<true>(W, o < kb1 ? o : N, N, K, cbase);  // (row N: zeros)
                            }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int k0 = kb + 16 * u;
                            const f32x4 a = *reinterpret_cast<const f32x4*>(dZ + l15 * ds + (k0 < Np ? k0 : 0) + 4 * g);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int c = 0; c < 4; ++c) acc[c] = mfma4(a[j], b[u][j][c], acc[c]);
                        }
                    }
                    // accumulator c, register r: cell 4g + r, column 64 blk + 4 l15 + c
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<f32x4*>(red + (wave * AR + 4 * g + r) * 64 + 4 * l15) =
                            f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                }
                __syncthreads();
                for (int e = tid; e < nblk * AR * 64; e += AT) {  // the two k halves, first + second
                    const int bq = e / (AR * 64), rem = e - bq * AR * 64, cell = rem / 64, c = rem - cell * 64;
                    float v = red[(bq * kh * AR + cell) * 64 + c];
                    if (kh == 2) v += red[((bq * kh + 1) * AR + cell) * 64 + c];
                    const int n = 64 * bq + c, row = row0 + cell;
                    if (l > 0) {
                        if (n < Kp) dA[cell * ds + n] = v;
                    } else if (row < B && n < K) {
                        J.gx[(int64_t)row * K + n] = -v;  // gradient reversal, alpha = 1
                    }
                }
                __syncthreads();
            } else {
                for (int t = wave; 16 * t < Kp; t += AW) {
                    const int n = 16 * t + l15;  // input feature
                    f32x4 acc = zero4();
                    for (int k0 = 0; k0 < Np; k0 += 16) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(dZ + l15 * ds + k0 + 4 * g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc = mfma4(a[j], ldm1<false>(W, k0 + 4 * g + j, N, K, n), acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cell = 4 * g + r, row = row0 + cell;
                        if (l > 0)
                            dA[cell * ds + n] = acc[r];
                        else if (row < B && n < K)
                            J.gx[(int64_t)row * K + n] = -acc[r];  // gradient reversal, alpha = 1
                    }
                }
                __syncthreads();
            }
        }
    }
    ADV_STAMP(1, sbid, 3);
}

// ------------------------------------------------------------------------------------------------ weight gradients
constexpr int DW_TM = 64, DW_TN = 64, DW_KC = 128;
constexpr int DW_LS = 80;          // LDS row stride (floats): 4 consecutive k rows land in different bank quarters
constexpr int DW_RS = DW_KC / 64;  // chunk rows staged per thread

struct DwHead {  // the head a thread's 4 classes belong to (heads job)
    int h, colh, C;
};

template <bool FAST>  // every job: M, N, leading dimensions multiples of 4, 16-byte aligned operands
__global__ __launch_bounds__(AT) void adv_dw_kernel(const mmvae_adv_dw_job* __restrict__ jobs, int n_jobs,
                                                    const mmvae_adv_opt* __restrict__ opts, int n_opts,
                                                    float* __restrict__ partials, unsigned* __restrict__ ticket,
                                                    const mmvae_adv_job* __restrict__ adv, int n_adv) {
    __shared__ float As[DW_KC * DW_LS];
    __shared__ float Bs[DW_KC * DW_LS];
    __shared__ float red[AW + 1];
    __shared__ double dred[AT];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    int ji = 0;
    while (ji + 1 < n_jobs && (int)blockIdx.x >= jobs[ji + 1].first_block) ++ji;
    const mmvae_adv_dw_job& J = jobs[ji];
    const int M = J.M, N = J.N, B = J.B;
    const int tiles_n = (N + DW_TN - 1) / DW_TN;
    const int tb = blockIdx.x - J.first_block, tm = tb / tiles_n, tn = tb - tm * tiles_n;
    const int m0 = tm * DW_TM, n0 = tn * DW_TN;
    const float* __restrict__ dz = J.dz;
    const float* __restrict__ inp = J.inp;
    const int64_t ldz = J.ld_dz, ldi = J.ld_inp;
    const bool heads = J.lse != nullptr;
    const float gscale = J.gscale;
    // staging role: rows sr, sr + 64 of the chunk; columns 4 sq .. + 3 and 32 + 4 sq .. + 3 of both tiles
    const int sr = tid >> 3, sq = tid & 7;
    const int amq[2] = {m0 + 4 * sq, m0 + 32 + 4 * sq};  // first of this thread's 4 output features (classes), twice
    const int bnq[2] = {n0 + 4 * sq, n0 + 32 + 4 * sq};
    DwHead hd[2] = {{-1, 0, 0}, {-1, 0, 0}};
    if (heads)
        for (int h = 0; h < J.H; ++h)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (amq[c] >= J.col[h] && amq[c] < J.col[h] + pad4(J.classes[h])) hd[c] = DwHead{h, J.col[h], J.classes[h]};

    // Two chunks are requested ahead of the one being multiplied (two FIXED register sets, the loop unrolled by two:
    // rotating the sets would make each chunk wait for the newest loads): the kernel is bound by load latency.
    struct Regs {
        f32x4 a[DW_RS][2], b[DW_RS][2];
        float lse[DW_RS][2];
        int y[DW_RS][2];
    };
    auto fetch = [&](int k0, Regs& X) {
#pragma unroll
        for (int q = 0; q < DW_RS; ++q) {
            const int row = k0 + sr + 64 * q;
            const bool in = row < B;
            const int rr = in ? row : B - 1;  // clamped: unconditional loads, selected below (no branch around a load)
            const float* ap = dz + (int64_t)rr * ldz;
            const float* bp = inp + (int64_t)rr * ldi;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (FAST) {
                    const int am = amq[c], bn = bnq[c];
                    X.a[q][c] = sel4(in && am < M, *reinterpret_cast<const f32x4*>(ap + (am < M ? am : M - 4)));
                    X.b[q][c] = sel4(in && bn < N, *reinterpret_cast<const f32x4*>(bp + (bn < N ? bn : N - 4)));
                } else {
                    X.a[q][c] = X.b[q][c] = zero4();
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (in && amq[c] + j < M) X.a[q][c][j] = ap[amq[c] + j];
                        if (in && bnq[c] + j < N) X.b[q][c][j] = bp[bnq[c] + j];
                    }
                }
                X.lse[q][c] = 0.f;
                X.y[q][c] = -1;
                if (heads) {
                    const int hh = hd[c].h >= 0 ? hd[c].h : 0;
                    const float lv = J.lse[(int64_t)hh * B + rr];
                    const int64_t y = J.labels[(int64_t)hh * B + rr];
                    X.lse[q][c] = lv;
                    X.y[q][c] = (in && hd[c].h >= 0 && y >= 0 && y < hd[c].C) ? (int)y : -1;
                }
            }
        }
    };
    f32x4 asum[2] = {zero4(), zero4()};  // column sums of this thread's rows (the bias gradient), chunks in row order
    auto stage = [&](int k0, const Regs& X) {
#pragma unroll
        for (int q = 0; q < DW_RS; ++q) {
            const int r = sr + 64 * q, row = k0 + r;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x4 a = X.a[q][c];
                if (heads) {
                    // dlogits = gscale * (softmax - onehot); padding columns hold -inf -> 0
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cls = amq[c] + j - hd[c].colh;
                        const float d = gscale * (expf(X.a[q][c][j] - X.lse[q][c]) - (cls == X.y[q][c] ? 1.f : 0.f));
                        a[j] = (hd[c].h >= 0 && row < B && cls < hd[c].C) ? d : 0.f;
                    }
                }
                asum[c] += a;
                *reinterpret_cast<f32x4*>(As + r * DW_LS + 32 * c + 4 * sq) = a;
                *reinterpret_cast<f32x4*>(Bs + r * DW_LS + 32 * c + 4 * sq) = X.b[q][c];
            }
        }
    };

    const int tr = wave >> 2, tc = wave & 3;  // this wave: output rows 32 tr .. + 31 (two MFMA tiles), columns 16 tc .. + 15
    ADV_STAMP(2, blockIdx.x, 0);
    f32x4 acc0 = zero4(), acc1 = zero4();
    Regs R0, R1;
#if MMVAE_ADV_STAMPS
    long long csum[4] = {0, 0, 0, 0};
#endif
    auto chunk = [&](int k0, Regs& X) {
#if MMVAE_ADV_STAMPS
        const long long c0 = wall_clock64();
#endif
        stage(k0, X);
#if MMVAE_ADV_STAMPS
        const long long c1 = wall_clock64();
#endif
        __syncthreads();
#if MMVAE_ADV_STAMPS
        const long long c2 = wall_clock64();
#endif
        fetch(k0 + 2 * DW_KC, X);  // (rows past B: clamped loads, dropped)
        // fragments of 8 k-steps are requested together, a block ahead of the MFMAs that use them (the compiler's own
        // schedule waited for every read before its two MFMAs: one LDS latency per k-step)
        float fa[2][8][2], fb[2][8];
        auto lds_block = [&](int kb, int set) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = 4 * (kb + u) + g;
                fb[set][u] = Bs[k * DW_LS + 16 * tc + l15];
                fa[set][u][0] = As[k * DW_LS + 32 * tr + l15];
                fa[set][u][1] = As[k * DW_LS + 32 * tr + 16 + l15];
            }
        };
        lds_block(0, 0);
#pragma unroll
        for (int blk = 0; blk < DW_KC / 32; ++blk) {
            if (blk + 1 < DW_KC / 32) lds_block(8 * (blk + 1), (blk + 1) & 1);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = mfma4(fa[blk & 1][u][0], fb[blk & 1][u], acc0);
                acc1 = mfma4(fa[blk & 1][u][1], fb[blk & 1][u], acc1);
            }
        }
#if MMVAE_ADV_STAMPS
        const long long c3 = wall_clock64() + (acc1[0] == 12345.678f ? 1 : 0);
#endif
        __syncthreads();
#if MMVAE_ADV_STAMPS
        const long long c4 = wall_clock64();
        csum[0] += c1 - c0, csum[1] += c2 - c1, csum[2] += c3 - c2, csum[3] += c4 - c3;
#endif
    };
    fetch(0, R0);
    fetch(DW_KC, R1);
    for (int k0 = 0; k0 < B; k0 += 2 * DW_KC) {  // (an odd chunk count stages one chunk of zeros)
        chunk(k0, R0);
        chunk(k0 + DW_KC, R1);
    }
    ADV_STAMP(2, blockIdx.x, 1);
#if MMVAE_ADV_STAMPS
    if (tid == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 4; ++i) g_adv_trace[2][blockIdx.x * 8 + 3 + i] = csum[i];
#endif
    // bias gradient: the 64 row-threads of each column group summed in thread order (As is free now)
    float bsum = 0.f;
    if (tn == 0) {
        *reinterpret_cast<f32x4*>(As + sr * DW_LS + 4 * sq) = asum[0];
        *reinterpret_cast<f32x4*>(As + sr * DW_LS + 32 + 4 * sq) = asum[1];
        __syncthreads();
        if (tid < DW_TM)
            for (int r = 0; r < 64; ++r) bsum += As[r * DW_LS + tid];
    }
    // accumulator t, register r: output feature m0 + 32 tr + 16 t + 4g + r, input feature n0 + 16 tc + l15
    float sq_sum = 0.f;
    {
        const int n = n0 + 16 * tc + l15;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * tr + 16 * t + 4 * g + r;
                const float v = t == 0 ? acc0[r] : acc1[r];
                if (m < M && n < N) {
                    J.gW[(int64_t)m * N + n] = v;
                    sq_sum += v * v;
                }
            }
    }
    if (tn == 0 && tid < DW_TM && m0 + tid < M && J.gb) {
        J.gb[m0 + tid] = bsum;
        sq_sum += bsum * bsum;
    }
    sq_sum = wave_sum(sq_sum);
    if (lane == 0) red[wave] = sq_sum;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < AW; ++w) t += red[w];
        partials[blockIdx.x] = t;
        __threadfence();
        const unsigned old = atomicAdd(ticket, 1u);
        s_last = old == gridDim.x - 1u;
        if (s_last) *ticket = 0u;
    }
    __syncthreads();
    ADV_STAMP(2, blockIdx.x, 2);
    if (blockIdx.x == 0 && adv) {
        // the pass's per-cell losses (written by the launch before this one) -> the logged sums of every adversary, in
        // order: fp64, fixed tree
        float adv_sum = 0.f;  // sum over the adversaries of total_scale * loss_total, in order
        float* adv_out = nullptr;
        for (int j = 0; j < n_adv; ++j) {
            const mmvae_adv_job* Jj = adv + j;
            const int Hj = Jj->H, Bj = Jj->B;
            float total = 0.f;
            for (int h = 0; h < Hj; ++h) {
                const float* rows = Jj->loss_rows + (int64_t)h * Bj;
                double s = 0.0;
                for (int i = tid; i < Bj; i += AT) s += (double)rows[i];
                __syncthreads();
                dred[tid] = s;
                __syncthreads();
                for (int st = AT / 2; st >= 1; st >>= 1) {
                    if (tid < st) dred[tid] += dred[tid + st];
                    __syncthreads();
                }
                const float r = (float)dred[0];
                if (tid == 0) Jj->loss_each[h] = r;
                total = h == 0 ? r : total + r;
            }
            if (tid == 0) Jj->loss_total[0] = total;
            if (Jj->total_loss) {
                adv_sum += Jj->total_scale * total;
                adv_out = Jj->total_loss;
            }
        }
        if (tid == 0 && adv_out) adv_out[0] = adv_sum;
        __syncthreads();
    }
    if (!s_last) return;
    __threadfence();
    // the launch's last workgroup: every optimiser's norm from the partials of its jobs, in block order
    const volatile float* pv = partials;
    for (int o = 0; o < n_opts; ++o) {
        double s = 0.0;
        for (int j = 0; j < n_jobs; ++j) {
            if (jobs[j].opt != o) continue;
            const int fb = jobs[j].first_block, nb = jobs[j].n_blocks;
            for (int b = fb + tid; b < fb + nb; b += AT) s += (double)pv[b];
        }
        dred[tid] = s;
        __syncthreads();
        for (int st = AT / 2; st >= 1; st >>= 1) {  // fixed tree (a serial sum by one thread cost ~10 us per optimiser)
            if (tid < st) dred[tid] += dred[tid + st];
            __syncthreads();
        }
        if (tid == 0) {
            const double t = dred[0];
            const mmvae_adv_opt op = opts[o];
            if (op.flags) adam_state_finish(op.state, t, op.flags, op.max_norm, op.grad_scale, op.beta1, op.beta2);
            if (op.norm_out) op.norm_out[0] = (float)(sqrt(t) * (double)fabsf(op.grad_scale));
        }
        __syncthreads();
    }
}

template <int NET, bool FAST>
int launch_pass(int n_jobs, const mmvae_adv_job* jobs_dev, int B, int splits, size_t lds_bytes, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(adv_fwd_kernel<NET, FAST>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(adv_bwd_kernel<NET, FAST>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048) != hipSuccess)
            return MMVAE_ERR_LAUNCH;
        attr = true;
    }
    const int n_rt = (B + AR - 1) / AR;
    MMVAE_LAUNCH((adv_fwd_kernel<NET, FAST>), dim3(n_rt * splits, n_jobs), dim3(AT), lds_bytes, stream, jobs_dev, B, splits);
    MMVAE_LAUNCH_CHECK();
    MMVAE_LAUNCH((adv_bwd_kernel<NET, FAST>), dim3(n_rt, n_jobs), dim3(AT), lds_bytes, stream, jobs_dev, B, splits);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

}  // namespace

// Which class tiles of which head a split works on.  A head costs a round of fixed overhead (first loads, the waves'
// merge) plus its tiles / 8 waves; the rounds of a split run one after the other.  Small heads go whole to the least
// loaded split, the tiles of big heads then fill the splits up to a common level.
static void plan_segments(mmvae_adv_job* job, int splits) {
    const int H = job->H;
    const double round_cost = 2.5;  // in units of one tile per wave (~4 us against ~1.7 us, measured at config C4)
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int tiles[MMVAE_ADV_MAX_HEADS];
    bool big[MMVAE_ADV_MAX_HEADS];
    for (int s = 0; s < 8; ++s)
        for (int h = 0; h < MMVAE_ADV_MAX_HEADS; ++h) job->seg_lo[s][h] = job->seg_hi[s][h] = 0;
    for (int h = 0; h < H; ++h) {
        tiles[h] = head_tiles(job->classes[h]);
        big[h] = tiles[h] >= 16 * splits;
    }
    for (int pass = 0; pass < H; ++pass) {  // small heads, largest first
        int h = -1;
        for (int q = 0; q < H; ++q)
            if (!big[q] && tiles[q] > 0 && (h < 0 || tiles[q] > tiles[h])) h = q;
        if (h < 0) break;
        int best = 0;
        for (int s = 1; s < splits; ++s)
            if (load[s] < load[best]) best = s;
        job->seg_lo[best][h] = 0;
        job->seg_hi[best][h] = tiles[h];
        load[best] += round_cost + (tiles[h] + 7) / 8;
        tiles[h] = 0;
    }
    for (int h = 0; h < H; ++h) {
        if (!big[h]) continue;
        // level the splits: every split takes a share; rounds(s) = level - load(s) - round_cost
        double sum = 0;
        for (int s = 0; s < splits; ++s) sum += load[s] + round_cost;
        const double level = (sum + tiles[h] / 8.0) / splits;
        int next = 0;
        for (int s = 0; s < splits; ++s) {
            double want = (level - load[s] - round_cost) * 8.0;
            int n = s == splits - 1 ? tiles[h] - next : (int)(want + 0.5);
            if (n < 0) n = 0;
            if (n > tiles[h] - next) n = tiles[h] - next;
            job->seg_lo[s][h] = next;
            job->seg_hi[s][h] = next + n;
            next += n;
            load[s] += n > 0 ? round_cost + (n + 7) / 8 : 0;
        }
    }
}

extern "C" int mmvae_adv_pass_plan(mmvae_adv_job* job, int splits, int* net, size_t* lds_bytes, int64_t* partial_floats,
                                   int* fast) {
    if (!job || splits < 1 || splits > 8) return MMVAE_ERR_ARG;
    const int L = job->n_layers, H = job->H;
    if (L < 1 || L > MMVAE_ADV_MAX_LAYERS || H < 1 || H > MMVAE_ADV_MAX_HEADS || job->B < 1) return MMVAE_ERR_ARG;
    for (int l = 0; l <= L; ++l)
        if (job->width[l] < 1 || job->width[l] > 1024) return MMVAE_ERR_ARG;
    const int ne = job->width[L];
    const int nt = (ne + 15) / 16;
    const int allowed[] = {1, 2, 4, 8};
    int pick = 0;
    for (int a : allowed)
        if (!pick && nt <= a) pick = a;
    if (!pick) return MMVAE_ERR_ARG;
    if (net && *net) {  // the launch's common tile count (the widest job's): this job is planned at it
        if (*net < pick || (*net != 1 && *net != 2 && *net != 4 && *net != 8)) return MMVAE_ERR_ARG;
        pick = *net;
    }
    int end = 0;
    for (int h = 0; h < H; ++h) {
        if (job->classes[h] < 1 || job->col[h] < end) return MMVAE_ERR_ARG;
        if (H > 1 && (job->col[h] & 3)) return MMVAE_ERR_ARG;
        end = job->col[h] + (H > 1 ? pad4(job->classes[h]) : job->classes[h]);
    }
    if (end > job->Ct) return MMVAE_ERR_ARG;
    AdvLds Y;
    adv_lds_layout(Y, job->width, L, pick, H, splits);
    const size_t bytes = (size_t)Y.total * 4;
    if (bytes > 160 * 1024 - 2048 - sizeof(mmvae_adv_job) - 64) return MMVAE_ERR_ARG;
    plan_segments(job, splits);
    if (net) *net = pick;
    if (lds_bytes) *lds_bytes = bytes;
    if (fast) {
        // every row the kernels read in 16-byte groups starts on a 16-byte boundary and holds a multiple of 4 floats
        // (pointers the caller has not filled in yet count as aligned)
        bool ok = (job->Ct & 3) == 0 && (job->ldx & 3) == 0 && aligned16(job->x) && aligned16(job->Wh) &&
                  aligned16(job->bh) && aligned16(job->logits);
        for (int l = 0; l <= L; ++l) ok = ok && (job->width[l] & 3) == 0;
        for (int l = 0; l < L; ++l) ok = ok && aligned16(job->W[l]) && aligned16(job->act[l]);
        for (int h = 0; h < H; ++h) ok = ok && (job->col[h] & 3) == 0;
        *fast = ok ? 1 : 0;
    }
    if (partial_floats) {
        const int64_t n_rt = (job->B + AR - 1) / AR;
        *partial_floats = n_rt * splits * H * (32 + AR * 16 * (int64_t)pick) + 256;
    }
    return MMVAE_OK;
}

extern "C" int mmvae_adv_pass_f32(int n_jobs, const mmvae_adv_job* jobs_dev, int B, int splits, int net, int fast,
                                  size_t lds_bytes, mmvae_stream_t stream) {
    if (n_jobs < 1 || n_jobs > 64 || !jobs_dev || B < 1 || splits < 1 || splits > 8) return MMVAE_ERR_ARG;
    if (lds_bytes > 160 * 1024 - 2048) return MMVAE_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
#define ADV_CASE(N)                                                                            \
    case N:                                                                                    \
        return fast ? launch_pass<N, true>(n_jobs, jobs_dev, B, splits, lds_bytes, st)         \
                    : launch_pass<N, false>(n_jobs, jobs_dev, B, splits, lds_bytes, st);
    switch (net) {
        ADV_CASE(1)
        ADV_CASE(2)
        ADV_CASE(4)
        ADV_CASE(8)
    }
#undef ADV_CASE
    return MMVAE_ERR_ARG;
}

extern "C" int mmvae_adv_dw_prepare(int n_jobs, mmvae_adv_dw_job* jobs, int* total_blocks, int* fast) {
    if (n_jobs < 1 || !jobs) return MMVAE_ERR_ARG;
    int next = 0;
    bool all_fast = true;
    for (int j = 0; j < n_jobs; ++j) {
        mmvae_adv_dw_job& J = jobs[j];
        if (J.M < 1 || J.N < 1 || J.B < 1 || !J.dz || !J.inp || !J.gW || J.opt < 0) return MMVAE_ERR_ARG;
        if (J.lse && (!J.labels || J.H < 1 || J.H > MMVAE_ADV_MAX_HEADS)) return MMVAE_ERR_ARG;
        if (J.lse && J.H > 1)
            for (int h = 0; h < J.H; ++h)
                if (J.col[h] & 3) return MMVAE_ERR_ARG;
        if (J.lse && J.H == 1 && (J.col[0] & 3)) return MMVAE_ERR_ARG;
        J.first_block = next;
        J.n_blocks = ((J.M + DW_TM - 1) / DW_TM) * ((J.N + DW_TN - 1) / DW_TN);
        next += J.n_blocks;
        all_fast = all_fast && (J.M & 3) == 0 && (J.N & 3) == 0 && (J.ld_dz & 3) == 0 && (J.ld_inp & 3) == 0 &&
                   aligned16(J.dz) && aligned16(J.inp);
    }
    if (total_blocks) *total_blocks = next;
    if (fast) *fast = all_fast ? 1 : 0;
    return MMVAE_OK;
}

extern "C" int mmvae_adv_dw_f32(int n_jobs, const mmvae_adv_dw_job* jobs_dev, int total_blocks, int n_opts,
                                const mmvae_adv_opt* opts_dev, float* partials, uint32_t* ticket, int n_adv,
                                const mmvae_adv_job* adv_jobs_dev, int fast, mmvae_stream_t stream) {
    if (n_jobs < 1 || !jobs_dev || total_blocks < 1 || n_opts < 0 || (n_opts && !opts_dev) || !partials || !ticket ||
        n_adv < 0 || (n_adv && !adv_jobs_dev))
        return MMVAE_ERR_ARG;
    if (fast)
        MMVAE_LAUNCH(adv_dw_kernel<true>, dim3(total_blocks), dim3(AT), 0, (hipStream_t)stream, jobs_dev, n_jobs, opts_dev,
                     n_opts, partials, ticket, n_adv ? adv_jobs_dev : nullptr, n_adv);
    else
        MMVAE_LAUNCH(adv_dw_kernel<false>, dim3(total_blocks), dim3(AT), 0, (hipStream_t)stream, jobs_dev, n_jobs, opts_dev,
                     n_opts, partials, ticket, n_adv ? adv_jobs_dev : nullptr, n_adv);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}
