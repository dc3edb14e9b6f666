// Row-owner adversary passes for gfx950 (include/mmvae_hip.h, "Row-owner adversary passes").
//
// Replaces (reference): one adversarial phase of CMMVAEModel.grf, models/cmmvae_model.py:59-101 -- Adversarial.encoder
// (modules/base/components.py:638-674: an FCBlock of Linear -> ReLU -> Dropout layers without BatchNorm), every head,
// CrossEntropyLoss(reduction="sum") (cmmvae_model.py:54,85), its autograd, GradientReversalFunction
// (components.py:879-899) -- and the clip_gradients / grad-norm bookkeeping of :118-131 that follows the backward.
//
// Nothing in such a stack couples cells except the weight gradients: a workgroup that owns 16 cells runs the forward,
// the cross-entropy (log-sum-exp is row-local) and the backward down to the hidden representation without a grid
// barrier.  The step had 12-17 latency-bound launches per adversary and phase; this file makes it two per phase for all
// adversaries together (adv_pass_kernel, adv_dw_kernel).  Exact-f32 MFMA (v_mfma_f32_16x16x4_f32): the work is small
// (< 0.5 % of the step's FLOPs) and bound by launch latency and L2 -> CU operand traffic, not by the matrix cores.
#include "common.h"

namespace {

constexpr int AR = 16;       // cells per workgroup
constexpr int AW = 8;        // waves per workgroup
constexpr int AT = AW * 64;  // threads

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// row[k .. k + 3] of a row with `len` valid elements (k % 4 == 0), zeros past the end.  `vec`: len % 4 == 0 and the
// row is 16-byte aligned.
__device__ __forceinline__ f32x4 ld4(const float* __restrict__ row, int k, int len, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
        if (k < len) v = *reinterpret_cast<const f32x4*>(row + k);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j < len) v[j] = row[k + j];
    }
    return v;
}

__host__ __device__ inline int pad16(int v) { return (v + 15) & ~15; }
__host__ __device__ inline int pad4(int v) { return (v + 3) & ~3; }
__host__ __device__ inline int head_tiles(int classes) { return (pad4(classes) + 15) / 16; }

// LDS plan of adv_pass_kernel (float offsets).  Activations [16][stride] with 4 floats of row padding (conflict-free
// 16-byte fragment reads), two gradient buffers, one weight tile per wave -- reused as that wave's merge record.
struct AdvLds {
    int act_off[MMVAE_ADV_MAX_LAYERS + 1], act_stride[MMVAE_ADV_MAX_LAYERS + 1];
    int da_off, dz_off, d_stride;
    int wt_off, wt_stride;  // per wave: 16 rows x wt_stride
    int misc_off;
    int total;
};
__host__ __device__ inline AdvLds adv_lds_layout(const int32_t* width, int L, int net, int H, int splits) {
    AdvLds o;
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
    p += AW * 16 * o.wt_stride;
    o.misc_off = p;
    p += H * AR + splits * H * AR + 64;
    if (p < 2 * AT + 64) p = 2 * AT + 64;  // the closing loss sums reuse the start of the buffer as AT doubles
    o.total = p;
    return o;
}

// One adversarial phase of one adversary per blockIdx.y; blockIdx.x = cell tile * splits + class split.
template <int NET>
__global__ __launch_bounds__(AT) void adv_pass_kernel(const mmvae_adv_job* __restrict__ jobs, int B, int splits,
                                                      unsigned* __restrict__ launch_ticket) {
    extern __shared__ float lds[];
    __shared__ mmvae_adv_job J;
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    {
        const int* src = reinterpret_cast<const int*>(jobs + blockIdx.y);
        int* dst = reinterpret_cast<int*>(&J);
        for (int i = tid; i < (int)(sizeof(mmvae_adv_job) / 4); i += AT) dst[i] = src[i];
    }
    __syncthreads();
    const int L = J.n_layers, H = J.H, NE = J.width[L], Ct = J.Ct;
    const AdvLds Y = adv_lds_layout(J.width, L, NET, H, splits);
    const int rt = blockIdx.x / splits, sp = blockIdx.x - rt * splits;
    const int row0 = rt * AR;
    constexpr int NEP = 16 * NET;
    constexpr int PST = 32 + AR * NEP;  // floats of one (cell tile, split, head) partial: max[16], sum[16], product[16][NEP]

    // ---- the hidden representation of this workgroup's cells -> LDS, zero padded
    {
        float* a0 = lds + Y.act_off[0];
        const int st = Y.act_stride[0], K0 = J.width[0], Kp = st - 4;
        for (int e = tid; e < AR * Kp; e += AT) {
            const int r = e / Kp, k = e - r * Kp, row = row0 + r;
            a0[r * st + k] = (row < B && k < K0) ? J.x[(int64_t)row * J.ldx + k] : 0.f;
        }
    }
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
        const bool vec = ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15u) == 0);
        for (int t = wave; t * 16 < Npad; t += AW) {
            const int n = 16 * t + l15;
            const float* wrow = W + (int64_t)n * K;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < Kp; k0 += 16) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(in + l15 * ist + k0 + 4 * g);
                f32x4 w = {0.f, 0.f, 0.f, 0.f};
                if (n < N) w = ld4(wrow, k0 + 4 * g, K, vec);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = mfma4(a[j], w[j], acc);
            }
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
    const bool hvec = ((NE & 3) == 0) && ((reinterpret_cast<uintptr_t>(Wh) & 15u) == 0);
    const bool lvec = ((Ct & 3) == 0) && ((reinterpret_cast<uintptr_t>(J.logits) & 15u) == 0);
    int T = 0;
    for (int h = 0; h < H; ++h) T += head_tiles(J.classes[h]);
    const int t_lo = (int)((int64_t)sp * T / splits), t_hi = (int)((int64_t)(sp + 1) * T / splits);
    float* part = J.partials + (int64_t)(rt * splits + sp) * H * PST;
    int base = 0;
    for (int h = 0; h < H; ++h) {
        const int C = J.classes[h], col = J.col[h], C4 = pad4(C), th = head_tiles(C);
        const int lo = (t_lo > base ? t_lo : base) - base, hi = (t_hi < base + th ? t_hi : base + th) - base;
        base += th;
        float m_run = -INFINITY, s_run = 0.f;
        f32x4 acc[NET];
#pragma unroll
        for (int t = 0; t < NET; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int tt = lo + wave; tt < hi; tt += AW) {
            const int c0 = 16 * tt;
            const int cls = c0 + l15;  // this lane's class as a ROW of the weight tile (A fragment of product 1)
            const float* wrow = Wh + (int64_t)(col + cls) * NE;
            f32x4 wf[NET];
#pragma unroll
            for (int i = 0; i < NET; ++i) {
                wf[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (cls < C) wf[i] = ld4(wrow, 16 * i + 4 * g, NE, hvec);
                *reinterpret_cast<f32x4*>(wt + l15 * WS + 16 * i + 4 * g) = wf[i];
            }
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NET; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s = mfma4(wf[i][j], ef[i][j], s);
            // s[r]: cell l15, class c0 + 4g + r
            const int cb = c0 + 4 * g;
            float lg[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                lg[r] = (cb + r < C) ? s[r] + (bh ? bh[col + cb + r] : 0.f) : -INFINITY;
            const int row = row0 + l15;
            if (row < B && cb < C4) {
                float* lp = J.logits + (int64_t)row * Ct + col + cb;
                if (lvec) {
                    *reinterpret_cast<f32x4*>(lp) = f32x4{lg[0], lg[1], lg[2], lg[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + cb + r < Ct) lp[r] = lg[r];
                }
            }
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
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int t = 0; t < NET; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = mfma4(p[j], wt[(4 * g + j) * WS + 16 * t + l15], acc[t]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
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
                if (M > -INFINITY)
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
                if (M > -INFINITY)
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

    // ---- ticket of the cell tile: the split that arrives last owns the rest of the pass
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned old = atomicAdd(&J.tickets[rt], 1u);
        s_last = old == (unsigned)splits - 1u;
        if (s_last) J.tickets[rt] = 0u;
    }
    __syncthreads();
    if (s_last) {
        __threadfence();
        const volatile float* P = J.partials + (int64_t)rt * splits * H * PST;
        float* lseb = lds + Y.misc_off;  // [H][16]
        float* wgt = lseb + H * AR;      // [splits][H][16]: exp(max_s - max) / sum
        if (tid < AR * H) {
            const int h = tid / AR, cell = tid - h * AR;
            float M = -INFINITY, S = 0.f;
            for (int s = 0; s < splits; ++s) M = fmaxf(M, P[(int64_t)(s * H + h) * PST + cell]);
            for (int s = 0; s < splits; ++s) {
                const float ms = P[(int64_t)(s * H + h) * PST + cell];
                if (ms > -INFINITY) S += P[(int64_t)(s * H + h) * PST + 16 + cell] * expf(ms - M);
            }
            for (int s = 0; s < splits; ++s) {
                const float ms = P[(int64_t)(s * H + h) * PST + cell];
                wgt[(s * H + h) * AR + cell] = (ms > -INFINITY) ? expf(ms - M) / S : 0.f;
            }
            const float lse = M + logf(S);
            lseb[h * AR + cell] = lse;
            const int row = row0 + cell;
            if (row < B) {
                const int64_t y = J.labels[(int64_t)h * B + row];
                const bool valid = y >= 0 && y < J.classes[h];
                const float ly =
                    valid ? reinterpret_cast<const volatile float*>(J.logits)[(int64_t)row * Ct + J.col[h] + y] : 0.f;
                J.lse[(int64_t)h * B + row] = lse;
                J.loss_rows[(int64_t)h * B + row] = valid ? lse - ly : 0.f;
            }
        }
        __syncthreads();
        // d(loss)/d(encoded) = gscale * sum over heads (softmax . W - W[label])
        float* dA = lds + Y.da_off;
        float* dZ = lds + Y.dz_off;
        const int ds = Y.d_stride;
        for (int e = tid; e < AR * NEP; e += AT) {
            const int cell = e / NEP, k = e - cell * NEP, row = row0 + cell;
            float v = 0.f;
            for (int h = 0; h < H; ++h)
                for (int s = 0; s < splits; ++s) {
                    const float w = wgt[(s * H + h) * AR + cell];
                    if (w != 0.f) v += P[(int64_t)(s * H + h) * PST + 32 + e] * w;
                }
            const bool live = row < B && k < NE;
            if (live)
                for (int h = 0; h < H; ++h) {
                    const int64_t y = J.labels[(int64_t)h * B + row];
                    if (y >= 0 && y < J.classes[h]) v -= Wh[(int64_t)(J.col[h] + y) * NE + k];
                }
            dA[cell * ds + k] = live ? J.gscale * v : 0.f;
        }
        __syncthreads();
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
                for (int t = wave; 16 * t < Kp; t += AW) {
                    const int n = 16 * t + l15;  // input feature
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    for (int k0 = 0; k0 < Np; k0 += 16) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(dZ + l15 * ds + k0 + 4 * g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int o = k0 + 4 * g + j;
                            const float b = (o < N && n < K) ? W[(int64_t)o * K + n] : 0.f;
                            acc = mfma4(a[j], b, acc);
                        }
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

    // ---- ticket of the launch: the workgroup that finishes last closes the losses of every job, in job order
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        const unsigned old = atomicAdd(launch_ticket, 1u);
        s_last = old == gridDim.x * gridDim.y - 1u;
        if (s_last) *launch_ticket = 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double* red = reinterpret_cast<double*>(lds);
    for (int j = 0; j < (int)gridDim.y; ++j) {
        const mmvae_adv_job* Jj = jobs + j;
        const int Hj = Jj->H;
        float total = 0.f;
        for (int h = 0; h < Hj; ++h) {
            const volatile float* rows = Jj->loss_rows + (int64_t)h * B;
            double s = 0.0;
            for (int i = tid; i < B; i += AT) s += (double)rows[i];
            red[tid] = s;
            __syncthreads();
            for (int st = AT / 2; st >= 1; st >>= 1) {
                if (tid < st) red[tid] += red[tid + st];
                __syncthreads();
            }
            const float r = (float)red[0];
            if (tid == 0) Jj->loss_each[h] = r;
            total = h == 0 ? r : total + r;
            __syncthreads();
        }
        if (tid == 0) {
            Jj->loss_total[0] = total;
            if (Jj->total_loss) Jj->total_loss[0] += Jj->total_scale * total;
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradients
constexpr int DW_TM = 32, DW_TN = 64, DW_KC = 64;
constexpr int DW_AS = 48, DW_BS = 80;  // LDS row strides: 4 consecutive k rows land in different bank quarters

struct DwHead {  // the head a thread's 4 classes belong to (heads job)
    int h, colh, C;
};

__global__ __launch_bounds__(AT) void adv_dw_kernel(const mmvae_adv_dw_job* __restrict__ jobs, int n_jobs,
                                                    const mmvae_adv_opt* __restrict__ opts, int n_opts,
                                                    float* __restrict__ partials, unsigned* __restrict__ ticket) {
    __shared__ float As[DW_KC * DW_AS];
    __shared__ float Bs[DW_KC * DW_BS];
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
    // staging role: row r of the chunk, 4 consecutive columns
    const int sr = tid >> 3, sq = tid & 7;
    const int am = m0 + 4 * sq;  // first of this thread's 4 output features (classes)
    const bool avec = ((ldz & 3) == 0) && ((reinterpret_cast<uintptr_t>(dz) & 15u) == 0) && am + 3 < M;
    const bool bvec = ((ldi & 3) == 0) && ((reinterpret_cast<uintptr_t>(inp) & 15u) == 0);
    DwHead hd = {-1, 0, 0};
    if (heads)
        for (int h = 0; h < J.H; ++h)
            if (am >= J.col[h] && am < J.col[h] + pad4(J.classes[h])) hd = DwHead{h, J.col[h], J.classes[h]};

    f32x4 ra, rb0, rb1;
    float r_lse = 0.f;
    int r_y = -1;
    auto fetch = [&](int k0) {
        const int row = k0 + sr;
        ra = f32x4{0.f, 0.f, 0.f, 0.f};
        rb0 = rb1 = ra;
        if (row >= B) return;
        const float* ap = dz + (int64_t)row * ldz + am;
        if (avec)
            ra = *reinterpret_cast<const f32x4*>(ap);
        else
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (am + j < M) ra[j] = ap[j];
        if (heads && hd.h >= 0) {
            r_lse = J.lse[(int64_t)hd.h * B + row];
            const int64_t y = J.labels[(int64_t)hd.h * B + row];
            r_y = (y >= 0 && y < hd.C) ? (int)y : -1;
        }
        const float* bp = inp + (int64_t)row * ldi + n0 + 4 * sq;
        if (bvec && n0 + 4 * sq + 3 < N)
            rb0 = *reinterpret_cast<const f32x4*>(bp);
        else
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n0 + 4 * sq + j < N) rb0[j] = bp[j];
        if (bvec && n0 + 32 + 4 * sq + 3 < N)
            rb1 = *reinterpret_cast<const f32x4*>(bp + 32);
        else
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n0 + 32 + 4 * sq + j < N) rb1[j] = bp[32 + j];
    };
    auto stage = [&](int k0) {
        f32x4 a = ra;
        if (heads) {
            // dlogits = gscale * (softmax - onehot); padding columns hold -inf -> 0
            const int row = k0 + sr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cls = am + j - hd.colh;
                float d = 0.f;
                if (hd.h >= 0 && row < B && cls < hd.C) d = gscale * (expf(ra[j] - r_lse) - (cls == r_y ? 1.f : 0.f));
                a[j] = d;
            }
        }
        *reinterpret_cast<f32x4*>(As + sr * DW_AS + 4 * sq) = a;
        *reinterpret_cast<f32x4*>(Bs + sr * DW_BS + 4 * sq) = rb0;
        *reinterpret_cast<f32x4*>(Bs + sr * DW_BS + 32 + 4 * sq) = rb1;
    };

    const int tr = wave >> 2, tc = wave & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    fetch(0);
    for (int k0 = 0; k0 < B; k0 += DW_KC) {
        stage(k0);
        __syncthreads();
        if (k0 + DW_KC < B) fetch(k0 + DW_KC);
#pragma unroll
        for (int ks = 0; ks < DW_KC / 4; ++ks) {
            const float a = As[(4 * ks + g) * DW_AS + 16 * tr + l15];
            const float b = Bs[(4 * ks + g) * DW_BS + 16 * tc + l15];
            acc = mfma4(a, b, acc);
        }
        if (tn == 0 && tid < DW_TM)
            for (int r = 0; r < DW_KC; ++r) bsum += As[r * DW_AS + tid];
        __syncthreads();
    }
    // accumulator register r: output feature m0 + 16 tr + 4g + r, input feature n0 + 16 tc + l15
    float sq_sum = 0.f;
    {
        const int n = n0 + 16 * tc + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * tr + 4 * g + r;
            if (m < M && n < N) {
                J.gW[(int64_t)m * N + n] = acc[r];
                sq_sum += acc[r] * acc[r];
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
        if (tid == 0) {
            double t = 0.0;
            for (int i = 0; i < AT; ++i) t += dred[i];
            const mmvae_adv_opt op = opts[o];
            if (op.flags) adam_state_finish(op.state, t, op.flags, op.max_norm, op.grad_scale, op.beta1, op.beta2);
            if (op.norm_out) op.norm_out[0] = (float)(sqrt(t) * (double)fabsf(op.grad_scale));
        }
        __syncthreads();
    }
}

template <int NET>
int launch_pass(int n_jobs, const mmvae_adv_job* jobs_dev, int B, int splits, size_t lds_bytes, unsigned* launch_ticket,
                hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(adv_pass_kernel<NET>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048) != hipSuccess)
            return MMVAE_ERR_LAUNCH;
        attr = true;
    }
    const int n_rt = (B + AR - 1) / AR;
    MMVAE_LAUNCH(adv_pass_kernel<NET>, dim3(n_rt * splits, n_jobs), dim3(AT), lds_bytes, stream, jobs_dev, B, splits,
                 launch_ticket);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

}  // namespace

extern "C" int mmvae_adv_pass_plan(const mmvae_adv_job* job, int splits, int* net, size_t* lds_bytes,
                                   int64_t* partial_floats) {
    if (!job || splits < 1 || splits > 64) return MMVAE_ERR_ARG;
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
    const AdvLds Y = adv_lds_layout(job->width, L, pick, H, splits);
    const size_t bytes = (size_t)Y.total * 4;
    if (bytes > 160 * 1024 - 2048 - sizeof(mmvae_adv_job) - 64) return MMVAE_ERR_ARG;
    if (net) *net = pick;
    if (lds_bytes) *lds_bytes = bytes;
    if (partial_floats) {
        const int64_t n_rt = (job->B + AR - 1) / AR;
        *partial_floats = n_rt * splits * H * (32 + AR * 16 * (int64_t)pick);
    }
    return MMVAE_OK;
}

extern "C" int mmvae_adv_pass_f32(int n_jobs, const mmvae_adv_job* jobs_dev, int B, int splits, int net, size_t lds_bytes,
                                  uint32_t* launch_ticket, mmvae_stream_t stream) {
    if (n_jobs < 1 || n_jobs > 64 || !jobs_dev || B < 1 || splits < 1 || splits > 64 || !launch_ticket) return MMVAE_ERR_ARG;
    if (lds_bytes > 160 * 1024 - 2048) return MMVAE_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (net) {
        case 1: return launch_pass<1>(n_jobs, jobs_dev, B, splits, lds_bytes, launch_ticket, st);
        case 2: return launch_pass<2>(n_jobs, jobs_dev, B, splits, lds_bytes, launch_ticket, st);
        case 4: return launch_pass<4>(n_jobs, jobs_dev, B, splits, lds_bytes, launch_ticket, st);
        case 8: return launch_pass<8>(n_jobs, jobs_dev, B, splits, lds_bytes, launch_ticket, st);
    }
    return MMVAE_ERR_ARG;
}

extern "C" int mmvae_adv_dw_prepare(int n_jobs, mmvae_adv_dw_job* jobs, int* total_blocks) {
    if (n_jobs < 1 || !jobs) return MMVAE_ERR_ARG;
    int next = 0;
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
    }
    if (total_blocks) *total_blocks = next;
    return MMVAE_OK;
}

extern "C" int mmvae_adv_dw_f32(int n_jobs, const mmvae_adv_dw_job* jobs_dev, int total_blocks, int n_opts,
                                const mmvae_adv_opt* opts_dev, float* partials, uint32_t* ticket, mmvae_stream_t stream) {
    if (n_jobs < 1 || !jobs_dev || total_blocks < 1 || n_opts < 0 || (n_opts && !opts_dev) || !partials || !ticket)
        return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(adv_dw_kernel, dim3(total_blocks), dim3(AT), 0, (hipStream_t)stream, jobs_dev, n_jobs, opts_dev, n_opts,
                 partials, ticket);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}
