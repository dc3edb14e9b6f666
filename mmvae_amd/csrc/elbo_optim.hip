// Row kernels of the MMVAE step for gfx950: reparameterisation + Gaussian KL (k8), sum-of-squares loss (k9),
// ELBO finalisation + K-sample log-mean-exp (k10), cross-entropy for the adversarial heads (k11),
// global-norm clip + Adam over a flat arena (k12, k13), Philox RNG (k15).
//
// Replaces (reference): components.py:795-801 (v = exp(a)+eps, rsample), vae.py:136-152 (kl_divergence,
// mse_loss(sum), loss), cmmvae_model.py:54,85 (CrossEntropyLoss(sum)), :126-131,:203-213 (clip_grad_norm_, Adam).
//
// All of these are HBM/latency-bound.  Per-cell (row) reductions use one 64-lane wavefront per row with
// shuffle reductions; global scalars are accumulated in fp64 in a fixed order (bitwise reproducible).
#include "common.h"

namespace {

// ---------------------------------------------------------------- reparam + KL
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(int B, int Z, int K, const float* __restrict__ mu,
                                                             const float* __restrict__ a_raw,
                                                             const float* __restrict__ eps, float var_eps,
                                                             float* __restrict__ std_out, float* __restrict__ z_out,
                                                             float* __restrict__ kl_row, float* __restrict__ stat_row) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float kl = 0.f, smu = 0.f, svar = 0.f;
    for (int j = lane; j < Z; j += 64) {
        const int64_t o = (int64_t)b * Z + j;
        const float m = mu[o];
        const float v = expf(a_raw[o]) + var_eps;
        const float s = sqrtf(v);
        if (std_out) std_out[o] = s;
        const float vr = s * s;  // the reference squares the scale again inside kl_divergence / .variance
        kl += 0.5f * (vr + m * m - 1.f - logf(vr));
        smu += m;
        svar += vr;
        if (z_out)
            for (int k = 0; k < K; ++k) {
                const int64_t ok = ((int64_t)k * B + b) * Z + j;
                z_out[ok] = m + s * eps[ok];
            }
    }
    kl = wave_sum(kl);
    smu = wave_sum(smu);
    svar = wave_sum(svar);
    if (lane == 0) {
        if (kl_row) kl_row[b] = kl;
        if (stat_row) {
            stat_row[b] = smu;
            stat_row[B + b] = svar;
        }
    }
}

__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(int B, int Z, int K, const float* __restrict__ mu,
                                                             const float* __restrict__ sd,
                                                             const float* __restrict__ eps, const float* __restrict__ dz,
                                                             const float* __restrict__ dmu_extra,
                                                             const float* __restrict__ dstd_extra,
                                                             const float* __restrict__ dkl_row,
                                                             const float* __restrict__ kl_scale_dev, float kl_scale_host,
                                                             float var_eps, float* __restrict__ dmu,
                                                             float* __restrict__ da_raw) {
    const int64_t n = (int64_t)B * Z;
    const float cs = (kl_scale_dev ? *kl_scale_dev : 1.f) * kl_scale_host;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(o / Z);
        const float c = cs * (dkl_row ? dkl_row[b] : 1.f);
        const float m = mu[o], s = sd[o];
        const float v = s * s;
        float gm = dmu_extra ? dmu_extra[o] : 0.f;
        float gs = dstd_extra ? dstd_extra[o] : 0.f;
        if (dz)
            for (int k = 0; k < K; ++k) {
                const int64_t ok = (int64_t)k * n + o;
                const float g = dz[ok];
                gm += g;
                gs += g * eps[ok];
            }
        dmu[o] = gm + c * m;
        da_raw[o] = (gs / (2.f * s) + c * 0.5f * (1.f - 1.f / v)) * (v - var_eps);
    }
}

// ---------------------------------------------------------------- stand-alone SE loss + grad: one workgroup per row strip
__global__ __launch_bounds__(256) void mse_rows_kernel(int B, int G, const float* __restrict__ xhat, int64_t ldxhat,
                                                       const float* __restrict__ x, int64_t ldx,
                                                       float* __restrict__ se_row, float* __restrict__ dxhat,
                                                       int64_t lddx, const float* __restrict__ gscale_dev,
                                                       float gscale_host) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float gs = 2.f * gscale_host * (gscale_dev ? *gscale_dev : 1.f);
    const float* xh = xhat + (int64_t)b * ldxhat;
    const float* xr = x + (int64_t)b * ldx;
    float* dr = dxhat ? dxhat + (int64_t)b * lddx : nullptr;
    float s = 0.f;
    for (int j = threadIdx.x; j < G; j += 256) {
        const float d = xh[j] - xr[j];
        s += d * d;
        if (dr) dr[j] = gs * d;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && se_row) se_row[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------- ELBO finalisation
// Stage 1: one wavefront per cell.  Sums the cell's squared-error partials over the T column tiles (lanes stride over
// tiles, then a butterfly: fixed order), applies the K-sample log-mean-exp, writes recon_row[b] and w[k, b].
constexpr int ELBO_MAXK = 64;
__global__ __launch_bounds__(256) void elbo_rows_kernel(int B, int K, int T, const float* __restrict__ se_part,
                                                        float* __restrict__ recon_row, float* __restrict__ w_out) {
    __shared__ float se_s[4][ELBO_MAXK];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    if (b >= B) return;
    const int64_t KB = (int64_t)K * B;
    if (K == 1) {
        float s = 0.f;
        for (int t = lane; t < T; t += 64) s += se_part[(int64_t)t * KB + b];
        s = wave_sum(s);
        if (lane == 0) {
            recon_row[b] = s;
            if (w_out) w_out[b] = 1.f;
        }
        return;
    }
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int t = lane; t < T; t += 64) s += se_part[(int64_t)t * KB + (int64_t)k * B + b];
        s = wave_sum(s);
        if (lane == 0) se_s[wv][k] = s;
    }
    __builtin_amdgcn_wave_barrier();
    // recon_b = -log(mean_k exp(-SE_bk)), max-subtracted; w_bk = softmax_k(-SE_b.)
    float v = (lane < K) ? -se_s[wv][lane] : -INFINITY;
    const float mx = wave_max(v);
    const float ex = (lane < K) ? expf(v - mx) : 0.f;
    const float lse = mx + logf(wave_sum(ex));
    if (lane == 0) recon_row[b] = -(lse - logf((float)K));
    if (w_out && lane < K) w_out[(int64_t)lane * B + b] = expf(v - lse);
}

// ---- opt-in "full IWAE" objective of the K-sample extension (not in the reference).
// r[k,b] = log q(z_kb | x_b) - log p(z_kb) = sum_j (-log s_bj - eps_kbj^2 / 2 + z_kbj^2 / 2); one wavefront per (k, b).
__global__ __launch_bounds__(256) void iwae_logratio_kernel(int B, int Z, int K, const float* __restrict__ std,
                                                            const float* __restrict__ eps, const float* __restrict__ z,
                                                            float* __restrict__ out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= K * B) return;
    const int b = row % B;
    const float* sr = std + (int64_t)b * Z;
    const float* er = eps + (int64_t)row * Z;
    const float* zr = z + (int64_t)row * Z;
    float s = 0.f;
    for (int j = lane; j < Z; j += 64) s += -logf(sr[j]) - 0.5f * er[j] * er[j] + 0.5f * zr[j] * zr[j];
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}

// Stage 1 of the full-IWAE finalisation: log-weights lw_k = -SE[k,b] - c r[k,b], bound_b = -logmeanexp_k lw_k,
// w[k,b] = softmax_k, and the w-weighted SE / r of the cell (reported as recon / kl).  rows3: [3, B].
__global__ __launch_bounds__(256) void elbo_rows_iwae_kernel(int B, int K, int T, const float* __restrict__ se_part,
                                                             const float* __restrict__ logratio,
                                                             const float* __restrict__ klw_dev, float c_host,
                                                             float* __restrict__ rows3, float* __restrict__ w_out) {
    __shared__ float se_s[4][ELBO_MAXK];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    if (b >= B) return;
    const int64_t KB = (int64_t)K * B;
    const float c = (klw_dev ? *klw_dev : 1.f) * c_host;
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int t = lane; t < T; t += 64) s += se_part[(int64_t)t * KB + (int64_t)k * B + b];
        s = wave_sum(s);
        if (lane == 0) se_s[wv][k] = s;
    }
    __builtin_amdgcn_wave_barrier();
    const float se = (lane < K) ? se_s[wv][lane] : 0.f;
    const float r = (lane < K) ? logratio[(int64_t)lane * B + b] : 0.f;
    const float v = (lane < K) ? -se - c * r : -INFINITY;
    const float mx = wave_max(v);
    const float ex = (lane < K) ? expf(v - mx) : 0.f;
    const float lse = mx + logf(wave_sum(ex));
    const float w = (lane < K) ? expf(v - lse) : 0.f;
    const float wse = wave_sum(w * se), wr = wave_sum(w * r);
    if (lane == 0) {
        rows3[b] = -(lse - logf((float)K));
        rows3[B + b] = wse;
        rows3[2 * (int64_t)B + b] = wr;
    }
    if (w_out && lane < K) w_out[(int64_t)lane * B + b] = w;
}

// Stage 2: out[0] = sum_b bound_b, out[1] = sum_b sum_k w SE, out[2] = mean_b sum_k w r, out[3] = kl weight, out[4..5] stats.
__global__ __launch_bounds__(256) void elbo_reduce_iwae_kernel(int B, const float* __restrict__ rows3,
                                                               const float* __restrict__ stat_row, int Z,
                                                               const float* __restrict__ klw_dev, float klw_host,
                                                               float* __restrict__ out6) {
    __shared__ double red[5][256];
    const int tid = threadIdx.x;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, smu = 0.0, svar = 0.0;
    for (int b = tid; b < B; b += 256) {
        a0 += (double)rows3[b];
        a1 += (double)rows3[B + b];
        a2 += (double)rows3[2 * (int64_t)B + b];
        if (stat_row) {
            smu += (double)stat_row[b];
            svar += (double)stat_row[B + b];
        }
    }
    red[0][tid] = a0;
    red[1][tid] = a1;
    red[2][tid] = a2;
    red[3][tid] = smu;
    red[4][tid] = svar;
    __syncthreads();
    if (tid == 0) {
        double t[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 256; ++i)
            for (int q = 0; q < 5; ++q) t[q] += red[q][i];
        out6[0] = (float)t[0];
        out6[1] = (float)t[1];
        out6[2] = (float)(t[2] / (double)B);
        out6[3] = (klw_dev ? *klw_dev : 1.f) * klw_host;
        out6[4] = (float)(t[3] / ((double)B * (double)(Z > 0 ? Z : 1)));
        out6[5] = (float)(t[4] / ((double)B * (double)(Z > 0 ? Z : 1)));
    }
}

// Backward terms of the log-ratio: dz[k,b,j] += c w[k,b] z[k,b,j]  (d r / d z = z), and the direct dependence on the
// variance, -log s, handed to mmvae_reparam_kl_bwd as dstd_extra[b,j] = -c / s[b,j]  (sum_k w = 1).
__global__ __launch_bounds__(256) void iwae_bwd_terms_kernel(int B, int Z, int K, const float* __restrict__ klw_dev,
                                                             float c_host, const float* __restrict__ w,
                                                             const float* __restrict__ z, const float* __restrict__ std,
                                                             float* __restrict__ dz, float* __restrict__ dstd_extra) {
    const float c = (klw_dev ? *klw_dev : 1.f) * c_host;
    const int64_t n = (int64_t)K * B * Z, nb = (int64_t)B * Z;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / Z;  // k * B + b
        dz[i] += c * w[row] * z[i];
        if (i < nb) dstd_extra[i] = -c / std[i];
    }
}

// Stage 2: single workgroup, fp64 accumulation in a fixed order.
__global__ __launch_bounds__(256) void elbo_reduce_kernel(int B, const float* __restrict__ recon_row,
                                                          const float* __restrict__ kl_row,
                                                          const float* __restrict__ stat_row, int Z,
                                                          const float* __restrict__ klw_dev, float klw_host,
                                                          float* __restrict__ out6) {
    __shared__ double red[4][256];
    const int tid = threadIdx.x;
    double recon = 0.0, kl = 0.0, smu = 0.0, svar = 0.0;
    for (int b = tid; b < B; b += 256) {
        recon += (double)recon_row[b];
        if (kl_row) kl += (double)kl_row[b];
        if (stat_row) {
            smu += (double)stat_row[b];
            svar += (double)stat_row[B + b];
        }
    }
    red[0][tid] = recon;
    red[1][tid] = kl;
    red[2][tid] = smu;
    red[3][tid] = svar;
    __syncthreads();
    if (tid == 0) {
        double r = 0.0, k = 0.0, m = 0.0, v = 0.0;
        for (int i = 0; i < 256; ++i) {
            r += red[0][i];
            k += red[1][i];
            m += red[2][i];
            v += red[3][i];
        }
        const float klw = (klw_dev ? *klw_dev : 1.f) * klw_host;
        const float reconf = (float)r;
        const float klf = (float)(k / (double)B);
        out6[0] = reconf + klw * klf;
        out6[1] = reconf;
        out6[2] = klf;
        out6[3] = klw;
        out6[4] = (float)(m / ((double)B * (double)(Z > 0 ? Z : 1)));
        out6[5] = (float)(v / ((double)B * (double)(Z > 0 ? Z : 1)));
    }
}

// ---------------------------------------------------------------- cross entropy (sum) + gradient, one wavefront per row
__global__ __launch_bounds__(256) void ce_rows_kernel(int B, int C, const float* __restrict__ logits, int64_t ld,
                                                      const int64_t* __restrict__ labels, float* __restrict__ loss_rows,
                                                      float* __restrict__ dlogits, int64_t ldd,
                                                      const float* __restrict__ gscale_dev, float gscale_host) {
    const float gscale = gscale_host * (gscale_dev ? *gscale_dev : 1.f);
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* lr = logits + (int64_t)b * ld;
    float mx = -INFINITY;
    for (int j = lane; j < C; j += 64) mx = fmaxf(mx, lr[j]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < C; j += 64) s += expf(lr[j] - mx);
    s = wave_sum(s);
    const float lse = mx + logf(s);
    const int64_t y = labels[b];
    if (lane == 0 && loss_rows) loss_rows[b] = (y >= 0 && y < C) ? lse - lr[y] : 0.f;
    if (dlogits) {
        float* dr = dlogits + (int64_t)b * ldd;
        for (int j = lane; j < C; j += 64) {
            const float p = expf(lr[j] - lse);
            dr[j] = gscale * (p - ((int64_t)j == y ? 1.f : 0.f));
        }
    }
}

// Wide heads (e.g. donor_id: 4644 classes): one WORKGROUP per row, the row held in registers (16-byte loads, all in
// flight at once), block-wide max and sum of exponentials, gradient written from the registers: one pass over the
// logits instead of three latency-bound strided sweeps by a single wavefront (37 us -> a few us at 512 x 4644).
// Requires C <= 256 * 4 * VPT.
template <int VPT>
__device__ __forceinline__ void ce_row_block(int b, int C, const float* __restrict__ logits, int64_t ld,
                                             const int64_t* __restrict__ labels, float* __restrict__ loss_rows,
                                             float* __restrict__ dlogits, int64_t ldd, float gscale) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* lr = logits + (int64_t)b * ld;
    const int nv = C >> 2;  // whole float4 groups; the (C & 3) tail elements are handled by the first threads
    f32x4 v[VPT];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = tid + 256 * u;
        v[u] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (i < nv) v[u] = *reinterpret_cast<const f32x4*>(lr + 4 * i);
    }
    float tail = -INFINITY;
    const int ti = 4 * nv + tid;
    if (ti < C) tail = lr[ti];
#pragma unroll
    for (int u = 0; u < VPT; ++u) mx = fmaxf(fmaxf(mx, fmaxf(v[u][0], v[u][1])), fmaxf(v[u][2], v[u][3]));
    mx = wave_max(fmaxf(mx, tail));
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < VPT; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += expf(v[u][j] - mx);  // exp(-inf) = 0 for the padding
    s += expf(tail - mx);
    s = wave_sum(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float lse = mx + logf(s);
    const int64_t y = labels[b];
    if (tid == 0 && loss_rows) loss_rows[b] = (y >= 0 && y < C) ? lse - lr[y] : 0.f;
    if (dlogits) {
        float* dr = dlogits + (int64_t)b * ldd;
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int i = tid + 256 * u;
            if (i < nv) {
                f32x4 d;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    d[j] = gscale * (expf(v[u][j] - lse) - ((int64_t)(4 * i + j) == y ? 1.f : 0.f));
                *reinterpret_cast<f32x4*>(dr + 4 * i) = d;
            }
        }
        if (ti < C) dr[ti] = gscale * (expf(tail - lse) - ((int64_t)ti == y ? 1.f : 0.f));
    }
}

template <int VPT>
__global__ __launch_bounds__(256) void ce_rows_block_kernel(int B, int C, const float* __restrict__ logits, int64_t ld,
                                                            const int64_t* __restrict__ labels,
                                                            float* __restrict__ loss_rows, float* __restrict__ dlogits,
                                                            int64_t ldd, const float* __restrict__ gscale_dev,
                                                            float gscale_host) {
    ce_row_block<VPT>(blockIdx.x, C, logits, ld, labels, loss_rows, dlogits, ldd,
                      gscale_host * (gscale_dev ? *gscale_dev : 1.f));
}

// All heads of an adversary in one launch: head h = columns [col[h], col[h] + classes[h]) of one logits matrix, labels
// and per-row losses in rows h of [H, B] arrays.  Workgroup (b, h) is ce_rows_block_kernel's workgroup b of head h.
template <int VPT>
__global__ __launch_bounds__(256) void ce_heads_kernel(int B, const int32_t* __restrict__ col,
                                                       const int32_t* __restrict__ classes,
                                                       const float* __restrict__ logits, int64_t ld,
                                                       const int64_t* __restrict__ labels, float* __restrict__ loss_rows,
                                                       float* __restrict__ dlogits, int64_t ldd, float gscale) {
    const int h = blockIdx.y, c0 = col[h];
    ce_row_block<VPT>(blockIdx.x, classes[h], logits + c0, ld, labels + (int64_t)h * B,
                      loss_rows ? loss_rows + (int64_t)h * B : nullptr, dlogits ? dlogits + c0 : nullptr, ldd, gscale);
}

// fixed-order fp64 sum of n floats by one workgroup
__global__ __launch_bounds__(1024) void sum_kernel(int64_t n, const float* __restrict__ v, float* __restrict__ out,
                                                   int accumulate) {
    __shared__ double red[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 512; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)red[0];
}

// H sums of n floats each (rows of v, stride ld) by one workgroup, each reduced exactly like sum_kernel, plus their
// float total in row order: the per-head losses of one adversarial phase and their sum in one launch instead of a
// sum + an accumulate per head.
__global__ __launch_bounds__(1024) void sum_rows_kernel(int H, int64_t n, const float* __restrict__ v, int64_t ld,
                                                        float* __restrict__ out_each, float* __restrict__ out_total) {
    __shared__ double red[1024];
    float total = 0.f;
    for (int h = 0; h < H; ++h) {
        const float* row = v + (int64_t)h * ld;
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)row[i];
        __syncthreads();
        red[threadIdx.x] = s;
        __syncthreads();
        for (int st = 512; st >= 1; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        const float r = (float)red[0];
        if (threadIdx.x == 0 && out_each) out_each[h] = r;
        total = h == 0 ? r : total + r;
    }
    if (threadIdx.x == 0 && out_total) out_total[0] = total;
}

// ---------------------------------------------------------------- clip + Adam over a flat arena
constexpr int64_t SQN_CHUNK = 1 << 16;  // floats per workgroup of the norm pass (fixed -> reproducible)

// Sum of squares of chunk `chunk` (SQN_CHUNK floats) of g[0..n): the value of one partial.  Valid in thread 0.
__device__ __forceinline__ float sqnorm_chunk(int64_t n, const float* __restrict__ g, int64_t chunk, float* red) {
    const int64_t beg = chunk * SQN_CHUNK;
    int64_t end = beg + SQN_CHUNK;
    if (end > n) end = n;
    float s = 0.f;
    const bool vec = ((reinterpret_cast<uintptr_t>(g) & 15u) == 0);
    if (vec) {
        const int64_t nv = (end - beg) / 4;
        const f32x4* gv = reinterpret_cast<const f32x4*>(g + beg);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int64_t i = threadIdx.x;
        for (; i + 7 * 256 < nv; i += 8 * 256) {  // 8 loads in flight per thread; accumulation order unchanged
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = gv[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s0 += v[u].x * v[u].x;
                s1 += v[u].y * v[u].y;
                s2 += v[u].z * v[u].z;
                s3 += v[u].w * v[u].w;
            }
        }
        for (; i < nv; i += 256) {
            const f32x4 v = gv[i];
            s0 += v.x * v.x;
            s1 += v.y * v.y;
            s2 += v.z * v.z;
            s3 += v.w * v.w;
        }
        s = (s0 + s1) + (s2 + s3);
        for (int64_t i = beg + nv * 4 + threadIdx.x; i < end; i += 256) s += g[i] * g[i];
    } else {
        for (int64_t i = beg + threadIdx.x; i < end; i += 256) s += g[i] * g[i];
    }
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sqnorm_kernel(int64_t n, const float* __restrict__ g,
                                                     float* __restrict__ partials) {
    __shared__ float red[4];
    const float s = sqnorm_chunk(n, g, blockIdx.x, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// Global norm (fp64 sum of the partials, fixed order) -> clip coefficient, step count, bias corrections in state[].
__device__ __forceinline__ void adam_prepare_body(int64_t np, const float* partials, float max_norm, float grad_scale,
                                                  float beta1, float beta2, float* __restrict__ state, unsigned flags,
                                                  double* red) {
    double s = 0.0;
    if (flags & MMVAE_PREPARE_NORM)
        for (int64_t i = threadIdx.x; i < np; i += 256) s += (double)partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        if (flags & MMVAE_PREPARE_NORM)
            for (int i = 0; i < 256; ++i) t += red[i];
        adam_state_finish(state, t, flags, max_norm, grad_scale, beta1, beta2);
    }
}

__global__ __launch_bounds__(256) void adam_prepare_kernel(int64_t np, const float* __restrict__ partials,
                                                           float max_norm, float grad_scale, float beta1, float beta2,
                                                           float* __restrict__ state, unsigned flags) {
    __shared__ double red[256];
    adam_prepare_body(np, partials, max_norm, grad_scale, beta1, beta2, state, flags, red);
}

// The norm pass over up to 4 ranges of a gradient arena (what no fused GEMM epilogue covers) and adam_prepare in ONE
// launch: every workgroup leaves its chunk's partial, the workgroup that arrives last (a ticket counter, reset for the
// next step) sums ALL the optimiser's partials in their fixed order.  Same numbers as the separate launches.
struct SqRanges {
    const float* g[4];
    int64_t n[4];
    int nb[4];
    int nr;
};
__global__ __launch_bounds__(256) void sqnorm_ranges_prepare_kernel(SqRanges r, float* __restrict__ partials,
                                                                    unsigned* __restrict__ ticket, int64_t np_all,
                                                                    const float* partials_all, float max_norm,
                                                                    float grad_scale, float beta1, float beta2,
                                                                    float* __restrict__ state, unsigned flags) {
    __shared__ float red[4];
    __shared__ double dred[256];
    __shared__ int last;
    int b = blockIdx.x, k = 0;
    while (k + 1 < r.nr && b >= r.nb[k]) {
        b -= r.nb[k];
        ++k;
    }
    const float s = sqnorm_chunk(r.n[k], r.g[k], b, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = s;
        __threadfence();  // the partial is visible device-wide before the ticket is taken
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) *ticket = 0u;
    __threadfence();  // acquire: the other workgroups' partials
    adam_prepare_body(np_all, const_cast<const float*>(reinterpret_cast<const volatile float*>(partials_all)), max_norm,
                      grad_scale, beta1, beta2, state, flags, dred);
}

#ifndef MMVAE_ADAM_NT
#define MMVAE_ADAM_NT 1  // 1: nontemporal loads of g, m, v and stores of m, v (2: p as well); 0: plain accesses
#endif
// cv > 0: gradient_clip_algorithm "value" (config.py:8; Lightning clip_gradients -> clip_grad_value_): every element
// of the (averaged) gradient is clamped to [-cv, cv] ahead of the weight decay; the norm coefficient is 1 then.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float gmul, float wd, float b1, float b2,
                                         float step_size, float inv_bc2_sqrt, float eps, float cv = 0.f) {
    const float gs = g * gmul;
    const float gc = (gs != gs ? gs : fminf(fmaxf(gs, -cv), cv)) + wd * p;  // (torch.clamp keeps a NaN; fminf would hide it)
    g = g * gmul + wd * p;
    if (cv > 0.f) g = gc;
    m = m + (1.f - b1) * (g - m);
    v = b2 * v + (1.f - b2) * g * g;
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__device__ __forceinline__ void adam_step_body(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                               float* __restrict__ m, float* __restrict__ v,
                                               const float* __restrict__ state, float lr, float b1, float b2, float eps,
                                               float wd, float grad_scale, int vec, int copy_n, const float* copy_src,
                                               float* copy_dst) {
    // optional rider: a small device-to-device copy (the step's logged scalars into the plan's log buffer) done by
    // workgroup 0 -- instead of a launch of its own behind the longest kernel of the step
    if (copy_n > 0 && blockIdx.x == 0)
        for (int i = threadIdx.x; i < copy_n; i += blockDim.x) copy_dst[i] = copy_src[i];
    const float gmul = state[2] * grad_scale;
    const float step_size = lr / state[3];
    const float inv_bc2_sqrt = 1.f / sqrtf(state[4]);
    const float cv = state[5];  // clip-by-value bound (0: off), mmvae_adam_set_clip_value
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t nv = n / 4;
        f32x4* pv = reinterpret_cast<f32x4*>(p);
        const f32x4* gv = reinterpret_cast<const f32x4*>(g);
        f32x4* mv = reinterpret_cast<f32x4*>(m);
        f32x4* vv = reinterpret_cast<f32x4*>(v);
        for (int64_t i = tid0; i < nv; i += stride) {
#if MMVAE_ADAM_NT
            // streaming operands: read once, written once per step -- keep them out of the caches' way
            // (178 us against 238 us per 41 M parameters back to back: 6.6 TB/s, profiles/r2_adam_nt.txt)
            f32x4 pp = MMVAE_ADAM_NT == 2 ? __builtin_nontemporal_load(pv + i) : pv[i];
            f32x4 gg = __builtin_nontemporal_load(gv + i), mm = __builtin_nontemporal_load(mv + i),
                  vw = __builtin_nontemporal_load(vv + i);
#else
            f32x4 pp = pv[i], gg = gv[i], mm = mv[i], vw = vv[i];
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pp[e], me = mm[e], ve = vw[e];
                adam_one(pe, gg[e], me, ve, gmul, wd, b1, b2, step_size, inv_bc2_sqrt, eps, cv);
                pp[e] = pe;
                mm[e] = me;
                vw[e] = ve;
            }
#if MMVAE_ADAM_NT == 2
            __builtin_nontemporal_store(pp, pv + i);
#else
            pv[i] = pp;
#endif
#if MMVAE_ADAM_NT
            __builtin_nontemporal_store(mm, mv + i);
            __builtin_nontemporal_store(vw, vv + i);
#else
            mv[i] = mm;
            vv[i] = vw;
#endif
        }
        for (int64_t i = nv * 4 + tid0; i < n; i += stride)
            adam_one(p[i], g[i], m[i], v[i], gmul, wd, b1, b2, step_size, inv_bc2_sqrt, eps, cv);
    } else {
        for (int64_t i = tid0; i < n; i += stride)
            adam_one(p[i], g[i], m[i], v[i], gmul, wd, b1, b2, step_size, inv_bc2_sqrt, eps, cv);
    }
}

__global__ __launch_bounds__(256) void adam_step_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ state, float lr, float b1, float b2,
                                                        float eps, float wd, float grad_scale, int vec, int copy_n,
                                                        const float* copy_src, float* copy_dst) {
    adam_step_body(n, p, g, m, v, state, lr, b1, b2, eps, wd, grad_scale, vec, copy_n, copy_src, copy_dst);
}

// The same pass confined to a chosen number of compute units (mmvae_adam_set_workgroups): that many workgroups of 1024
// threads, each holding enough LDS that a compute unit takes exactly one -- for callers that run the update beside
// another kernel whose grid is capped to the remaining units (the engine's deferred expert update).  Elementwise work:
// identical results for any grid.
constexpr int ADAM_WIDE_LDS = 84 * 1024;
__global__ __launch_bounds__(1024) void adam_step_wide_kernel(int64_t n, float* __restrict__ p,
                                                              const float* __restrict__ g, float* __restrict__ m,
                                                              float* __restrict__ v, const float* __restrict__ state,
                                                              float lr, float b1, float b2, float eps, float wd,
                                                              float grad_scale, int vec, int copy_n,
                                                              const float* copy_src, float* copy_dst) {
    adam_step_body(n, p, g, m, v, state, lr, b1, b2, eps, wd, grad_scale, vec, copy_n, copy_src, copy_dst);
}

// The update of several optimisers' arenas in one launch (blockIdx.y = arena): the adversaries of a step.  Arithmetic of
// adam_step_kernel per arena.
__global__ __launch_bounds__(256) void adam_step_multi_kernel(const mmvae_adam_arena* __restrict__ arenas) {
    const mmvae_adam_arena a = arenas[blockIdx.y];
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.p) | reinterpret_cast<uintptr_t>(a.g) |
                         reinterpret_cast<uintptr_t>(a.m) | reinterpret_cast<uintptr_t>(a.v);
    adam_step_body(a.n, a.p, a.g, a.m, a.v, a.state, a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.grad_scale,
                   (al & 15u) == 0, 0, nullptr, nullptr);
}

inline int grid_for(int64_t n, int per_block, int cap);
constexpr int STREAM_GRID_CAP = 1 << 22;  // long streaming passes: one workgroup per block of elements (see launch_adam_step)
static int g_adam_workgroups = 0;  // 0: the chip-filling grid of 256-thread workgroups

extern "C" int mmvae_adam_set_workgroups(int workgroups) {
    if (workgroups < 0 || workgroups > 256) return MMVAE_ERR_ARG;
    g_adam_workgroups = workgroups;
    return MMVAE_OK;
}

extern "C" int mmvae_adam_get_workgroups(void) { return g_adam_workgroups; }

template <typename... Args>
static int launch_adam_step(int64_t n, hipStream_t stream, Args... args) {
    if (g_adam_workgroups > 0) {
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(adam_step_wide_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, ADAM_WIDE_LDS) != hipSuccess)
                return MMVAE_ERR_LAUNCH;
            attr = true;
        }
        MMVAE_LAUNCH(adam_step_wide_kernel, dim3(g_adam_workgroups), dim3(1024), (size_t)ADAM_WIDE_LDS, stream, n,
                     args...);
    } else {
        // One workgroup per 1024 elements, no grid-stride loop: the hardware hands out workgroups in order, so the seven
        // streams of the pass (p, g, m, v in; p, m, v out) move through memory as one narrow window.  With the grid capped
        // at 4096 workgroups and a loop (r1-r3), the resident half of the grid and the half that waits for it drift apart
        // over the iterations: 42 M parameters 176-183 us against 166 (7.1 TB/s), 124 M (the reference's 60 530 genes)
        // 729 us against 578 (tools/debug/adam_bw.py); C2 step 0.980 -> 0.960 ms, 60 530 / 52 437 genes 2.49 -> 2.36.
        MMVAE_LAUNCH(adam_step_kernel, dim3(grid_for(n, 1024, STREAM_GRID_CAP)), dim3(256), 0, stream, n, args...);
    }
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

// Adam over a LIST of arena segments, one workgroup per job (a job = at most ADAM_JOB_ELEMS consecutive elements of
// one parameter tensor, with that tensor's own bias corrections): the optimiser of a conditional-layer model updates
// only the condition blocks that took part in the step (torch.optim.Adam skips parameters without a gradient and counts
// steps per parameter) -- hundreds of 128 x 128 blocks out of thousands -- in ONE launch.
__global__ __launch_bounds__(256) void adam_step_jobs_kernel(const mmvae_adam_job* __restrict__ jobs, float* __restrict__ p,
                                                             const float* __restrict__ g, float* __restrict__ m,
                                                             float* __restrict__ v, const float* __restrict__ state,
                                                             float lr, float b1, float b2, float eps, float wd,
                                                             float grad_scale) {
    const mmvae_adam_job job = jobs[blockIdx.x];
    if ((job.reserved & 3) == 2) return;  // a retired segment (zeroed for the exchange, takes no step)
    const float gmul = state[2] * grad_scale;
    const float step_size = lr / job.bc1;
    const float inv_bc2_sqrt = 1.f / sqrtf(job.bc2);
    const float cv = state[5];
    const int64_t o = job.offset;
    const int n = job.len;
    if ((o & 3) == 0) {  // arena tensors start on 16-byte boundaries: 16-byte accesses, scalar tail
        const int nv = n >> 2;
        f32x4* pv = reinterpret_cast<f32x4*>(p + o);
        const f32x4* gv = reinterpret_cast<const f32x4*>(g + o);
        f32x4* mv = reinterpret_cast<f32x4*>(m + o);
        f32x4* vv = reinterpret_cast<f32x4*>(v + o);
        for (int i = threadIdx.x; i < nv; i += 256) {
            f32x4 pp = pv[i], gg = gv[i], mm = mv[i], vw = vv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pp[e], me = mm[e], ve = vw[e];
                adam_one(pe, gg[e], me, ve, gmul, wd, b1, b2, step_size, inv_bc2_sqrt, eps, cv);
                pp[e] = pe;
                mm[e] = me;
                vw[e] = ve;
            }
            pv[i] = pp;
            mv[i] = mm;
            vv[i] = vw;
        }
        for (int i = 4 * nv + threadIdx.x; i < n; i += 256)
            adam_one(p[o + i], g[o + i], m[o + i], v[o + i], gmul, wd, b1, b2, step_size, inv_bc2_sqrt, eps, cv);
    } else {
        for (int i = threadIdx.x; i < n; i += 256)
            adam_one(p[o + i], g[o + i], m[o + i], v[o + i], gmul, wd, b1, b2, step_size, inv_bc2_sqrt, eps, cv);
    }
}

// Sum of squares of one job's arena segment -> partials[job] (0 for an empty job): the global norm of a step in which
// only the listed tensors have a gradient, without reading (or zeroing) the rest of the arena.  Fixed order: thread-
// strided partial sums, then the wave / workgroup reduction in a fixed tree.
__global__ __launch_bounds__(256) void sqnorm_jobs_kernel(const mmvae_adam_job* __restrict__ jobs,
                                                          const float* __restrict__ g, float* __restrict__ partials) {
    __shared__ float red[4];
    const mmvae_adam_job job = jobs[blockIdx.x];
    const int64_t o = job.offset;
    const int n = (job.reserved & 3) == 2 ? 0 : job.len;  // retired segments do not count
    float s = 0.f;
    if ((o & 3) == 0) {
        const int nv = n >> 2;
        const f32x4* gv = reinterpret_cast<const f32x4*>(g + o);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int i = threadIdx.x; i < nv; i += 256) {
            const f32x4 v = gv[i];
            s0 += v.x * v.x;
            s1 += v.y * v.y;
            s2 += v.z * v.z;
            s3 += v.w * v.w;
        }
        s = (s0 + s1) + (s2 + s3);
        for (int i = 4 * nv + threadIdx.x; i < n; i += 256) s += g[o + i] * g[o + i];
    } else {
        for (int i = threadIdx.x; i < n; i += 256) s += g[o + i] * g[o + i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Zero the gradient segment of every job whose `reserved` word is 1 -- the condition blocks that step on this rank only
// because ANOTHER rank saw them (data parallelism: this rank contributes zeros to their all-reduce) -- or 2 -- segments
// that stepped last time and do not now ("retired": zeroed once so that the dense all-reduce never sums stale values;
// the norm and Adam job kernels skip them).  Same fixed launch size as the other job kernels.
__global__ __launch_bounds__(256) void zero_flagged_jobs_kernel(const mmvae_adam_job* __restrict__ jobs,
                                                                float* __restrict__ g) {
    const mmvae_adam_job job = jobs[blockIdx.x];
    if ((job.reserved & 3) != 1 && (job.reserved & 3) != 2) return;
    for (int i = threadIdx.x; i < job.len; i += 256) g[job.offset + i] = 0.f;
}

// Gather / scatter the segments of a job table into / out of a staging buffer: the gradient exchange of a step in which
// only some tensors took part moves the staging buffer instead of the whole arena.  Job j's place in it is written by
// the host into the upper bits of its `reserved` word: (reserved >> 2) * 128 floats (segments are laid out back to back,
// each rounded up to 128 floats; the low 2 bits stay the zero / retire flags).
template <bool PACK>
__global__ __launch_bounds__(256) void jobs_pack_kernel(const mmvae_adam_job* __restrict__ jobs, float* __restrict__ arena,
                                                        float* __restrict__ staging) {
    const mmvae_adam_job job = jobs[blockIdx.x];
    float* a = arena + job.offset;
    float* s = staging + (int64_t)((unsigned)job.reserved >> 2) * 128;
    const int n = job.len, padded = (n + 127) & ~127;
    for (int i = threadIdx.x; i < n; i += 256) {
        if (PACK)
            s[i] = a[i];
        else
            a[i] = s[i];
    }
    if (PACK)  // the padding is part of the exchanged range: defined (zero) on every rank
        for (int i = n + threadIdx.x; i < padded; i += 256) s[i] = 0.f;
}

// Diagnostics: stand-in for a collective that runs beside the step.  Each workgroup holds `lds`
// bytes of LDS (so that a CU cannot host its usual two GEMM workgroups beside it) and spins for `micros` microseconds of
// the constant 100 MHz wall clock; every wave reaches the exit condition, the grid always drains.
__global__ __launch_bounds__(256) void occupy_kernel(long long ticks, float* __restrict__ sink) {
    extern __shared__ float hold[];
    hold[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    float acc = hold[(threadIdx.x + 1) & 255];
    while (wall_clock64() - t0 < ticks) acc = acc * 1.0000001f + 1e-7f;
    if (acc == -1.f && sink) sink[0] = acc;  // keeps the loop alive; never true
}

// ---------------------------------------------------------------- Philox4x32-10
struct u32x4 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ u32x4 philox4x32_10(uint64_t counter, uint64_t stream_id, uint64_t seed) {
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = (uint32_t)stream_id,
             c3 = (uint32_t)(stream_id >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(256) void philox_mask_kernel(int64_t n, float p_drop, uint8_t* __restrict__ mask,
                                                          const uint64_t* __restrict__ rng, uint64_t stream_id) {
    const uint64_t seed = rng[0], off = rng[1];
    const int64_t nq = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 r = philox4x32_10(off + (uint64_t)q, stream_id, seed);
        const uint32_t rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = q * 4 + j;
            if (i < n) mask[i] = (u01(rv[j]) >= p_drop) ? 1 : 0;
        }
    }
}

__global__ __launch_bounds__(256) void philox_normal_kernel(int64_t n, float* __restrict__ out,
                                                            const uint64_t* __restrict__ rng, uint64_t stream_id) {
    const uint64_t seed = rng[0], off = rng[1];
    const int64_t nq = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 r = philox4x32_10(off + (uint64_t)q, stream_id, seed);
        const float u0 = u01(r.x), u1 = u01(r.y), u2 = u01(r.z), u3 = u01(r.w);
        const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
        float s0, c0, s1, c1;
        sincosf(6.283185307179586f * u1, &s0, &c0);
        sincosf(6.283185307179586f * u3, &s1, &c1);
        const float nv[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = q * 4 + j;
            if (i < n) out[i] = nv[j];
        }
    }
}

__global__ void philox_advance_kernel(uint64_t* rng, uint64_t by) { rng[1] += by; }

// Several fills of one step (the dropout keep-masks and the rsample noise) in one launch: workgroup column blockIdx.y
// serves job y with the arithmetic of philox_mask_kernel / philox_normal_kernel (counter-based: the numbers depend on
// (offset, stream id, element) only, not on the launch shape).
// advance_by > 0 (with a ticket word): the workgroup that finishes last adds it to the counter -- every workgroup has read
// the counter by then -- so that the step needs no separate mmvae_philox_advance launch.
__global__ __launch_bounds__(256) void philox_fill_jobs_kernel(const mmvae_philox_job* __restrict__ jobs, uint64_t* rng,
                                                               uint64_t advance_by, unsigned* ticket) {
    const mmvae_philox_job job = jobs[blockIdx.y];
    const uint64_t seed = rng[0], off = rng[1];
    const int64_t n = job.n, nq = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 r = philox4x32_10(off + (uint64_t)q, job.stream_id, seed);
        if (job.kind == 0) {
            uint8_t* mask = static_cast<uint8_t*>(job.out);
            const uint32_t rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t i = q * 4 + j;
                if (i < n) mask[i] = (u01(rv[j]) >= job.p_drop) ? 1 : 0;
            }
        } else {
            float* out = static_cast<float*>(job.out);
            const float u0 = u01(r.x), u1 = u01(r.y), u2 = u01(r.z), u3 = u01(r.w);
            const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
            float s0, c0, s1, c1;
            sincosf(6.283185307179586f * u1, &s0, &c0);
            sincosf(6.283185307179586f * u3, &s1, &c1);
            const float nv[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t i = q * 4 + j;
                if (i < n) out[i] = nv[j];
            }
        }
    }
    if (advance_by && ticket) {
        __syncthreads();  // every thread of this workgroup is done with `off`
        if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x * gridDim.y - 1) {
            *ticket = 0u;
            rng[1] = off + advance_by;
        }
    }
}

__global__ __launch_bounds__(256) void axpby_kernel(int64_t n, float alpha, const float* __restrict__ x, float beta,
                                                    float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = alpha * x[i] + (beta != 0.f ? beta * y[i] : 0.f);
}

// 16-byte variant (n % 4 == 0, both pointers 16-byte aligned); beta == 0 never reads y (plain copy / scaled copy).
__global__ __launch_bounds__(256) void axpby4_kernel(int64_t n4, float alpha, const float4* __restrict__ x, float beta,
                                                     float4* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = x[i];
        float4 r = make_float4(alpha * a.x, alpha * a.y, alpha * a.z, alpha * a.w);
        if (beta != 0.f) {
            const float4 b = y[i];
            r.x += beta * b.x, r.y += beta * b.y, r.z += beta * b.z, r.w += beta * b.w;
        }
        y[i] = r;
    }
}

// One workgroup column (blockIdx.y) per job; elements of a job are spread over gridDim.x workgroups.
__global__ __launch_bounds__(256) void sum_parts_batch_kernel(const mmvae_sum_job* __restrict__ jobs) {
    const mmvae_sum_job j = jobs[blockIdx.y];
    const bool acc = j.flags & MMVAE_GEMM_ACCUMULATE;
    const bool vec = j.cols % 4 == 0;  // whole 16-byte groups per row (global accesses need no alignment on gfx950)
    if (vec) {
        const int c4n = j.cols / 4;
        const int64_t total = (int64_t)j.rows * c4n;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int r = (int)(i / c4n), c = (int)(i - (int64_t)r * c4n) * 4;
            const float* sp = j.src + (int64_t)r * j.ld_src + c;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            int p = 0;
            for (; p + 8 <= j.n_parts; p += 8) {  // 8 loads in flight, added in part order (160 row-chunk partials at K x B rows)
                f32x4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(sp + (int64_t)(p + u) * j.part_stride);
#pragma unroll
                for (int u = 0; u < 8; ++u) s += t[u];
            }
            for (; p < j.n_parts; ++p) s += *reinterpret_cast<const f32x4*>(sp + (int64_t)p * j.part_stride);
            f32x4 v = s * j.alpha;
            float* dp = j.dst + (int64_t)r * j.ld_dst + c;
            if (acc) v += *reinterpret_cast<const f32x4*>(dp);
            *reinterpret_cast<f32x4*>(dp) = v;
        }
    } else {
        const int64_t total = (int64_t)j.rows * j.cols;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int r = (int)(i / j.cols), c = (int)(i - (int64_t)r * j.cols);
            const float* sp = j.src + (int64_t)r * j.ld_src + c;
            float s = 0.f;
            for (int p = 0; p < j.n_parts; ++p) s += sp[(int64_t)p * j.part_stride];
            float v = s * j.alpha;
            float* dp = j.dst + (int64_t)r * j.ld_dst + c;
            if (acc) v += *dp;
            *dp = v;
        }
    }
}

__global__ __launch_bounds__(256) void scale_rows_kernel(int B, int N, const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ rs, float* __restrict__ y,
                                                         int64_t ldy) {
    const int64_t total = (int64_t)B * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / N), j = (int)(i - (int64_t)b * N);
        y[(int64_t)b * ldy + j] = x[(int64_t)b * ldx + j] * rs[b];
    }
}

inline int grid_for(int64_t n, int per_block, int cap) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

}  // namespace

extern "C" int mmvae_abi_version(void) { return MMVAE_ABI_VERSION; }
extern "C" const char* mmvae_build_arch(void) { return "gfx950"; }

extern "C" int mmvae_reparam_kl_fwd(int B, int Z, int K, const float* mu, const float* a_raw, const float* eps,
                                    float var_eps, float* std_out, float* z_out, float* kl_row, float* stat_row,
                                    mmvae_stream_t stream) {
    if (B <= 0 || Z <= 0 || K < 1 || !mu || !a_raw) return MMVAE_ERR_ARG;
    if (z_out && !eps) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(reparam_kl_fwd_kernel, dim3(ceil_div_i(B, 4)), dim3(256), 0, (hipStream_t)stream, B, Z, K, mu,
                       a_raw, eps, var_eps, std_out, z_out, kl_row, stat_row);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_reparam_kl_bwd(int B, int Z, int K, const float* mu, const float* sd, const float* eps,
                                    const float* dz, const float* dmu_extra, const float* dstd_extra,
                                    const float* dkl_row, const float* kl_scale_dev, float kl_scale_host, float var_eps,
                                    float* dmu, float* da_raw, mmvae_stream_t stream) {
    if (B <= 0 || Z <= 0 || K < 1 || !mu || !sd || !dmu || !da_raw) return MMVAE_ERR_ARG;
    if (dz && !eps) return MMVAE_ERR_ARG;
    const int64_t n = (int64_t)B * Z;
    MMVAE_LAUNCH(reparam_kl_bwd_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, B, Z, K,
                       mu, sd, eps, dz, dmu_extra, dstd_extra, dkl_row, kl_scale_dev, kl_scale_host, var_eps, dmu,
                       da_raw);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_mse_sum_fwd_bwd(int B, int G, const float* xhat, int64_t ldxhat, const float* x, int64_t ldx,
                                     float* se_row, float* dxhat, int64_t lddx, const float* gscale_dev,
                                     float gscale_host, mmvae_stream_t stream) {
    if (B <= 0 || G <= 0 || !xhat || !x || ldxhat < G || ldx < G) return MMVAE_ERR_ARG;
    if (!se_row && !dxhat) return MMVAE_ERR_ARG;
    if (dxhat && lddx < G) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(mse_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, B, G, xhat, ldxhat, x, ldx, se_row,
                       dxhat, lddx, gscale_dev, gscale_host);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_elbo_finalize(int B, int K, int T, const float* se_part, const float* kl_row,
                                   const float* stat_row, int Z, const float* kl_weight_dev, float kl_weight_host,
                                   float* out6, float* w_out, float* recon_row, mmvae_stream_t stream) {
    if (B <= 0 || K < 1 || K > ELBO_MAXK || T < 1 || !se_part || !out6 || !recon_row) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(elbo_rows_kernel, dim3(ceil_div_i(B, 4)), dim3(256), 0, (hipStream_t)stream, B, K, T, se_part,
                       recon_row, w_out);
    MMVAE_LAUNCH_CHECK();
    MMVAE_LAUNCH(elbo_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, B, recon_row, kl_row, stat_row, Z,
                       kl_weight_dev, kl_weight_host, out6);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_iwae_logratio(int B, int Z, int K, const float* std, const float* eps, const float* z, float* out,
                                   mmvae_stream_t stream) {
    if (B <= 0 || Z <= 0 || K < 1 || K > ELBO_MAXK || !std || !eps || !z || !out) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(iwae_logratio_kernel, dim3(ceil_div_i(K * B, 4)), dim3(256), 0, (hipStream_t)stream, B, Z, K, std, eps, z,
                 out);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_elbo_finalize_iwae(int B, int K, int T, const float* se_part, const float* logratio,
                                        const float* stat_row, int Z, const float* kl_weight_dev, float kl_weight_host,
                                        float* out6, float* w_out, float* rows3, mmvae_stream_t stream) {
    if (B <= 0 || K < 1 || K > ELBO_MAXK || T < 1 || !se_part || !logratio || !out6 || !rows3) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(elbo_rows_iwae_kernel, dim3(ceil_div_i(B, 4)), dim3(256), 0, (hipStream_t)stream, B, K, T, se_part,
                 logratio, kl_weight_dev, kl_weight_host / (float)B, rows3, w_out);
    MMVAE_LAUNCH_CHECK();
    MMVAE_LAUNCH(elbo_reduce_iwae_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, B, rows3, stat_row, Z,
                 kl_weight_dev, kl_weight_host, out6);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_iwae_bwd_terms(int B, int Z, int K, const float* kl_weight_dev, float kl_weight_host, const float* w,
                                    const float* z, const float* std, float* dz, float* dstd_extra,
                                    mmvae_stream_t stream) {
    if (B <= 0 || Z <= 0 || K < 1 || !w || !z || !std || !dz || !dstd_extra) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(iwae_bwd_terms_kernel, dim3(grid_for((int64_t)K * B * Z, 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                 B, Z, K, kl_weight_dev, kl_weight_host / (float)B, w, z, std, dz, dstd_extra);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_cross_entropy_sum(int B, int C, const float* logits, int64_t ld, const int64_t* labels,
                                       float* loss_rows, float* dlogits, int64_t ldd, const float* gscale_dev,
                                       float gscale_host, mmvae_stream_t stream) {
    if (B <= 0 || C <= 0 || !logits || !labels || ld < C) return MMVAE_ERR_ARG;
    if (!loss_rows && !dlogits) return MMVAE_ERR_ARG;
    if (dlogits && ldd < C) return MMVAE_ERR_ARG;
    // (16-byte row accesses need no alignment on gfx950: heads packed side by side in one logits matrix -- column
    // offsets and a leading dimension that are not multiples of 4 -- take the same kernel)
    if (C >= 1024 && C <= 256 * 4 * 8)  // wide heads: one workgroup per row, the row in registers
        MMVAE_LAUNCH(ce_rows_block_kernel<8>, dim3(B), dim3(256), 0, (hipStream_t)stream, B, C, logits, ld, labels,
                     loss_rows, dlogits, ldd, gscale_dev, gscale_host);
    else
        MMVAE_LAUNCH(ce_rows_kernel, dim3(ceil_div_i(B, 4)), dim3(256), 0, (hipStream_t)stream, B, C, logits, ld,
                     labels, loss_rows, dlogits, ldd, gscale_dev, gscale_host);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_cross_entropy_heads(int B, int H, int max_classes, const int32_t* col_dev, const int32_t* classes_dev,
                                         const float* logits, int64_t ld, const int64_t* labels, float* loss_rows,
                                         float* dlogits, int64_t ldd, float gscale, mmvae_stream_t stream) {
    if (B <= 0 || H <= 0 || H > 65535 || max_classes <= 0 || max_classes > 256 * 4 * 8 || !col_dev || !classes_dev ||
        !logits || !labels || (!loss_rows && !dlogits))
        return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(ce_heads_kernel<8>, dim3(B, H), dim3(256), 0, (hipStream_t)stream, B, col_dev, classes_dev, logits, ld,
                 labels, loss_rows, dlogits, ldd, gscale);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

// Diagnostics: one wall-clock stamp (100 MHz) into buf[slot] -- a marker launch between two launches of a captured
// program, for a timeline of the program as it runs WITHOUT a tracer attached (bench.py --stamps).
__global__ void debug_stamp_kernel(long long* __restrict__ buf, int slot) { buf[slot] = wall_clock64(); }
extern "C" int mmvae_debug_stamp(long long* buf, int slot, mmvae_stream_t stream) {
    if (!buf || slot < 0) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(debug_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, buf, slot);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_debug_occupy(int workgroups, int lds_bytes, int micros, float* sink, mmvae_stream_t stream) {
    if (workgroups <= 0 || workgroups > 256 || lds_bytes < 1024 || lds_bytes > 160 * 1024 || micros <= 0 ||
        micros > 20000)
        return MMVAE_ERR_ARG;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds_bytes) != hipSuccess)
        return MMVAE_ERR_LAUNCH;
    MMVAE_LAUNCH(occupy_kernel, dim3(workgroups), dim3(256), (size_t)lds_bytes, (hipStream_t)stream,
                 (long long)micros * 100, sink);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_sum_f32(int64_t n, const float* v, float* out, int accumulate, mmvae_stream_t stream) {
    if (n < 0 || (n > 0 && !v) || !out) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, v, out, accumulate);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_sum_rows_f32(int H, int64_t n, const float* v, int64_t ld, float* out_each, float* out_total,
                                  mmvae_stream_t stream) {
    if (H <= 0 || n < 0 || (n > 0 && !v) || ld < n || (!out_each && !out_total)) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(sum_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, H, n, v, ld, out_each, out_total);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int64_t mmvae_sqnorm_partials(int64_t n) { return n > 0 ? ceil_div_l(n, SQN_CHUNK) : 0; }

extern "C" int mmvae_grad_sqnorm(int64_t n, const float* grad, float* partials, mmvae_stream_t stream) {
    if (n <= 0 || !grad || !partials) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(sqnorm_kernel, dim3((unsigned)mmvae_sqnorm_partials(n)), dim3(256), 0, (hipStream_t)stream, n,
                       grad, partials);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_adam_prepare(int64_t n_partials, const float* partials, float max_norm, float grad_scale,
                                  float beta1, float beta2, float* state, unsigned flags, mmvae_stream_t stream) {
    if (n_partials < 0 || !state) return MMVAE_ERR_ARG;
    if ((flags & MMVAE_PREPARE_NORM) && n_partials > 0 && !partials) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(adam_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n_partials, partials, max_norm,
                       grad_scale, beta1, beta2, state, flags);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_grad_sqnorm_ranges_prepare(int n_ranges, const float* const* grads, const int64_t* lens,
                                                float* partials, unsigned* ticket, int64_t n_partials_all,
                                                const float* partials_all, float max_norm, float grad_scale,
                                                float beta1, float beta2, float* state, unsigned flags,
                                                mmvae_stream_t stream) {
    if (n_ranges < 1 || n_ranges > 4 || !grads || !lens || !partials || !ticket || !partials_all || !state ||
        n_partials_all <= 0)
        return MMVAE_ERR_ARG;
    SqRanges r = {};
    r.nr = n_ranges;
    int total = 0;
    for (int i = 0; i < n_ranges; ++i) {
        if (!grads[i] || lens[i] <= 0) return MMVAE_ERR_ARG;
        r.g[i] = grads[i];
        r.n[i] = lens[i];
        r.nb[i] = (int)mmvae_sqnorm_partials(lens[i]);
        total += r.nb[i];
    }
    MMVAE_LAUNCH(sqnorm_ranges_prepare_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, r, partials, ticket,
                 n_partials_all, partials_all, max_norm, grad_scale, beta1, beta2, state, flags | MMVAE_PREPARE_NORM);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                               const float* state, float lr, float beta1, float beta2, float eps, float weight_decay,
                               float grad_scale, mmvae_stream_t stream) {
    if (n <= 0 || !param || !grad || !exp_avg || !exp_avg_sq || !state) return MMVAE_ERR_ARG;
    const int vec = aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq);
    return launch_adam_step(n, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, state, lr, beta1, beta2, eps,
                            weight_decay, grad_scale, vec, 0, (const float*)nullptr, (float*)nullptr);
}

extern "C" int mmvae_adam_step_multi(int n_arenas, const mmvae_adam_arena* arenas_dev, int64_t max_n,
                                     mmvae_stream_t stream) {
    if (n_arenas < 1 || n_arenas > 64 || !arenas_dev || max_n <= 0) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(adam_step_multi_kernel, dim3(grid_for(max_n, 1024, 4096), n_arenas), dim3(256), 0, (hipStream_t)stream,
                 arenas_dev);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_adam_step_copy(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                    const float* state, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, float grad_scale, int copy_n, const float* copy_src,
                                    float* copy_dst, mmvae_stream_t stream) {
    if (n <= 0 || !param || !grad || !exp_avg || !exp_avg_sq || !state) return MMVAE_ERR_ARG;
    if (copy_n < 0 || copy_n > 65536 || (copy_n > 0 && (!copy_src || !copy_dst))) return MMVAE_ERR_ARG;
    const int vec = aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq);
    return launch_adam_step(n, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, state, lr, beta1, beta2, eps,
                            weight_decay, grad_scale, vec, copy_n, copy_src, copy_dst);
}

extern "C" int mmvae_adam_step_jobs(int n_jobs, const mmvae_adam_job* jobs_dev, float* param, const float* grad,
                                    float* exp_avg, float* exp_avg_sq, const float* state, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, float grad_scale, mmvae_stream_t stream) {
    if (n_jobs <= 0 || !jobs_dev || !param || !grad || !exp_avg || !exp_avg_sq || !state) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(adam_step_jobs_kernel, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, param, grad, exp_avg,
                 exp_avg_sq, state, lr, beta1, beta2, eps, weight_decay, grad_scale);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_grad_sqnorm_jobs(int n_jobs, const mmvae_adam_job* jobs_dev, const float* grad, float* partials,
                                      mmvae_stream_t stream) {
    if (n_jobs <= 0 || !jobs_dev || !grad || !partials) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(sqnorm_jobs_kernel, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, grad, partials);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_jobs_pack(int n_jobs, const mmvae_adam_job* jobs_dev, const float* arena, float* staging,
                               mmvae_stream_t stream) {
    if (n_jobs <= 0 || !jobs_dev || !arena || !staging) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(jobs_pack_kernel<true>, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev,
                 const_cast<float*>(arena), staging);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_jobs_unpack(int n_jobs, const mmvae_adam_job* jobs_dev, float* arena, const float* staging,
                                 mmvae_stream_t stream) {
    if (n_jobs <= 0 || !jobs_dev || !arena || !staging) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(jobs_pack_kernel<false>, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, arena,
                 const_cast<float*>(staging));
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_grad_zero_flagged_jobs(int n_jobs, const mmvae_adam_job* jobs_dev, float* grad, mmvae_stream_t stream) {
    if (n_jobs <= 0 || !jobs_dev || !grad) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(zero_flagged_jobs_kernel, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, grad);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_philox_keep_mask(int64_t n, float p_drop, uint8_t* mask, uint64_t* rng_state, uint64_t stream_id,
                                      int advance, mmvae_stream_t stream) {
    if (n <= 0 || !mask || !rng_state || p_drop < 0.f || p_drop >= 1.f) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(philox_mask_kernel, dim3(grid_for((n + 3) / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, n,
                       p_drop, mask, rng_state, stream_id);
    MMVAE_LAUNCH_CHECK();
    if (advance) {
        MMVAE_LAUNCH(philox_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state,
                           (uint64_t)((n + 3) / 4));
        MMVAE_LAUNCH_CHECK();
    }
    return MMVAE_OK;
}

extern "C" int mmvae_philox_normal(int64_t n, float* out, uint64_t* rng_state, uint64_t stream_id, int advance,
                                   mmvae_stream_t stream) {
    if (n <= 0 || !out || !rng_state) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(philox_normal_kernel, dim3(grid_for((n + 3) / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                       n, out, rng_state, stream_id);
    MMVAE_LAUNCH_CHECK();
    if (advance) {
        MMVAE_LAUNCH(philox_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state,
                           (uint64_t)((n + 3) / 4));
        MMVAE_LAUNCH_CHECK();
    }
    return MMVAE_OK;
}

extern "C" int mmvae_philox_fill_jobs(int n_jobs, const mmvae_philox_job* jobs_dev, int64_t max_n, uint64_t* rng_state,
                                      mmvae_stream_t stream) {
    if (n_jobs <= 0 || n_jobs > 65535 || !jobs_dev || max_n <= 0 || !rng_state) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(philox_fill_jobs_kernel, dim3(grid_for((max_n + 3) / 4, 256, 2048), n_jobs), dim3(256), 0,
                 (hipStream_t)stream, jobs_dev, rng_state, (uint64_t)0, (unsigned*)nullptr);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_philox_fill_jobs_advance(int n_jobs, const mmvae_philox_job* jobs_dev, int64_t max_n,
                                              uint64_t* rng_state, uint64_t advance_by, unsigned* ticket,
                                              mmvae_stream_t stream) {
    if (n_jobs <= 0 || n_jobs > 65535 || !jobs_dev || max_n <= 0 || !rng_state || !ticket || advance_by == 0)
        return MMVAE_ERR_ARG;
    // 128 workgroups per job (grid-stride): every workgroup takes a ticket on ONE word, and 2048 x jobs same-address
    // atomics took longer (19.7 us) than the launch they save
    MMVAE_LAUNCH(philox_fill_jobs_kernel, dim3(grid_for((max_n + 3) / 4, 256, 128), n_jobs), dim3(256), 0,
                 (hipStream_t)stream, jobs_dev, rng_state, advance_by, ticket);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_philox_advance(uint64_t* rng_state, uint64_t by, mmvae_stream_t stream) {
    if (!rng_state) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(philox_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state, by);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

// CSR -> dense rows.  One workgroup per (row, column chunk): the chunk is zero-filled with 16-byte
// stores, then the row's stored elements that fall into the chunk are scattered over it.  Both phases are coalesced
// streams (indices of a row are contiguous); the barrier between them orders the two stores to an element inside
// the workgroup.  Out-of-range column indices are dropped (never written), so a malformed matrix cannot fault.
constexpr int CSR_CHUNK = 8192;  // columns per workgroup

template <typename I>  // index type of the CSR arrays: int64 (torch's default) or int32 (what scipy-built tensors carry)
__global__ __launch_bounds__(256) void csr_to_dense_kernel(int G, int64_t nnz, const I* __restrict__ crow,
                                                           const I* __restrict__ col,
                                                           const float* __restrict__ val, float* __restrict__ out,
                                                           int64_t ldo, int vec) {
    const int row = blockIdx.y;
    const int c0 = blockIdx.x * CSR_CHUNK;
    const int c1 = min(c0 + CSR_CHUNK, G);
    float* o = out + (int64_t)row * ldo;
    if (vec) {  // 16-byte zero fill (no alignment needed on gfx950)
        const int n4 = (c1 - c0) >> 2;
        f32x4* o4 = reinterpret_cast<f32x4*>(o + c0);
        for (int i = threadIdx.x; i < n4; i += 256) o4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = c0 + 4 * n4 + threadIdx.x; c < c1; c += 256) o[c] = 0.f;
    } else {
        for (int c = c0 + threadIdx.x; c < c1; c += 256) o[c] = 0.f;
    }
    __syncthreads();
    int64_t beg = (int64_t)crow[row], end = (int64_t)crow[row + 1];
    beg = beg < 0 ? 0 : beg;
    end = end > nnz ? nnz : end;  // a malformed row pointer cannot read past the index / value arrays
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
        const int64_t c = (int64_t)col[i];
        if (c >= c0 && c < c1) o[c] = val[i];
    }
}

extern "C" int mmvae_csr_to_dense_f32(int B, int G, int64_t nnz, const int64_t* crow_indices,
                                      const int64_t* col_indices, const float* values, float* out, int64_t ldo,
                                      mmvae_stream_t stream) {
    if (B <= 0 || G <= 0 || nnz < 0 || !crow_indices || !out || ldo < G) return MMVAE_ERR_ARG;
    if (nnz > 0 && (!col_indices || !values)) return MMVAE_ERR_ARG;
    if (B > 65535) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(csr_to_dense_kernel<int64_t>, dim3(ceil_div_i(G, CSR_CHUNK), B), dim3(256), 0, (hipStream_t)stream, G,
                       nnz, crow_indices, col_indices, values, out, ldo, 1);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_csr_to_dense_i32_f32(int B, int G, int64_t nnz, const int32_t* crow_indices,
                                          const int32_t* col_indices, const float* values, float* out, int64_t ldo,
                                          mmvae_stream_t stream) {
    if (B <= 0 || G <= 0 || nnz < 0 || nnz > 0x7fffffffLL || !crow_indices || !out || ldo < G) return MMVAE_ERR_ARG;
    if (nnz > 0 && (!col_indices || !values)) return MMVAE_ERR_ARG;
    if (B > 65535) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(csr_to_dense_kernel<int32_t>, dim3(ceil_div_i(G, CSR_CHUNK), B), dim3(256), 0, (hipStream_t)stream, G,
                       nnz, crow_indices, col_indices, values, out, ldo, 1);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

// ---- f1 experiment: CSR x dense^T product of the first layer WITHOUT densifying.
// y[b, :] = sum over the stored elements (g, v) of row b of v * Wt[g, :]  (+ bias), Wt = the layer's weight TRANSPOSED
// ([G, N], N contiguous: a stored element then touches one contiguous 4 N-byte row of Wt).  One workgroup per cell row
// and 1024-column chunk; thread t owns columns 4 t .. 4 t + 3 of the chunk; the row's (column, value) pairs are staged
// through LDS 256 at a time and every pair costs each wave one 1-KB coalesced load of Wt: the kernel is a gather from
// L2 / Infinity Cache / HBM of nnz x 4 N bytes -- at 10 % density ~10x the bytes the dense GEMM streams, which is why
// the dense bf16x3 GEMM over the densified batch stays the product path (measured: profiles/r2_sparse_input.txt).
// Accumulation in stored order: bitwise reproducible.
template <typename I>
__global__ __launch_bounds__(256) void csr_spmm_rows_kernel(int N, int G, int64_t nnz, const I* __restrict__ crow,
                                                            const I* __restrict__ col, const float* __restrict__ val,
                                                            const float* __restrict__ Wt, int64_t ldwt,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int64_t ldy) {
    __shared__ int s_col[256];
    __shared__ float s_val[256];
    const int row = blockIdx.y;
    const int c = blockIdx.x * 1024 + 4 * threadIdx.x;
    const bool live = c + 3 < N;  // N % 4 == 0 is required: a 16-byte group is inside the row or outside it
    int64_t beg = (int64_t)crow[row], end = (int64_t)crow[row + 1];
    beg = beg < 0 ? 0 : beg;
    end = end > nnz ? nnz : end;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t base = beg; base < end; base += 256) {
        const int n = (int)((end - base) < 256 ? (end - base) : 256);
        __syncthreads();
        if ((int)threadIdx.x < n) {
            const int64_t g = (int64_t)col[base + threadIdx.x];
            const bool ok = g >= 0 && g < G;  // a malformed index contributes nothing
            s_col[threadIdx.x] = ok ? (int)g : 0;
            s_val[threadIdx.x] = ok ? val[base + threadIdx.x] : 0.f;
        }
        __syncthreads();
        if (live) {
            int i = 0;
            for (; i + 8 <= n; i += 8) {  // 8 gathered rows in flight per thread
                f32x4 w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const f32x4*>(Wt + (int64_t)s_col[i + u] * ldwt + c);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += w[u] * s_val[i + u];
            }
            for (; i < n; ++i) acc += *reinterpret_cast<const f32x4*>(Wt + (int64_t)s_col[i] * ldwt + c) * s_val[i];
        }
    }
    if (live) {
        if (bias) acc += *reinterpret_cast<const f32x4*>(bias + c);
        *reinterpret_cast<f32x4*>(y + (int64_t)row * ldy + c) = acc;
    }
}

extern "C" int mmvae_csr_spmm_wt_i32_f32(int B, int N, int G, int64_t nnz, const int32_t* crow_indices,
                                         const int32_t* col_indices, const float* values, const float* Wt, int64_t ldwt,
                                         const float* bias, float* y, int64_t ldy, mmvae_stream_t stream) {
    if (B <= 0 || B > 65535 || N <= 0 || N % 4 != 0 || G <= 0 || nnz < 0 || !crow_indices || !Wt || !y) return MMVAE_ERR_ARG;
    if (ldwt < N || ldy < N || (nnz > 0 && (!col_indices || !values))) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(csr_spmm_rows_kernel<int32_t>, dim3(ceil_div_i(N, 1024), B), dim3(256), 0, (hipStream_t)stream, N, G, nnz,
                 crow_indices, col_indices, values, Wt, ldwt, bias, y, ldy);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_axpby(int64_t n, float alpha, const float* x, float beta, float* y, mmvae_stream_t stream) {
    if (n <= 0 || !x || !y) return MMVAE_ERR_ARG;
    if (n % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)
        MMVAE_LAUNCH(axpby4_kernel, dim3(grid_for(n / 4, 256, STREAM_GRID_CAP)), dim3(256), 0, (hipStream_t)stream, n / 4,
                           alpha, reinterpret_cast<const float4*>(x), beta, reinterpret_cast<float4*>(y));
    else
        MMVAE_LAUNCH(axpby_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, n, alpha, x,
                           beta, y);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

// Table upload by a KERNEL: `host_src` is page-locked host memory (hipHostMalloc / torch pin_memory) that the device reads
// in place.  A hipMemcpyAsync of the same bytes goes to the SDMA engine, and on this runtime an SDMA copy enqueued behind a
// captured program makes the HOST wait for that program (0.5 ms per step of the conditional programs, whose host side is
// their limit; with HSA_ENABLE_SDMA=0 the same loop is device-bound) -- a kernel launch is just the next packet.
__global__ __launch_bounds__(256) void upload_words_kernel(int64_t n, const int32_t* __restrict__ src,
                                                           int32_t* __restrict__ dst, int vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int4* s4 = reinterpret_cast<const int4*>(src);
        int4* d4 = reinterpret_cast<int4*>(dst);
        for (int64_t i = t0; i < n / 4; i += stride) d4[i] = s4[i];
        for (int64_t i = n / 4 * 4 + t0; i < n; i += stride) dst[i] = src[i];
    } else {
        for (int64_t i = t0; i < n; i += stride) dst[i] = src[i];
    }
}

extern "C" int mmvae_upload_words(int64_t n_words, const void* host_src, void* dst, mmvae_stream_t stream) {
    if (n_words <= 0 || !host_src || !dst) return MMVAE_ERR_ARG;
    void* dev_src = nullptr;  // (fails for pageable memory: the caller must hand page-locked, mapped memory)
    if (hipHostGetDevicePointer(&dev_src, const_cast<void*>(host_src), 0) != hipSuccess || !dev_src) {
        (void)hipGetLastError();
        return MMVAE_ERR_ARG;
    }
    const int vec = ((reinterpret_cast<uintptr_t>(dev_src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    MMVAE_LAUNCH(upload_words_kernel, dim3(grid_for((n_words + 3) / 4, 256, 256)), dim3(256), 0, (hipStream_t)stream,
                 n_words, static_cast<const int32_t*>(dev_src), static_cast<int32_t*>(dst), vec);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_sum_parts_batch(int n_jobs, const mmvae_sum_job* jobs, int64_t max_elems, mmvae_stream_t stream) {
    if (n_jobs <= 0 || n_jobs > 65535 || !jobs || max_elems < 0) return MMVAE_ERR_ARG;
    // workgroups per job: sized for the largest job, 64 when the caller cannot say
    const int gx = max_elems > 0 ? grid_for((max_elems + 3) / 4, 256, 2048) : 64;  // one 16-byte group per thread
    MMVAE_LAUNCH(sum_parts_batch_kernel, dim3(gx < 64 ? (gx < 1 ? 1 : gx) : gx, n_jobs), dim3(256), 0, (hipStream_t)stream,
                 jobs);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

// Row-weighted column sums of a wide matrix, read once in 16-byte accesses: partials[chunk][c] = sum over the chunk's
// WCS_ROWS rows of w[r] * x[r][c] (fp32 chain in row order inside a wave's share, the four waves' shares added in wave
// order: reproducible).  A workgroup = 256 columns x one row chunk; wave v takes rows v, v + 4, ... of the chunk, eight
// rows in flight.  (K-sample ELBO: the decoder bias's gradient is w^T dP; the pass that scaled dP in place for it read
// and wrote 2 x 410 MB at C3 -- 350 us.)
constexpr int WCS_ROWS = 256, WCS_COLS = 256;
__global__ __launch_bounds__(256) void weighted_colsum_kernel(int B, int N, const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ w, float* __restrict__ partials,
                                                              int vec) {
    __shared__ f32x4 red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * WCS_COLS + 4 * lane;
    const int r_begin = blockIdx.y * WCS_ROWS, r_end = min(r_begin + WCS_ROWS, B);
    const bool whole = vec && c + 3 < N;  // (vec == 0: rows that are not 16-byte groups -- odd toy shapes -- go element-wise)
    const int cc = whole ? c : 0;  // (a straddling group of 4 is handled element-wise below)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = r_begin + wave; vec && r0 < r_end; r0 += 32) {
        f32x4 v[8];
        float ws[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // clamped, unconditional loads: eight rows in flight
            const int r = min(r0 + 4 * i, r_end - 1);
            ws[i] = (r0 + 4 * i < r_end) ? w[r] : 0.f;
            v[i] = *reinterpret_cast<const f32x4*>(x + (int64_t)r * ldx + cc);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(ws[i], v[i][e], acc[e]);
    }
    if (!whole) {  // the last, partial group of columns: scalar
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = r_begin + wave; r < r_end; r += 4)
            for (int e = 0; e < 4; ++e)
                if (c + e < N) acc[e] = fmaf(w[r], x[(int64_t)r * ldx + c + e], acc[e]);
    }
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += red[v][lane][e];
        float* out = partials + (int64_t)blockIdx.y * N + c;
        if (whole && (((uintptr_t)out) & 15u) == 0) {
            *reinterpret_cast<f32x4*>(out) = acc;
        } else {
            for (int e = 0; e < 4; ++e)
                if (c + e < N) out[e] = acc[e];
        }
    }
}

extern "C" int mmvae_weighted_colsum_chunks(int B) { return B > 0 ? (B + WCS_ROWS - 1) / WCS_ROWS : 0; }

extern "C" int mmvae_weighted_colsum_f32(int B, int N, const float* x, int64_t ldx, const float* row_weight, float* partials,
                                         mmvae_stream_t stream) {
    if (B <= 0 || N <= 0 || !x || !row_weight || !partials || ldx < N) return MMVAE_ERR_ARG;
    const int vec = !((((uintptr_t)x) & 15u) || (ldx & 3));  // 16-byte row groups
    MMVAE_LAUNCH(weighted_colsum_kernel, dim3((N + WCS_COLS - 1) / WCS_COLS, (B + WCS_ROWS - 1) / WCS_ROWS), dim3(256), 0,
                 (hipStream_t)stream, B, N, x, ldx, row_weight, partials, vec);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_scale_rows(int B, int N, const float* x, int64_t ldx, const float* row_scale, float* y,
                                int64_t ldy, mmvae_stream_t stream) {
    if (B <= 0 || N <= 0 || !x || !row_scale || !y || ldx < N || ldy < N) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(scale_rows_kernel, dim3(grid_for((int64_t)B * N, 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                       B, N, x, ldx, row_scale, y, ldy);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}
