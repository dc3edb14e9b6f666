// Weight gradient of the expert encoder's first layer from a SPARSE batch (reference: the datapipe yields
// CSR batches of log1p-normalised counts, ~5-10 % of the entries stored -- data/local/cellxgene_datapipe.py:169-193 --
// and the first Linear's autograd forms dW = dY^T x over the densified batch, modules/base/components.py:276).
//
//   dW[o][g] = sum over the cells r with x[r][g] != 0 of dY[r][o] * x[r][g]
//
// Unlike the forward product (which gathers 4-KB rows of the 82-MB weight per stored entry and loses to the dense GEMM,
// profiles/r2_sparse_input.txt), this product gathers rows of dY -- 2 MB, of which a workgroup's 64-output slice fits
// the LDS.  The batch is held gene-major (ELL: every gene owns `cap` slots of (cell, value), cells ascending; built from
// the dense batch in one pass, no prefix sum); a wave owns one gene at a time, lane = output: one LDS read + one FMA
// per stored entry and 64 outputs, the entry itself arrives by scalar loads.  fp32 FMA chain in cell order: bitwise
// reproducible.
#include "common.h"

namespace {

constexpr int SP_OS = 64;      // outputs per workgroup slice (= lanes)
constexpr int SP_GT = 64;      // genes per output tile (staged through LDS so that the stores run along genes)
constexpr int SP_WAVES = 16;
constexpr int SP_THREADS = SP_WAVES * 64;

// ---- dense batch [B, G] -> gene-major ELL.  Workgroup: 64 genes x 8 cell groups of 32 cells (256 cells per round, the
// values held in registers); one pass over x.  A gene's list is padded with (offset 0, value 0) to a multiple of 8 entries:
// the consumer reads it in groups of 8.  `rows` holds cell * 64: the consumer's LDS index of that cell's dY row.
__global__ __launch_bounds__(512) void ell_from_dense_kernel(int B, int G, const float* __restrict__ x, int64_t ldx,
                                                             int cap, int32_t* __restrict__ rows,
                                                             float* __restrict__ vals, int32_t* __restrict__ cnt) {
    __shared__ int counts[8][64];
    const int gl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int g = blockIdx.x * 64 + gl;
    const int gc = g < G ? g : G - 1;
    int base = 0;
    for (int r0 = 0; r0 < B; r0 += 256) {
        float v[32];
        int c = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {  // (clamped, unconditional loads: all 32 in flight)
            const int r = r0 + rg * 32 + i;
            const float t = x[(int64_t)(r < B ? r : B - 1) * ldx + gc];
            v[i] = (g < G && r < B) ? t : 0.f;
            c += v[i] != 0.f;
        }
        counts[rg][gl] = c;
        __syncthreads();
        int off = base, total = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = counts[q][gl];
            if (q < rg) off += n;
            total += n;
        }
        if (g < G) {
            int32_t* rp = rows + (int64_t)g * cap;
            float* vp = vals + (int64_t)g * cap;
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (v[i] != 0.f) {
                    rp[off] = (r0 + rg * 32 + i) * SP_OS;
                    vp[off] = v[i];
                    ++off;
                }
        }
        base += total;
        __syncthreads();
    }
    if (rg == 0 && g < G) {
        cnt[g] = base;
        for (int k = base; k < ((base + 7) & ~7); ++k) {
            rows[(int64_t)g * cap + k] = 0;
            vals[(int64_t)g * cap + k] = 0.f;
        }
    }
}

// ---- dW[M, G] = dY^T . x from the ELL batch.  blockIdx.x = output slice + n_slices * gene partition.
__global__ __launch_bounds__(SP_THREADS) void dw_sparse_ell_kernel(int B, int G, int M, const float* __restrict__ dY,
                                                                   int64_t ldy, const int32_t* __restrict__ rows,
                                                                   const float* __restrict__ vals,
                                                                   const int32_t* __restrict__ cnt, int cap,
                                                                   float* __restrict__ dW, int64_t ldw, int n_parts) {
    extern __shared__ float dYs[];             // [B][64]: this slice of dY
    __shared__ float tile[SP_OS][SP_GT + 1];   // [output][gene of the tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_slices = (M + SP_OS - 1) / SP_OS;
    const int slice = blockIdx.x % n_slices, part = blockIdx.x / n_slices;
    const int o0 = slice * SP_OS;
    for (int e = tid; e < B * SP_OS; e += SP_THREADS) {
        const int r = e >> 6, c = e & 63;
        dYs[e] = (o0 + c < M) ? dY[(int64_t)r * ldy + o0 + c] : 0.f;
    }
    __syncthreads();
    const int n_tiles = (G + SP_GT - 1) / SP_GT;
    for (int t = part; t < n_tiles; t += n_parts) {
        const int g0 = t * SP_GT;
#pragma unroll
        for (int q = 0; q < SP_GT / SP_WAVES; ++q) {
            const int gl = wave + SP_WAVES * q, g = g0 + gl;
            float acc0 = 0.f, acc1 = 0.f;
            if (g < G) {
                const int n = cnt[g];
                // (wave-uniform addresses: the entries arrive by scalar loads, eight at a time; the list is zero-padded)
                const int4* __restrict__ rp = reinterpret_cast<const int4*>(rows + (int64_t)g * cap);
                const f32x4* __restrict__ vp = reinterpret_cast<const f32x4*>(vals + (int64_t)g * cap);
                for (int k = 0; k < n; k += 8) {
                    const int4 ra = rp[k >> 2], rb = rp[(k >> 2) + 1];
                    const f32x4 va = vp[k >> 2], vb = vp[(k >> 2) + 1];
                    const float d0 = dYs[ra.x + lane], d1 = dYs[ra.y + lane], d2 = dYs[ra.z + lane], d3 = dYs[ra.w + lane];
                    const float d4 = dYs[rb.x + lane], d5 = dYs[rb.y + lane], d6 = dYs[rb.z + lane], d7 = dYs[rb.w + lane];
                    acc0 = fmaf(va[0], d0, acc0);
                    acc1 = fmaf(va[1], d1, acc1);
                    acc0 = fmaf(va[2], d2, acc0);
                    acc1 = fmaf(va[3], d3, acc1);
                    acc0 = fmaf(vb[0], d4, acc0);
                    acc1 = fmaf(vb[1], d5, acc1);
                    acc0 = fmaf(vb[2], d6, acc0);
                    acc1 = fmaf(vb[3], d7, acc1);
                }
            }
            tile[lane][gl] = acc0 + acc1;
        }
        __syncthreads();
        for (int e = tid; e < SP_OS * SP_GT; e += SP_THREADS) {
            const int o = e >> 6, gq = e & 63;
            if (o0 + o < M && g0 + gq < G) dW[(int64_t)(o0 + o) * ldw + g0 + gq] = tile[o][gq];
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int mmvae_ell_from_dense_f32(int B, int G, const float* x, int64_t ldx, int cap, int32_t* rows, float* vals,
                                        int32_t* cnt, mmvae_stream_t stream) {
    if (B < 1 || G < 1 || !x || ldx < G || cap < ((B + 7) & ~7) || (cap & 7) || !rows || !vals || !cnt) return MMVAE_ERR_ARG;
    MMVAE_LAUNCH(ell_from_dense_kernel, dim3((G + 63) / 64), dim3(512), 0, (hipStream_t)stream, B, G, x, ldx, cap, rows,
                 vals, cnt);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

extern "C" int mmvae_dw_sparse_ell_f32(int B, int G, int M, const float* dY, int64_t ldy, const int32_t* rows,
                                       const float* vals, const int32_t* cnt, int cap, float* dW, int64_t ldw,
                                       mmvae_stream_t stream) {
    if (B < 1 || G < 1 || M < 1 || !dY || ldy < M || !rows || !vals || !cnt || cap < 8 || (cap & 7) || !dW || ldw < G)
        return MMVAE_ERR_ARG;
    const size_t lds = (size_t)B * SP_OS * sizeof(float);
    if (lds > 140 * 1024) return MMVAE_ERR_ARG;  // the dY slice must fit the LDS beside the output tile
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dw_sparse_ell_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess)
            return MMVAE_ERR_LAUNCH;
        attr = true;
    }
    const int n_slices = (M + SP_OS - 1) / SP_OS;
    const int n_tiles = (G + SP_GT - 1) / SP_GT;
    int n_parts = 256 / n_slices;  // one workgroup per CU
    if (n_parts < 1) n_parts = 1;
    if (n_parts > n_tiles) n_parts = n_tiles;
    MMVAE_LAUNCH(dw_sparse_ell_kernel, dim3(n_slices * n_parts), dim3(SP_THREADS), lds, (hipStream_t)stream, B, G, M, dY,
                 ldy, rows, vals, cnt, cap, dW, ldw, n_parts);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}
