// Pre-split operands of the bf16x3 GEMMs (r3): the exact three-way bf16 split of an fp32 matrix done ONCE by its
// producer (mmvae_split_planes_f32, or a fused producer epilogue) instead of once per tile that reads it, and the
// instantiations of the wave-specialised kernel (gemm_dev.h) whose stagers move such planes global -> LDS by LDS-DMA.
//
// Replaces the same nn.Linear forward / backward dispatches as gemm_f32.hip (components.py:276 and its autograd): the
// products, their order and therefore the results are bit-identical to the in-kernel split -- only where the split
// happens changes.
#include "gemm_dev.h"

namespace {

// fp32 [rows, cols] -> planes[p][row][col] (bf16), p = 0 (top 16 bits), 1, 2 (residuals): a = p0 + p1 + p2 exactly.
// One thread per 8 consecutive columns: two 16-byte loads, three 16-byte stores (HBM-bound: 4 B read, 6 B written per
// element).  The arithmetic is x3_pack4_lean's -- the same split the GEMM stagers perform.  A column count that is not a
// multiple of 8 (the reference's 60 530 / 52 437 genes): the planes' leading dimension is rounded up to 8 and the last
// group of a row is read element by element, its columns beyond `cols` written as zeros -- the rows-contiguous DMA of
// the planes kernels fetches whole 16-byte groups and multiplies those zeros into columns the epilogue never stores.
__global__ __launch_bounds__(256) void split_planes_kernel(int rows, int cols, int groups, const float* __restrict__ src,
                                                           int64_t ld_src, unsigned short* __restrict__ planes,
                                                           int64_t ld, int64_t pstride) {
    const int64_t total = (int64_t)rows * groups;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / groups), gq = (int)(idx - (int64_t)row * groups);
        const float* sp = src + (int64_t)row * ld_src + 8 * gq;
        f32x4 v0, v1;
        if (8 * gq + 8 <= cols) {
            v0 = *reinterpret_cast<const f32x4*>(sp);
            v1 = *reinterpret_cast<const f32x4*>(sp + 4);
        } else {
            const int left = cols - 8 * gq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v0[j] = j < left ? sp[j] : 0.f;
                v1[j] = 4 + j < left ? sp[4 + j] : 0.f;
            }
        }
        uint2 a[3], b[3];
        x3_pack4_lean(f32x2{v0[0], v0[1]}, f32x2{v0[2], v0[3]}, a);
        x3_pack4_lean(f32x2{v1[0], v1[1]}, f32x2{v1[2], v1[3]}, b);
        unsigned short* dp = planes + (int64_t)row * ld + 8 * gq;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint4*>(dp + p * pstride) = make_uint4(a[p].x, a[p].y, b[p].x, b[p].y);
    }
}

}  // namespace

extern "C" int mmvae_split_planes_f32(int rows, int cols, const float* src, int64_t ld_src, uint16_t* planes, int64_t ld,
                                      int64_t plane_stride, mmvae_stream_t stream) {
    if (rows <= 0 || cols <= 0 || !src || !planes) return MMVAE_ERR_ARG;
    const int groups = (cols + 7) / 8;  // (the last group of a row may be partial: zero-filled up to ld)
    if (ld_src < cols || ld < 8 * (int64_t)groups || ld % 8 != 0 || plane_stride % 8 != 0 ||
        plane_stride < (int64_t)rows * ld)
        return MMVAE_ERR_ARG;
    if (!aligned16(planes)) return MMVAE_ERR_ARG;
    const int64_t total = (int64_t)rows * groups;
    // <= 3 workgroups per CU: the pass is HBM-bound long before that, and a grid that fills every wave slot starves the
    // latency-bound kernels the engine runs beside it
    int blocks = (int)((total + 255) / 256);
    if (blocks > 768) blocks = 768;
    MMVAE_LAUNCH(split_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows, cols, groups, src, ld_src,
                 reinterpret_cast<unsigned short*>(planes), ld, plane_stride);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

#if MMVAE_X3_STAMPS
extern "C" int mmvae_debug_x3p_stamps(long long* out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_x3_stamps), sizeof(long long) * 32) == hipSuccess ? 0 : 1;
}
extern "C" int mmvae_debug_x3p_trace(long long* out, int n_blocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x3_trace), sizeof(long long) * 4 * n_blocks) == hipSuccess ? 0 : 1;
}
#endif

namespace mmvae_detail {

// Which (layout, operand source) pairs have a kernel: TN with both operands pre-split (the weight gradients: dY^T . x and
// dP^T . h -- both operands are activations / gradients whose producers can write planes), NT and NN with a pre-split A
// (x, dP) against the fp32 weights (split in the kernel: the optimiser rewrites them every step).
bool x3w_planes_combo(int layout, bool a_pl, bool b_pl) {
    // (TN with only B pre-split: dW = dP^T h with the small operand h from its layer tail -- on the 160x256 tile B is
    // 256 of the 416 rows the stagers would otherwise split per k-tile)
    // (late r5: TN with only A pre-split -- dW = dY^T x with the small dY from its column kernel and the batch x read as
    // fp32, split by the stagers: no split pass over x at all)
    if (layout == MMVAE_GEMM_TN) return a_pl || b_pl;
    // (r5) NN with a pre-split B: the weights of the last decoder layer, split ONCE per step for programs whose products
    // read them from many row tiles (K-sample programs: 20 row tiles of 256 at C3) -- dX = dP . W with dP fp32
    if (layout == MMVAE_GEMM_NN) return a_pl != b_pl;
    return a_pl && !b_pl;
}

int launch_x3w_planes(int layout, int tile_id, bool a_pl, bool b_pl, int epi, const GemmArgs& g0, int nwork, int slots,
                      hipStream_t s) {
    // (the fused reconstruction launch also exists with both operands pre-split)
    const bool combo = epi == EPI_RECON ? (layout == MMVAE_GEMM_NT && a_pl) : x3w_planes_combo(layout, a_pl, b_pl);
    if (!combo || tile_id < 6 || tile_id > 8 || slots <= 0) return MMVAE_ERR_ARG;
    GemmArgs g = g0;
    g.nwork = nwork;
    const int nblocks = nwork < slots ? nwork : slots;  // persistent over the work items
    if (epi == EPI_RECON && b_pl) {  // (r5) fused last decoder layer, h AND W pre-split: the stagers only move planes
        if (layout != MMVAE_GEMM_NT || !a_pl) return MMVAE_ERR_ARG;
        if (tile_id == 6)
            MMVAE_LAUNCH((gemm_x3w_kernel<FORM_KC, FORM_KC, 256, 160, 4, 1, EPI_RECON, SRC_PLANES, SRC_PLANES>), dim3(nblocks),
                         dim3(512), 0, s, g);
        else if (tile_id == 8)
            MMVAE_LAUNCH((gemm_x3w_kernel<FORM_KC, FORM_KC, 256, 128, 2, 2, EPI_RECON, SRC_PLANES, SRC_PLANES>), dim3(nblocks),
                         dim3(512), 0, s, g);
        else
            return MMVAE_ERR_ARG;
        MMVAE_LAUNCH_CHECK();
        return MMVAE_OK;
    }
    if (epi == EPI_RECON) {  // fused last decoder layer: h pre-split, W fp32
        if (layout != MMVAE_GEMM_NT) return MMVAE_ERR_ARG;
        if (tile_id == 6)
            MMVAE_LAUNCH((gemm_x3w_kernel<FORM_KC, FORM_KC, 256, 160, 4, 1, EPI_RECON, SRC_PLANES, SRC_F32>), dim3(nblocks),
                         dim3(512), 0, s, g);
        else if (tile_id == 8)
            MMVAE_LAUNCH((gemm_x3w_kernel<FORM_KC, FORM_KC, 256, 128, 2, 2, EPI_RECON, SRC_PLANES, SRC_F32>), dim3(nblocks),
                         dim3(512), 0, s, g);
        else
            return MMVAE_ERR_ARG;
        MMVAE_LAUNCH_CHECK();
        return MMVAE_OK;
    }
#define XWP(AF, BF, AS, BS)                                                                                             \
    do {                                                                                                                \
        if (tile_id == 6)                                                                                               \
            MMVAE_LAUNCH((gemm_x3w_kernel<AF, BF, 256, 160, 4, 1, EPI_STD, AS, BS>), dim3(nblocks), dim3(512), 0, s, g); \
        else if (tile_id == 7)                                                                                          \
            MMVAE_LAUNCH((gemm_x3w_kernel<AF, BF, 160, 256, 1, 4, EPI_STD, AS, BS>), dim3(nblocks), dim3(512), 0, s, g); \
        else                                                                                                            \
            MMVAE_LAUNCH((gemm_x3w_kernel<AF, BF, 256, 128, 2, 2, EPI_STD, AS, BS>), dim3(nblocks), dim3(512), 0, s, g); \
    } while (0)
    if (layout == MMVAE_GEMM_TN && a_pl && !b_pl)
        XWP(FORM_RC, FORM_RC, SRC_PLANES, SRC_F32);
    else if (layout == MMVAE_GEMM_TN && a_pl)
        XWP(FORM_RC, FORM_RC, SRC_PLANES, SRC_PLANES);
    else if (layout == MMVAE_GEMM_TN)
        XWP(FORM_RC, FORM_RC, SRC_F32, SRC_PLANES);
    else if (layout == MMVAE_GEMM_NT)
        XWP(FORM_KC, FORM_KC, SRC_PLANES, SRC_F32);
    else if (a_pl)
        XWP(FORM_KC, FORM_RC, SRC_PLANES, SRC_F32);
    else
        XWP(FORM_KC, FORM_RC, SRC_F32, SRC_PLANES);
#undef XWP
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

}  // namespace mmvae_detail
