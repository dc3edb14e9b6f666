// fp32 MFMA GEMMs of the MMVAE step (k1-k5 of SURVEY 2b) for gfx950.
//
// Replaces the nn.Linear forward/backward dispatches of the reference (components.py:276 and its autograd).
//
// Design (MI355X-first, not a port of any BLAS):
//   * v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate (bitwise an fmaf chain) -> parity with the
//     reference's fp32 path, at the 157 TFLOP/s matrix peak.
//   * 256-thread workgroups = 4 wavefronts in a 2x2 grid; block tile 128x128 (or 64x64 for the small core
//     layers), BK = 32.  Each wave owns (BM/2)x(BN/2) as 32x32 MFMA blocks.
//   * Operand tiles go HBM -> VGPR (16 B per lane, coalesced along the contiguous axis) -> LDS, double buffered:
//     the global loads of k-tile t+1 are in flight while the MFMAs of k-tile t run, one barrier per k-tile.
//   * Two LDS images, chosen per operand by which axis is contiguous in HBM, so that no transposing pass exists:
//       KC ("K contiguous", x / W rows):   [rows][BK+4]  read with ONE ds_read_b128 per lane = 4 k-steps
//       RC ("row contiguous", k-slices):   [BK][rows]    read with 4 ds_read_b32, lanes along the row axis
//     The MFMA's k index is permuted (lane-half h of step j reads k = 8*kk + 4*h + j) identically for A and B,
//     which is legal because k is summed over; it is what lets the KC image feed 4 MFMAs from one 16-byte read.
//     The +4 float row pad makes those reads bank-conflict free (stride 144 B over the 64-bank b128 groups).
//   * Split-K over gridDim-level slices for the K = G (20k gene) reductions that only have 32 output tiles;
//     partial slabs are reduced in fixed order (no atomics: bitwise reproducible).
//   * Workgroup ids are remapped so each XCD (private 4 MiB L2) gets a contiguous run of tiles, M fastest:
//     neighbouring tiles share the same weight / activation panel in that XCD's L2.
//   * The last decoder layer has its own epilogue: bias + ReLU + (xhat - x)^2 + dP, with the per-cell squared
//     error reduced across the wavefront by shuffles (mmvae_decoder_recon_f32).
#include "gemm_dev.h"

namespace {

template <int AFORM, int BFORM, int BM, int BN, int BK>
struct F32Lds {
    static constexpr int A_FLOATS = Tile<AFORM, BM, BK>::LDS_FLOATS, B_FLOATS = Tile<BFORM, BN, BK>::LDS_FLOATS;
    static constexpr int FLOATS = 2 * A_FLOATS + 2 * B_FLOATS;
};

// One work item (output tile x split-K slice, linear index L: z slowest, M fastest) of the exact-f32 MFMA GEMM.
template <int AFORM, int BFORM, int BM, int BN, int BK, int WGM, int WGN, bool VEC, int EPI>
__device__ __forceinline__ void gemm_f32_item(const GemmArgs& g, int L, float* lds) {
    static_assert(WGM * WGN == 4, "4 wavefronts per workgroup");
    constexpr int WTM = BM / WGM, WTN = BN / WGN;  // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;    // 32x32 MFMA blocks per wave
    static_assert(TM * 32 == WTM && TN * 32 == WTN && BK % 8 == 0, "tile shape");
    using TA = Tile<AFORM, BM, BK>;
    using TB = Tile<BFORM, BN, BK>;
    constexpr int A_FLOATS = TA::LDS_FLOATS, B_FLOATS = TB::LDS_FLOATS;
    constexpr int KK = BK / 8;
    float* As = lds;
    float* Bs = lds + 2 * A_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;

    const int tiles = g.mt * g.nt;
    const int z = L / tiles;
    const int t = L - z * tiles;
    const int bm = t % g.mt, bn = t / g.mt;

    const int kt_beg = z * g.ktiles_per_split;
    int kt_end = kt_beg + g.ktiles_per_split;
    if (kt_end > g.ktiles) kt_end = g.ktiles;
    const int nkt = kt_end - kt_beg;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][n][e] = 0.f;

    f32x4 ra[TA::VECS], rb[TB::VECS];
    unsigned va[TA::VECS], vb[TB::VECS];
    if (nkt > 0) {
        load_tile<AFORM, BM, BK, VEC>(ra, va, g.A, g.lda, bm * BM, g.M, kt_beg * BK, g.K, tid);
        load_tile<BFORM, BN, BK, VEC>(rb, vb, g.B, g.ldb, bn * BN, g.N, kt_beg * BK, g.K, tid);
        store_tile<AFORM, BM, BK>(As, ra, va, tid);
        store_tile<BFORM, BN, BK>(Bs, rb, vb, tid);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            const bool more = (kt + 1 < nkt);
            if (more) {  // next tile's loads: in flight during this k-tile's MFMAs
                const int k0 = (kt_beg + kt + 1) * BK;
                load_tile<AFORM, BM, BK, VEC>(ra, va, g.A, g.lda, bm * BM, g.M, k0, g.K, tid);
                load_tile<BFORM, BN, BK, VEC>(rb, vb, g.B, g.ldb, bn * BN, g.N, k0, g.K, tid);
            }
            const float* Ac = As + cur * A_FLOATS;
            const float* Bc = Bs + cur * B_FLOATS;
#if MMVAE_GEMM_PRELOAD
            // all of this k-tile's operand fragments are requested from LDS up front, so the MFMAs below only wait
            // for the group they consume (counted lgkmcnt) instead of exposing the LDS latency once per 4 MFMAs
            f32x4 fa[KK][TM], fb[KK][TN];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[kk][i] = load_frag<AFORM, BM, BK>(Ac, wm * WTM + i * 32 + l31, kk, half);
#pragma unroll
                for (int n = 0; n < TN; ++n) fb[kk][n] = load_frag<BFORM, BN, BK>(Bc, wn * WTN + n * 32 + l31, kk, half);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int n = 0; n < TN; ++n)
                            acc[i][n] =
                                __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i][j], fb[kk][n][j], acc[i][n], 0, 0, 0);
#else
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = load_frag<AFORM, BM, BK>(Ac, wm * WTM + i * 32 + l31, kk, half);
#pragma unroll
                for (int n = 0; n < TN; ++n) fb[n] = load_frag<BFORM, BN, BK>(Bc, wn * WTN + n * 32 + l31, kk, half);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int n = 0; n < TN; ++n)
                            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][j], fb[n][j], acc[i][n], 0, 0, 0);
            }
#endif
            if (more) {
                store_tile<AFORM, BM, BK>(As + (cur ^ 1) * A_FLOATS, ra, va, tid);
                store_tile<BFORM, BN, BK>(Bs + (cur ^ 1) * B_FLOATS, rb, vb, tid);
            }
            __syncthreads();
        }
    }

    gemm_epilogue<BM, BN, WGM, WGN, EPI>(acc, g, bm, bn, z, lds);
}

// Block tile BM x BN, k-tile BK, 4 waves arranged WGM x WGN; each wave owns (BM/WGM) x (BN/WGN) as 32x32 MFMA blocks.
template <int AFORM, int BFORM, int BM, int BN, int BK, int WGM, int WGN, bool VEC, int EPI>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[F32Lds<AFORM, BFORM, BM, BN, BK>::FLOATS];
    // z (split-K slice) slowest, M fastest
    gemm_f32_item<AFORM, BFORM, BM, BN, BK, WGM, WGN, VEC, EPI>(g, xcd_remap(blockIdx.x, gridDim.x), lds);
}

// Grouped launch of independent small GEMMs (64x64 tiles, exact-f32 MFMA, unsplit): one grid covers the tiles of
// every job, so a backward pass's weight-gradient GEMMs of the core layers cost one launch (a dependent launch is
// ~4-5 us on the step's critical path) and fill the chip together instead of 8-128 workgroups at a time.
__global__ __launch_bounds__(NT) void gemm_f32_batch_kernel(const mmvae_gemm_job* __restrict__ jobs, int n_jobs) {
    constexpr int LDS_FLOATS = F32Lds<FORM_KC, FORM_KC, 64, 64, 32>::FLOATS;  // the largest of the three layouts
    static_assert(LDS_FLOATS >= F32Lds<FORM_KC, FORM_RC, 64, 64, 32>::FLOATS &&
                      LDS_FLOATS >= F32Lds<FORM_RC, FORM_RC, 64, 64, 32>::FLOATS, "LDS size");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    int j = 0;
    while (j + 1 < n_jobs && L >= jobs[j + 1].first_block) ++j;  // wave-uniform
    const mmvae_gemm_job& job = jobs[j];
    GemmArgs g = {};
    g.A = job.A;
    g.B = job.B;
    g.C = job.C;
    g.bias = job.bias;
    g.lda = job.lda;
    g.ldb = job.ldb;
    g.ldc = job.ldc;
    g.M = job.M;
    g.N = job.N;
    g.K = job.K;
    g.mt = (job.M + 63) / 64;
    g.nt = (job.N + 63) / 64;
    g.ktiles = (job.K + 31) / 32;
    g.ktiles_per_split = g.ktiles;
    g.alpha = job.alpha;
    g.flags = job.flags;
    g.aligned = 2;
    g.c_vec = 1;
    g.x_rows = 1;
    const int l = L - job.first_block;
    if (job.layout == MMVAE_GEMM_NT)
        gemm_f32_item<FORM_KC, FORM_KC, 64, 64, 32, 2, 2, true, EPI_STD>(g, l, lds);
    else if (job.layout == MMVAE_GEMM_NN)
        gemm_f32_item<FORM_KC, FORM_RC, 64, 64, 32, 2, 2, true, EPI_STD>(g, l, lds);
    else
        gemm_f32_item<FORM_RC, FORM_RC, 64, 64, 32, 2, 2, true, EPI_STD>(g, l, lds);
}

// ---- bf16x3 twin of gemm_f32_item for the 64x64 tiles of the core layers (r3).  These launches are latency-bound, but
// a quarter to a half of a workgroup's life was its MFMA chain: 16 x v_mfma_f32_32x32x2_f32 (64 cycles each) per k-tile
// and wave, with 2-4 workgroups sharing a CU's matrix cores.  The same products as six bf16 MFMAs per 16 k (the split
// done once per element on the way to LDS, 88 VALU instructions per thread and k-tile): 384 cycles instead of 1 024.
// One LDS image (three bf16 planes per operand, 80-byte rows), next tile's loads in flight during the MFMAs, two
// barriers per k-tile; fp32-GEMM accuracy as in the chip-filling kernels (MMVAE_GEMM_PRECISION_F32 keeps the exact path).
constexpr int X3S_LDS_BYTES = 3 * (64 + 64) * X3_LD;
template <int AFORM, int BFORM, bool VEC, int EPI>
__device__ __forceinline__ void gemm_x3s_item(const GemmArgs& g, int L, char* lds) {
    constexpr int BM = 64, BN = 64, WGN = 2;
    constexpr int PA = BM * X3_LD, PB = BN * X3_LD;
    char* As = lds;
    char* Bs = lds + 3 * PA;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const int tiles = g.mt * g.nt;
    const int z = L / tiles;
    const int t = L - z * tiles;
    const int bm = t % g.mt, bn = t / g.mt;
    const int kt_beg = z * g.ktiles_per_split;
    int kt_end = kt_beg + g.ktiles_per_split;
    if (kt_end > g.ktiles) kt_end = g.ktiles;
    const int nkt = kt_end - kt_beg;
    f32x16 acc[1][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
    constexpr int NVA = X3Regs<AFORM, BM>::NV, NVB = X3Regs<BFORM, BN>::NV;
    f32x4 ra[NVA], rb[NVB];
    unsigned va[NVA], vb[NVB];
    auto load_ab = [&](int kt) {
        const int k0 = kt * X3_BK;
        if constexpr (AFORM == FORM_KC)
            load_tile<FORM_KC, BM, X3_BK, VEC>(ra, va, g.A, g.lda, bm * BM, g.M, k0, g.K, tid);
        else
            x3_load_rc<BM, VEC>(ra, va, g.A, g.lda, bm * BM, g.M, k0, g.K, tid);
        if constexpr (BFORM == FORM_KC)
            load_tile<FORM_KC, BN, X3_BK, VEC>(rb, vb, g.B, g.ldb, bn * BN, g.N, k0, g.K, tid);
        else
            x3_load_rc<BN, VEC>(rb, vb, g.B, g.ldb, bn * BN, g.N, k0, g.K, tid);
    };
    if (nkt > 0) {
        load_ab(kt_beg);
        x3_store<AFORM, BM, VEC>(As, ra, va, tid, bm * BM, g.M, kt_beg * X3_BK, g.K);
        x3_store<BFORM, BN, VEC>(Bs, rb, vb, tid, bn * BN, g.N, kt_beg * X3_BK, g.K);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const bool more = kt + 1 < nkt;
            if (more) load_ab(kt_beg + kt + 1);  // in flight during this k-tile's MFMAs
            f32x16 c = acc[0][0];
#pragma unroll
            for (int ks = 0; ks < X3_BK / 16; ++ks) {
                bf16x8 fa[3], fb[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    fa[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(
                                                           As + p * PA + (wm * 32 + l31) * X3_LD + ks * 32 + half * 16));
                    fb[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(
                                                           Bs + p * PB + (wn * 32 + l31) * X3_LD + ks * 32 + half * 16));
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], c, 0, 0, 0);  // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], c, 0, 0, 0);
            }
            acc[0][0] = c;
            __syncthreads();
            if (more) {
                const int k0 = (kt_beg + kt + 1) * X3_BK;
                x3_store<AFORM, BM, VEC>(As, ra, va, tid, bm * BM, g.M, k0, g.K);
                x3_store<BFORM, BN, VEC>(Bs, rb, vb, tid, bn * BN, g.N, k0, g.K);
            }
            __syncthreads();
        }
    }
    gemm_epilogue<BM, BN, 2, 2, EPI>(acc, g, bm, bn, z, reinterpret_cast<float*>(lds));
}

template <int AFORM, int BFORM, bool VEC, int EPI>
__global__ __launch_bounds__(NT) void gemm_x3s_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char lds[X3S_LDS_BYTES];
    gemm_x3s_item<AFORM, BFORM, VEC, EPI>(g, xcd_remap(blockIdx.x, gridDim.x), lds);
}

// gemm_f32_batch_kernel on the bf16x3 item
__global__ __launch_bounds__(NT) void gemm_x3s_batch_kernel(const mmvae_gemm_job* __restrict__ jobs, int n_jobs) {
    __shared__ __attribute__((aligned(16))) char lds[X3S_LDS_BYTES];
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    int j = 0;
    while (j + 1 < n_jobs && L >= jobs[j + 1].first_block) ++j;  // wave-uniform
    const mmvae_gemm_job& job = jobs[j];
    GemmArgs g = {};
    g.A = job.A;
    g.B = job.B;
    g.C = job.C;
    g.bias = job.bias;
    g.lda = job.lda;
    g.ldb = job.ldb;
    g.ldc = job.ldc;
    g.M = job.M;
    g.N = job.N;
    g.K = job.K;
    g.mt = (job.M + 63) / 64;
    g.nt = (job.N + 63) / 64;
    g.ktiles = (job.K + 31) / 32;
    g.ktiles_per_split = g.ktiles;
    g.alpha = job.alpha;
    g.flags = job.flags;
    g.aligned = 2;
    g.c_vec = 1;
    g.x_rows = 1;
    const int l = L - job.first_block;
    if (job.layout == MMVAE_GEMM_NT)
        gemm_x3s_item<FORM_KC, FORM_KC, true, EPI_STD>(g, l, lds);
    else if (job.layout == MMVAE_GEMM_NN)
        gemm_x3s_item<FORM_KC, FORM_RC, true, EPI_STD>(g, l, lds);
    else
        gemm_x3s_item<FORM_RC, FORM_RC, true, EPI_STD>(g, l, lds);
}

template <int AFORM, int BFORM, int BM, int BN, int WGM, int WGN, bool VEC, int EPI>
__global__ __launch_bounds__(NT, 2) void gemm_x3_kernel(const GemmArgs g) {
    static_assert(WGM * WGN == 4, "4 wavefronts per workgroup");
    static_assert(VEC || (BM == 128 && BN == 128), "element-guarded matrices use the square tile");
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_BYTES = 3 * BM * X3_LD, B_BYTES = 3 * BN * X3_LD;
    constexpr int PA = BM * X3_LD, PB = BN * X3_LD;
    __shared__ __attribute__((aligned(16))) char lds[A_BYTES + B_BYTES];
    char* As = lds;
    char* Bs = lds + A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;

    const int nwg = gridDim.x, bid = blockIdx.x;
#if MMVAE_X3_STAMPS
    if (tid == 0 && bid < 4096) g_x3_trace[bid * 4 + 0] = wall_clock64();
#endif
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tiles = g.mt * g.nt;
    // Persistent over work items (output tile x split-K slice): the grid is at most the chip's resident slots, and a
    // workgroup's epilogue stores drain behind the main loop of its next item instead of idling the matrix cores
    // (a 128x160 tile's epilogue is 14 us of a 77 us item when 512 workgroups store at once).
    for (int w = L; w < g.nwork; w += nwg) {
    const int z = w / tiles;
    const int t = w - z * tiles;
    const int bm = t % g.mt, bn = t / g.mt;

    const int kt_beg = z * g.ktiles_per_split;
    int kt_end = kt_beg + g.ktiles_per_split;
    if (kt_end > g.ktiles) kt_end = g.ktiles;
    const int nkt = kt_end - kt_beg;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][n][e] = 0.f;

    auto mfma_group = [&](int i, int n, const bf16x8 (&fa)[3][TM], const bf16x8 (&fb)[3][TN]) {
        f32x16 c = acc[i][n];  // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][n], c, 0, 0, 0);
        acc[i][n] = c;
    };
    auto load_frags = [&](int ks, bf16x8 (&fa)[3][TM], bf16x8 (&fb)[3][TN]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[p][i] = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const f32x4*>(As + p * PA + (wm * WTM + i * 32 + l31) * X3_LD + ks * 32 +
                                                            half * 16));
#pragma unroll
            for (int n = 0; n < TN; ++n)
                fb[p][n] = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const f32x4*>(Bs + p * PB + (wn * WTN + n * 32 + l31) * X3_LD + ks * 32 +
                                                            half * 16));
        }
    };
    if (nkt > 0) {
        if constexpr (VEC) {
            // Software pipeline over 16-byte-regular operands with K % 32 == 0 (one raw staging set + packed planes).
            // While k-tile t is multiplied, the raw registers of tile t+1 (loaded during tile t-1) are split into
            // packed bf16 planes, one chunk of <= 4 VALU instructions behind every MFMA; the operand with more split
            // units is split during k-step 0, the other one during k-step 1, and each operand's raw registers are
            // re-issued as the loads of tile t+2 as soon as its last unit is split; after the barrier only the
            // ds_write burst remains.  The body is the same for every k-tile: the last one splits and writes a tile
            // that nobody reads (its loads are clamped into the matrix), which is cheaper than a second copy of the
            // loop body with guards (the guarded copies pushed the kernel into scratch spills).
            constexpr int NUA = BM / 32, NUB = BN / 32, NG = TM * TN;
            static_assert(NG >= NUA && NG >= NUB, "one split unit per MFMA group");
            constexpr bool A_FIRST = NUA > NUB;
            f32x4 ra[NUA], rb[NUB];
            uint2 pka[NUA][3], pkb[NUB][3];
            unsigned offa[NUA], offb[NUB];
            x3p_offsets<AFORM, BM>(offa, g.lda, bm * BM, g.M, tid);
            x3p_offsets<BFORM, BN>(offb, g.ldb, bn * BN, g.N, tid);
            const int kt_last = g.ktiles - 1;  // loads past the end re-read the last k-tile (never multiplied)
            auto load_a = [&](int kt) { x3p_load<NUA>(ra, x3p_base<AFORM>(g.A, g.lda, bm * BM, min(kt, kt_last)), offa); };
            auto load_b = [&](int kt) { x3p_load<NUB>(rb, x3p_base<BFORM>(g.B, g.ldb, bn * BN, min(kt, kt_last)), offb); };
            auto reload = [&](int phase, int kt) {
                if ((phase == 0) == A_FIRST)
                    load_a(kt);
                else
                    load_b(kt);
            };
            auto write_all = [&]() {
#pragma unroll
                for (int u = 0; u < NUA; ++u) x3p_write<AFORM, BM>(As, u, pka[u], tid);
#pragma unroll
                for (int u = 0; u < NUB; ++u) x3p_write<BFORM, BN>(Bs, u, pkb[u], tid);
            };
            load_a(kt_beg);
            load_b(kt_beg);
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                if (u < NUA) x3p_split<AFORM, BM, false>(ra, u, pka[u < NUA ? u : 0], tid, kt_beg * X3_BK, g.K);
                if (u < NUB) x3p_split<BFORM, BN, false>(rb, u, pkb[u < NUB ? u : 0], tid, kt_beg * X3_BK, g.K);
            }
            load_a(kt_beg + 1);
            load_b(kt_beg + 1);
            write_all();
            __syncthreads();
#if MMVAE_X3_STAMPS
            if (tid == 0 && bid < 4096) g_x3_trace[bid * 4 + 1] = wall_clock64();
            long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            long long tprev = clock64();
#define X3_STAMP(i)                       \
    {                                     \
        const long long tn_ = clock64();  \
        st[i] += tn_ - tprev;             \
        tprev = tn_;                      \
    }
#else
#define X3_STAMP(i)
#endif
            auto frag1 = [&](bf16x8& f, const char* S, int plane_bytes, int row, int ks, int p) {
                f = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(S + p * plane_bytes + row * X3_LD + ks * 32 +
                                                                               half * 16));
            };
            // One 16-deep k-step: 6 MFMAs per 32x32 block (smallest terms first), one split chunk behind every MFMA,
            // the next block's fragment reads behind the last three MFMAs, sched_barrier fences pin the order.
            bf16x8 fa[3][TM == 2 ? 2 : 1], fb[3][TN == 2 ? 2 : 1];  // 2x2 wave tiles: all fragments of a k-step
            bf16x8 fs[3], fl[2][3];                                  // 1x5 / 5x1 wave tiles: short side, streamed long side
            auto kstep_il = [&](int ks) {
                const bool first_is_a = (ks == 0) == A_FIRST;  // which operand this k-step splits
                X3Stage st_;
                auto fetch = [&](int u) {
                    if (first_is_a) {
                        if (u < NUA) x3s_fetch<AFORM, BM>(st_, ra, u, tid);
                    } else {
                        if (u < NUB) x3s_fetch<BFORM, BN>(st_, rb, u, tid);
                    }
                };
                auto chunk = [&](int u, auto P) {
                    constexpr int PH = decltype(P)::value;
                    if (first_is_a) {
                        if (u < NUA) x3s_phase<PH>(st_, pka[u < NUA ? u : 0]);
                    } else {
                        if (u < NUB) x3s_phase<PH>(st_, pkb[u < NUB ? u : 0]);
                    }
                };
                using I0 = std::integral_constant<int, 0>;
                using I1 = std::integral_constant<int, 1>;
                using I2 = std::integral_constant<int, 2>;
                using I3 = std::integral_constant<int, 3>;
                using I4 = std::integral_constant<int, 4>;
                using I5 = std::integral_constant<int, 5>;
// (two 16x16x32 MFMAs in place of each of these, as in the wave-specialised kernel, bought this loop 4 %: its MFMAs
// share their wave's issue slots with the in-kernel split -- not worth a second reconstruction epilogue)
#define X3_MFMA(C, A, B) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#define X3_GROUP_FS(C, A2, A1, A0, B0, B1, B2, U, RD, FS)                  \
    X3_MFMA(C, A2, B0);       \
    chunk(U, I0{});                                                        \
    FS(0);                                                                 \
    X3_SB();                                                               \
    X3_MFMA(C, A1, B1);       \
    chunk(U, I1{});                                                        \
    X3_SB();                                                               \
    X3_MFMA(C, A0, B2);       \
    chunk(U, I2{});                                                        \
    FS(2);                                                                 \
    X3_SB();                                                               \
    X3_MFMA(C, A1, B0);       \
    chunk(U, I3{});                                                        \
    RD(0);                                                                 \
    FS(3);                                                                 \
    X3_SB();                                                               \
    X3_MFMA(C, A0, B1);       \
    chunk(U, I4{});                                                        \
    RD(1);                                                                 \
    FS(4);                                                                 \
    X3_SB();                                                               \
    X3_MFMA(C, A0, B0);       \
    chunk(U, I5{});                                                        \
    RD(2);                                                                 \
    FS(5);                                                                 \
    X3_SB();
#define X3_GROUP(C, A2, A1, A0, B0, B1, B2, U, RD)                        \
    X3_MFMA(C, A2, B0);       \
    chunk(U, I0{});                                                        \
    X3_SB();                                                               \
    X3_MFMA(C, A1, B1);       \
    chunk(U, I1{});                                                        \
    X3_SB();                                                               \
    X3_MFMA(C, A0, B2);       \
    chunk(U, I2{});                                                        \
    X3_SB();                                                               \
    X3_MFMA(C, A1, B0);       \
    chunk(U, I3{});                                                        \
    RD(0);                                                                 \
    X3_SB();                                                               \
    X3_MFMA(C, A0, B1);       \
    chunk(U, I4{});                                                        \
    RD(1);                                                                 \
    X3_SB();                                                               \
    X3_MFMA(C, A0, B0);       \
    chunk(U, I5{});                                                        \
    RD(2);                                                                 \
    X3_SB();
                if constexpr (TM == 2 && TN == 2) {
                    // k-step 0 reads the fragments of block (0,0) up front; k-step 1 finds them prefetched (behind the
                    // MFMAs of k-step 0's last two blocks, into the registers those blocks no longer need)
                    if (ks == 0) {
#pragma unroll
                        for (int p = 2; p >= 0; --p) frag1(fa[p][0], As, PA, wm * WTM + l31, 0, p);
#pragma unroll
                        for (int p = 0; p < 3; ++p) frag1(fb[p][0], Bs, PB, wn * WTN + l31, 0, p);
                    }
                    X3_SB();
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int i = gq >> 1, n = gq & 1;
                        fetch(gq);
                        // reads behind this block's last three MFMAs: block 0 -> b[.][1], block 1 -> a[.][1];
                        // in k-step 0, blocks 2 and 3 -> a[.][0] and b[.][0] of k-step 1
                        auto rd = [&](int slot) {
                            if (gq == 0) frag1(fb[slot][1], Bs, PB, wn * WTN + 32 + l31, ks, slot);
                            if (gq == 1) frag1(fa[2 - slot][1], As, PA, wm * WTM + 32 + l31, ks, 2 - slot);
                            if (ks == 0 && gq == 2) frag1(fa[2 - slot][0], As, PA, wm * WTM + l31, 1, 2 - slot);
                            if (ks == 0 && gq == 3) frag1(fb[slot][0], Bs, PB, wn * WTN + l31, 1, slot);
                        };
                        f32x16 c = acc[i][n];
                        X3_GROUP(c, fa[2][i], fa[1][i], fa[0][i], fb[0][n], fb[1][n], fb[2][n], gq, rd)
                        acc[i][n] = c;
                    }
                } else {
                    constexpr bool STREAM_B = TM == 1;
                    static_assert(TM == 1 || TN == 1, "streamed fragments: one side has a single 32-row block");
                    constexpr int NL = STREAM_B ? TN : TM;
                    // short side fs (kept over the k-step), long side fl (double-buffered, one block ahead).  k-step 0
                    // reads fs and the first long-side block up front; its last block prefetches both for k-step 1:
                    // the long-side block into the fl buffer that block does not use, fs IN PLACE -- a plane of fs is
                    // re-read as soon as the last MFMA that consumes it has been issued.
                    const int par = ks == 0 ? 0 : (NL & 1);  // fl buffer of this k-step's first block
                    if (ks == 0) {
                        if (STREAM_B) {
#pragma unroll
                            for (int p = 2; p >= 0; --p) frag1(fs[p], As, PA, wm * WTM + l31, 0, p);
#pragma unroll
                            for (int p = 0; p < 3; ++p) frag1(fl[0][p], Bs, PB, wn * WTN + l31, 0, p);
                        } else {
#pragma unroll
                            for (int p = 0; p < 3; ++p) frag1(fs[p], Bs, PB, wn * WTN + l31, 0, p);
#pragma unroll
                            for (int p = 2; p >= 0; --p) frag1(fl[0][p], As, PA, wm * WTM + l31, 0, p);
                        }
                    }
                    X3_SB();
#pragma unroll
                    for (int j = 0; j < NL; ++j) {
                        fetch(j);
                        const int cur = (j + par) & 1;
                        const bool last0 = ks == 0 && j == NL - 1;  // the block that prefetches for k-step 1
                        auto rd = [&](int slot) {  // the next block's planes in the order its MFMAs consume them
                            if (j + 1 < NL) {
                                if (STREAM_B)
                                    frag1(fl[cur ^ 1][slot], Bs, PB, wn * WTN + (j + 1) * 32 + l31, ks, slot);
                                else
                                    frag1(fl[cur ^ 1][2 - slot], As, PA, wm * WTM + (j + 1) * 32 + l31, ks, 2 - slot);
                            } else if (last0) {
                                if (STREAM_B)
                                    frag1(fl[cur ^ 1][slot], Bs, PB, wn * WTN + l31, 1, slot);
                                else
                                    frag1(fl[cur ^ 1][2 - slot], As, PA, wm * WTM + l31, 1, 2 - slot);
                            }
                        };
                        // in-place prefetch of fs for k-step 1, behind the last MFMA that reads each plane:
                        // STREAM_B (fs = a): a2 after MFMA 0, a1 after MFMA 3, a0 after MFMA 5
                        // else     (fs = b): b2 after MFMA 2, b1 after MFMA 4, b0 after MFMA 5
                        auto fsrd = [&](int m) {
                            if (!last0) return;
                            if (STREAM_B) {
                                if (m == 0) frag1(fs[2], As, PA, wm * WTM + l31, 1, 2);
                                if (m == 3) frag1(fs[1], As, PA, wm * WTM + l31, 1, 1);
                                if (m == 5) frag1(fs[0], As, PA, wm * WTM + l31, 1, 0);
                            } else {
                                if (m == 2) frag1(fs[2], Bs, PB, wn * WTN + l31, 1, 2);
                                if (m == 4) frag1(fs[1], Bs, PB, wn * WTN + l31, 1, 1);
                                if (m == 5) frag1(fs[0], Bs, PB, wn * WTN + l31, 1, 0);
                            }
                        };
                        if (STREAM_B) {
                            f32x16 c = acc[0][j < TN ? j : 0];
                            X3_GROUP_FS(c, fs[2], fs[1], fs[0], fl[cur][0], fl[cur][1], fl[cur][2], j, rd, fsrd)
                            acc[0][j < TN ? j : 0] = c;
                        } else {
                            f32x16 c = acc[j < TM ? j : 0][0];
                            X3_GROUP_FS(c, fl[cur][2], fl[cur][1], fl[cur][0], fs[0], fs[1], fs[2], j, rd, fsrd)
                            acc[j < TM ? j : 0][0] = c;
                        }
                    }
                }
#undef X3_GROUP
#undef X3_GROUP_FS
            };
#if MMVAE_X3_PRIO_TOGGLE
            // slot of this wave in its SIMD's wave buffer (HW_REG_HW_ID bits 3:0): the two co-resident waves of a SIMD
            // (one per workgroup) sit in different slots
            const int wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
#endif
            for (int kt = 0; kt < nkt; ++kt) {
#if MMVAE_X3_PRIO_TOGGLE
                // the two workgroups of a CU share each SIMD's issue and matrix pipe; at equal priority the older wave
                // wins every arbitration and the younger workgroup runs ~35 % longer (the kernel ends with it).  Alternating
                // the priority per k-tile, in opposite phase for the two wave slots, time-shares the SIMD between them.
                if ((kt ^ wave_slot) & 1)
                    __builtin_amdgcn_s_setprio(1);
                else
                    __builtin_amdgcn_s_setprio(0);
#endif
                X3_STAMP(0)
                kstep_il(0);
                X3_STAMP(1)
                // unconditional on purpose: guarding the last two (useless, L2-resident) re-loads with a wave-uniform
                // branch cuts the body into blocks with conservative vmcnt(0) waits: measured +8...+20 % per kernel
                reload(0, kt_beg + kt + 2);  // this operand's raw registers are free again: next-next tile in flight
                X3_STAMP(2)
                kstep_il(1);
                reload(1, kt_beg + kt + 2);
                X3_STAMP(3)  // k-step 1 MFMAs issued
                __syncthreads();  // every wave is done reading this k-tile
                X3_STAMP(4)
                write_all();
                X3_STAMP(5)  // plane writes issued and landed
                __syncthreads();
                X3_STAMP(6)
            }
#if MMVAE_X3_STAMPS
            if (bid == 0 && lane == 0) {
                for (int i = 0; i < 7; ++i) g_x3_stamps[wave * 8 + i] = st[i];
                g_x3_stamps[wave * 8 + 7] = nkt;
            }
            if (tid == 0 && bid < 4096) g_x3_trace[bid * 4 + 2] = wall_clock64();
#endif
        } else {  // element-guarded (unaligned) matrices: plain loop, square tile
            constexpr int NVA = X3Regs<AFORM, BM>::NV, NVB = X3Regs<BFORM, BN>::NV;
            f32x4 ra[NVA], rb[NVB];
            unsigned va[NVA], vb[NVB];
            auto load_ab = [&](int kt) {
                const int k0 = kt * X3_BK;
                if (AFORM == FORM_KC)
                    load_tile<FORM_KC, BM, X3_BK, VEC>(ra, va, g.A, g.lda, bm * BM, g.M, k0, g.K, tid);
                else
                    x3_load_rc<BM, VEC>(ra, va, g.A, g.lda, bm * BM, g.M, k0, g.K, tid);
                if (BFORM == FORM_KC)
                    load_tile<FORM_KC, BN, X3_BK, VEC>(rb, vb, g.B, g.ldb, bn * BN, g.N, k0, g.K, tid);
                else
                    x3_load_rc<BN, VEC>(rb, vb, g.B, g.ldb, bn * BN, g.N, k0, g.K, tid);
            };
            load_ab(kt_beg);
            x3_store<AFORM, BM, VEC>(As, ra, va, tid, bm * BM, g.M, kt_beg * X3_BK, g.K);
            x3_store<BFORM, BN, VEC>(Bs, rb, vb, tid, bn * BN, g.N, kt_beg * X3_BK, g.K);
            __syncthreads();
            for (int kt = 0; kt < nkt; ++kt) {
                const bool more = kt + 1 < nkt;
                if (more) load_ab(kt_beg + kt + 1);
#pragma unroll
                for (int ks = 0; ks < X3_BK / 16; ++ks) {
                    bf16x8 fa[3][TM], fb[3][TN];
                    load_frags(ks, fa, fb);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int n = 0; n < TN; ++n) mfma_group(i, n, fa, fb);
                }
                __syncthreads();
                if (more) {
                    const int k0 = (kt_beg + kt + 1) * X3_BK;
                    x3_store<AFORM, BM, VEC>(As, ra, va, tid, bm * BM, g.M, k0, g.K);
                    x3_store<BFORM, BN, VEC>(Bs, rb, vb, tid, bn * BN, g.N, k0, g.K);
                }
                __syncthreads();
            }
        }
    }
    gemm_epilogue<BM, BN, WGM, WGN, EPI>(acc, g, bm, bn, z, reinterpret_cast<float*>(lds));
    if (EPI == EPI_RECON) __syncthreads();  // the epilogue's LDS scratch is overwritten by the next item's prologue
    }  // work items
#if MMVAE_X3_STAMPS
    if (tid == 0 && bid < 4096) {
        __builtin_amdgcn_s_waitcnt(0);  // stores of this wave issued and acknowledged
        g_x3_trace[bid * 4 + 3] = wall_clock64();
    }
#endif
}


// Fixed-order reduction of split-K slabs + the standard epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int S, int64_t slab_stride,
                                                            int M, int N, float alpha, const float* __restrict__ bias,
                                                            unsigned flags, float* __restrict__ C, int64_t ldc) {
    const int64_t total = (int64_t)M * N;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / N), col = (int)(idx - (int64_t)row * N);
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += slabs[(int64_t)k * slab_stride + idx];
        float v = s * alpha + (bias ? bias[col] : 0.f);
        float* cp = C + (int64_t)row * ldc + col;
        if (flags & MMVAE_GEMM_ACCUMULATE) v += *cp;
        if (flags & MMVAE_GEMM_RELU) v = fmaxf(v, 0.f);
        *cp = v;
    }
}

// ---------------------------------------------------------------------------------------------------- host side
// Tile shapes.  id 0: 128x128 (2x2 waves), 1: 128x160 (4x1 waves; 20000 = 125 x 160 -> no ragged last wave of tiles),
// 2: 64x64 (2x2 waves, the small core layers).  BK = 32 except NT 128x160 (BK = 16 keeps LDS at 46 KB so that two
// workgroups stay resident per CU).
struct TileShape {
    int bm, bn, blocks_per_cu;
};

TileShape tile_shape(int layout, int id) {
    if (id == 0 || id == 3) return {128, 128, 2};
    if (id == 1 || id == 4) return {128, 160, 2};
    if (id == 5) return {160, 128, 2};
    if (id == 6) return {256, 160, 1};  // 6, 7, 8: the wave-specialised bf16x3 kernel, one 512-thread workgroup per CU
    if (id == 7) return {160, 256, 1};
    if (id == 8) return {256, 128, 1};
    return {64, 64, 4};
}

extern int g_precision;
// 64x64-tile launches on the bf16 matrix cores: opt-in (MMVAE_X3S=1, bf16x3 mode).  Measured r3 at C2, interleaved on
// one box: 0.984 / 0.984 ms per step with it against 0.989 / 0.985 with the exact-f32 MFMA kernel -- these launches are
// bound by their launch / prologue / epilogue latency, not by the MFMA chain, so the default stays the exact kernel.
bool x3s_enabled() {
    const char* e = getenv("MMVAE_X3S");
    return g_precision == MMVAE_GEMM_PRECISION_BF16X3 && e && e[0] == '1';
}

template <int AFORM, int BFORM, bool VEC, int EPI>
int launch_gemm_vec(int tile_id, const GemmArgs& g, int nblocks, hipStream_t s) {
    if (tile_id == 0)
        MMVAE_LAUNCH((gemm_f32_kernel<AFORM, BFORM, 128, 128, MMVAE_GEMM_BK0, 2, 2, VEC, EPI>), dim3(nblocks),
                           dim3(NT), 0, s, g);
    else if (tile_id == 1) {
        if (AFORM == FORM_KC && BFORM == FORM_KC)  // BK = 16 keeps two workgroups resident per CU
            MMVAE_LAUNCH((gemm_f32_kernel<FORM_KC, FORM_KC, 128, 160, 16, 4, 1, VEC, EPI>), dim3(nblocks),
                               dim3(NT), 0, s, g);
        else if (AFORM == FORM_RC)
            MMVAE_LAUNCH((gemm_f32_kernel<FORM_RC, FORM_RC, 128, 160, 32, 4, 1, VEC, EPI>), dim3(nblocks),
                               dim3(NT), 0, s, g);
        else
            return MMVAE_ERR_ARG;
    }
    else if (x3s_enabled() && EPI == EPI_STD)
        MMVAE_LAUNCH((gemm_x3s_kernel<AFORM, BFORM, VEC, EPI_STD>), dim3(nblocks), dim3(NT), 0, s, g);
    else
        MMVAE_LAUNCH((gemm_f32_kernel<AFORM, BFORM, 64, 64, 32, 2, 2, VEC, EPI>), dim3(nblocks), dim3(NT), 0, s,
                           g);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

int g_precision = MMVAE_GEMM_PRECISION_BF16X3;  // process-wide, set by mmvae_gemm_set_precision

// Compute units of the current device, asked once (256 on MI355X; also the answer when no device can be asked: the
// planner is callable on a host without a GPU).
int device_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            return 256;
        }
        return n;
    }();
    return cus;
}


template <int AFORM, int BFORM, bool VEC, int EPI>
int launch_gemm_x3(int tile_id, const GemmArgs& g0, int nwork, hipStream_t s) {
    GemmArgs g = g0;
    g.nwork = nwork;
#if MMVAE_X3_PERSISTENT
    const int nblocks = nwork < 2 * device_cus() ? nwork : 2 * device_cus();  // 2 resident workgroups per CU: the rest is looped over
#else
    const int nblocks = nwork;
#endif
    if (VEC && tile_id == 4)
        MMVAE_LAUNCH((gemm_x3_kernel<AFORM, BFORM, 128, 160, 4, 1, true, EPI>), dim3(nblocks), dim3(NT), 0, s, g);
    else if (VEC && tile_id == 5 && EPI == EPI_STD)
        MMVAE_LAUNCH((gemm_x3_kernel<AFORM, BFORM, 160, 128, 1, 4, true, EPI_STD>), dim3(nblocks), dim3(NT), 0, s,
                           g);
    else if (tile_id == 3)
        MMVAE_LAUNCH((gemm_x3_kernel<AFORM, BFORM, 128, 128, 2, 2, VEC, EPI>), dim3(nblocks), dim3(NT), 0, s, g);
    else
        return MMVAE_ERR_ARG;
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

// The wave-specialised kernel is taken where its tiles fill the chip.  Host-side launch state like the precision switch:
// -1 = follow the environment (MMVAE_X3W=0 keeps the 2 x 4-wave kernel; read at every launch: tests and A/B runs toggle
// it), 0 / 1 = set by mmvae_gemm_set_x3w (the engine switches it off under a gradient exchange).  Interleaved A/B of the
// C2 step on one box: 1.130 against 1.150 ms.
int g_x3w = -1;
bool x3w_enabled() {
    if (g_x3w >= 0) return g_x3w != 0;
    const char* e = getenv("MMVAE_X3W");
    return !(e && e[0] == '0');
}

int g_wg_cap = 0;  // mmvae_gemm_set_workgroup_cap: > 0 caps the persistent kernel's grid
// one resident workgroup per CU
int x3w_slots() { return (g_wg_cap > 0 && g_wg_cap < device_cus()) ? g_wg_cap : device_cus(); }

template <int AFORM, int BFORM, int EPI>
int launch_gemm_x3w(int tile_id, const GemmArgs& g0, int nwork, hipStream_t s) {
    GemmArgs g = g0;
    g.nwork = nwork;
    const int nblocks = nwork < x3w_slots() ? nwork : x3w_slots();  // persistent over the work items
    if (tile_id == 6)
        MMVAE_LAUNCH((gemm_x3w_kernel<AFORM, BFORM, 256, 160, 4, 1, EPI>), dim3(nblocks), dim3(512), 0, s, g);
    else if (tile_id == 7 && EPI == EPI_STD)
        MMVAE_LAUNCH((gemm_x3w_kernel<AFORM, BFORM, 160, 256, 1, 4, EPI_STD>), dim3(nblocks), dim3(512), 0, s, g);
    else if (tile_id == 8)
        MMVAE_LAUNCH((gemm_x3w_kernel<AFORM, BFORM, 256, 128, 2, 2, EPI>), dim3(nblocks), dim3(512), 0, s, g);
    else
        return MMVAE_ERR_ARG;
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

// tile_id 3, 4, 5 = the bf16x3 kernel (128x128, 128x160, 160x128 tiles; the last two need 16-byte-regular operands);
// 6, 7, 8 = its wave-specialised form (256x160, 160x256, 256x128; 16-byte-regular operands and whole k-tiles)
template <int AFORM, int BFORM, int EPI>
int launch_gemm_forms(int tile_id, const GemmArgs& g, int nblocks, hipStream_t s) {
    if (tile_id >= 6) {
        if (!(g.x3_vec && g.K % X3_BK == 0)) return MMVAE_ERR_ARG;
        return launch_gemm_x3w<AFORM, BFORM, EPI>(tile_id, g, nblocks, s);
    }
    if (tile_id >= 3)  // the pipelined bf16x3 loop needs 16-byte-regular operands and whole k-tiles
        return (g.x3_vec && g.K % X3_BK == 0) ? launch_gemm_x3<AFORM, BFORM, true, EPI>(tile_id, g, nblocks, s)
                                                    : launch_gemm_x3<AFORM, BFORM, false, EPI>(tile_id, g, nblocks, s);
    return g.aligned == 2 ? launch_gemm_vec<AFORM, BFORM, true, EPI>(tile_id, g, nblocks, s)
                          : launch_gemm_vec<AFORM, BFORM, false, EPI>(tile_id, g, nblocks, s);
}

// The pipelined bf16x3 loader reads operands in 16-byte groups.  gfx950 serves 16-byte global accesses at any 4-byte
// aligned address at full rate (tools/ubench/unaligned.hip), so neither the base pointers nor the leading dimensions
// need 16-byte alignment.  What is needed: groups along K must not cross the K edge (whole k-tiles: K % 32 == 0, or
// the tail slab), and a rows-contiguous operand (A of TN along M, B of NN / TN along N) whose extent is not a multiple
// of 4 has an edge group that reads up to 12 bytes past the end of a row -- past the matrix for its last row: legal
// only when the caller vouches for the slack (MMVAE_GEMM_OPERAND_SLACK).
bool x3_vec_ok(int layout, int M, int N, bool slack) {
    const bool a_rc = layout == MMVAE_GEMM_TN, b_rc = layout != MMVAE_GEMM_NT;
    return (!a_rc || M % 4 == 0 || slack) && (!b_rc || N % 4 == 0 || slack);
}

int bk_of(int layout, int tile_id) {
    if (tile_id >= 3) return X3_BK;
    if (tile_id == 0) return MMVAE_GEMM_BK0;
    return (tile_id == 1 && layout == MMVAE_GEMM_NT) ? 16 : 32;
}

// Picks tile id and split-K.  Large outputs: the tile whose (rounds x tile area) is smallest, rounds = number of
// times the chip's resident-workgroup slots are filled.  Few output tiles (K = G reductions): split-K.
void plan(int layout, int M, int N, int K, int* tile_id, int* splitk) {
    const int CUS = device_cus();
    long best_cost = -1;
    int best = 0;
    for (int id = 0; id < (layout == MMVAE_GEMM_NN ? 1 : 2); ++id) {
        const TileShape ts = tile_shape(layout, id);
        const long tiles = (long)ceil_div_i(M, ts.bm) * ceil_div_i(N, ts.bn);
        const long slots = (long)CUS * ts.blocks_per_cu;
        const long rounds = (tiles + slots - 1) / slots;
        const long cost = rounds * ts.bm * ts.bn;  // every round costs one tile time per resident slot
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = id;
        }
    }
    const TileShape tb = tile_shape(layout, best);
    const int tbig = ceil_div_i(M, tb.bm) * ceil_div_i(N, tb.bn);
    if (tbig >= 192) {
        *tile_id = best;
        // Fewer tiles than the chip's 512 resident slots and a long K (dX of the last decoder layer over K x B sample
        // rows: 320 tiles, K = 20 000): one round would leave slots idle for its whole length.  Take the split whose
        // rounds-per-slice is smallest (320 tiles x 3 slices = 1.9 rounds of a third of the work each: -33 %).
        int s = 1;
        const int t128b = ceil_div_i(M, 128) * ceil_div_i(N, 128);  // split launches use the square tile
        if (t128b < 512) {
            const int ktl = K / X3_BK;
            double best_cost = 1.0;
            for (int c = 2; c <= 8; ++c) {
                if (ktl / c < 16) break;  // keep slices long enough to amortise prologue and epilogue
                const double cost = (double)ceil_div_i(t128b * c, 512) / c;
                if (cost < best_cost - 0.05) {
                    best_cost = cost;
                    s = c;
                }
            }
        }
        // a K that is not a multiple of the k-tile (30 000 genes): one more slab for the K tail, so that the others
        // cover whole k-tiles on the pipelined kernel (gemm_f32_impl, "tail slab")
        if (K % X3_BK != 0 && K >= 8 * X3_BK) s += 1;
        *splitk = s;
        return;
    }
    const int kt = ceil_div_i(K, 32);
    const int t128 = ceil_div_i(M, 128) * ceil_div_i(N, 128);
    const int t64 = ceil_div_i(M, 64) * ceil_div_i(N, 64);
    int sA = ceil_div_i(512, t128);
    if (sA > kt / 8) sA = kt / 8;
    if (sA < 1) sA = 1;
    if (sA > 64) sA = 64;
    int sB = ceil_div_i(512, t64);
    if (sB > kt / 4) sB = kt / 4;
    if (sB < 1) sB = 1;
    if (sB > 64) sB = 64;
    if (t128 * sA >= 256 || t128 * sA >= t64 * sB) {
        *tile_id = 0;
        *splitk = sA;
        // K not a multiple of the 32-wide k-tile (30 000 genes): one more slab for the K tail, so that the slabs over
        // the whole k-tiles can run on the pipelined bf16x3 kernel (gemm_f32_impl, "tail slab")
        if (sA > 1 && K % X3_BK != 0 && sA < 64) *splitk = sA + 1;
    } else {
        *tile_id = 2;
        *splitk = sB;
    }
}

// bf16x3 tile for an unsplit GEMM over 16-byte-regular operands: the shape whose (rounds x tile area) is smallest
// (20000 = 125 x 160: a 512 x 20000 output is 628 square tiles = 2 rounds of the 512 resident slots, but 500 tiles of
// 128x160 = 1 round).  allow_tall: the 160x128 shape has no fused-recon instantiation.
int x3_tile_for(int M, int N, bool allow_tall) {
    const long slots = 2L * device_cus();  // 2 resident workgroups per CU
    int best = 3;
    long best_cost = -1;
    for (int id = 3; id <= (allow_tall ? 5 : 4); ++id) {
        const TileShape ts = tile_shape(0, id);
        const long tiles = (long)ceil_div_i(M, ts.bm) * ceil_div_i(N, ts.bn);
        const long cost = ((tiles + slots - 1) / slots) * ts.bm * ts.bn;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = id;
        }
    }
    return best;
}

// Tile of the wave-specialised kernel for `slabs` split-K slices of an M x N output over 16-byte-regular operands with
// whole k-tiles, or 0 when that kernel would leave too many CUs without a workgroup (it runs ONE workgroup per CU).
// Cost of a shape = rounds of the chip's 256 slots x tile area.
int x3w_tile_for(int M, int N, int slabs, bool allow_tall) {
    if (!x3w_enabled()) return 0;
    int best = 0;
    long best_cost = -1;
    for (int id = 6; id <= 8; ++id) {
        if (id == 7 && !allow_tall) continue;
        const TileShape ts = tile_shape(0, id);
        const long work = (long)ceil_div_i(M, ts.bm) * ceil_div_i(N, ts.bn) * slabs;
        if (work < 160) continue;
        const long cost = ((work + x3w_slots() - 1) / x3w_slots()) * ts.bm * ts.bn;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = id;
        }
    }
    return best;
}

// the bf16x3 tile an unsplit (or `slabs`-way split) launch over regular operands takes
int x3_tile_regular(int M, int N, int slabs, bool allow_tall) {
    const int w = x3w_tile_for(M, N, slabs, allow_tall);
    if (w) return w;
    return slabs == 1 ? x3_tile_for(M, N, allow_tall) : 3;
}

}  // namespace

extern "C" int mmvae_gemm_plan(int layout, int M, int N, int K, int* tile_out, int* splitk_out) {
    if (layout < 0 || layout > 2 || M <= 0 || N <= 0 || K <= 0) return MMVAE_ERR_ARG;
    int tile, sk;
    plan(layout, M, N, K, &tile, &sk);
    if (tile_out) *tile_out = tile;
    if (splitk_out) *splitk_out = sk;
    return MMVAE_OK;
}

extern "C" size_t mmvae_gemm_workspace_bytes(int layout, int M, int N, int K, int splitk) {
    if (splitk == 0) {
        int tile;
        if (mmvae_gemm_plan(layout, M, N, K, &tile, &splitk) != MMVAE_OK) return 0;
    }
    return splitk > 1 ? (size_t)splitk * (size_t)M * (size_t)N * sizeof(float) : 0;
}

// optional pre-split forms of the operands (mmvae_gemm_planes_f32)
struct PlaneArgs {
    const uint16_t* Ap;
    const uint16_t* Bp;
    int64_t ldap, ldbp, a_pstride, b_pstride;
};

static int gemm_f32_impl(int layout, int M, int N, int K, float alpha, const float* A, int64_t lda, const float* B,
                         int64_t ldb, float* C, int64_t ldc, const float* bias, unsigned flags, int splitk,
                         float* workspace, size_t workspace_bytes, float* sq_partials, int64_t sq_capacity,
                         mmvae_stream_t stream, const PlaneArgs* pl_in = nullptr);

extern "C" int mmvae_gemm_f32(int layout, int M, int N, int K, float alpha, const float* A, int64_t lda,
                              const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, unsigned flags,
                              int splitk, float* workspace, size_t workspace_bytes, mmvae_stream_t stream) {
    return gemm_f32_impl(layout, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, flags, splitk, workspace,
                         workspace_bytes, nullptr, 0, stream);
}

// A pre-split operand the LDS-DMA stagers can address: 16-byte aligned rows, 32-bit byte offsets inside the three
// planes, and -- for a rows-contiguous operand (A of TN along M, B of NN / TN along N), fetched in 16-byte groups of 8
// columns -- a leading dimension that covers the extent rounded up to 8 (the columns in between hold zeros).
static bool planes_usable(int layout, bool is_a, int M, int N, int K, const uint16_t* P, int64_t ld, int64_t pstride) {
    if (!P || !aligned16(P) || ld % 8 != 0 || pstride % 8 != 0 || K % X3_BK != 0) return false;
    const bool rc = is_a ? layout == MMVAE_GEMM_TN : layout != MMVAE_GEMM_NT;
    const int64_t rows = rc ? K : (is_a ? M : N), inner = rc ? (is_a ? M : N) : K;
    if (ld < inner || pstride < rows * ld) return false;
    if (rc && ld < (inner + 7) / 8 * 8) return false;  // (zero columns up to the next multiple of 8: see XwPlanes::offsets)
    return 3 * pstride * 2 < (int64_t)0xFFFFFFFF;
}

// Tiles of the unsplit launch the library would make for this shape (= partial sums mmvae_gemm_f32_sq writes).
static int sq_tiles(int layout, int M, int N, int K, bool aligned2) {
    int tile_id, sk;
    plan(layout, M, N, K, &tile_id, &sk);
    if (sk != 1) return 0;
    if (g_precision == MMVAE_GEMM_PRECISION_BF16X3 && tile_id != 2)
        tile_id = (aligned2 && K % X3_BK == 0) ? x3_tile_regular(M, N, 1, true) : 3;
    const TileShape ts = tile_shape(layout, tile_id);
    return ceil_div_i(M, ts.bm) * ceil_div_i(N, ts.bn);
}

extern "C" int mmvae_gemm_sq_partials(int layout, int M, int N, int K, int operands_regular) {
    if (layout < 0 || layout > 2 || M <= 0 || N <= 0 || K <= 0) return 0;
    // the tile shape depends on operand alignment: exact when the caller vouches for 16-byte-regular operands
    // (aligned bases, leading dimensions and M, N, K multiples of 4), otherwise the larger of the two counts (the
    // launch zero-fills the slots it does not use)
    const int a = sq_tiles(layout, M, N, K, true), b = sq_tiles(layout, M, N, K, false);
    (void)operands_regular;  // (alignment no longer decides the tile shape; kept for ABI stability)
    if (x3_vec_ok(layout, M, N, false)) return a;
    return a > b ? a : b;
}

extern "C" int mmvae_gemm_f32_sq(int layout, int M, int N, int K, float alpha, const float* A, int64_t lda,
                                 const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, unsigned flags,
                                 float* sq_partials, int64_t sq_capacity, mmvae_stream_t stream) {
    if (!sq_partials || sq_capacity <= 0 || (flags & MMVAE_GEMM_RAW_SLABS)) return MMVAE_ERR_ARG;
    return gemm_f32_impl(layout, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, flags, 1, nullptr, 0, sq_partials,
                         sq_capacity, stream);
}

extern "C" int mmvae_gemm_planes_f32(int layout, int M, int N, int K, float alpha, const float* A, int64_t lda,
                                     const uint16_t* Ap, int64_t ldap, int64_t a_plane_stride, const float* B,
                                     int64_t ldb, const uint16_t* Bp, int64_t ldbp, int64_t b_plane_stride, float* C,
                                     int64_t ldc, const float* bias, unsigned flags, int splitk, float* workspace,
                                     size_t workspace_bytes, float* sq_partials, int64_t sq_capacity,
                                     mmvae_stream_t stream) {
    if (sq_partials && (sq_capacity <= 0 || (flags & MMVAE_GEMM_RAW_SLABS))) return MMVAE_ERR_ARG;
    const PlaneArgs pl = {Ap, Bp, ldap, ldbp, a_plane_stride, b_plane_stride};
    return gemm_f32_impl(layout, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, flags, sq_partials ? 1 : splitk,
                         workspace, workspace_bytes, sq_partials, sq_partials ? sq_capacity : 0, stream, &pl);
}

// 1 when a launch of this shape would read the named operands from planes (both flags as passed to the launch)
extern "C" int mmvae_gemm_planes_supported(int layout, int M, int N, int K, int splitk, int a_planes, int b_planes) {
    if (layout < 0 || layout > 2 || M <= 0 || N <= 0 || K <= 0 || K % X3_BK != 0) return 0;
    if (g_precision != MMVAE_GEMM_PRECISION_BF16X3) return 0;
    if (!mmvae_detail::x3w_planes_combo(layout, a_planes != 0, b_planes != 0)) return 0;
    // (a rows-contiguous planes operand whose extent is off a multiple of 8 needs its leading dimension rounded up to 8
    // with zero columns: checked at the launch, planes_usable)
    int tile_id, sk;
    plan(layout, M, N, K, &tile_id, &sk);
    if (splitk == 0) splitk = sk;
    if (tile_id == 2 || !x3_vec_ok(layout, M, N, true)) return 0;
    tile_id = x3_tile_regular(M, N, splitk, true);
    if (tile_id < 6) return 0;
    const int kt32 = K / X3_BK, kps = ceil_div_i(kt32, splitk);
    return (int64_t)(splitk - 1) * kps < kt32 ? 1 : 0;
}

static int gemm_f32_impl(int layout, int M, int N, int K, float alpha, const float* A, int64_t lda, const float* B,
                         int64_t ldb, float* C, int64_t ldc, const float* bias, unsigned flags, int splitk,
                         float* workspace, size_t workspace_bytes, float* sq_partials, int64_t sq_capacity,
                         mmvae_stream_t stream, const PlaneArgs* pl_in) {
    const bool want_ap = pl_in && pl_in->Ap, want_bp = pl_in && pl_in->Bp;
    if (layout < 0 || layout > 2 || M <= 0 || N <= 0 || K <= 0 || (!A && !want_ap) || (!B && !want_bp) || !C)
        return MMVAE_ERR_ARG;
    if (splitk < 0 || ldc < N) return MMVAE_ERR_ARG;
    // leading-dimension sanity: contiguous axis extent must fit in the stride
    const int64_t a_inner = (layout == MMVAE_GEMM_TN) ? M : K;
    const int64_t b_inner = (layout == MMVAE_GEMM_NT) ? K : N;
    if ((A && lda < a_inner) || (B && ldb < b_inner)) return MMVAE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;

    int tile_id, sk_auto;
    plan(layout, M, N, K, &tile_id, &sk_auto);
    if (splitk == 0) splitk = sk_auto;
    if (splitk > 1 && tile_id == 1) tile_id = 0;  // split-K slices use the square tiles
    int aligned = A && B && aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    if (aligned && M % 4 == 0 && N % 4 == 0 && K % 4 == 0) aligned = 2;
    const bool slack = (flags & MMVAE_GEMM_OPERAND_SLACK) != 0;
    flags &= ~MMVAE_GEMM_OPERAND_SLACK;
    // Pre-split operands run on the wave-specialised kernel only (whole k-tiles, chip-filling shapes); an operand whose
    // planes cannot be used falls back to its fp32 form when the caller passed one.
    bool a_pl = false, b_pl = false;
    if ((want_ap || want_bp) && g_precision == MMVAE_GEMM_PRECISION_BF16X3) {
        a_pl = want_ap && planes_usable(layout, true, M, N, K, pl_in->Ap, pl_in->ldap, pl_in->a_pstride);
        b_pl = want_bp && planes_usable(layout, false, M, N, K, pl_in->Bp, pl_in->ldbp, pl_in->b_pstride);
        if (!mmvae_detail::x3w_planes_combo(layout, a_pl, b_pl)) {
            // the nearest combination with a kernel: drop what has an fp32 form
            if (a_pl && b_pl && B && mmvae_detail::x3w_planes_combo(layout, true, false))
                b_pl = false;
            else
                a_pl = b_pl = false;
        }
    }
    bool x3v = false;
    const int tile_planned = tile_id;
    // (a pre-split operand is not read in its fp32 form: only the fp32 operands' rows need the 16-byte-group conditions)
    auto pick_tile = [&](bool apl, bool bpl) {
        const bool a_rc = layout == MMVAE_GEMM_TN, b_rc = layout != MMVAE_GEMM_NT;
        x3v = (!a_rc || apl || M % 4 == 0 || slack) && (!b_rc || bpl || N % 4 == 0 || slack);
        int t = tile_planned;
        if (g_precision == MMVAE_GEMM_PRECISION_BF16X3 && t != 2)  // chip-filling GEMMs: bf16x3 cores
            t = (x3v && K % X3_BK == 0) ? x3_tile_regular(M, N, splitk, true) : 3;
        if (t >= 6) {  // the persistent kernel's k-tile stream has no empty items: every slice must own a k-tile
            const int kt32 = K / X3_BK, kps = ceil_div_i(kt32, splitk);
            if ((int64_t)(splitk - 1) * kps >= kt32) t = 3;
        }
        return t;
    };
    tile_id = pick_tile(a_pl, b_pl);
    if ((a_pl || b_pl) && tile_id < 6) {
        a_pl = b_pl = false;
        tile_id = pick_tile(false, false);
    }
    if (want_ap || want_bp) {
        if ((!a_pl && !A) || (!b_pl && !B)) return MMVAE_ERR_ARG;
    }
    const TileShape ts = tile_shape(layout, tile_id);
    const int ktiles = ceil_div_i(K, bk_of(layout, tile_id));
    const bool raw = (flags & MMVAE_GEMM_RAW_SLABS) != 0;
    if (!raw && splitk > ktiles) splitk = ktiles;
    if (raw && (flags & (MMVAE_GEMM_RELU | MMVAE_GEMM_ACCUMULATE))) return MMVAE_ERR_ARG;
    if (raw && bias) return MMVAE_ERR_ARG;

    GemmArgs g = {};
    g.A = A;
    g.B = B;
    g.bias = bias;
    g.lda = lda;
    g.ldb = ldb;
    g.M = M;
    g.N = N;
    g.K = K;
    g.mt = ceil_div_i(M, ts.bm);
    g.nt = ceil_div_i(N, ts.bn);
    g.ktiles = ktiles;
    g.ktiles_per_split = ceil_div_i(ktiles, splitk);
    if (!raw) splitk = ceil_div_i(ktiles, g.ktiles_per_split);  // drop empty trailing slices (raw: caller sized the slabs)
    g.alpha = alpha;
    g.flags = flags;
    g.x_rows = 1;
    g.aligned = aligned;
    g.x3_vec = x3v;
    if (a_pl) {
        g.Ap = pl_in->Ap;
        g.ldap = pl_in->ldap;
        g.a_pstride = pl_in->a_pstride;
    }
    if (b_pl) {
        g.Bp = pl_in->Bp;
        g.ldbp = pl_in->ldbp;
        g.b_pstride = pl_in->b_pstride;
    }
    if (raw) {
        g.C = C;
        g.ldc = ldc;
        g.slab_stride = (int64_t)M * ldc;
    } else if (splitk > 1) {
        if (!workspace || workspace_bytes < (size_t)splitk * M * N * sizeof(float)) return MMVAE_ERR_WORKSPACE;
        g.C = workspace;
        g.ldc = N;
        g.slab_stride = (int64_t)M * N;
    } else {
        g.C = C;
        g.ldc = ldc;
        g.slab_stride = 0;
    }
    g.c_vec = 1;  // 16-byte epilogue accesses need no alignment on gfx950; the N-edge group goes element-wise
    if (sq_partials) {  // fused sum of squares: one partial per output tile, the rest of the caller's slots zeroed
        if (splitk != 1 || (int64_t)g.mt * g.nt > sq_capacity) return MMVAE_ERR_ARG;
        g.sq_part = sq_partials;
        if ((int64_t)g.mt * g.nt < sq_capacity &&
            hipMemsetAsync(sq_partials + (int64_t)g.mt * g.nt, 0, (size_t)(sq_capacity - (int64_t)g.mt * g.nt) * sizeof(float),
                           s) != hipSuccess)
            return MMVAE_ERR_LAUNCH;
    }
    auto launch_tile = [&](int tid_, const GemmArgs& ga, int slabs) {
        const int nblocks = ga.mt * ga.nt * slabs;
        if (a_pl || b_pl) {
            if (tid_ < 6) return (int)MMVAE_ERR_ARG;
            return mmvae_detail::launch_x3w_planes(layout, tid_, a_pl, b_pl, EPI_STD, ga, nblocks, x3w_slots(), s);
        }
        if (layout == MMVAE_GEMM_NT) return launch_gemm_forms<FORM_KC, FORM_KC, EPI_STD>(tid_, ga, nblocks, s);
        if (layout == MMVAE_GEMM_NN) return launch_gemm_forms<FORM_KC, FORM_RC, EPI_STD>(tid_, ga, nblocks, s);
        return launch_gemm_forms<FORM_RC, FORM_RC, EPI_STD>(tid_, ga, nblocks, s);
    };
    auto launch = [&](const GemmArgs& ga, int slabs) { return launch_tile(tile_id, ga, slabs); };
    int rc;
    // Tail slab: slab outputs of a bf16x3 split over 16-byte-regular operands whose K is not a multiple of the k-tile.
    // The pipelined kernel needs whole k-tiles, so slabs 0 .. splitk-2 cover the whole k-tiles [0, K_main) and run on
    // it; the last slab is the K tail (< 32 columns) on the element-guarded variant.  The slab count is unchanged.
    const int K_main = K / X3_BK * X3_BK;
    if (tile_id == 3 && splitk > 1 && x3v && K % X3_BK != 0 && K_main / X3_BK >= splitk - 1 && g.slab_stride > 0) {
        GemmArgs gm = g;
        gm.K = K_main;
        gm.ktiles = K_main / X3_BK;
        gm.ktiles_per_split = ceil_div_i(gm.ktiles, splitk - 1);
        int sk_main = splitk - 1;
        if (!raw) sk_main = ceil_div_i(gm.ktiles, gm.ktiles_per_split);
        {  // the slabs over the whole k-tiles are free to take the tile shape that fills the chip best
            const int t_main = x3_tile_regular(M, N, sk_main, true);
            const TileShape tm = tile_shape(layout, t_main);
            gm.mt = ceil_div_i(M, tm.bm);
            gm.nt = ceil_div_i(N, tm.bn);
            rc = launch_tile(t_main, gm, sk_main);
        }
        if (rc != MMVAE_OK) return rc;
        GemmArgs gt = g;
        const bool a_kc = layout != MMVAE_GEMM_TN, b_kc = layout == MMVAE_GEMM_NT;
        gt.A = A + (a_kc ? (int64_t)K_main : (int64_t)K_main * lda);
        gt.B = B + (b_kc ? (int64_t)K_main : (int64_t)K_main * ldb);
        gt.K = K - K_main;
        gt.ktiles = 1;
        gt.ktiles_per_split = 1;
        gt.C = g.C + (int64_t)sk_main * g.slab_stride;
        rc = launch(gt, 1);
        if (rc != MMVAE_OK) return rc;
        if (raw && sk_main + 1 < splitk) {  // (cannot happen: K_main / 32 >= splitk - 1 slabs of >= 1 k-tile)
            return MMVAE_ERR_ARG;
        }
        if (!raw) splitk = sk_main + 1;
    } else {
        rc = launch(g, splitk);
        if (rc != MMVAE_OK) return rc;
    }
    if (!raw && splitk > 1) {
        const int64_t total = (int64_t)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        MMVAE_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, workspace, splitk, (int64_t)M * N, M,
                           N, alpha, bias, flags, C, ldc);
        MMVAE_LAUNCH_CHECK();
    }
    return MMVAE_OK;
}

// ---- grouped small GEMMs
static bool batch_job_ok(const mmvae_gemm_job& j) {
    if (j.layout < 0 || j.layout > 2 || j.M <= 0 || j.N <= 0 || j.K <= 0 || !j.A || !j.B || !j.C) return false;
    if ((j.M | j.N | j.K) & 3) return false;
    if (!aligned16(j.A) || !aligned16(j.B) || !aligned16(j.C) || (j.bias && !aligned16(j.bias))) return false;
    if ((j.lda | j.ldb | j.ldc) & 3) return false;
    const int64_t a_inner = (j.layout == MMVAE_GEMM_TN) ? j.M : j.K;
    const int64_t b_inner = (j.layout == MMVAE_GEMM_NT) ? j.K : j.N;
    if (j.lda < a_inner || j.ldb < b_inner || j.ldc < j.N) return false;
    if (j.flags & ~(MMVAE_GEMM_RELU | MMVAE_GEMM_ACCUMULATE)) return false;
    return true;
}

extern "C" int mmvae_gemm_batch_job_ok(const mmvae_gemm_job* job) { return job && batch_job_ok(*job) ? 1 : 0; }

extern "C" int mmvae_gemm_batch_prepare(int n_jobs, mmvae_gemm_job* jobs, int* total_blocks) {
    if (n_jobs <= 0 || n_jobs > 4096 || !jobs || !total_blocks) return MMVAE_ERR_ARG;
    int64_t first = 0;
    for (int i = 0; i < n_jobs; ++i) {
        if (!batch_job_ok(jobs[i])) return MMVAE_ERR_ARG;
        jobs[i].first_block = (int32_t)first;
        jobs[i].n_blocks = ceil_div_i(jobs[i].M, 64) * ceil_div_i(jobs[i].N, 64);
        first += jobs[i].n_blocks;
        if (first > (1 << 30)) return MMVAE_ERR_ARG;
    }
    *total_blocks = (int)first;
    return MMVAE_OK;
}

extern "C" int mmvae_gemm_batch_f32(int n_jobs, const mmvae_gemm_job* jobs_dev, int total_blocks,
                                    mmvae_stream_t stream) {
    if (n_jobs <= 0 || !jobs_dev || total_blocks <= 0) return MMVAE_ERR_ARG;
    if (x3s_enabled())
        MMVAE_LAUNCH(gemm_x3s_batch_kernel, dim3(total_blocks), dim3(NT), 0, (hipStream_t)stream, jobs_dev, n_jobs);
    else
        MMVAE_LAUNCH(gemm_f32_batch_kernel, dim3(total_blocks), dim3(NT), 0, (hipStream_t)stream, jobs_dev, n_jobs);
    MMVAE_LAUNCH_CHECK();
    return MMVAE_OK;
}

// Rows of se_part the fused decoder/recon kernel defines (one row per column tile; the tile is 128 or 160 genes wide
// depending on mode and shape -- rows beyond the tiles actually used are written as zeros).
extern "C" int mmvae_recon_tiles(int G) {
    if (G <= 0) return 0;
    return ceil_div_i(G, g_precision == MMVAE_GEMM_PRECISION_BF16X3 ? 128 : 160);
}

#if MMVAE_X3_STAMPS
extern "C" int mmvae_debug_x3_stamps(long long* out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_x3_stamps), sizeof(long long) * 32) == hipSuccess ? 0 : 1;
}
extern "C" int mmvae_debug_x3_trace(long long* out, int n_blocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x3_trace), sizeof(long long) * 4 * n_blocks) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int mmvae_gemm_set_precision(int mode) {
    if (mode != MMVAE_GEMM_PRECISION_F32 && mode != MMVAE_GEMM_PRECISION_BF16X3) return MMVAE_ERR_ARG;
    g_precision = mode;
    return MMVAE_OK;
}

extern "C" int mmvae_gemm_get_precision(void) { return g_precision; }

extern "C" int mmvae_gemm_set_x3w(int mode) {
    if (mode < -1 || mode > 1) return MMVAE_ERR_ARG;
    g_x3w = mode;
    return MMVAE_OK;
}

extern "C" int mmvae_gemm_get_x3w(void) { return x3w_enabled() ? 1 : 0; }

extern "C" int mmvae_gemm_set_workgroup_cap(int max_workgroups) {
    if (max_workgroups < 0) return MMVAE_ERR_ARG;
    g_wg_cap = max_workgroups;
    return MMVAE_OK;
}

extern "C" int mmvae_recon_row_tiles(int rows) { return rows > 0 ? ceil_div_i(rows, 128) : 0; }

extern "C" int mmvae_decoder_recon_rows_f32(int rows, int x_rows, int G, int H, const float* h, int64_t ldh,
                                            const float* W, int64_t ldw, const float* bias, const float* x, int64_t ldx,
                                            float* xhat, int64_t ldxhat, float* dP, int64_t lddp, float* se_part,
                                            mmvae_stream_t stream) {
    return mmvae_decoder_recon_rows_colsum_f32(rows, x_rows, G, H, h, ldh, W, ldw, bias, x, ldx, xhat, ldxhat, dP, lddp,
                                               se_part, nullptr, stream);
}

// h pre-split (hp != NULL): the wave-specialised kernel with LDS-DMA staging of h (mmvae_decoder_recon_planes_f32)
// Launch state (host side, like the workgroup cap): the caller vouches that h has ZERO columns from H up to the next
// multiple of the 32-wide k-tile inside its leading dimension and that W's rows may be read that far (finite values: the
// next row, or the slack behind an optimiser arena).  A hidden width that is not a multiple of 32 (1000) then runs the
// pipelined kernels over the padded K -- what lies beyond H meets exact zeros -- instead of the guarded loop
// (242 against 123 us at C2's sizes).
static int g_recon_kpad = 0;
extern "C" int mmvae_recon_set_h_kpad(int on) {
    g_recon_kpad = on ? 1 : 0;
    return MMVAE_OK;
}

static int decoder_recon_impl(int rows, int x_rows, int G, int H, const float* h, int64_t ldh, const uint16_t* hp,
                              int64_t ldhp, int64_t h_pstride, const float* W, int64_t ldw, const uint16_t* Wp,
                              int64_t ldwp, int64_t w_pstride, const float* bias,
                              const float* x, int64_t ldx, float* xhat, int64_t ldxhat, float* dP, int64_t lddp,
                              uint16_t* dPp, int64_t lddpp, int64_t dp_pstride, float* se_part, float* col_part,
                              mmvae_stream_t stream) {
    if (rows <= 0 || x_rows <= 0 || rows % x_rows != 0 || G <= 0 || H <= 0 || (!h && !hp) || !W || !x || !se_part)
        return MMVAE_ERR_ARG;
    if (dPp && (G % 8 != 0 || lddpp < G || lddpp % 8 != 0 || dp_pstride % 8 != 0 || dp_pstride < (int64_t)rows * lddpp ||
                !aligned16(dPp) || g_precision != MMVAE_GEMM_PRECISION_BF16X3))
        return MMVAE_ERR_ARG;
    if ((h && ldh < H) || ldw < H || ldx < G) return MMVAE_ERR_ARG;
    if (xhat && ldxhat < G) return MMVAE_ERR_ARG;
    if (dP && lddp < G) return MMVAE_ERR_ARG;
    GemmArgs g = {};
    g.A = h;
    g.B = W;
    g.bias = bias;
    g.lda = ldh;
    g.ldb = ldw;
    if (g_recon_kpad && H % X3_BK != 0 && h && !hp && ldh >= (int64_t)ceil_div_i(H, X3_BK) * X3_BK)
        H = ceil_div_i(H, X3_BK) * X3_BK;  // (zero columns of h against whatever follows a row of W)
    g.M = rows;
    g.N = G;
    g.K = H;
    const bool x3 = g_precision == MMVAE_GEMM_PRECISION_BF16X3;
    g.aligned = h && aligned16(h) && aligned16(W) && (ldh % 4 == 0) && (ldw % 4 == 0);
    if (g.aligned && rows % 4 == 0 && G % 4 == 0 && H % 4 == 0) g.aligned = 2;
    g.x3_vec = 1;  // both operands K-contiguous: rows are clamped one by one, no edge groups
    // The wave-specialised kernel, now that its multipliers run 16x16x32 MFMAs with a reconstruction epilogue for that
    // accumulator layout (r4): C2 0.967 against 0.990 ms per step, C3 3.578 against 3.62 on one box.  On 32x32x16 MFMAs
    // it lost to the 2 x 4-wave kernel, whose two workgroups per CU overlap one's heavy epilogue -- 160 KB of x read,
    // 160 KB of dP written per tile -- with the other's main loop (r3: 1.097 against 1.067 ms).  MMVAE_X3W_RECON=0
    // restores the 2 x 4-wave kernel (diagnostics).
    const char* e_recon = getenv("MMVAE_X3W_RECON");
    const bool w_recon = !(e_recon && e_recon[0] == '0');
    int tile_id = !x3 ? 1 : ((H % X3_BK == 0) ? (w_recon ? x3_tile_regular(rows, G, 1, false) : x3_tile_for(rows, G, false)) : 3);
    bool h_pl = false;
    if (hp) {
        const int tw = (x3 && H % X3_BK == 0) ? x3w_tile_for(rows, G, 1, false) : 0;
        h_pl = tw >= 6 && planes_usable(MMVAE_GEMM_NT, true, rows, G, H, hp, ldhp, h_pstride);
        if (h_pl) tile_id = tw;
        if (!h_pl && !h) return MMVAE_ERR_ARG;
    }
    const int tbm = tile_shape(0, tile_id).bm;
    if (col_part && tbm != 128 && tbm != 256) return MMVAE_ERR_ARG;  // [mmvae_recon_row_tiles(rows)][G] partials
    g.mt = ceil_div_i(rows, tile_shape(0, tile_id).bm);
    g.nt = ceil_div_i(G, tile_shape(0, tile_id).bn);
    g.se_tiles = mmvae_recon_tiles(G);  // rows nt .. se_tiles-1 of se_part are zeroed by the last column tile
    g.ktiles = ceil_div_i(H, x3 ? X3_BK : 16);
    g.ktiles_per_split = g.ktiles;
    g.alpha = 1.f;
    g.x = x;
    g.xhat = xhat;
    g.dP = dP;
    g.se_part = se_part;
    g.col_part = col_part;
    g.ldx = ldx;
    g.ldxhat = ldxhat;
    g.lddp = lddp;
    g.x_rows = x_rows;
    g.c_vec = 1;
    g.dPp = reinterpret_cast<unsigned short*>(dPp);
    g.lddpp = lddpp;
    g.dp_pstride = dp_pstride;
    if (h_pl) {
        g.Ap = hp;
        g.ldap = ldhp;
        g.a_pstride = h_pstride;
        // (r5) W pre-split too -- only together with h: the kernel whose stagers do no vector work at all
        const bool w_pl = Wp && planes_usable(MMVAE_GEMM_NT, false, rows, G, H, Wp, ldwp, w_pstride);
        if (w_pl) {
            g.Bp = Wp;
            g.ldbp = ldwp;
            g.b_pstride = w_pstride;
        }
        return mmvae_detail::launch_x3w_planes(MMVAE_GEMM_NT, tile_id, true, w_pl, EPI_RECON, g, g.mt * g.nt, x3w_slots(),
                                               (hipStream_t)stream);
    }
    return launch_gemm_forms<FORM_KC, FORM_KC, EPI_RECON>(tile_id, g, g.mt * g.nt, (hipStream_t)stream);
}

extern "C" int mmvae_decoder_recon_rows_colsum_f32(int rows, int x_rows, int G, int H, const float* h, int64_t ldh,
                                                   const float* W, int64_t ldw, const float* bias, const float* x,
                                                   int64_t ldx, float* xhat, int64_t ldxhat, float* dP, int64_t lddp,
                                                   float* se_part, float* col_part, mmvae_stream_t stream) {
    return decoder_recon_impl(rows, x_rows, G, H, h, ldh, nullptr, 0, 0, W, ldw, nullptr, 0, 0, bias, x, ldx, xhat, ldxhat, dP,
                              lddp, nullptr, 0, 0, se_part, col_part, stream);
}

extern "C" int mmvae_decoder_recon_planes_f32(int rows, int x_rows, int G, int H, const float* h, int64_t ldh,
                                              const uint16_t* hp, int64_t ldhp, int64_t h_plane_stride, const float* W,
                                              int64_t ldw, const float* bias, const float* x, int64_t ldx, float* xhat,
                                              int64_t ldxhat, float* dP, int64_t lddp, uint16_t* dP_planes,
                                              int64_t lddpp, int64_t dp_plane_stride, float* se_part, float* col_part,
                                              mmvae_stream_t stream) {
    return decoder_recon_impl(rows, x_rows, G, H, h, ldh, hp, ldhp, h_plane_stride, W, ldw, nullptr, 0, 0, bias, x, ldx, xhat,
                              ldxhat, dP, lddp, dP_planes, lddpp, dp_plane_stride, se_part, col_part, stream);
}

// (r5) ... with the weights pre-split as well (Wp; used together with hp, otherwise W is read): K-sample programs read
// every weight tile from rows / 256 row tiles -- 20 at C3 -- and split it there each time
extern "C" int mmvae_decoder_recon_wplanes_f32(int rows, int x_rows, int G, int H, const float* h, int64_t ldh,
                                               const uint16_t* hp, int64_t ldhp, int64_t h_plane_stride, const float* W,
                                               int64_t ldw, const uint16_t* Wp, int64_t ldwp, int64_t w_plane_stride,
                                               const float* bias, const float* x, int64_t ldx, float* xhat, int64_t ldxhat,
                                               float* dP, int64_t lddp, float* se_part, float* col_part,
                                               mmvae_stream_t stream) {
    return decoder_recon_impl(rows, x_rows, G, H, h, ldh, hp, ldhp, h_plane_stride, W, ldw, Wp, ldwp, w_plane_stride, bias, x,
                              ldx, xhat, ldxhat, dP, lddp, nullptr, 0, 0, se_part, col_part, stream);
}

extern "C" int mmvae_decoder_recon_f32(int B, int G, int H, const float* h, int64_t ldh, const float* W, int64_t ldw,
                                       const float* bias, const float* x, int64_t ldx, float* xhat, int64_t ldxhat,
                                       float* dP, int64_t lddp, float* se_part, mmvae_stream_t stream) {
    return mmvae_decoder_recon_rows_f32(B, B, G, H, h, ldh, W, ldw, bias, x, ldx, xhat, ldxhat, dP, lddp, se_part,
                                        stream);
}
