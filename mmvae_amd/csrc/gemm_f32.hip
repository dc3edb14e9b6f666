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
#include "common.h"
#include <type_traits>

namespace {

#ifndef MMVAE_GEMM_BK0
#define MMVAE_GEMM_BK0 32  // k-tile of the 128x128 block tile
#endif
#ifndef MMVAE_X3_PERSISTENT
#define MMVAE_X3_PERSISTENT 0  // 1: grid capped at 512 workgroups looping over work items (measured: no gain)
#endif
#ifndef MMVAE_X3_STAMPS
#define MMVAE_X3_STAMPS 0  // diagnostic build: per-phase cycle sums of the bf16x3 loop (block 0, one lane per wave)
#endif
#ifndef MMVAE_GEMM_PRELOAD
#define MMVAE_GEMM_PRELOAD 0
#endif

constexpr int NT = 256;  // threads per workgroup (4 wavefronts)

enum { FORM_KC = 0, FORM_RC = 1 };
enum { EPI_STD = 0, EPI_RECON = 1 };

// LDS image of one operand tile with R rows/columns along the non-K axis and BK along K.
//   KC ("K contiguous" in HBM):  [R][BK + 4]   (+4 floats: conflict-free ds_read_b128)
//   RC ("row contiguous"):        [BK][R]
template <int FORM, int R, int BK>
struct Tile {
    static constexpr int KC_LD = BK + 4;
    static constexpr int LDS_FLOATS = (FORM == FORM_KC) ? R * KC_LD : BK * R;
    static constexpr int NVEC = R * BK / 4;               // float4 per tile
    static constexpr int VECS = (NVEC + NT - 1) / NT;     // float4 per thread
    static constexpr bool EXACT = (NVEC % NT) == 0;
    static constexpr int C4 = (FORM == FORM_KC) ? BK / 4 : R / 4;  // float4 per contiguous run
};

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int64_t lda, ldb, ldc;
    int M, N, K;
    int mt, nt;            // tiles along M, N
    int ktiles;            // total k-tiles
    int ktiles_per_split;  // k-tiles per split-K slice
    int64_t slab_stride;   // elements between split-K slabs of C (0 when splitk == 1)
    float alpha;
    unsigned flags;
    int aligned;           // 1: A, B 16-byte aligned with leading dimensions % 4 == 0; 2: and M, N, K % 4 == 0
    int x3_vec;            // bf16x3 kernel: operands may be read in 16-byte groups (x3_vec_ok)
    // recon epilogue
    const float* x;
    float* xhat;
    float* dP;
    float* se_part;
    int64_t ldx, ldxhat, lddp;
    int x_rows;    // x row = output row % x_rows (K-sample decode stacks K copies of the batch)
    int se_tiles;  // rows of se_part the caller reads (>= nt); the last column tile zeroes rows nt .. se_tiles-1
    int c_vec;     // 16-byte epilogue accesses (a group straddling the N edge falls back to elements); 0: element path
    int nwork;     // bf16x3 kernel: work items (output tiles x split-K slices), looped over by <= 512 workgroups
    float* sq_part;  // optional [mt * nt]: sum of squares of the C values this tile stores (unsplit launches only)
    float* col_part;  // recon epilogue, optional [mt][N]: column sums of dP over this row tile (bias gradient partials)
};

// HBM -> registers.  r0: first row (KC) / column (RC) of this tile along the non-K axis, Rtot its extent.
// One code path, no branches: every 16-byte group is loaded from a clamped (always valid) address and a validity
// mask is kept beside it; store_tile zeroes the out-of-matrix elements on the way to LDS.  Loads therefore issue back
// to back and stay in flight across the k-tile's MFMAs (an exec-masked load per element, or a control-flow join
// between loads and consumers, makes the compiler drain them one by one).
//   VEC = true   A, B 16-byte aligned, leading dimensions and M, N, K multiples of 4: a group never straddles an edge
//   VEC = false  anything else (e.g. 60530 / 52437-gene matrices): clamped element loads
template <int FORM, int R, int BK, bool VEC, int NV>
__device__ __forceinline__ void load_tile(f32x4 (&reg)[NV], unsigned (&valid)[NV], const float* __restrict__ P,
                                          int64_t ld, int r0, int Rtot, int k0, int Kend, int tid) {
    using T = Tile<FORM, R, BK>;
    static_assert(NV == T::VECS, "register tile size");
#pragma unroll
    for (int i = 0; i < T::VECS; ++i) {
        const int f = tid + NT * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        unsigned vm = 0u;
        if (T::EXACT || f < T::NVEC) {
            const int run = f / T::C4, c4 = f % T::C4;
            // (o, x): index along the strided (outer) axis and along the contiguous axis, with their extents
            const int o = (FORM == FORM_KC) ? r0 + run : k0 + run;
            const int x = (FORM == FORM_KC) ? k0 + c4 * 4 : r0 + c4 * 4;
            const int olim = (FORM == FORM_KC) ? Rtot : Kend;
            const int xlim = (FORM == FORM_KC) ? Kend : Rtot;
            const bool ov = o < olim;
            const float* p = P + (int64_t)(ov ? o : olim - 1) * ld;
            if (VEC) {
                const bool full = ov && (x + 3 < xlim);
                v = *reinterpret_cast<const f32x4*>(p + (full ? x : 0));
                vm = full ? 0xFu : 0u;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = p[min(x + j, xlim - 1)];
                    vm |= (ov && x + j < xlim) ? (1u << j) : 0u;
                }
            }
        }
        reg[i] = v;
        valid[i] = vm;
    }
}

// registers -> LDS image (zeroing out-of-matrix elements for the guarded modes)
template <int FORM, int R, int BK, int NV>
__device__ __forceinline__ void store_tile(float* S, const f32x4 (&reg)[NV], const unsigned (&valid)[NV], int tid) {
    static_assert(NV == Tile<FORM, R, BK>::VECS, "register tile size");
    using T = Tile<FORM, R, BK>;
#pragma unroll
    for (int i = 0; i < T::VECS; ++i) {
        const int f = tid + NT * i;
        if (T::EXACT || f < T::NVEC) {
            const int run = f / T::C4, c4 = f % T::C4;
            f32x4 v = reg[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((valid[i] >> j) & 1u) ? v[j] : 0.f;
            if (FORM == FORM_KC)
                *reinterpret_cast<f32x4*>(&S[run * T::KC_LD + c4 * 4]) = v;
            else
                *reinterpret_cast<f32x4*>(&S[run * R + c4 * 4]) = v;
        }
    }
}

// LDS -> MFMA operand fragment: element j of the result feeds MFMA step j of k-group kk and carries
// k = 8*kk + 4*half + j for row/column `row` of the tile.
template <int FORM, int R, int BK>
__device__ __forceinline__ f32x4 load_frag(const float* S, int row, int kk, int half) {
    if (FORM == FORM_KC) {
        return *reinterpret_cast<const f32x4*>(&S[row * Tile<FORM, R, BK>::KC_LD + kk * 8 + 4 * half]);
    } else {
        const float* q = &S[(kk * 8 + 4 * half) * R + row];
        f32x4 f;
        f.x = q[0];
        f.y = q[R];
        f.z = q[2 * R];
        f.w = q[3 * R];
        return f;
    }
}

// value of lane (quad base + S_q) for lane q of each quad (DPP quad_perm)
template <int S0, int S1, int S2, int S3>
__device__ __forceinline__ float quad_perm(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), S0 | (S1 << 2) | (S2 << 4) | (S3 << 6), 0xF, 0xF, true));
}

// Sum of `v` over the 256 threads of the workgroup in a fixed order (wave shuffles, then the 4 wave sums in wave order)
// -> *dst.  `lds` is free: the operand tiles are dead after the k-loop's final barrier.
__device__ __forceinline__ void tile_sum_to(float* dst, float v, float* lds) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) *dst = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    __syncthreads();  // a persistent workgroup's next item reuses the LDS
}

// Epilogue shared by the fp32-MFMA and the bf16x3-MFMA kernels (the C/D register layout of the 32x32 MFMAs is
// dtype-independent): col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
template <int BM, int BN, int WGM, int WGN, int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[TM][TN], const GemmArgs& g, int bm, int bn, int z,
                                              float* lds) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    if (EPI == EPI_STD) {
        float* C = g.C + (int64_t)z * g.slab_stride;
        const bool raw = (g.flags & MMVAE_GEMM_RAW_SLABS) || g.slab_stride != 0;
        const bool accum = !raw && (g.flags & MMVAE_GEMM_ACCUMULATE);
        const bool relu = !raw && (g.flags & MMVAE_GEMM_RELU);
        const float alpha = raw ? 1.f : g.alpha;
        float sq = 0.f;  // sum of squares of what this thread stores (feeds the fused gradient-norm partial)
        if (g.c_vec) {
            // 16-byte stores: the 4 lanes of a quad hold a 4 row x 4 column patch column-wise (one column each, rows
            // e = 4 gq .. 4 gq + 3); a quad transpose (2 DPP exchange stages) gives every lane one row x 4 columns.
            // A quarter of the store instructions of the dword path: the epilogue is store-issue bound when every
            // workgroup of a round stores at once.
            const int q = lane & 3;
            const bool odd = q & 1, hi = q & 2;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    const int col = bn * BN + wn * WTN + n * 32 + (l31 & ~3);
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    const bool whole = col + 3 < g.N;  // N % 4 != 0: the group straddling the edge goes element-wise
                    if (!raw && g.bias && col < g.N) {
                        if (whole) {
                            bv = *reinterpret_cast<const f32x4*>(g.bias + col);
                        } else {
                            for (int j = 0; j < g.N - col; ++j) bv[j] = g.bias[col + j];
                        }
                    }
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float a0 = acc[i][n][4 * gq], a1 = acc[i][n][4 * gq + 1], a2 = acc[i][n][4 * gq + 2],
                              a3 = acc[i][n][4 * gq + 3];
                        // stage 1: exchange with lane ^ 1
                        const float r0 = quad_perm<1, 0, 3, 2>(odd ? a0 : a1), r1 = quad_perm<1, 0, 3, 2>(odd ? a2 : a3);
                        const float c0 = odd ? r0 : a0, c1 = odd ? a1 : r0, c2 = odd ? r1 : a2, c3 = odd ? a3 : r1;
                        // stage 2: exchange with lane ^ 2
                        const float t0 = quad_perm<2, 3, 0, 1>(hi ? c0 : c2), t1 = quad_perm<2, 3, 0, 1>(hi ? c1 : c3);
                        f32x4 v;
                        v[0] = hi ? t0 : c0;
                        v[1] = hi ? t1 : c1;
                        v[2] = hi ? c2 : t0;
                        v[3] = hi ? c3 : t1;
                        const int row = bm * BM + wm * WTM + i * 32 + 8 * gq + 4 * half + q;
                        if (row < g.M && col < g.N) {
                            float* cp = C + (int64_t)row * g.ldc + col;
                            v = v * alpha + bv;
                            if (whole) {
                                if (accum) v += *reinterpret_cast<const f32x4*>(cp);
                                if (relu) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                                }
                                *reinterpret_cast<f32x4*>(cp) = v;
                                sq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                            } else {
                                for (int j = 0; j < g.N - col; ++j) {
                                    float e = v[j];
                                    if (accum) e += cp[j];
                                    if (relu) e = fmaxf(e, 0.f);
                                    cp[j] = e;
                                    sq += e * e;
                                }
                            }
                        }
                    }
                }
            if (g.sq_part) tile_sum_to(g.sq_part + ((int64_t)bn * g.mt + bm), sq, lds);
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int col = bn * BN + wn * WTN + n * 32 + l31;
                if (col >= g.N) continue;
                const float bv = (!raw && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = bm * BM + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (row >= g.M) continue;
                    float* cp = C + (int64_t)row * g.ldc + col;
                    float v = acc[i][n][e] * alpha + bv;
                    if (accum) v += *cp;
                    if (relu) v = fmaxf(v, 0.f);
                    *cp = v;
                    sq += v * v;
                }
            }
        if (g.sq_part) tile_sum_to(g.sq_part + ((int64_t)bn * g.mt + bm), sq, lds);
    } else {
        // bias + ReLU + squared error + dP; per-cell SE reduced over the 32 lanes that share a row.
        float* rowsum = lds;  // [WGN][BM] scratch: the operand tiles are dead after the final barrier
        if (g.c_vec) {
            // 16-byte path (x, xhat, dP 16-byte regular): quad transpose as in the standard epilogue, so that x is
            // read and xhat / dP are written as one row x 4 genes per lane; the row's SE is then the sum over the 8
            // lanes of a half that share q = lane & 3 (strides 4, 8, 16).
            const int q = lane & 3;
            const bool odd = q & 1, hi = q & 2;
            // 256-row tiles (wave-specialised kernel): 160 accumulator registers are live -- the bias vectors are
            // re-read per row group (L1 hits) instead of being held in 20 more
            constexpr bool KEEP_BIAS = BM < 256;
            auto load_bias = [&](int n) {
                const int col = bn * BN + wn * WTN + n * 32 + (l31 & ~3);
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (g.bias && col < g.N) {
                    if (col + 3 < g.N) {
                        b = *reinterpret_cast<const f32x4*>(g.bias + col);
                    } else {
                        for (int j = 0; j < g.N - col; ++j) b[j] = g.bias[col + j];
                    }
                }
                return b;
            };
            f32x4 cs[TN];  // column sums of dP over this lane's rows (col_part)
#pragma unroll
            for (int n = 0; n < TN; ++n) cs[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 bv[KEEP_BIAS ? TN : 1];
            if (KEEP_BIAS) {
#pragma unroll
                for (int n = 0; n < TN; ++n) bv[KEEP_BIAS ? n : 0] = load_bias(n);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int rloc = wm * WTM + i * 32 + 8 * gq + 4 * half + q;
                    const int row = bm * BM + rloc;
                    const int xr = row % g.x_rows;
                    float sr = 0.f;
#pragma unroll
                    for (int n = 0; n < TN; ++n) {
                        const float a0 = acc[i][n][4 * gq], a1 = acc[i][n][4 * gq + 1], a2 = acc[i][n][4 * gq + 2],
                                    a3 = acc[i][n][4 * gq + 3];
                        const float r0 = quad_perm<1, 0, 3, 2>(odd ? a0 : a1), r1 = quad_perm<1, 0, 3, 2>(odd ? a2 : a3);
                        const float c0 = odd ? r0 : a0, c1 = odd ? a1 : r0, c2 = odd ? r1 : a2, c3 = odd ? a3 : r1;
                        const float t0 = quad_perm<2, 3, 0, 1>(hi ? c0 : c2), t1 = quad_perm<2, 3, 0, 1>(hi ? c1 : c3);
                        f32x4 p;
                        p[0] = hi ? t0 : c0;
                        p[1] = hi ? t1 : c1;
                        p[2] = hi ? c2 : t0;
                        p[3] = hi ? c3 : t1;
                        const int col = bn * BN + wn * WTN + n * 32 + (l31 & ~3);
                        const f32x4 bn4 = KEEP_BIAS ? bv[KEEP_BIAS ? n : 0] : load_bias(n);
                        if (row < g.M && col + 3 < g.N) {
                            p += bn4;
                            const f32x4 xv = *reinterpret_cast<const f32x4*>(g.x + (int64_t)xr * g.ldx + col);
                            f32x4 xh, dp;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                xh[j] = fmaxf(p[j], 0.f);
                                const float d = xh[j] - xv[j];
                                sr += d * d;
                                dp[j] = (p[j] > 0.f) ? 2.f * d : 0.f;
                            }
                            if (g.xhat) *reinterpret_cast<f32x4*>(g.xhat + (int64_t)row * g.ldxhat + col) = xh;
                            if (g.dP) *reinterpret_cast<f32x4*>(g.dP + (int64_t)row * g.lddp + col) = dp;
                            cs[n] += dp;
                        } else if (row < g.M && col < g.N) {  // G % 4 != 0: the group straddling the edge, element-wise
                            p += bn4;
                            for (int j = 0; j < g.N - col; ++j) {
                                const float xhj = fmaxf(p[j], 0.f);
                                const float d = xhj - g.x[(int64_t)xr * g.ldx + col + j];
                                sr += d * d;
                                const float dpj = (p[j] > 0.f) ? 2.f * d : 0.f;
                                if (g.xhat) g.xhat[(int64_t)row * g.ldxhat + col + j] = xhj;
                                if (g.dP) g.dP[(int64_t)row * g.lddp + col + j] = dpj;
                                cs[n][j] += dpj;
                            }
                        }
                    }
                    sr += __shfl_xor(sr, 4, 64);
                    sr += __shfl_xor(sr, 8, 64);
                    sr += __shfl_xor(sr, 16, 64);
                    if (l31 < 4) rowsum[wn * BM + rloc] = sr;
                }
            }
            if (g.col_part) {
                // the 8 lanes that share a 4-column group (q = 0..3, both halves) -> one; then the WGM waves through LDS
                float* colbuf = lds + WGN * BM;  // [WGM][BN], behind the row sums
#pragma unroll
                for (int n = 0; n < TN; ++n) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = cs[n][j];
                        v += __shfl_xor(v, 1, 64);
                        v += __shfl_xor(v, 2, 64);
                        v += __shfl_xor(v, 32, 64);
                        cs[n][j] = v;
                    }
                    if (q == 0 && half == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) colbuf[wm * BN + wn * WTN + n * 32 + (l31 & ~3) + j] = cs[n][j];
                    }
                }
            }
        } else {
        float cs[TN];
#pragma unroll
        for (int n = 0; n < TN; ++n) cs[n] = 0.f;
        float bv[TN];
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = bn * BN + wn * WTN + n * 32 + l31;
            bv[n] = (g.bias && col < g.N) ? g.bias[col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rloc = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                const int row = bm * BM + rloc;
                float s = 0.f;
                if (row < g.M) {
                    const int xr = row % g.x_rows;
#pragma unroll
                    for (int n = 0; n < TN; ++n) {
                        const int col = bn * BN + wn * WTN + n * 32 + l31;
                        if (col < g.N) {
                            const float p = acc[i][n][e] + bv[n];
                            const float xh = fmaxf(p, 0.f);
                            const float d = xh - g.x[(int64_t)xr * g.ldx + col];
                            s += d * d;
                            const float dpv = (p > 0.f) ? 2.f * d : 0.f;
                            if (g.xhat) g.xhat[(int64_t)row * g.ldxhat + col] = xh;
                            if (g.dP) g.dP[(int64_t)row * g.lddp + col] = dpv;
                            cs[n] += dpv;
                        }
                    }
                }
                s = half_wave_sum(s);
                if (l31 == 0) rowsum[wn * BM + rloc] = s;
            }
        }
        if (g.col_part) {
            float* colbuf = lds + WGN * BM;
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const float v = cs[n] + __shfl_xor(cs[n], 32, 64);
                if (half == 0) colbuf[wm * BN + wn * WTN + n * 32 + l31] = v;
            }
        }
        }
        __syncthreads();
        if (tid < BM) {
            const int row = bm * BM + tid;
            if (row < g.M) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WGN; ++w) s += rowsum[w * BM + tid];
                g.se_part[(int64_t)bn * g.M + row] = s;
                if (bn == g.nt - 1)
                    for (int tz = g.nt; tz < g.se_tiles; ++tz) g.se_part[(int64_t)tz * g.M + row] = 0.f;
            }
        }
        if (g.col_part) {
            const float* colbuf = lds + WGN * BM;
            for (int c = tid; c < BN; c += WGM * WGN * 64) {
                const int col = bn * BN + c;
                if (col < g.N) {
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < WGM; ++w) sum += colbuf[w * BN + c];  // wave order: reproducible
                    g.col_part[(int64_t)bm * g.N + col] = sum;
                }
            }
        }
    }
}

// Block tile BM x BN, k-tile BK, 4 waves arranged WGM x WGN; each wave owns (BM/WGM) x (BN/WGN) as 32x32 MFMA blocks.
// XCD-aware (bijective) remap of the workgroup id: each XCD (private L2) gets a contiguous run of work items.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

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

// ====================================================================================================================
// bf16x3 GEMM: fp32 in, fp32 out, computed on the bf16 matrix cores.
//
// Every fp32 operand element a is split EXACTLY into three bf16 pieces by truncation, a = a0 + a1 + a2
// (a0 = top 16 bits of a, a1 = top 16 bits of a - a0, a2 = a - a0 - a1: 3 x 8 significant bits = the 24 of fp32), and
//     a * b  ~=  a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)
// (the dropped terms are below 2^-24 |a b|).  Each bf16 x bf16 product is exact in fp32 and the MFMA accumulates in
// fp32, so the result has fp32-GEMM accuracy (measured 1.5e-7 rel-L2 vs fp64 at K = 20000, plain fp32 GEMM 3.5e-7),
// while v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32: 6 MFMAs replace 8 -> 2.67x the
// matrix-core throughput of the exact-f32 path.  The split happens once per element while a tile is staged into LDS
// (its VALU instructions are interleaved with the wave's own MFMAs, 4 behind each: see X3Stage); LDS holds three bf16
// planes per operand, [plane][row][32 k]
// with 80-byte rows (64 B data + 16 B pad: conflict-free ds_read_b128 fragments).  Operands whose contiguous axis is
// not K (the "RC" images of the NN / TN layouts) are transposed in registers on the way to LDS (each thread owns a
// 4 k x 4 row patch), so the MFMA loop is identical for all three layouts.
// One LDS buffer + register prefetch of the next k-tile; two workgroups per CU.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#if MMVAE_X3_STAMPS
__device__ long long g_x3_stamps[32];
__device__ long long g_x3_trace[4096 * 4];  // per workgroup: 100 MHz wall clock at entry, loop start, loop end, exit
#endif
constexpr int X3_BK = 32;
constexpr int X3_LD = 80;  // bytes per LDS row per plane

__device__ __forceinline__ void x3_split(float a, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned u = __float_as_uint(a);
    p0 = u & 0xFFFF0000u;
    const float r1 = a - __uint_as_float(p0);
    p1 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(p1);
    p2 = __float_as_uint(r2);  // <= 8 significant bits left: already a bf16 value
}

// Four fp32 values that are consecutive in k -> three 8-byte groups of 4 bf16, written to the three planes.
__device__ __forceinline__ void x3_store4(char* S, int plane_bytes, int byte_off, float v0, float v1, float v2,
                                          float v3) {
    unsigned a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
    x3_split(v0, a0, a1, a2);
    x3_split(v1, b0, b1, b2);
    x3_split(v2, c0, c1, c2);
    x3_split(v3, d0, d1, d2);
    uint2 w;
    w.x = (a0 >> 16) | b0;
    w.y = (c0 >> 16) | d0;
    *reinterpret_cast<uint2*>(S + byte_off) = w;
    w.x = (a1 >> 16) | b1;
    w.y = (c1 >> 16) | d1;
    *reinterpret_cast<uint2*>(S + plane_bytes + byte_off) = w;
    w.x = (a2 >> 16) | (b2 & 0xFFFF0000u);
    w.y = (c2 >> 16) | (d2 & 0xFFFF0000u);
    *reinterpret_cast<uint2*>(S + 2 * plane_bytes + byte_off) = w;
}

// RC operand (k-slices contiguous along rows in HBM): each thread owns a 4 k x 4 row patch, so that after the split it
// holds 4 consecutive-k values for each of its 4 rows.  Patch p: row group c4 = p / 8, k group kq = p % 8.
template <int R, bool VEC, int NV>
__device__ __forceinline__ void x3_load_rc(f32x4 (&reg)[NV], unsigned (&valid)[NV], const float* __restrict__ P,
                                           int64_t ld, int r0, int Rtot, int k0, int Kend, int tid) {
    constexpr int NP = R / 4 * 8;  // patches per tile
    static_assert(NV == 4 * ((NP + NT - 1) / NT), "register tile size");
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) {
        const int pidx = tid + NT * i;
        const int c4 = pidx >> 3, kq = pidx & 7;
        const int x = r0 + c4 * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = k0 + kq * 4 + j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            unsigned vm = 0u;
            if (NP % NT == 0 || pidx < NP) {
                const bool ov = o < Kend;
                const float* p = P + (int64_t)(ov ? o : Kend - 1) * ld;
                if (VEC) {
                    const bool full = ov && (x + 3 < Rtot);
                    v = *reinterpret_cast<const f32x4*>(p + (full ? x : 0));
                    vm = full ? 0xFu : 0u;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = p[min(x + e, Rtot - 1)];
                        vm |= (ov && x + e < Rtot) ? (1u << e) : 0u;
                    }
                }
            }
            reg[i * 4 + j] = v;
            valid[i * 4 + j] = vm;
        }
    }
}

__device__ __forceinline__ f32x4 x3_mask(f32x4 v, unsigned valid, bool vec) {
    if (vec) {  // all-or-nothing per 16-byte group: one AND per element, no compares
        const unsigned m = (valid & 1u) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(__float_as_uint(v[j]) & m);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((valid >> j) & 1u) ? v[j] : 0.f;
    }
    return v;
}

// In VEC mode the validity of a 16-byte group is recomputed from its indices here (cheaper than carrying a mask
// register per group across the prefetch distance); `valid` is only read in the element-guarded mode.
template <int FORM, int R, bool VEC, int NV>
__device__ __forceinline__ void x3_store(char* S, const f32x4 (&reg)[NV], const unsigned (&valid)[NV], int tid, int r0,
                                         int Rtot, int k0, int Kend) {
    constexpr int PLANE = R * X3_LD;
    if (FORM == FORM_KC) {
        using T = Tile<FORM_KC, R, X3_BK>;
#pragma unroll
        for (int i = 0; i < T::VECS; ++i) {
            const int f = tid + NT * i;
            if (T::EXACT || f < T::NVEC) {
                const int row = f >> 3, c4 = f & 7;
                const unsigned vm = VEC ? ((r0 + row < Rtot && k0 + c4 * 4 + 3 < Kend) ? 0xFu : 0u) : valid[i];
                const f32x4 v = x3_mask(reg[i], vm, VEC);
                x3_store4(S, PLANE, row * X3_LD + c4 * 8, v[0], v[1], v[2], v[3]);
            }
        }
    } else {
        constexpr int NP = R / 4 * 8;
#pragma unroll
        for (int i = 0; i < NV / 4; ++i) {
            const int pidx = tid + NT * i;
            if (NP % NT == 0 || pidx < NP) {
                const int c4 = pidx >> 3, kq = pidx & 7;
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned vm =
                        VEC ? ((k0 + kq * 4 + j < Kend && r0 + c4 * 4 + 3 < Rtot) ? 0xFu : 0u) : valid[i * 4 + j];
                    v[j] = x3_mask(reg[i * 4 + j], vm, VEC);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    x3_store4(S, PLANE, (c4 * 4 + e) * X3_LD + kq * 8, v[0][e], v[1][e], v[2][e], v[3][e]);
            }
        }
    }
}

// Lean split for the pipelined loop: 4.5 VALU instructions per element (VALU and MFMA instructions of one SIMD do not
// overlap on gfx950 -- tools/ubench/overlap.hip -- so every instruction here is paid for in matrix-core time):
// 2 ANDs + 2 halves of a packed subtract per element, one v_perm_b32 per bf16 pair and plane.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 x3_top16(f32x2 v) {
    f32x2 h;
    h[0] = __uint_as_float(__float_as_uint(v[0]) & 0xFFFF0000u);
    h[1] = __uint_as_float(__float_as_uint(v[1]) & 0xFFFF0000u);
    return h;
}
// dword = [bf16(hi) : bf16(lo)] = the top halves of two fp32 (v_perm_b32: selector bytes 0-3 address the 2nd source)
__device__ __forceinline__ unsigned x3_pair(f32x2 v) {
    return __builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u);
}
// Scalar subtracts on purpose: v_pk_add_f32 beside MFMAs costs more than two v_sub_f32 (MI355X_MICROARCH.md, "price of
// one filler beside MFMAs"); the asm keeps the SLP vectoriser from re-packing them.
__device__ __forceinline__ float x3_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 x3_resid(f32x2 v) {
    const f32x2 h = x3_top16(v);
    f32x2 r;
    r[0] = x3_sub(v[0], h[0]);
    r[1] = x3_sub(v[1], h[1]);
    return r;
}
__device__ __forceinline__ void x3_pack4_lean(f32x2 lo, f32x2 hi, uint2 (&pk)[3]) {
    pk[0].x = x3_pair(lo);
    pk[0].y = x3_pair(hi);
    const f32x2 r1lo = x3_resid(lo), r1hi = x3_resid(hi);
    pk[1].x = x3_pair(r1lo);
    pk[1].y = x3_pair(r1hi);
    const f32x2 r2lo = x3_resid(r1lo), r2hi = x3_resid(r1hi);
    pk[2].x = x3_pair(r2lo);  // <= 8 significant bits left: the top half IS the value
    pk[2].y = x3_pair(r2hi);
}

// ---- pipelined (VEC-mode) staging of one operand tile of R rows x 32 k; R = 128 or 160, R / 32 "units" per thread.
// A unit = 4 consecutive-k fp32 values of one tile row = three uint2 of packed bf16 (one per plane).
//   KC operand (K contiguous in HBM): unit u = 16-byte group f = tid + 256 u: row f >> 3, k = 4 (f & 7).
//   RC operand (rows contiguous):     rows 0..127: each thread owns a 4 k x 4 row patch (row group c4 = tid >> 3,
//                                     k group kq = tid & 7), loaded as 4 row-vectors; unit u (< 4) = patch row u, its
//                                     4 k values are component u of the 4 loads (a register-name transpose).
//                                     rows 128..159 (R = 160): one row-vector per thread at k = 4 (tid >> 5) + (tid & 3),
//                                     row group rq = (tid >> 2) & 7; the 4 lanes of a quad hold 4 consecutive k of the
//                                     same 4 rows and transpose them with DPP quad broadcasts: unit 4 = row 4 rq + q.
// Rows/columns beyond the matrix are loaded from clamped (finite, in-matrix) addresses: they only reach accumulators
// the epilogue never stores.  Only the K tail must be zeroed, and only in the last k-tile (a wave-uniform branch).
// Loads are addressed as (wave-uniform tile base, advanced by the k-tile) + (per-thread byte offset, constant over the
// k-loop): the offsets are computed once per work item, the loop itself spends no VALU on addresses (global_load with an
// SGPR base and a 32-bit VGPR offset).  Requires K % 32 == 0 (no k clamp) and < 4 GiB between a tile's first and last
// byte of one k-tile (124 * ld bytes).
template <int FORM, int R>
__device__ __forceinline__ void x3p_offsets(unsigned (&off)[R / 32], int64_t ld, int r0, int Rtot, int tid) {
    constexpr int NU = R / 32;
    if (FORM == FORM_KC) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int f = tid + NT * u;
            const int row = r0 + (f >> 3);
            off[u] = (unsigned)(((int64_t)((row < Rtot ? row : Rtot - 1) - r0) * ld + 4 * (f & 7)) * 4);
        }
    } else {
        const int c4 = tid >> 3, kq = tid & 7;
        // a group that straddles the edge (extent % 4 != 0, operands with tail slack) is read whole: its rows beyond
        // the matrix only reach accumulators that are never stored
        const int Rpad = (Rtot + 3) & ~3;
        const int xo = (r0 + 4 * c4 + 3 < Rpad) ? 4 * c4 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) off[j] = (unsigned)(((int64_t)(4 * kq + j) * ld + xo) * 4);
        if (NU == 5) {
            const int xt = 128 + 4 * ((tid >> 2) & 7);
            off[NU - 1] = (unsigned)(((int64_t)(4 * (tid >> 5) + (tid & 3)) * ld + ((r0 + xt + 3 < Rpad) ? xt : 0)) * 4);
        }
    }
}
// tile base of k-tile kt: first row/column r0 of the tile, first k of the k-tile
template <int FORM>
__device__ __forceinline__ const char* x3p_base(const float* __restrict__ P, int64_t ld, int r0, int kt) {
    const int64_t k0 = (int64_t)kt * X3_BK;
    return reinterpret_cast<const char*>(FORM == FORM_KC ? P + (int64_t)r0 * ld + k0 : P + k0 * ld + r0);
}
template <int NU>
__device__ __forceinline__ void x3p_load(f32x4 (&reg)[NU], const char* __restrict__ base, const unsigned (&off)[NU]) {
#pragma unroll
    for (int u = 0; u < NU; ++u) reg[u] = *reinterpret_cast<const f32x4*>(base + off[u]);
}

template <int FORM, int R, bool TAILCHK>
__device__ __forceinline__ void x3p_split(const f32x4 (&reg)[R / 32], int u, uint2 (&pk)[3], int tid, int k0, int Kend) {
    float v[4];
    const bool tail = TAILCHK && (k0 + X3_BK > Kend);  // steady-state k-tiles are instantiated without the check
    if (FORM == FORM_KC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = reg[u][j];
        if (tail) {
            const unsigned m = (k0 + 4 * ((tid + NT * u) & 7) + 3 < Kend) ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(__float_as_uint(v[j]) & m);
        }
    } else if (u < 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = reg[j][u];
        if (tail) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (k0 + 4 * (tid & 7) + j < Kend) ? v[j] : 0.f;
        }
    } else {
        // 4 x 4 transpose inside each quad (lane q holds k = q, rows in its 4 components -> row q, k in v[0..3]):
        // two DPP exchange stages with plain selects (a 4-way select chain here compiles to divergent branches, which
        // would cut the k-loop body into basic blocks and stop the MFMA / VALU interleave)
        const f32x4 t = reg[R / 32 - 1];
        const int q = tid & 3;
        const bool odd = q & 1, hi2 = q & 2;
        const float r0 = quad_perm<1, 0, 3, 2>(odd ? t[0] : t[1]), r1 = quad_perm<1, 0, 3, 2>(odd ? t[2] : t[3]);
        const float c0 = odd ? r0 : t[0], c1 = odd ? t[1] : r0, c2 = odd ? r1 : t[2], c3 = odd ? t[3] : r1;
        const float t0 = quad_perm<2, 3, 0, 1>(hi2 ? c0 : c2), t1 = quad_perm<2, 3, 0, 1>(hi2 ? c1 : c3);
        v[0] = hi2 ? t0 : c0;
        v[1] = hi2 ? t1 : c1;
        v[2] = hi2 ? c2 : t0;
        v[3] = hi2 ? c3 : t1;
        if (tail) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (k0 + 4 * (tid >> 5) + j < Kend) ? v[j] : 0.f;
        }
    }
    f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
    x3_pack4_lean(lo, hi, pk);
}

template <int FORM, int R>
__device__ __forceinline__ void x3p_write(char* S, int u, const uint2 (&pk)[3], int tid) {
    constexpr int PLANE = R * X3_LD;
    int off;
    if (FORM == FORM_KC) {
        const int f = tid + NT * u;
        off = (f >> 3) * X3_LD + (f & 7) * 8;
    } else if (u < 4) {
        off = ((tid >> 3) * 4 + u) * X3_LD + (tid & 7) * 8;
    } else {
        off = (128 + 4 * ((tid >> 2) & 7) + (tid & 3)) * X3_LD + (tid >> 5) * 8;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(S + p * PLANE + off) = pk[p];
}

// ---- staged split.  The 22 VALU instructions that split one unit (4 consecutive-k fp32 values) into its three packed
// bf16 planes, cut into 6 chunks of <= 4: the hand-interleaved k-step issues one chunk behind each MFMA of a 6-MFMA
// group, so that a wave's own VALU work fits into the issue slots its MFMAs leave free (an MFMA holds the SIMD's vector
// issue for 8 of its 32 cycles; 4 single-issue VALU instructions take 16).  Left to the compiler, the 6 MFMAs are emitted
// back to back and the unit's instructions as one run of 25-45, and the matrix pipe idles through every run unless the
// partner wave of the SIMD happens to be in its MFMA phase.
struct X3Stage {
    float v[4], h[4], r[4];
};
template <int FORM, int R>
__device__ __forceinline__ void x3s_fetch(X3Stage& s, const f32x4 (&reg)[R / 32], int u, int tid) {
    if (FORM == FORM_KC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s.v[j] = reg[u][j];
    } else if (u < 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s.v[j] = reg[j][u];
    } else {  // rows 128..159 of a 160-row RC tile: 4 x 4 transpose inside each quad (see x3p_split)
        const f32x4 t = reg[R / 32 - 1];
        const int q = tid & 3;
        const bool odd = q & 1, hi2 = q & 2;
        const float r0 = quad_perm<1, 0, 3, 2>(odd ? t[0] : t[1]), r1 = quad_perm<1, 0, 3, 2>(odd ? t[2] : t[3]);
        const float c0 = odd ? r0 : t[0], c1 = odd ? t[1] : r0, c2 = odd ? r1 : t[2], c3 = odd ? t[3] : r1;
        const float t0 = quad_perm<2, 3, 0, 1>(hi2 ? c0 : c2), t1 = quad_perm<2, 3, 0, 1>(hi2 ? c1 : c3);
        s.v[0] = hi2 ? t0 : c0;
        s.v[1] = hi2 ? t1 : c1;
        s.v[2] = hi2 ? c2 : t0;
        s.v[3] = hi2 ? c3 : t1;
    }
}
__device__ __forceinline__ float x3_hi(float a) { return __uint_as_float(__float_as_uint(a) & 0xFFFF0000u); }
__device__ __forceinline__ unsigned x3_pair2(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
template <int P>
__device__ __forceinline__ void x3s_phase(X3Stage& s, uint2 (&pk)[3]) {
    if constexpr (P == 0) {
        pk[0].x = x3_pair2(s.v[0], s.v[1]);
        pk[0].y = x3_pair2(s.v[2], s.v[3]);
        s.h[0] = x3_hi(s.v[0]);
        s.h[1] = x3_hi(s.v[1]);
    } else if constexpr (P == 1) {
        s.h[2] = x3_hi(s.v[2]);
        s.h[3] = x3_hi(s.v[3]);
        s.r[0] = x3_sub(s.v[0], s.h[0]);
        s.r[1] = x3_sub(s.v[1], s.h[1]);
    } else if constexpr (P == 2) {
        s.r[2] = x3_sub(s.v[2], s.h[2]);
        s.r[3] = x3_sub(s.v[3], s.h[3]);
        pk[1].x = x3_pair2(s.r[0], s.r[1]);
        s.h[0] = x3_hi(s.r[0]);
    } else if constexpr (P == 3) {
        s.h[1] = x3_hi(s.r[1]);
        s.h[2] = x3_hi(s.r[2]);
        s.h[3] = x3_hi(s.r[3]);
        pk[1].y = x3_pair2(s.r[2], s.r[3]);
    } else if constexpr (P == 4) {
        s.r[0] = x3_sub(s.r[0], s.h[0]);
        s.r[1] = x3_sub(s.r[1], s.h[1]);
        s.r[2] = x3_sub(s.r[2], s.h[2]);
        s.r[3] = x3_sub(s.r[3], s.h[3]);
    } else {
        pk[2].x = x3_pair2(s.r[0], s.r[1]);  // <= 8 significant bits left: the top half IS the value
        pk[2].y = x3_pair2(s.r[2], s.r[3]);
    }
}
#ifndef MMVAE_X3_PRIO_TOGGLE
#define MMVAE_X3_PRIO_TOGGLE 0  // 1: alternate the wave priority per k-tile, opposite phase per wave slot (measured: the
                                // two workgroups of a CU finish closer together, the kernel does not get shorter)
#endif
#define X3_SB() __builtin_amdgcn_sched_barrier(0)

template <int FORM, int R>
struct X3Regs {
    static constexpr int NV = (FORM == FORM_KC) ? Tile<FORM_KC, R, X3_BK>::VECS : 4 * ((R / 4 * 8 + NT - 1) / NT);
};

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
#define X3_GROUP_FS(C, A2, A1, A0, B0, B1, B2, U, RD, FS)                  \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B0, C, 0, 0, 0);       \
    chunk(U, I0{});                                                        \
    FS(0);                                                                 \
    X3_SB();                                                               \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, C, 0, 0, 0);       \
    chunk(U, I1{});                                                        \
    X3_SB();                                                               \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B2, C, 0, 0, 0);       \
    chunk(U, I2{});                                                        \
    FS(2);                                                                 \
    X3_SB();                                                               \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, C, 0, 0, 0);       \
    chunk(U, I3{});                                                        \
    RD(0);                                                                 \
    FS(3);                                                                 \
    X3_SB();                                                               \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, C, 0, 0, 0);       \
    chunk(U, I4{});                                                        \
    RD(1);                                                                 \
    FS(4);                                                                 \
    X3_SB();                                                               \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, C, 0, 0, 0);       \
    chunk(U, I5{});                                                        \
    RD(2);                                                                 \
    FS(5);                                                                 \
    X3_SB();
#define X3_GROUP(C, A2, A1, A0, B0, B1, B2, U, RD)                        \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B0, C, 0, 0, 0);       \
    chunk(U, I0{});                                                        \
    X3_SB();                                                               \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, C, 0, 0, 0);       \
    chunk(U, I1{});                                                        \
    X3_SB();                                                               \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B2, C, 0, 0, 0);       \
    chunk(U, I2{});                                                        \
    X3_SB();                                                               \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, C, 0, 0, 0);       \
    chunk(U, I3{});                                                        \
    RD(0);                                                                 \
    X3_SB();                                                               \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, C, 0, 0, 0);       \
    chunk(U, I4{});                                                        \
    RD(1);                                                                 \
    X3_SB();                                                               \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, C, 0, 0, 0);       \
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

// =====================================================================================================================
// Wave-specialised bf16x3 kernel (r2): ONE 512-thread workgroup per CU, two roles.
//
//   waves 0-3  "multipliers": fragment reads + MFMAs only.  4 waves = one per SIMD; wave tile (BM/WGM) x (BN/WGN).
//   waves 4-7  "stagers":     global loads of the fp32 operand tiles, the exact 3-way bf16 split (22 VALU per 4
//                             elements) and the LDS plane writes -- for the k-tile AFTER the one being multiplied.
//
// What the 2 x 4-wave kernel above loses (profiles/r2_stamps: the first-dispatched workgroup of a CU needs 3 550 cycles
// per k-tile for 1 536 cycles of its own MFMAs; its CU partner gets what is left, finishes 30 us later and runs the last
// third of the kernel alone at 43 % matrix-core occupancy): every wave alternates between a matrix phase and a
// 1 450-cycle staging phase (fragment waits, LDS write burst, two barriers), and two independent workgroups only
// overlap those phases by luck.  Here the phases are different waves of the SAME SIMD: the stager's VALU / LDS-write /
// VMEM instructions issue beside the multiplier's MFMAs (separate pipes; the SIMD's vector issue is needed 8 of an
// MFMA's 32 cycles), LDS is double buffered (the whole 160 KiB: two images of (BM + BN) rows x 3 planes x 64 B), there
// is one barrier per k-tile, placed inside the multiplier's k-tile BEFORE its last streamed block, when all its
// fragment reads have returned -- the last block's MFMAs then cover the first fragment reads of the next k-tile.
// The loop is persistent over the workgroup's items, and the stagers run ahead across item boundaries: the prologue of
// the next output tile (HBM latency) hides behind the epilogue stores of the current one.
//
// LDS image: [plane][row][64 B] (32 k as bf16), no padding (2 x 79 872 B would not fit with 80-byte rows); the 16-byte
// chunk c of row r sits at chunk c ^ ((r >> 2) & 3): conflict-free for the multipliers' ds_read_b128 (its 16-lane
// groups cover rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}: (r >> 2) & 3 takes each value once per r % 4) and for
// the stagers' ds_write_b64 (16 lanes = 2 whole rows).  The last 4 KiB of LDS are the epilogue's scratch.
constexpr int XW_ROWB = 64;                   // bytes per row per plane
constexpr int XW_SCRATCH = 4096;              // epilogue scratch behind the two images
__device__ __forceinline__ int xw_off(int row, int chunk) { return row * XW_ROWB + ((chunk ^ ((row >> 2) & 3)) << 4); }

// One operand of the stager: R rows, NU = R / 32 units per thread (a unit = 4 consecutive-k fp32 values of one row).
// R = 256 over a rows-contiguous operand is handled as two 128-row halves of the 4 k x 4 row patch scheme of x3p_*.
template <int FORM, int R>
struct XwOperand {
    static constexpr int NU = R / 32;
    static constexpr bool SPLIT256 = (FORM == FORM_RC && R == 256);
    unsigned off[NU];
    f32x4 raw0[NU], raw1[NU];  // two raw register sets (two k-tiles in flight), selected at compile time

    __device__ __forceinline__ void offsets(int64_t ld, int r0, int Rtot, int st) {
        if constexpr (SPLIT256) {
            unsigned lo[4], hi[4];
            // the second half starts at r0 + 128 when that is inside the matrix; otherwise it re-reads the first half
            // (its rows only reach accumulators that are never stored) -- never an address beyond the operand
            const int r1 = (r0 + 128 < Rtot) ? r0 + 128 : r0;
            x3p_offsets<FORM_RC, 128>(lo, ld, r0, Rtot, st);
            x3p_offsets<FORM_RC, 128>(hi, ld, r1, Rtot, st);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                off[j] = lo[j];
                off[4 + j] = hi[j] + (unsigned)(r1 - r0) * 4u;  // hi[] is relative to r1 (x3p_base adds r0 only)
            }
        } else {
            x3p_offsets<FORM, R>(off, ld, r0, Rtot, st);
        }
    }
    template <int SET>
    __device__ __forceinline__ void load(const char* __restrict__ base) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + off[u]);
            if (SET == 0)
                raw0[u] = v;
            else
                raw1[u] = v;
        }
    }
    // split unit u of raw set SET and write its three planes into the image at S (plane stride R * 64 B)
    template <int SET>
    __device__ __forceinline__ void stage_unit(int u, char* S, int st) {
        const f32x4 (&raw)[NU] = SET == 0 ? raw0 : raw1;
        uint2 pk[3];
        int row, k4;
        if constexpr (FORM == FORM_KC) {
            x3p_split<FORM_KC, R, false>(raw, u, pk, st, 0, 0);
            const int f = st + NT * u;
            row = f >> 3;
            k4 = f & 7;
        } else if constexpr (SPLIT256) {
            f32x4 half[4];
            const int h = u >> 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) half[j] = raw[4 * h + j];
            x3p_split<FORM_RC, 128, false>(half, u & 3, pk, st, 0, 0);
            row = 128 * h + (st >> 3) * 4 + (u & 3);
            k4 = st & 7;
        } else {
            x3p_split<FORM_RC, R, false>(raw, u, pk, st, 0, 0);
            if (u < 4) {
                row = (st >> 3) * 4 + u;
                k4 = st & 7;
            } else {  // rows 128..159 of a 160-row tile (DPP-transposed unit)
                row = 128 + 4 * ((st >> 2) & 7) + (st & 3);
                k4 = st >> 5;
            }
        }
        const int o = xw_off(row, k4 >> 1) + (k4 & 1) * 8;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(S + p * (R * XW_ROWB) + o) = pk[p];
    }
    template <int SET>
    __device__ __forceinline__ void load_one(int u, const char* __restrict__ base) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + off[u]);
        if (SET == 0)
            raw0[u] = v;
        else
            raw1[u] = v;
    }
    // Stage every unit of raw set SET into the image at S and re-issue each raw register as the load of the k-tile
    // two ahead (base `next`) as soon as its last unit is split: the 13 loads of a k-tile then enter the memory
    // pipeline one at a time between ~100-cycle runs of VALU work instead of as one burst that the wave sits behind
    // (16 B x 64 lanes = 16 cycles of address processing each, shared by the CU's four stagers).
    template <int SET>
    __device__ __forceinline__ void stage_and_reload(char* S, int st, const char* __restrict__ next) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            stage_unit<SET>(u, S, st);
            if constexpr (FORM == FORM_KC) {
                load_one<SET>(u, next);
            } else {
                if (u < (NU / 4) * 4) {
                    if ((u & 3) == 3) {
#pragma unroll
                        for (int j = u - 3; j <= u; ++j) load_one<SET>(j, next);
                    }
                } else {
                    load_one<SET>(u, next);  // the 32 extra rows of a 160-row tile: one register, one unit
                }
            }
        }
    }
};

// position in the workgroup's stream of k-tiles (items = output tile x split-K slice, looped over persistently)
struct XwCursor {
    int w, kt, kt_end, bm, bn, z;
    __device__ __forceinline__ bool valid(const GemmArgs& g) const { return w < g.nwork; }
    __device__ __forceinline__ void open(const GemmArgs& g) {  // item w -> tile, slice, k-tile range
        if (w >= g.nwork) return;
        const int tiles = g.mt * g.nt;
        z = w / tiles;
        const int t = w - z * tiles;
        bm = t % g.mt;
        bn = t / g.mt;
        kt = z * g.ktiles_per_split;
        kt_end = min(kt + g.ktiles_per_split, g.ktiles);
    }
    __device__ __forceinline__ bool advance(const GemmArgs& g, int nwg) {  // next k-tile; true when a new item began
        if (++kt < kt_end) return false;
        w += nwg;
        open(g);
        return true;
    }
};

template <int AFORM, int BFORM, int BM, int BN, int WGM, int WGN, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_x3w_kernel(const GemmArgs g) {
    static_assert(WGM * WGN == 4, "4 multiplier wavefronts");
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr bool KEEP_A = TM <= TN;  // fragments of the short side stay in registers over a k-step, the long side streams
    constexpr int TK = KEEP_A ? TM : TN, TL = KEEP_A ? TN : TM;
    constexpr int PA = BM * XW_ROWB, PB = BN * XW_ROWB;       // plane strides
    constexpr int IMG = 3 * (PA + PB);                        // one image (A planes, then B planes)
    static_assert(2 * IMG + XW_SCRATCH <= 160 * 1024, "two images + scratch must fit the CU's LDS");
    __shared__ __attribute__((aligned(16))) char lds[2 * IMG + XW_SCRATCH];

    const int tid = threadIdx.x;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    float* scratch = reinterpret_cast<float*>(lds + 2 * IMG);
    // workgroup barriers an item's epilogue contains (the stagers take part in them)
    const int epi_barriers = (EPI == EPI_RECON) ? 2 : (g.sq_part ? 2 : 0);

    if (tid >= NT) {
        // ------------------------------------------------------------------------------------------------ stagers
        const int st = tid - NT;
#ifndef MMVAE_XW_STAGER_PRIO
#define MMVAE_XW_STAGER_PRIO 0
#endif
        // (the stagers are the longer role -- ~4 400 against ~3 900 cycles per k-tile, profiles/r2_x3w_stamps.txt)
        if (MMVAE_XW_STAGER_PRIO) __builtin_amdgcn_s_setprio(MMVAE_XW_STAGER_PRIO);
        XwOperand<AFORM, BM> oa;
        XwOperand<BFORM, BN> ob;
        XwCursor ld_c;  // the k-tile the next load fetches
        XwCursor br_c;  // the k-tile whose barrier comes next (= the one the multipliers work on)
        ld_c.w = br_c.w = L;
        ld_c.open(g);
        br_c.open(g);
        if (!br_c.valid(g)) return;
        // The loads themselves sit in straight-line code (a wave-uniform branch AROUND a load makes the compiler drain
        // the memory pipeline at the join); a load past the end of the stream re-reads the last k-tile.  The per-thread
        // offsets (64-bit multiplies: ~1 000 cycles for both operands) are recomputed only when the stream enters a new
        // item, in a branch that holds VALU work only.
#if MMVAE_X3_STAMPS
        long long sst[4] = {0, 0, 0, 0};
        long long tp = clock64();
#define XW_SSTAMP(i)                     \
    {                                    \
        const long long tn_ = clock64(); \
        sst[i] += tn_ - tp;              \
        tp = tn_;                        \
    }
#else
#define XW_SSTAMP(i)
#endif
        int off_w = -1;  // item the offsets were computed for
        auto refresh_offsets = [&]() {
            if (ld_c.w != off_w) {
                oa.offsets(g.lda, ld_c.bm * BM, g.M, st);
                ob.offsets(g.ldb, ld_c.bn * BN, g.N, st);
                off_w = ld_c.w;
            }
        };
        auto bump = [&]() {
            XwCursor nx = ld_c;
            nx.advance(g, nwg);
            if (nx.valid(g)) ld_c = nx;  // (a select per field, not a branch around the loads)
        };
        auto issue_load = [&](auto SET) {  // prologue only
            refresh_offsets();
            oa.template load<decltype(SET)::value>(x3p_base<AFORM>(g.A, g.lda, ld_c.bm * BM, ld_c.kt));
            ob.template load<decltype(SET)::value>(x3p_base<BFORM>(g.B, g.ldb, ld_c.bn * BN, ld_c.kt));
            bump();
        };
        auto stage = [&](auto SET, int img) {  // prologue only
            constexpr int S_ = decltype(SET)::value;
            char* As = lds + img * IMG;
            char* Bs = As + 3 * PA;
#pragma unroll
            for (int u = 0; u < XwOperand<AFORM, BM>::NU; ++u) oa.template stage_unit<S_>(u, As, st);
#pragma unroll
            for (int u = 0; u < XwOperand<BFORM, BN>::NU; ++u) ob.template stage_unit<S_>(u, Bs, st);
        };
        // steady state: stage raw set SET into image `img`, re-issuing its registers as the loads of the k-tile at ld_c
        auto stage_reload = [&](auto SET, int img) {
            constexpr int S_ = decltype(SET)::value;
            char* As = lds + img * IMG;
            char* Bs = As + 3 * PA;
#if MMVAE_X3_STAMPS
            {  // diagnostic: how long does the oldest raw set still take to arrive?  (its NU_A + NU_B loads are the oldest
               // outstanding ones; the other set's are younger)
                constexpr int YOUNGER = XwOperand<AFORM, BM>::NU + XwOperand<BFORM, BN>::NU;
                const long long t0_ = clock64();
                if (YOUNGER == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
                if (YOUNGER == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                sst[1] += clock64() - t0_;
                tp = clock64();
            }
#endif
            // NOTE: the offsets in registers belong to the item of ld_c; the raw data being split was loaded with the
            // offsets valid at ITS load time -- only the new loads use the refreshed ones
            refresh_offsets();
            oa.template stage_and_reload<S_>(As, st, x3p_base<AFORM>(g.A, g.lda, ld_c.bm * BM, ld_c.kt));
            ob.template stage_and_reload<S_>(Bs, st, x3p_base<BFORM>(g.B, g.ldb, ld_c.bn * BN, ld_c.kt));
            bump();
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        issue_load(S0{});  // element 0 -> set 0
        issue_load(S1{});  // element 1 -> set 1
        stage(S0{}, 0);
        issue_load(S0{});  // element 2 -> set 0
        __syncthreads();   // image 0 holds element 0
        // per element s of the stream (the one the multipliers work on): stage element s + 1 (it arrived in raw set
        // (s + 1) & 1) into image (s + 1) & 1, re-issue that set as the load of element s + 3, then barrier #s.  Past the
        // end of the stream the same code stages and loads data nobody reads.
        while (true) {
            stage_reload(S1{}, 1);
            XW_SSTAMP(0)
            __syncthreads();
            XW_SSTAMP(2)
#if MMVAE_X3_STAMPS
            sst[3] += 1;
#endif
            if (br_c.advance(g, nwg))
                for (int e = 0; e < epi_barriers; ++e) __syncthreads();
            if (!br_c.valid(g)) break;
            stage_reload(S0{}, 0);
            XW_SSTAMP(0)
            __syncthreads();
            XW_SSTAMP(2)
#if MMVAE_X3_STAMPS
            sst[3] += 1;
#endif
            if (br_c.advance(g, nwg))
                for (int e = 0; e < epi_barriers; ++e) __syncthreads();
            if (!br_c.valid(g)) break;
        }
#if MMVAE_X3_STAMPS
        if (bid == 0 && (tid & 63) == 0)
            for (int i = 0; i < 4; ++i) g_x3_stamps[16 + (tid >> 6) - 4 + 4 * i - 0] = sst[i];  // slots 16..31: [i][stager wave]
#endif
        return;
    }

    // -------------------------------------------------------------------------------------------------- multipliers
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const int swz = (l31 >> 2) & 3;
    // byte offset of this lane's fragment inside a plane for k-step ks: row l31 of a 32-row block, chunk 2 ks + half
    const int coff[2] = {l31 * XW_ROWB + (((0 + half) ^ swz) << 4), l31 * XW_ROWB + (((2 + half) ^ swz) << 4)};
    const int a_row0 = wm * WTM * XW_ROWB, b_row0 = wn * WTN * XW_ROWB;
    auto frag = [&](const char* plane_base, int block, int ks) {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(plane_base + block * (32 * XW_ROWB) + coff[ks]));
    };
    // kept side: all TK blocks x 3 planes of a k-step; streamed side: one block x 3 planes, double buffered
    bf16x8 fk[2][3][TK], fs[2][3];
    auto load_kept = [&](int set, const char* img, int ks) {
        const char* base = KEEP_A ? img + a_row0 : img + 3 * PA + b_row0;
        constexpr int P = KEEP_A ? PA : PB;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int k = 0; k < TK; ++k) fk[set][p][k] = frag(base + p * P, k, ks);
    };
    auto load_stream = [&](int set, const char* img, int l, int ks) {
        const char* base = KEEP_A ? img + 3 * PA + b_row0 : img + a_row0;
        constexpr int P = KEEP_A ? PB : PA;
#pragma unroll
        for (int p = 0; p < 3; ++p) fs[set][p] = frag(base + p * P, l, ks);
    };

    XwCursor c;
    c.w = L;
    c.open(g);
    if (!c.valid(g)) return;
    int s = 0;
    // the multipliers' MFMAs and fragment reads win every issue arbitration against the stager wave of their SIMD: the
    // matrix pipe sets the pace, the stagers fill the slots it leaves
#ifndef MMVAE_XW_PRIO
#define MMVAE_XW_PRIO 0
#endif
    if (MMVAE_XW_PRIO) __builtin_amdgcn_s_setprio(MMVAE_XW_PRIO);
    __syncthreads();  // image 0 holds element 0
    load_kept(0, lds, 0);
    load_stream(0, lds, 0, 0);
#if MMVAE_X3_STAMPS
    long long mst[4] = {0, 0, 0, 0};  // before the barrier, in the barrier, behind it, k-tiles
    long long tq = clock64();
    if (tid == 0 && bid < 4096) g_x3_trace[bid * 4 + 0] = wall_clock64();
#endif
    while (c.valid(g)) {
        const int bm = c.bm, bn = c.bn, z = c.z;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][n][e] = 0.f;
        bool more = true;
        while (more) {
            const char* img = lds + (s & 1) * IMG;
            const char* nxt = lds + ((s + 1) & 1) * IMG;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int l = 0; l < TL; ++l) {
                    const int cur = (ks * TL + l) & 1;  // fs set of this block
                    const bool last = ks == 1 && l == TL - 1;
                    if (last) {
                        // every fragment of this k-tile has been requested: wait for them, then the k-tile's barrier;
                        // behind it the other image holds the next k-tile, whose first fragments are read under the
                        // MFMAs of this last block
#if MMVAE_X3_STAMPS
                        {
                            const long long tn_ = clock64();
                            mst[0] += tn_ - tq;
                            tq = tn_;
                        }
#endif
                        __syncthreads();
#if MMVAE_X3_STAMPS
                        {
                            const long long tn_ = clock64();
                            mst[1] += tn_ - tq;
                            tq = tn_;
                            mst[3] += 1;
                        }
#endif
                        load_kept(0, nxt, 0);
                        load_stream(cur ^ 1, nxt, 0, 0);
                    } else if (l + 1 < TL) {
                        load_stream(cur ^ 1, img, l + 1, ks);
                    } else {  // last block of k-step 0: first streamed block of k-step 1
                        load_stream(cur ^ 1, img, 0, 1);
                    }
                    if (ks == 0 && l == 0) load_kept(1, img, 1);  // kept fragments of k-step 1, a whole k-step ahead
#pragma unroll
                    for (int k = 0; k < TK; ++k) {
                        f32x16 cc = KEEP_A ? acc[k][l] : acc[l][k];
                        const bf16x8 a0 = KEEP_A ? fk[ks][0][k] : fs[cur][0], a1 = KEEP_A ? fk[ks][1][k] : fs[cur][1],
                                     a2 = KEEP_A ? fk[ks][2][k] : fs[cur][2];
                        const bf16x8 b0 = KEEP_A ? fs[cur][0] : fk[ks][0][k], b1 = KEEP_A ? fs[cur][1] : fk[ks][1][k],
                                     b2 = KEEP_A ? fs[cur][2] : fk[ks][2][k];
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, cc, 0, 0, 0);  // smallest terms first
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, cc, 0, 0, 0);
                        if (KEEP_A)
                            acc[k][l] = cc;
                        else
                            acc[l][k] = cc;
                    }
                }
            }
            ++s;
            more = !c.advance(g, nwg);
#if MMVAE_X3_STAMPS
            {
                const long long tn_ = clock64();
                mst[2] += tn_ - tq;
                tq = tn_;
            }
#endif
        }
#if MMVAE_X3_STAMPS
        if (tid == 0 && bid < 4096) g_x3_trace[bid * 4 + 2] = wall_clock64();
#endif
        gemm_epilogue<BM, BN, WGM, WGN, EPI>(acc, g, bm, bn, z, scratch);
        if (EPI == EPI_RECON) __syncthreads();
#if MMVAE_X3_STAMPS
        tq = clock64();
#endif
    }
#if MMVAE_X3_STAMPS
    if (bid == 0 && lane == 0)
        for (int i = 0; i < 4; ++i) g_x3_stamps[wave + 4 * i] = mst[i];  // slots 0..15: [i][multiplier wave]
    if (tid == 0 && bid < 4096) {
        __builtin_amdgcn_s_waitcnt(0);
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

// The wave-specialised kernel is taken where its tiles fill the chip; MMVAE_X3W=0 keeps the 2 x 4-wave kernel (read at
// every launch: tests and A/B runs toggle it).  Interleaved A/B of the C2 step on one box: 1.130 against 1.150 ms.
bool x3w_enabled() {
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

static int gemm_f32_impl(int layout, int M, int N, int K, float alpha, const float* A, int64_t lda, const float* B,
                         int64_t ldb, float* C, int64_t ldc, const float* bias, unsigned flags, int splitk,
                         float* workspace, size_t workspace_bytes, float* sq_partials, int64_t sq_capacity,
                         mmvae_stream_t stream);

extern "C" int mmvae_gemm_f32(int layout, int M, int N, int K, float alpha, const float* A, int64_t lda,
                              const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, unsigned flags,
                              int splitk, float* workspace, size_t workspace_bytes, mmvae_stream_t stream) {
    return gemm_f32_impl(layout, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, flags, splitk, workspace,
                         workspace_bytes, nullptr, 0, stream);
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

static int gemm_f32_impl(int layout, int M, int N, int K, float alpha, const float* A, int64_t lda, const float* B,
                         int64_t ldb, float* C, int64_t ldc, const float* bias, unsigned flags, int splitk,
                         float* workspace, size_t workspace_bytes, float* sq_partials, int64_t sq_capacity,
                         mmvae_stream_t stream) {
    if (layout < 0 || layout > 2 || M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return MMVAE_ERR_ARG;
    if (splitk < 0 || ldc < N) return MMVAE_ERR_ARG;
    // leading-dimension sanity: contiguous axis extent must fit in the stride
    const int64_t a_inner = (layout == MMVAE_GEMM_TN) ? M : K;
    const int64_t b_inner = (layout == MMVAE_GEMM_NT) ? K : N;
    if (lda < a_inner || ldb < b_inner) return MMVAE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;

    int tile_id, sk_auto;
    plan(layout, M, N, K, &tile_id, &sk_auto);
    if (splitk == 0) splitk = sk_auto;
    if (splitk > 1 && tile_id == 1) tile_id = 0;  // split-K slices use the square tiles
    int aligned = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    if (aligned && M % 4 == 0 && N % 4 == 0 && K % 4 == 0) aligned = 2;
    const bool x3v = x3_vec_ok(layout, M, N, (flags & MMVAE_GEMM_OPERAND_SLACK) != 0);
    flags &= ~MMVAE_GEMM_OPERAND_SLACK;
    if (g_precision == MMVAE_GEMM_PRECISION_BF16X3 && tile_id != 2)  // chip-filling GEMMs: bf16x3 cores
        tile_id = (x3v && K % X3_BK == 0) ? x3_tile_regular(M, N, splitk, true) : 3;
    if (tile_id >= 6) {  // the persistent kernel's k-tile stream has no empty items: every slice must own a k-tile
        const int kt32 = K / X3_BK, kps = ceil_div_i(kt32, splitk);
        if ((int64_t)(splitk - 1) * kps >= kt32) tile_id = 3;
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

extern "C" int mmvae_decoder_recon_rows_colsum_f32(int rows, int x_rows, int G, int H, const float* h, int64_t ldh,
                                                   const float* W, int64_t ldw, const float* bias, const float* x,
                                                   int64_t ldx, float* xhat, int64_t ldxhat, float* dP, int64_t lddp,
                                                   float* se_part, float* col_part, mmvae_stream_t stream) {
    if (rows <= 0 || x_rows <= 0 || rows % x_rows != 0 || G <= 0 || H <= 0 || !h || !W || !x || !se_part)
        return MMVAE_ERR_ARG;
    if (ldh < H || ldw < H || ldx < G) return MMVAE_ERR_ARG;
    if (xhat && ldxhat < G) return MMVAE_ERR_ARG;
    if (dP && lddp < G) return MMVAE_ERR_ARG;
    GemmArgs g = {};
    g.A = h;
    g.B = W;
    g.bias = bias;
    g.lda = ldh;
    g.ldb = ldw;
    g.M = rows;
    g.N = G;
    g.K = H;
    const bool x3 = g_precision == MMVAE_GEMM_PRECISION_BF16X3;
    g.aligned = aligned16(h) && aligned16(W) && (ldh % 4 == 0) && (ldw % 4 == 0);
    if (g.aligned && rows % 4 == 0 && G % 4 == 0 && H % 4 == 0) g.aligned = 2;
    g.x3_vec = 1;  // both operands K-contiguous: rows are clamped one by one, no edge groups
    // (the 2 x 4-wave kernel keeps this launch: its two workgroups per CU overlap one's heavy epilogue -- 160 KB of x
    // read, 160 KB of dP written per tile -- with the other's main loop, while the wave-specialised kernel's epilogue is
    // done by 4 of its 8 waves with the stagers idle: 1.097 against 1.067 ms per C2 step, also after its spills were
    // cut from 63 to 9 registers; MMVAE_X3W_RECON=1 to compare)
    const char* e_recon = getenv("MMVAE_X3W_RECON");
    const bool w_recon = e_recon && e_recon[0] == '1';
    const int tile_id = !x3 ? 1 : ((H % X3_BK == 0) ? (w_recon ? x3_tile_regular(rows, G, 1, false) : x3_tile_for(rows, G, false)) : 3);
    if (col_part && tile_shape(0, tile_id).bm != 128) return MMVAE_ERR_ARG;  // [mmvae_recon_row_tiles(rows)][G] partials
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
    return launch_gemm_forms<FORM_KC, FORM_KC, EPI_RECON>(tile_id, g, g.mt * g.nt, (hipStream_t)stream);
}

extern "C" int mmvae_decoder_recon_f32(int B, int G, int H, const float* h, int64_t ldh, const float* W, int64_t ldw,
                                       const float* bias, const float* x, int64_t ldx, float* xhat, int64_t ldxhat,
                                       float* dP, int64_t lddp, float* se_part, mmvae_stream_t stream) {
    return mmvae_decoder_recon_rows_f32(B, B, G, H, h, ldh, W, ldw, bias, x, ldx, xhat, ldxhat, dP, lddp, se_part,
                                        stream);
}
