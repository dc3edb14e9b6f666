// Device-side building blocks shared by the GEMM translation units (gemm_f32.hip, gemm_planes.hip): tile / LDS image
// helpers, the shared epilogues, the exact 3-way bf16 split and the wave-specialised bf16x3 kernel template.
// Everything lives in an anonymous namespace: each translation unit instantiates what it launches.
#pragma once
#include "common.h"
#include <type_traits>

namespace mmvae_detail {
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
    // pre-split operands (SRC_PLANES): three bf16 planes of the row-major fp32 matrix, plane p at Xp + p * x_pstride,
    // leading dimension ldxp (bf16 elements; 16-byte aligned rows)
    const unsigned short* Ap;
    const unsigned short* Bp;
    int64_t ldap, ldbp, a_pstride, b_pstride;
    // recon epilogue, optional: dP also (or only) as pre-split planes for the two GEMMs that consume it (N % 8 == 0)
    unsigned short* dPp;
    int64_t lddpp, dp_pstride;
};
// gemm_planes.hip: the instantiations of the wave-specialised kernel over pre-split operands
bool x3w_planes_combo(int layout, bool a_pl, bool b_pl);
int launch_x3w_planes(int layout, int tile_id, bool a_pl, bool b_pl, int epi, const GemmArgs& g, int nwork, int slots,
                      hipStream_t s);

}  // namespace mmvae_detail

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
#ifndef MMVAE_SLAB_STORE_NT
#define MMVAE_SLAB_STORE_NT 0  // 1: raw split-K slabs stored non-temporally (measured r3: 1.036 against 1.033 ms per C2 step -- no gain)
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

using mmvae_detail::GemmArgs;


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

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x3_pack4_lean(f32x2 lo, f32x2 hi, uint2 (&pk)[3]);  // (defined with the bf16x3 helpers)

// Epilogue shared by the fp32-MFMA and the bf16x3-MFMA kernels (the C/D register layout of the 32x32 MFMAs is
// dtype-independent): col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
// MF16 (standard epilogue only): a 32x32 block was accumulated as 2 x 2 sub-blocks by v_mfma_f32_16x16x32 -- element
// e = 4 (2 si + sj) + j of the block's 16 registers is row 16 si + 4 (lane >> 4) + j, col 16 sj + (lane & 15).  Either
// way a lane holds 4 consecutive rows of one column per register group, so the quad transposes below are common.
template <int BM, int BN, int WGM, int WGN, int EPI, int TM, int TN, bool MF16 = false>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[TM][TN], const GemmArgs& g, int bm, int bn, int z,
                                              float* lds) {
    static_assert(!MF16 || EPI == EPI_STD || EPI == EPI_RECON, "16x16 accumulator layout: standard and reconstruction epilogues");
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
                    // (MF16: the two 16-column halves of the block have a column group each)
                    int cols[MF16 ? 2 : 1];
                    f32x4 bvs[MF16 ? 2 : 1];
#pragma unroll
                    for (int h = 0; h < (MF16 ? 2 : 1); ++h) {
                        const int col = bn * BN + wn * WTN + n * 32 + (MF16 ? 16 * h + (lane & 12) : (l31 & ~3));
                        cols[h] = col;
                        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                        if (!raw && g.bias && col < g.N) {
                            if (col + 3 < g.N) {
                                bv = *reinterpret_cast<const f32x4*>(g.bias + col);
                            } else {
                                for (int j = 0; j < g.N - col; ++j) bv[j] = g.bias[col + j];
                            }
                        }
                        bvs[h] = bv;
                    }
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int col = cols[MF16 ? (gq & 1) : 0];
                        const f32x4 bv = bvs[MF16 ? (gq & 1) : 0];
                        const bool whole = col + 3 < g.N;  // N % 4 != 0: the group straddling the edge goes element-wise
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
                        const int row = bm * BM + wm * WTM + i * 32 +
                                        (MF16 ? 16 * (gq >> 1) + 4 * (lane >> 4) + q : 8 * gq + 4 * half + q);
                        if (row < g.M && col < g.N) {
                            float* cp = C + (int64_t)row * g.ldc + col;
                            v = v * alpha + bv;
                            if (whole) {
                                if (accum) v += *reinterpret_cast<const f32x4*>(cp);
                                if (relu) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                                }
#if MMVAE_SLAB_STORE_NT
                                // raw split-K slabs (16 x 2 MB for the G-wide K reductions) are read exactly once, by the
                                // next launch: stored non-temporally they do not sit dirty in L2 at the kernel boundary
                                if (raw)
                                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(cp));
                                else
#endif
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
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int col = bn * BN + wn * WTN + n * 32 + (MF16 ? 16 * ((e >> 2) & 1) + (lane & 15) : l31);
                    if (col >= g.N) continue;
                    const float bv = (!raw && g.bias) ? g.bias[col] : 0.f;
                    const int row = bm * BM + wm * WTM + i * 32 +
                                    (MF16 ? 16 * (e >> 3) + 4 * (lane >> 4) + (e & 3) : (e & 3) + 8 * (e >> 2) + 4 * half);
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
        if constexpr (MF16) {
            // 16x16x32 accumulators (16-byte path only: the launcher of the fused kernel sets c_vec).  Registers
            // 4 gq .. 4 gq + 3 of a 32x32 block are sub-block (si, sj) = (gq >> 1, gq & 1): rows 16 si + 4 (lane >> 4) + j,
            // column 16 sj + (lane & 15); after the quad transpose a lane holds row 16 si + 4 (lane >> 4) + q and the four
            // columns 16 sj + (lane & 12) .. + 3.  Column groups outermost (one bias vector and one column-sum vector
            // live at a time), the per-row squared errors of the wave's 2 TM row groups in 2 TM scalars.
            const int q = lane & 3, lg = lane >> 4, c12 = lane & 12;
            const bool odd = q & 1, hi = q & 2;
            float sr[TM][2];
#pragma unroll
            for (int i = 0; i < TM; ++i) sr[i][0] = sr[i][1] = 0.f;
            float* colbuf = lds + WGN * BM;  // [WGM][BN], behind the row sums
#pragma unroll
            for (int n = 0; n < TN; ++n) {
#pragma unroll
                for (int sj = 0; sj < 2; ++sj) {
                    const int cloc = wn * WTN + n * 32 + 16 * sj + c12;
                    const int col = bn * BN + cloc;
                    f32x4 bn4 = {0.f, 0.f, 0.f, 0.f};
                    if (g.bias && col < g.N) {
                        if (col + 3 < g.N) {
                            bn4 = *reinterpret_cast<const f32x4*>(g.bias + col);
                        } else {
                            for (int j = 0; j < g.N - col; ++j) bn4[j] = g.bias[col + j];
                        }
                    }
                    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
                    // the x rows of this column group's 2 TM row groups, requested together (clamped addresses: no
                    // branch around the loads) -- one load at a time cost a memory round trip per 16 x 16 sub-block, 40
                    // in a row per 256 x 160 tile (C3: 3.53 -> 3.455 ms; requested one column group AHEAD instead: more
                    // spills, 3.79 against 3.72)
                    f32x4 xpre[TM][2];
                    const bool col_whole = col + 3 < g.N;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int si = 0; si < 2; ++si) {
                            const int row = bm * BM + wm * WTM + i * 32 + 16 * si + 4 * lg + q;
                            const int xr = min(row, g.M - 1) % g.x_rows;
                            xpre[i][si] = *reinterpret_cast<const f32x4*>(g.x + (int64_t)xr * g.ldx + (col_whole ? col : 0));
                        }
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
#pragma unroll
                        for (int si = 0; si < 2; ++si) {
                            const int gq = 2 * si + sj;
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
                            const int row = bm * BM + wm * WTM + i * 32 + 16 * si + 4 * lg + q;
                            const int xr = row % g.x_rows;
                            if (row < g.M && col + 3 < g.N) {
                                p += bn4;
                                const f32x4 xv = xpre[i][si];
                                f32x4 xh, dp;
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    xh[j] = fmaxf(p[j], 0.f);
                                    const float d = xh[j] - xv[j];
                                    sr[i][si] += d * d;
                                    dp[j] = (p[j] > 0.f) ? 2.f * d : 0.f;
                                }
                                if (g.xhat) *reinterpret_cast<f32x4*>(g.xhat + (int64_t)row * g.ldxhat + col) = xh;
                                if (g.dP) *reinterpret_cast<f32x4*>(g.dP + (int64_t)row * g.lddp + col) = dp;
                                if (g.dPp) {
                                    uint2 pk[3];
                                    x3_pack4_lean(f32x2{dp[0], dp[1]}, f32x2{dp[2], dp[3]}, pk);
                                    unsigned short* pp = g.dPp + (int64_t)row * g.lddpp + col;
#pragma unroll
                                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(pp + pl * g.dp_pstride) = pk[pl];
                                }
                                csum += dp;
                            } else if (row < g.M && col < g.N) {  // G % 4 != 0: the group straddling the edge
                                p += bn4;
                                for (int j = 0; j < g.N - col; ++j) {
                                    const float xhj = fmaxf(p[j], 0.f);
                                    const float d = xhj - g.x[(int64_t)xr * g.ldx + col + j];
                                    sr[i][si] += d * d;
                                    const float dpj = (p[j] > 0.f) ? 2.f * d : 0.f;
                                    if (g.xhat) g.xhat[(int64_t)row * g.ldxhat + col + j] = xhj;
                                    if (g.dP) g.dP[(int64_t)row * g.lddp + col + j] = dpj;
                                    csum[j] += dpj;
                                }
                            }
                        }
                    }
                    if (g.col_part) {  // the 16 lanes that share this column group (q, lane >> 4) -> one, per wave
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = csum[j];
                            v += __shfl_xor(v, 1, 64);
                            v += __shfl_xor(v, 2, 64);
                            v += __shfl_xor(v, 16, 64);
                            v += __shfl_xor(v, 32, 64);
                            csum[j] = v;
                        }
                        if (q == 0 && lg == 0) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) colbuf[wm * BN + cloc + j] = csum[j];
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int si = 0; si < 2; ++si) {
                    float v = sr[i][si];  // the 4 lanes of a row (lane & 12) -> one
                    v += __shfl_xor(v, 4, 64);
                    v += __shfl_xor(v, 8, 64);
                    if (c12 == 0) rowsum[wn * BM + wm * WTM + i * 32 + 16 * si + 4 * lg + q] = v;
                }
        } else if (g.c_vec) {
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
                            if (g.dPp) {  // the same values, split once here for the dW / dX GEMMs that read them
                                uint2 pk[3];
                                x3_pack4_lean(f32x2{dp[0], dp[1]}, f32x2{dp[2], dp[3]}, pk);
                                unsigned short* pp = g.dPp + (int64_t)row * g.lddpp + col;
#pragma unroll
                                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(pp + pl * g.dp_pstride) = pk[pl];
                            }
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
            // col_part is [ceil(M / 128)][N]: a 256-row tile leaves one partial per 128-row half (its waves along M in order)
            constexpr int HALVES = BM >= 256 ? BM / 128 : 1;
            constexpr int WPH = WGM >= HALVES ? WGM / HALVES : 1;
            static_assert(HALVES == 1 || (BM % 128 == 0 && WGM % HALVES == 0), "col_part rows are 128-row partials");
            const float* colbuf = lds + WGN * BM;
            for (int c = tid; c < BN; c += WGM * WGN * 64) {
                const int col = bn * BN + c;
                if (col < g.N) {
#pragma unroll
                    for (int hh = 0; hh < HALVES; ++hh) {
                        float sum = 0.f;
#pragma unroll
                        for (int w = hh * WPH; w < (hh + 1) * WPH; ++w) sum += colbuf[w * BN + c];  // wave order: reproducible
                        const int prow = bm * HALVES + hh;
                        if (HALVES == 1 || prow * 128 < g.M) g.col_part[(int64_t)prow * g.N + col] = sum;
                    }
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

// (x3_split: common.h)

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
// chunk c of row r sits at chunk c ^ xw_swz(r), xw_swz(r) = -(r >> 2) & 3.  A ds_read_b128 is served in four groups of
// 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- each of which must touch 16 distinct
// 16-byte bank groups (row % 4 picks the 64-byte quarter, the chunk the 16 bytes inside it):
//   32x32x16 fragments (lane & 31 = row, lane >> 5 = chunk pair): a group covers row blocks r >> 2 = {0, 3, 5, 6} or
//       {1, 2, 4, 7} with one chunk: any bijection of (r >> 2) & 3 separates them;
//   16x16x32 fragments (lane & 15 = row, lane >> 4 = chunk): a group covers row blocks {0, 3} with chunk c and {1, 2}
//       with chunk c ^ 1: needs {s(0), s(3), s(1) ^ 1, s(2) ^ 1} distinct -- s = (0, 3, 2, 1), not the identity (r3's
//       c ^ (r >> 2) & 3 made rows 4-7 collide with rows 0-3 and 8-11 with 12-15: SQ_LDS_BANK_CONFLICT 5.8 M cycles per
//       launch of the NT kernel = a third of its LDS cycles, 0 now; the step did not change: profiles/r4_pmc_extra.csv).
// The stagers' ds_write_b64 cover whole rows (32 lanes = 4 rows): any chunk permutation is conflict-free for them.
// The last 4 KiB of LDS are the epilogue's scratch.
constexpr int XW_ROWB = 64;                   // bytes per row per plane
constexpr int XW_SCRATCH = 4096;              // epilogue scratch behind the two images
__device__ __forceinline__ int xw_swz(int row) { return -(row >> 2) & 3; }
__device__ __forceinline__ int xw_off(int row, int chunk) { return row * XW_ROWB + ((chunk ^ xw_swz(row)) << 4); }

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
    // `between(u)` runs behind the split of unit u and AHEAD of the loads that follow it (mixed kernels issue their
    // LDS-DMA pieces there); TRAIL = the loads issued behind the last between() call.
    static constexpr int TRAIL = (FORM == FORM_KC) ? 1 : ((NU % 4 == 0) ? 4 : 1);
    template <int SET, class F>
    __device__ __forceinline__ void stage_and_reload(char* S, int st, const char* __restrict__ next, F&& between) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            stage_unit<SET>(u, S, st);
            between(u);
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
    template <bool NF>
    __device__ __forceinline__ void open(const GemmArgs& g) {  // item w -> tile, slice, k-tile range
        if (w >= g.nwork) return;
        const int tiles = g.mt * g.nt;
        z = w / tiles;
        const int t = w - z * tiles;
        // Consecutive items (one XCD's share of a round, running together over one L2) cycle through the tiles of the
        // axis with FEWER tiles: the panels they cycle through stay in that L2, and the other operand's panel is fetched
        // once for all of them.  (dW = dP^T h: 125 x 4 tiles -- M fastest walked all 125 dP panels per h panel and
        // re-read dP four times: 179 MB fetched against 43 MB of operands.)  NF is the tile's shape, not the launch's tile
        // counts: the tall-output tile (160 x 256) is the one planned for many row tiles x few column tiles, and a
        // run-time choice cost the register-bound planes kernels their spill-free loops (3 x slower).
        if (NF) {
            bn = t % g.nt;
            bm = t / g.nt;
        } else {
            bm = t % g.mt;
            bn = t / g.mt;
        }
        kt = z * g.ktiles_per_split;
        kt_end = min(kt + g.ktiles_per_split, g.ktiles);
    }
    template <bool NF>
    __device__ __forceinline__ bool advance(const GemmArgs& g, int nwg) {  // next k-tile; true when a new item began
        if (++kt < kt_end) return false;
        w += nwg;
        open<NF>(g);
        return true;
    }
};

// ---- pre-split operands (r3).  What bounded the kernel in r2 was the split itself: 19.1 M VALU instructions beside 3.84 M
// MFMAs per launch on SIMDs that issue one or the other, every operand element split once per tile that reads it (10 x).
// An operand whose producer has already written its three bf16 planes to HBM (SRC_PLANES; same row-major orientation as
// the fp32 matrix) is moved global -> LDS by LDS-DMA: global_load_lds_dwordx4, 64 lanes x 16 B = one 1-KiB piece of LDS
// per wave-instruction, no VGPR destination, no VALU.  The LDS destination of a piece is linear (M0 + 16 * lane), so the
// image's swizzle is applied to the per-lane SOURCE address and again by the reader.
//   K-contiguous planes (FORM_KC): the multipliers' usual [plane][row][64 B] image, chunk c of row r at c ^ xw_swz(r);
//       a piece = 16 rows x 64 B.
//   rows-contiguous planes (FORM_RC): the image is [plane][32 k][R rows] bf16 -- the operand's own orientation -- and the
//       multipliers read their fragments with ds_read_b64_tr_b16 (two per 8-k fragment): no transposing pass anywhere.
//       The four k-rows of a transposed read must fall on different 64-byte bank slots: rows of 512 B / 256 B (R = 256 /
//       128) XOR the slot index with k & 3; rows of 320 B (R = 160) already do.
enum { SRC_F32 = 0, SRC_PLANES = 1 };
#ifndef MMVAE_MFMA16
#define MMVAE_MFMA16 1  // multipliers of the standard-epilogue kernels: v_mfma_f32_16x16x32_bf16 (0: 32x32x16)
#endif
#ifndef MMVAE_XW_INTERLEAVE
#define MMVAE_XW_INTERLEAVE 1  // multipliers: one fragment read pinned behind each MFMA (0: compiler-scheduled reads)
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// LDS[lds_addr + 16 * lane .. + 15] <- 16 bytes at base + off (base, lds_addr wave-uniform).  Hidden from the compiler's
// vmcnt bookkeeping (it would drain the memory pipeline at every use of an ordinary load while a DMA is in flight): the
// stager waits for its pieces itself before the k-tile's barrier.  M0 is written in the statement that reads it.
__device__ __forceinline__ void xw_glds16(const char* base, unsigned off, unsigned lds_addr) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);  // (wave-uniform by construction; the "s" operands need it provable)
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(base);
    base = reinterpret_cast<const char*>(
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64));
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(off), "s"(base), "s"(lds_addr)
        : "memory");
}

// first byte of k-tile kt of a planes operand (plane 0, row / column 0)
template <int FORM>
__device__ __forceinline__ const char* xwp_base(const unsigned short* P, int64_t ld, int kt) {
    return reinterpret_cast<const char*>(P) + (FORM == FORM_KC ? (int64_t)kt * (X3_BK * 2) : (int64_t)kt * X3_BK * ld * 2);
}

template <int FORM, int R>
struct XwPlanes {
    static constexpr int PIECES = 3 * R * XW_ROWB / 1024;  // 1-KiB pieces of one k-tile (3 planes)
    static constexpr int NP = (PIECES + 3) / 4;            // per stager wave; a wave without a last piece repeats its previous one
    static_assert(R % 16 == 0 && (R * XW_ROWB) % 1024 == 0, "whole pieces per plane");
    unsigned off[NP];  // per-lane source byte offsets relative to xwp_base (constant over an item's k-loop)

    __device__ __forceinline__ static int piece(int wv, int j) {
        const int pi = wv + 4 * j;
        return (PIECES % 4 != 0 && pi >= PIECES) ? pi - 4 : pi;
    }
    // r0: first row (KC) / column (RC) of the tile along the non-K axis, Rtot its extent.  Rows / columns beyond the
    // matrix are fetched from in-matrix addresses: they only reach accumulators the epilogue never stores.
    __device__ __forceinline__ void offsets(int64_t ld, int64_t pstride, int r0, int Rtot, int lane, int wv) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int pi = piece(wv, j);
            if (FORM == FORM_KC) {
                constexpr int PPP = R / 16;  // pieces per plane
                const int plane = pi / PPP, row = 16 * (pi % PPP) + (lane >> 2);
                const int chunk = (lane & 3) ^ xw_swz(row);
                const int rc = min(r0 + row, Rtot - 1);
                off[j] = (unsigned)((plane * pstride + (int64_t)rc * ld) * 2 + chunk * 16);
            } else {
                const int pb = pi * 1024 + 16 * lane;
                const int plane = pb / (R * XW_ROWB), pin = pb % (R * XW_ROWB);
                const int k = pin / (2 * R);
                int b = pin % (2 * R);
                if (R != 160) b = (((b >> 6) ^ (k & 3)) << 6) | (b & 63);
                int col = r0 + (b >> 1);
                // (an extent off a multiple of 8: the planes' leading dimension is rounded up to 8 and the columns between
                // hold zeros -- mmvae_split_planes_f32 -- so the last 16-byte group is fetched like any other)
                if (col + 8 > ((Rtot + 7) & ~7)) col = r0;
                off[j] = (unsigned)((plane * pstride + (int64_t)k * ld + col) * 2);
            }
        }
    }
    __device__ __forceinline__ void issue(const char* base, unsigned lds_opnd, int wv) const {
#pragma unroll
        for (int j = 0; j < NP; ++j) xw_glds16(base, off[j], lds_opnd + piece(wv, j) * 1024);
    }
    // pieces [j0, j1) of this wave (compile-time bounds after unrolling)
    __device__ __forceinline__ void issue_range(const char* base, unsigned lds_opnd, int wv, int j0, int j1) const {
#pragma unroll
        for (int j = 0; j < NP; ++j)
            if (j >= j0 && j < j1) xw_glds16(base, off[j], lds_opnd + piece(wv, j) * 1024);
    }
};

template <int AFORM, int BFORM, int BM, int BN, int WGM, int WGN, int EPI, int ASRC = SRC_F32, int BSRC = SRC_F32>
__global__ __launch_bounds__(512, 2) void gemm_x3w_kernel(const GemmArgs g) {
    static_assert(WGM * WGN == 4, "4 multiplier wavefronts");
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr bool KEEP_A = TM <= TN;  // fragments of the short side stay in registers over a k-step, the long side streams
    constexpr int TK = KEEP_A ? TM : TN, TL = KEEP_A ? TN : TM;
    constexpr int PA = BM * XW_ROWB, PB = BN * XW_ROWB;       // plane strides
    constexpr int IMG = 3 * (PA + PB);                        // one image (A planes, then B planes)
    constexpr bool A_PL = ASRC == SRC_PLANES, B_PL = BSRC == SRC_PLANES;
    constexpr bool NFAST = BM < BN;  // walk order of the work items (XwCursor::open)
    constexpr bool ANY_F32 = !A_PL || !B_PL, ANY_PL = A_PL || B_PL;
    constexpr bool A_TR = A_PL && AFORM == FORM_RC, B_TR = B_PL && BFORM == FORM_RC;  // transposed-read images
    // 16x16x32 MFMAs: the same flops with a quarter of the accumulator registers per instruction -- the chip is
    // power-limited under this kernel and holds a ~10 % higher rate with them (profiles/r3_mfma_shape.txt, measured in
    // this kernel: forward 103 -> 91 us).  The fused reconstruction epilogue keeps the 32x32 layout.
    // Not for TN.  With fp32 operands that kernel is bound by its stagers (in-kernel split + transposing store) and the
    // short MFMAs take twice the issue slots of the SIMD the stager wave shares (160x256: 127 -> 137 us).  With planes
    // operands the 16-row fragments come out of the DMA'd k-major image by transposed reads whose four 16-lane groups
    // address the same 16 operand rows at k-rows 8 apart -- the same banks: 108 -> 128 us; trading the two 32-byte
    // halves of a slot in every other k-octet to separate them made it 320 us (profiles/r3_mfma16.txt).
    constexpr bool IS_TN = AFORM == FORM_RC && BFORM == FORM_RC;
    constexpr bool MF16 = MMVAE_MFMA16 && MMVAE_XW_INTERLEAVE && !IS_TN;  // (both epilogues know the 16x16 layout)
    static_assert(!MF16 || TK >= 2, "the in-place reload of the kept fragments needs two kept blocks");
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
        using OA = std::conditional_t<A_PL, XwPlanes<AFORM, BM>, XwOperand<AFORM, BM>>;
        using OB = std::conditional_t<B_PL, XwPlanes<BFORM, BN>, XwOperand<BFORM, BN>>;
        OA oa;
        OB ob;
        XwCursor ld_c;  // fp32 operands: the k-tile the next load fetches (two ahead of the one being staged)
        XwCursor dm_c;  // planes operands: the k-tile the next DMA fetches (= the one being staged)
        XwCursor br_c;  // the k-tile whose barrier comes next (= the one the multipliers work on)
        ld_c.w = dm_c.w = br_c.w = L;
        ld_c.template open<NFAST>(g);
        dm_c.template open<NFAST>(g);
        br_c.template open<NFAST>(g);
        if (!br_c.valid(g)) return;
        const int dlane = st & 63;
        const int wv = __builtin_amdgcn_readfirstlane(st >> 6);
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
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
        int off_w = -1;  // item the fp32 operands' offsets were computed for
        int dma_w = -1;  // item the planes operands' offsets were computed for
        auto refresh_offsets = [&]() {
            if (ld_c.w != off_w) {
                if constexpr (!A_PL) oa.offsets(g.lda, ld_c.bm * BM, g.M, st);
                if constexpr (!B_PL) ob.offsets(g.ldb, ld_c.bn * BN, g.N, st);
                off_w = ld_c.w;
            }
        };
        auto bump = [&](XwCursor& c) {
            XwCursor nx = c;
            nx.template advance<NFAST>(g, nwg);
            if (nx.valid(g)) c = nx;  // (a select per field, not a branch around the loads)
        };
        auto issue_load = [&](auto SET) {  // prologue only
            refresh_offsets();
            if constexpr (!A_PL) oa.template load<decltype(SET)::value>(x3p_base<AFORM>(g.A, g.lda, ld_c.bm * BM, ld_c.kt));
            if constexpr (!B_PL) ob.template load<decltype(SET)::value>(x3p_base<BFORM>(g.B, g.ldb, ld_c.bn * BN, ld_c.kt));
            bump(ld_c);
        };
        auto stage = [&](auto SET, int img) {  // prologue only
            constexpr int S_ = decltype(SET)::value;
            char* As = lds + img * IMG;
            char* Bs = As + 3 * PA;
            if constexpr (!A_PL) {
#pragma unroll
                for (int u = 0; u < XwOperand<AFORM, BM>::NU; ++u) oa.template stage_unit<S_>(u, As, st);
            }
            if constexpr (!B_PL) {
#pragma unroll
                for (int u = 0; u < XwOperand<BFORM, BN>::NU; ++u) ob.template stage_unit<S_>(u, Bs, st);
            }
        };
        // planes operands of the k-tile at dm_c -> image `img` (LDS-DMA; the caller waits before the barrier)
        auto dma = [&](int img) {
            if (dm_c.w != dma_w) {
                if constexpr (A_PL) oa.offsets(g.ldap, g.a_pstride, dm_c.bm * BM, g.M, dlane, wv);
                if constexpr (B_PL) ob.offsets(g.ldbp, g.b_pstride, dm_c.bn * BN, g.N, dlane, wv);
                dma_w = dm_c.w;
            }
            const unsigned ia = lds0 + img * IMG;
            if constexpr (A_PL) oa.issue(xwp_base<AFORM>(g.Ap, g.ldap, dm_c.kt), ia, wv);
            if constexpr (B_PL) ob.issue(xwp_base<BFORM>(g.Bp, g.ldbp, dm_c.kt), ia + 3 * PA, wv);
            bump(dm_c);
        };
        // steady state: stage raw set SET into image `img`, re-issuing its registers as the loads of the k-tile at ld_c
        auto stage_reload = [&](auto SET, int img) {
            constexpr int S_ = decltype(SET)::value;
            char* As = lds + img * IMG;
            char* Bs = As + 3 * PA;
#if MMVAE_X3_STAMPS
            if constexpr (!ANY_PL) {  // diagnostic: how long does the oldest raw set still take to arrive?  (its NU_A + NU_B loads are the oldest
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
            auto nothing = [](int) {};
            if constexpr (!A_PL) oa.template stage_and_reload<S_>(As, st, x3p_base<AFORM>(g.A, g.lda, ld_c.bm * BM, ld_c.kt), nothing);
            if constexpr (!B_PL) ob.template stage_and_reload<S_>(Bs, st, x3p_base<BFORM>(g.B, g.ldb, ld_c.bn * BN, ld_c.kt), nothing);
            bump(ld_c);
        };
        // mixed kernels (one operand pre-split, the other fp32): the DMA pieces are dealt out behind the split of the fp32
        // operand's units -- a piece
        // costs ~125 cycles of VMEM issue while every CU streams (stamps of r3: 12 pieces back to back held the wave for
        // 1 450 cycles before its split work even began) -- and ahead of the re-loads that follow each unit, so that
        // only TRAIL compiler-visible loads are younger than the last piece: the wait before the barrier is counted.
        auto step_mixed = [&](auto SET, int img) {
            constexpr int S_ = decltype(SET)::value;
            if constexpr (A_PL != B_PL) {
                // F = the fp32 operand (split here), P = the pre-split one (DMA)
                using XF = std::conditional_t<A_PL, XwOperand<BFORM, BN>, XwOperand<AFORM, BM>>;
                constexpr int NPP = std::conditional_t<A_PL, XwPlanes<AFORM, BM>, XwPlanes<BFORM, BN>>::NP;
                constexpr int PPU = (NPP + XF::NU - 1) / XF::NU;  // pieces per unit of the fp32 operand
                if (dm_c.w != dma_w) {
                    if constexpr (A_PL) oa.offsets(g.ldap, g.a_pstride, dm_c.bm * BM, g.M, dlane, wv);
                    if constexpr (B_PL) ob.offsets(g.ldbp, g.b_pstride, dm_c.bn * BN, g.N, dlane, wv);
                    dma_w = dm_c.w;
                }
                const unsigned ip = lds0 + img * IMG + (A_PL ? 0 : 3 * PA);
                refresh_offsets();
                if constexpr (A_PL) {
                    const char* pbase = xwp_base<AFORM>(g.Ap, g.ldap, dm_c.kt);
                    ob.template stage_and_reload<S_>(lds + img * IMG + 3 * PA, st,
                                                     x3p_base<BFORM>(g.B, g.ldb, ld_c.bn * BN, ld_c.kt),
                                                     [&](int u) { oa.issue_range(pbase, ip, wv, u * PPU, (u + 1) * PPU); });
                } else {
                    const char* pbase = xwp_base<BFORM>(g.Bp, g.ldbp, dm_c.kt);
                    oa.template stage_and_reload<S_>(lds + img * IMG, st,
                                                     x3p_base<AFORM>(g.A, g.lda, ld_c.bm * BM, ld_c.kt),
                                                     [&](int u) { ob.issue_range(pbase, ip, wv, u * PPU, (u + 1) * PPU); });
                }
                bump(ld_c);
                bump(dm_c);
                if (XF::TRAIL == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // every piece has landed
                if (XF::TRAIL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
        };
        // one k-tile of the stream into image `img`
        auto step = [&](auto SET, int img) {
            if constexpr (ANY_PL && ANY_F32) {
                step_mixed(SET, img);
            } else if constexpr (ANY_PL) {
                dma(img);
                XW_SSTAMP(1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every piece has landed
            } else {
                stage_reload(SET, img);
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        if constexpr (ANY_F32) {
            issue_load(S0{});  // element 0 -> set 0
            issue_load(S1{});  // element 1 -> set 1
        }
        if constexpr (ANY_PL) dma(0);  // element 0 -> image 0
        if constexpr (ANY_F32) {
            stage(S0{}, 0);
            issue_load(S0{});  // element 2 -> set 0
        }
        if constexpr (ANY_PL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // image 0 holds element 0
        // per element s of the stream (the one the multipliers work on): stage element s + 1 (it arrived in raw set
        // (s + 1) & 1) into image (s + 1) & 1, re-issue that set as the load of element s + 3, then barrier #s.  Past the
        // end of the stream the same code stages and loads data nobody reads.
        while (true) {
            step(S1{}, 1);
            XW_SSTAMP(0)
            __syncthreads();
            XW_SSTAMP(2)
#if MMVAE_X3_STAMPS
            sst[3] += 1;
#endif
            if (br_c.template advance<NFAST>(g, nwg))
                for (int e = 0; e < epi_barriers; ++e) __syncthreads();
            if (!br_c.valid(g)) break;
            step(S0{}, 0);
            XW_SSTAMP(0)
            __syncthreads();
            XW_SSTAMP(2)
#if MMVAE_X3_STAMPS
            sst[3] += 1;
#endif
            if (br_c.template advance<NFAST>(g, nwg))
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
    const int swz = xw_swz(l31);
    // byte offset of this lane's fragment inside a plane for k-step ks: row l31 of a 32-row block, chunk 2 ks + half
    const int coff[2] = {l31 * XW_ROWB + (((0 + half) ^ swz) << 4), l31 * XW_ROWB + (((2 + half) ^ swz) << 4)};
    const int a_row0 = wm * WTM * XW_ROWB, b_row0 = wn * WTN * XW_ROWB;
    // transposed-read images: lane 4 q + p of a 16-lane group addresses k-row q, elements 4 p .. 4 p + 3 of the group's
    // 16 operand rows (lanes 0-15 / 16-31: rows 0-15 / 16-31 of the 32-row block; lanes 32-63: k + 8)
    const int tq = (lane & 15) >> 2;
    auto tr_lane_off = [&](int R, int bg) {  // first transposed read of block bg (32 operand rows from 32 bg), k-step 0
        const int slot = (R == 160) ? bg : (bg ^ tq);
        return (8 * half + tq) * (2 * R) + slot * 64 + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
    };
    int troA[A_TR ? TM : 1], troB[B_TR ? TN : 1];
    if constexpr (A_TR) {
#pragma unroll
        for (int i = 0; i < TM; ++i) troA[i] = tr_lane_off(BM, wm * TM + i);
    }
    if constexpr (B_TR) {
#pragma unroll
        for (int n = 0; n < TN; ++n) troB[n] = tr_lane_off(BN, wn * TN + n);
    }
    auto tr_frag = [&](const char* a, int dk) {  // k = 8 half + 0..3 from a, + 4..7 from a + dk
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + dk));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    auto frag = [&](const char* plane_base, int block, int ks) {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(plane_base + block * (32 * XW_ROWB) + coff[ks]));
    };
    auto fragA = [&](const char* img, int p, int block, int ks) {
        if constexpr (A_TR)
            return tr_frag(img + p * PA + ks * (16 * 2 * BM) + troA[block], 4 * 2 * BM);
        else
            return frag(img + a_row0 + p * PA, block, ks);
    };
    auto fragB = [&](const char* img, int p, int block, int ks) {
        if constexpr (B_TR)
            return tr_frag(img + 3 * PA + p * PB + ks * (16 * 2 * BN) + troB[block], 4 * 2 * BN);
        else
            return frag(img + 3 * PA + b_row0 + p * PB, block, ks);
    };
    // 16x16x32 fragments: row l15 of a 16-row sub-block, k = 8 grp .. 8 grp + 7 = chunk grp of the row's 64 bytes; the
    // transposed reads address k-row 8 grp + q, the sub-block's half of the 64-byte slot
    const int l15 = lane & 15, grp = lane >> 4;
    const int coff16 = l15 * XW_ROWB + ((grp ^ xw_swz(l15)) << 4);
    auto tr_lane_off16 = [&](int R, int bg) {  // first transposed read of sub-block 0 of block bg
        const int slot = (R == 160) ? bg : (bg ^ tq);
        return (8 * grp + tq) * (2 * R) + slot * 64 + 8 * (lane & 3);
    };
    constexpr int tr_odd = 32;  // from sub-block 0 to sub-block 1 of a block
    int tro16A[(MF16 && A_TR) ? TM : 1], tro16B[(MF16 && B_TR) ? TN : 1];
    if constexpr (MF16 && A_TR) {
#pragma unroll
        for (int i = 0; i < TM; ++i) tro16A[i] = tr_lane_off16(BM, wm * TM + i);
    }
    if constexpr (MF16 && B_TR) {
#pragma unroll
        for (int n = 0; n < TN; ++n) tro16B[n] = tr_lane_off16(BN, wn * TN + n);
    }
    auto frag16 = [&](const char* plane_base, int sb) {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(plane_base + sb * (16 * XW_ROWB) + coff16));
    };
    auto frag16A = [&](const char* img, int p, int sb) {  // sub-block sb (16 rows) of this wave's A rows, plane p
        if constexpr (A_TR)
            return tr_frag(img + p * PA + tro16A[sb >> 1] + ((sb & 1) ? tr_odd : 0), 4 * 2 * BM);
        else
            return frag16(img + a_row0 + p * PA, sb);
    };
    auto frag16B = [&](const char* img, int p, int sb) {
        if constexpr (B_TR)
            return tr_frag(img + 3 * PA + p * PB + tro16B[sb >> 1] + ((sb & 1) ? tr_odd : 0), 4 * 2 * BN);
        else
            return frag16(img + 3 * PA + b_row0 + p * PB, sb);
    };
    auto frag16K = [&](const char* img, int p, int sb) {  // kept operand
        if constexpr (KEEP_A)
            return frag16A(img, p, sb);
        else
            return frag16B(img, p, sb);
    };
    auto frag16S = [&](const char* img, int p, int sb) {  // streamed operand
        if constexpr (KEEP_A)
            return frag16B(img, p, sb);
        else
            return frag16A(img, p, sb);
    };
    // kept side: all TK blocks x 3 planes of a k-step; streamed side: one block x 3 planes, double buffered
    // (MF16: fk[0] / fk[1] hold the even / odd 16-row sub-blocks of the kept blocks for the whole k-tile)
    bf16x8 fk[2][3][TK], fs[2][3];
    auto load_kept = [&](int set, const char* img, int ks) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int k = 0; k < TK; ++k) {
                if constexpr (KEEP_A)
                    fk[set][p][k] = fragA(img, p, k, ks);
                else
                    fk[set][p][k] = fragB(img, p, k, ks);
            }
    };
    auto load_stream = [&](int set, const char* img, int l, int ks) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if constexpr (KEEP_A)
                fs[set][p] = fragB(img, p, l, ks);
            else
                fs[set][p] = fragA(img, p, l, ks);
        }
    };

    XwCursor c;
    c.w = L;
    c.template open<NFAST>(g);
    if (!c.valid(g)) return;
    int s = 0;
    // the multipliers' MFMAs and fragment reads win every issue arbitration against the stager wave of their SIMD: the
    // matrix pipe sets the pace, the stagers fill the slots it leaves
#ifndef MMVAE_XW_PRIO
#define MMVAE_XW_PRIO 0
#endif
    if (MMVAE_XW_PRIO) __builtin_amdgcn_s_setprio(MMVAE_XW_PRIO);
    __syncthreads();  // image 0 holds element 0
    if constexpr (MF16) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int sb = 0; sb < 2 * TK; ++sb) fk[sb & 1][p][sb >> 1] = frag16K(lds, p, sb);
            fs[0][p] = frag16S(lds, p, 0);
        }
    } else {
        load_kept(0, lds, 0);
        load_stream(0, lds, 0, 0);
    }
#if MMVAE_X3_STAMPS
    long long mst[4] = {0, 0, 0, 0};  // before the barrier, in the barrier, behind it, k-tiles
    long long tq_ = clock64();
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
#if MMVAE_XW_INTERLEAVE
            // Fragment reads placed by hand: ONE fragment (a ds_read_b128, or the two transposed reads of a rows-contiguous
            // planes operand) behind each MFMA, in the issue slots the MFMA leaves free, the order pinned by
            // sched_barrier -- left to itself the compiler gathers the reads of several blocks into bursts of ~20 and
            // the MFMAs behind a burst wait for it (37.8 cycles per MFMA, stamps of r3).  Same reads, same MFMAs, same
            // accumulation order per accumulator: identical results.
            if constexpr (MF16) {
                // One k-step of 32 per k-tile; blocks of 16 streamed rows x all kept blocks, 2 x 6 MFMAs of 16 cycles per
                // kept block.  Reads behind the MFMAs: the next block's streamed fragments behind the first three; the
                // kept fragments are reloaded in place -- block kb - 1 behind the first MFMAs of block kb in the last
                // streamed block (behind the k-tile's barrier, from the next image), the last kept block at the start
                // of the next k-tile's first streamed block.
#pragma unroll
                for (int l16 = 0; l16 < 2 * TL; ++l16) {
                    const int cur = l16 & 1;
                    const bool last = l16 == 2 * TL - 1;
                    const char* s_img = last ? nxt : img;
                    const int s_sb = last ? 0 : l16 + 1;
                    if (last) {
#if MMVAE_X3_STAMPS
                        {
                            const long long tn_ = clock64();
                            mst[0] += tn_ - tq_;
                            tq_ = tn_;
                        }
#endif
                        __syncthreads();
#if MMVAE_X3_STAMPS
                        {
                            const long long tn_ = clock64();
                            mst[1] += tn_ - tq_;
                            tq_ = tn_;
                            mst[3] += 1;
                        }
#endif
                    }
                    X3_SB();
#pragma unroll
                    for (int kb = 0; kb < TK; ++kb) {
                        // the two 16-row sub-blocks of kept block kb against this streamed sub-block: two independent
                        // accumulator chains, their MFMAs alternating
                        const int lb = l16 >> 1, lsub = l16 & 1;
                        f32x16& blk = KEEP_A ? acc[kb][lb] : acc[lb][kb];
                        const int gq0 = KEEP_A ? lsub : 2 * lsub, gq1 = KEEP_A ? 2 + lsub : 2 * lsub + 1;
                        f32x4 c0 = {blk[4 * gq0], blk[4 * gq0 + 1], blk[4 * gq0 + 2], blk[4 * gq0 + 3]};
                        f32x4 c1 = {blk[4 * gq1], blk[4 * gq1 + 1], blk[4 * gq1 + 2], blk[4 * gq1 + 3]};
#pragma unroll
                        for (int m = 0; m < 6; ++m) {
                            const int pk = (m == 0) ? 2 : (m == 1 || m == 3) ? 1 : 0;               // plane of A
                            const int ps = (m == 0 || m == 3 || m == 5) ? 0 : (m == 1 || m == 4) ? 1 : 2;  // plane of B
#pragma unroll
                            for (int ksub = 0; ksub < 2; ++ksub) {
                                const bf16x8 am = KEEP_A ? fk[ksub][pk][kb] : fs[cur][pk];
                                const bf16x8 bq = KEEP_A ? fs[cur][ps] : fk[ksub][ps][kb];
                                if (ksub == 0)  // smallest terms first
                                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bq, c0, 0, 0, 0);
                                else
                                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bq, c1, 0, 0, 0);
                                const int u = 2 * m + ksub;  // read unit behind this MFMA
                                if (kb == 0) {
                                    if (u < 3) {
                                        fs[cur ^ 1][u] = frag16S(s_img, u, s_sb);
                                    } else if (l16 == 0 && u - 3 < 6) {
                                        // the last kept block's fragments of THIS k-tile (its image is the current one now)
                                        const int sb = 2 * (TK - 1) + (u - 3) / 3;
                                        fk[sb & 1][(u - 3) % 3][TK - 1] = frag16K(img, (u - 3) % 3, sb);
                                    }
                                } else if (last && u < 6) {  // kept block kb - 1 is done with: the next k-tile's
                                    const int sb = 2 * (kb - 1) + u / 3;
                                    fk[sb & 1][u % 3][kb - 1] = frag16K(nxt, u % 3, sb);
                                }
                                X3_SB();
                            }
                        }
                        blk[4 * gq0] = c0[0], blk[4 * gq0 + 1] = c0[1], blk[4 * gq0 + 2] = c0[2], blk[4 * gq0 + 3] = c0[3];
                        blk[4 * gq1] = c1[0], blk[4 * gq1 + 1] = c1[1], blk[4 * gq1 + 2] = c1[2], blk[4 * gq1 + 3] = c1[3];
                    }
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int l = 0; l < TL; ++l) {
                    const int cur = (ks * TL + l) & 1;  // fs set of this block
                    const bool last = ks == 1 && l == TL - 1;
                    // the block's reads: units 0..2 = the next block's streamed fragments (planes 0..2); the first block
                    // of a k-tile also reads the kept fragments of k-step 1, the last one (behind the k-tile's barrier)
                    // those of the next k-tile's k-step 0
                    const bool kept_here = last || (ks == 0 && l == 0);
                    const char* s_img = last ? nxt : img;
                    const int s_blk = last ? 0 : (l + 1 < TL ? l + 1 : 0);
                    const int s_ks = last ? 0 : (l + 1 < TL ? ks : 1);
                    if (last) {
#if MMVAE_X3_STAMPS
                        {
                            const long long tn_ = clock64();
                            mst[0] += tn_ - tq_;
                            tq_ = tn_;
                        }
#endif
                        __syncthreads();
#if MMVAE_X3_STAMPS
                        {
                            const long long tn_ = clock64();
                            mst[1] += tn_ - tq_;
                            tq_ = tn_;
                            mst[3] += 1;
                        }
#endif
                    }
                    X3_SB();
#pragma unroll
                    for (int k = 0; k < TK; ++k) {
                        f32x16 cc = KEEP_A ? acc[k][l] : acc[l][k];
                        const bf16x8 a0 = KEEP_A ? fk[ks][0][k] : fs[cur][0], a1 = KEEP_A ? fk[ks][1][k] : fs[cur][1],
                                     a2 = KEEP_A ? fk[ks][2][k] : fs[cur][2];
                        const bf16x8 b0 = KEEP_A ? fs[cur][0] : fk[ks][0][k], b1 = KEEP_A ? fs[cur][1] : fk[ks][1][k],
                                     b2 = KEEP_A ? fs[cur][2] : fk[ks][2][k];
#pragma unroll
                        for (int m = 0; m < 6; ++m) {
                            const bf16x8 am = (m == 0) ? a2 : (m == 1 || m == 3) ? a1 : a0;
                            const bf16x8 bq = (m == 0 || m == 3 || m == 5) ? b0 : (m == 1 || m == 4) ? b1 : b2;
                            cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bq, cc, 0, 0, 0);  // smallest terms first
                            const int u = k * 6 + m;  // read unit issued behind this MFMA
                            if (u < 3) {
                                if constexpr (KEEP_A)
                                    fs[cur ^ 1][u] = fragB(s_img, u, s_blk, s_ks);
                                else
                                    fs[cur ^ 1][u] = fragA(s_img, u, s_blk, s_ks);
                            } else if (kept_here && u - 3 < 3 * TK) {
                                const int p = (u - 3) / TK, kk = (u - 3) % TK;
                                const char* k_img = last ? nxt : img;
                                const int k_ks = last ? 0 : 1;
                                if constexpr (KEEP_A)
                                    fk[k_ks][p][kk] = fragA(k_img, p, kk, k_ks);
                                else
                                    fk[k_ks][p][kk] = fragB(k_img, p, kk, k_ks);
                            }
                            X3_SB();
                        }
                        if (KEEP_A)
                            acc[k][l] = cc;
                        else
                            acc[l][k] = cc;
                    }
                }
            }
            }  // (32x32x16 path)
#else
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
                            mst[0] += tn_ - tq_;
                            tq_ = tn_;
                        }
#endif
                        __syncthreads();
#if MMVAE_X3_STAMPS
                        {
                            const long long tn_ = clock64();
                            mst[1] += tn_ - tq_;
                            tq_ = tn_;
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
#endif
            ++s;
            more = !c.template advance<NFAST>(g, nwg);
#if MMVAE_X3_STAMPS
            {
                const long long tn_ = clock64();
                mst[2] += tn_ - tq_;
                tq_ = tn_;
            }
#endif
        }
#if MMVAE_X3_STAMPS
        if (tid == 0 && bid < 4096) g_x3_trace[bid * 4 + 2] = wall_clock64();
#endif
        gemm_epilogue<BM, BN, WGM, WGN, EPI, TM, TN, MF16>(acc, g, bm, bn, z, scratch);
        if (EPI == EPI_RECON) __syncthreads();
#if MMVAE_X3_STAMPS
        tq_ = clock64();
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

}  // namespace
