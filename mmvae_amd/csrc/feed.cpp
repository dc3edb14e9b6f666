// libmmvae_feed.so: host-side row gather of the data feed (include/mmvae_feed.h).  No HIP: plain C++17, g++.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

#include "../../include/mmvae_feed.h"

namespace {

template <typename I>
int64_t row_len(const I* indptr, int64_t r) {
    return (int64_t)indptr[r + 1] - (int64_t)indptr[r];
}

template <typename I, typename O>
void gather_range(const I* indptr, const I* indices, const float* data, const int64_t* rows, int64_t lo, int64_t hi,
                  const O* crow, O* out_col, float* out_val) {
    for (int64_t i = lo; i < hi; ++i) {
        const int64_t s = (int64_t)indptr[rows[i]], n = (int64_t)indptr[rows[i] + 1] - s;
        O* c = out_col + (int64_t)crow[i];
        const I* src = indices + s;
        for (int64_t k = 0; k < n; ++k) c[k] = (O)src[k];  // converting copy (vectorised by the compiler)
        std::memcpy(out_val + (int64_t)crow[i], data + s, (size_t)n * sizeof(float));
    }
}

template <typename I, typename O>
int gather(const I* indptr, const I* indices, const float* data, int64_t n_chunk_rows, const int64_t* rows,
           int64_t n_rows, O* out_crow, O* out_col, float* out_val, int64_t capacity, int n_threads,
           int64_t* out_nnz) {
    out_crow[0] = 0;
    int64_t total = 0;
    for (int64_t i = 0; i < n_rows; ++i) {
        if (rows[i] < 0 || rows[i] >= n_chunk_rows) return MMVAE_FEED_ERR_ARG;
        const int64_t n = row_len(indptr, rows[i]);
        if (n < 0) return MMVAE_FEED_ERR_ARG;
        total += n;
        if (sizeof(O) == 4 && total > 0x7fffffffLL) return MMVAE_FEED_ERR_CAPACITY;  // does not fit int32 row pointers
        out_crow[i + 1] = (O)total;
    }
    const int64_t nnz = total;
    if (out_nnz) *out_nnz = nnz;
    if (nnz > capacity) return MMVAE_FEED_ERR_CAPACITY;
    if (n_threads <= 1 || n_rows < 2 * n_threads) {
        gather_range(indptr, indices, data, rows, 0, n_rows, out_crow, out_col, out_val);
        return MMVAE_FEED_OK;
    }
    std::vector<std::thread> pool;
    const int64_t per = (n_rows + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        const int64_t lo = t * per, hi = lo + per < n_rows ? lo + per : n_rows;
        if (lo >= hi) break;
        pool.emplace_back(gather_range<I, O>, indptr, indices, data, rows, lo, hi, out_crow, out_col, out_val);
    }
    for (auto& th : pool) th.join();
    return MMVAE_FEED_OK;
}

}  // namespace

extern "C" int mmvae_feed_abi_version(void) { return 3; }

extern "C" int64_t mmvae_feed_rows_nnz(const void* indptr, int index_bytes, int64_t n_chunk_rows, const int64_t* rows,
                                       int64_t n_rows) {
    if (!indptr || !rows || n_rows < 0 || (index_bytes != 4 && index_bytes != 8)) return -1;
    int64_t total = 0;
    for (int64_t i = 0; i < n_rows; ++i) {
        if (rows[i] < 0 || rows[i] >= n_chunk_rows) return -1;
        total += index_bytes == 4 ? row_len(static_cast<const int32_t*>(indptr), rows[i])
                                  : row_len(static_cast<const int64_t*>(indptr), rows[i]);
    }
    return total;
}

extern "C" int mmvae_feed_gather_rows(const void* indptr, const void* indices, int index_bytes, const float* data,
                                      int64_t n_chunk_rows, const int64_t* rows, int64_t n_rows, int64_t* out_crow,
                                      int64_t* out_col, float* out_val, int64_t capacity, int n_threads,
                                      int64_t* out_nnz) {
    if (!indptr || !rows || !out_crow || n_rows <= 0 || n_chunk_rows <= 0 || capacity < 0) return MMVAE_FEED_ERR_ARG;
    if (capacity > 0 && (!indices || !data || !out_col || !out_val)) return MMVAE_FEED_ERR_ARG;
    if (index_bytes == 4)
        return gather(static_cast<const int32_t*>(indptr), static_cast<const int32_t*>(indices), data, n_chunk_rows,
                      rows, n_rows, out_crow, out_col, out_val, capacity, n_threads, out_nnz);
    if (index_bytes == 8)
        return gather(static_cast<const int64_t*>(indptr), static_cast<const int64_t*>(indices), data, n_chunk_rows,
                      rows, n_rows, out_crow, out_col, out_val, capacity, n_threads, out_nnz);
    return MMVAE_FEED_ERR_ARG;
}

extern "C" int mmvae_feed_gather_rows_i32(const void* indptr, const void* indices, int index_bytes, const float* data,
                                          int64_t n_chunk_rows, const int64_t* rows, int64_t n_rows, int32_t* out_crow,
                                          int32_t* out_col, float* out_val, int64_t capacity, int n_threads,
                                          int64_t* out_nnz) {
    if (!indptr || !rows || !out_crow || n_rows <= 0 || n_chunk_rows <= 0 || capacity < 0) return MMVAE_FEED_ERR_ARG;
    if (capacity > 0 && (!indices || !data || !out_col || !out_val)) return MMVAE_FEED_ERR_ARG;
    if (index_bytes == 4)
        return gather(static_cast<const int32_t*>(indptr), static_cast<const int32_t*>(indices), data, n_chunk_rows,
                      rows, n_rows, out_crow, out_col, out_val, capacity, n_threads, out_nnz);
    if (index_bytes == 8)  // (column indices of such a chunk must fit int32: the caller checks the gene count)
        return gather(static_cast<const int64_t*>(indptr), static_cast<const int64_t*>(indices), data, n_chunk_rows,
                      rows, n_rows, out_crow, out_col, out_val, capacity, n_threads, out_nnz);
    return MMVAE_FEED_ERR_ARG;
}

// ---- index tables of the conditional layers (mmvae_amd/cond_tables.py states the layout; csrc/cond_layers.hip reads it)
namespace {
constexpr int COND_CHUNK = 32;  // MMVAE_COND_DW_CHUNK of include/mmvae_hip.h
}

extern "C" int mmvae_feed_cond_tables(int n_pos, int R, const int32_t* local, const int32_t* base, int32_t* seg,
                                      int64_t seg_stride, int32_t* present, int32_t* n_present) {
    if (n_pos < 0 || R <= 0 || !local || !base || !seg || !present || !n_present) return MMVAE_FEED_ERR_ARG;
    const int nc = R + R / COND_CHUNK + 1, nr = R / (COND_CHUNK + 1) + 1;
    if (seg_stride < 2 * (int64_t)R + 3 * (int64_t)nc + 3 * (int64_t)nr) return MMVAE_FEED_ERR_ARG;
    std::vector<int32_t> order((size_t)R), counts;
    for (int j = 0; j < n_pos; ++j) {
        const int32_t* loc = local + (int64_t)j * R;
        int32_t* s = seg + (int64_t)j * seg_stride;
        int32_t *cond = s, *rows = s + R, *cdst = s + 2 * R, *cbeg = cdst + nc, *cend = cbeg + nc, *rcond = cend + nc,
                *rslot = rcond + nr, *rn = rslot + nr;
        for (int i = 0; i < R; ++i) {
            if (loc[i] < 0) return MMVAE_FEED_ERR_ARG;
            cond[i] = loc[i] + base[j];
        }
        // stable counting sort of the cells by block (banks hold up to a few thousand blocks; falls back to a comparison
        // sort for sparse huge indices)
        int32_t top = 0;
        for (int i = 0; i < R; ++i) top = std::max(top, loc[i]);
        if ((int64_t)top <= 64 * (int64_t)R + 4096) {
            counts.assign((size_t)top + 2, 0);
            for (int i = 0; i < R; ++i) ++counts[(size_t)loc[i] + 1];
            for (int32_t b = 0; b <= top; ++b) counts[(size_t)b + 1] += counts[(size_t)b];
            for (int i = 0; i < R; ++i) order[(size_t)counts[(size_t)loc[i]]++] = i;
        } else {
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [loc](int32_t a, int32_t b) { return loc[a] < loc[b]; });
        }
        std::memcpy(rows, order.data(), (size_t)R * sizeof(int32_t));
        int32_t* pres = present + (int64_t)j * R;
        int n_chunks = 0, n_red = 0, n_blocks = 0, slots = 0;
        for (int start = 0; start < R;) {
            const int32_t blk = loc[order[start]];
            int end = start + 1;
            while (end < R && loc[order[end]] == blk) ++end;
            const int count = end - start, pieces = (count + COND_CHUNK - 1) / COND_CHUNK;
            pres[n_blocks++] = blk;
            if (pieces > 1) {
                if (n_red >= nr) return MMVAE_FEED_ERR_CAPACITY;
                rcond[n_red] = blk + base[j];
                rslot[n_red] = slots;
                rn[n_red] = pieces;
                ++n_red;
            }
            for (int k = 0; k < pieces; ++k) {
                if (n_chunks >= nc) return MMVAE_FEED_ERR_CAPACITY;
                cdst[n_chunks] = pieces > 1 ? -2 - (slots + k) : blk + base[j];
                cbeg[n_chunks] = start + k * COND_CHUNK;
                cend[n_chunks] = std::min(start + (k + 1) * COND_CHUNK, end);
                ++n_chunks;
            }
            if (pieces > 1) slots += pieces;
            start = end;
        }
        n_present[j] = n_blocks;
        for (int c = n_chunks; c < nc; ++c) {
            cdst[c] = -1;
            cbeg[c] = 0;
            cend[c] = 0;
        }
        for (int r = n_red; r < nr; ++r) {
            rcond[r] = -1;
            rslot[r] = 0;
            rn[r] = 0;
        }
    }
    return MMVAE_FEED_OK;
}
