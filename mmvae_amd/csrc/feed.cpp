// libmmvae_feed.so: host-side row gather of the data feed (include/mmvae_feed.h).  No HIP: plain C++17, g++.
#include <cstring>
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

extern "C" int mmvae_feed_abi_version(void) { return 2; }

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
