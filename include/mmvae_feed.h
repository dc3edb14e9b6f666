/*
 * mmvae_feed.h -- C-ABI of libmmvae_feed.so: host-side helpers of the data feed.
 *
 * The reference builds every batch in Python: a row permutation of the chunk's scipy CSR matrix
 * (data/local/cellxgene_datapipe.py:110-122) followed by a row slice and torch.sparse_csr_tensor(...)
 * (:169-193).  This library does the one memory-bound step of that chain natively and without the interpreter
 * lock: gather B rows of a CSR chunk (given by index, i.e. the permutation is never materialised) into three
 * caller-owned staging arrays in the layout torch.sparse_csr uses (int64 or int32 row pointers and column indices, fp32
 * values) -- typically page-locked buffers that are then copied to the GPU asynchronously.
 *
 * Plain C, host pointers only, no allocation, no global state; thread-safe (distinct output buffers per call).
 */
#ifndef MMVAE_FEED_H
#define MMVAE_FEED_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMVAE_FEED_OK 0
#define MMVAE_FEED_ERR_ARG 1
#define MMVAE_FEED_ERR_CAPACITY 2 /* the gathered rows hold more stored elements than `capacity` */

int mmvae_feed_abi_version(void);

/* Stored elements of the selected rows: sum over rows[i] of (indptr[r + 1] - indptr[r]).
 * index_bytes: 4 or 8 = width of the chunk's indptr / indices arrays (scipy uses int32 while they fit). */
int64_t mmvae_feed_rows_nnz(const void* indptr, int index_bytes, int64_t n_chunk_rows, const int64_t* rows,
                            int64_t n_rows);

/* Gather rows `rows[0..n_rows)` of a CSR chunk (indptr [n_chunk_rows + 1], indices / data [nnz]) into
 *   out_crow [n_rows + 1] int64 (starts at 0), out_col [capacity] int64, out_val [capacity] fp32.
 * n_threads <= 1: single-threaded; > 1: the rows are split over that many threads (disjoint output ranges).
 * Returns MMVAE_FEED_OK and the element count in *out_nnz. */
int mmvae_feed_gather_rows(const void* indptr, const void* indices, int index_bytes, const float* data,
                           int64_t n_chunk_rows, const int64_t* rows, int64_t n_rows, int64_t* out_crow,
                           int64_t* out_col, float* out_val, int64_t capacity, int n_threads, int64_t* out_nnz);

/* Same with int32 row pointers / column indices in the output -- the index type of the reference's batches
 * (torch.sparse_csr_tensor keeps the int32 arrays of the scipy slice, cellxgene_datapipe.py:178-183) and a third less
 * data to copy to the GPU.  MMVAE_FEED_ERR_CAPACITY when the batch holds more than 2^31 - 1 stored elements. */
int mmvae_feed_gather_rows_i32(const void* indptr, const void* indices, int index_bytes, const float* data,
                               int64_t n_chunk_rows, const int64_t* rows, int64_t n_rows, int32_t* out_crow,
                               int32_t* out_col, float* out_val, int64_t capacity, int n_threads, int64_t* out_nnz);

/* Index tables of the conditional layers for one step (reference ConditionalLayer.forward, components.py:
 * 365-413, groups the cells of a batch by condition with Python masks): from the per-cell block index of each of n_pos
 * layer applications -- local [n_pos][R], >= 0 -- the padded table set the kernels of csrc/cond_layers.hip read
 * (mmvae_amd/cond_tables.py documents it and holds the numpy statement of the same function: group_tables +
 * fill_padded, tested equal).  Position j writes seg + j * seg_stride, in this order:
 *   cond [R] (block + base[j]), rows [R] (cells sorted by block, stable), chunk_dst / chunk_beg / chunk_end [nc],
 *   red_cond / red_slot / red_n [nr],   nc = R + R / 32 + 1, nr = R / 33 + 1   (unused slots: dst / cond -1, else 0);
 * a block's cells are cut into pieces of at most 32; a block of several pieces gets scratch slots (dst = -2 - slot) and
 * one reduction entry.  present [n_pos][R]: the local indices of the blocks that took part, ascending, n_present[j] of
 * them.  seg_stride >= 2 R + 3 nc + 3 nr words. */
int mmvae_feed_cond_tables(int n_pos, int R, const int32_t* local, const int32_t* base, int32_t* seg, int64_t seg_stride,
                           int32_t* present, int32_t* n_present);

#ifdef __cplusplus
}
#endif
#endif /* MMVAE_FEED_H */
