/*
 * spla_hip.h — the kernel-level C ABI of the MI355X (gfx950) backend: the "thin C-ABI dispatch" a host-side
 * RegistryAlgo calls once its operands are device-resident.
 *
 * Every entry point replaces the body of one reference accelerator algorithm `execute()` (the OpenCL ones,
 * which are NOT ported; the semantics are those of the CPU algorithms, the parity oracle):
 *
 *   splahip_mxv_masked   <- Algo_mxv_masked_cl<T>::execute   reference src/opencl/cl_mxv.hpp:65-74
 *                           semantics: Algo_mxv_masked_cpu   reference src/cpu/cpu_mxv.hpp:56-106
 *   splahip_vxm_masked   <- Algo_vxm_masked_cl<T>::execute   reference src/opencl/cl_vxm.hpp:60-173
 *                           semantics: Algo_vxm_masked_cpu   reference src/cpu/cpu_vxm.hpp:58-128
 *   splahip_mxm          <- (no accelerated version exists)  reference src/opencl/cl_algo_registry.cpp:82-95
 *                           semantics: Algo_mxm_cpu          reference src/cpu/cpu_mxm.hpp:58-117
 *   splahip_mxmT_masked  <- Algo_mxmT_masked_cl<T>::execute  reference src/opencl/cl_mxmT_masked.hpp:60-127
 *                           semantics: Algo_mxmT_masked_cpu  reference src/cpu/cpu_mxmT_masked.hpp:56-128
 *   splahip_csr_plan_*   <- cl_csr_init (device CSR upload)  reference src/opencl/cl_format_csr.hpp:40-59
 *   splahip_malloc/free/memcpy_* <- cl::Buffer create/read   reference src/opencl/cl_format_dense_vec.hpp:40-88
 *
 * Conventions
 *   - plain pointers and sizes only; all `d_` pointers are DEVICE pointers (hipMalloc'ed by anyone: this
 *     library, PyTorch-ROCm, ...); elements are 4-byte words typed by `splahip_type`; indices are uint32
 *     (reference include/spla/config.hpp:56); scalars (init) travel as their uint32 bit pattern.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls enqueue work and return;
 *     the only calls that block are the ones documented as returning a host-side count.
 *   - every function returns 0 on success or a non-zero splahip_status; splahip_last_error() describes it.
 *   - there is NO CPU fallback behind this ABI: without a usable gfx950 device every call fails loudly.
 *   - concurrency: a plan carries per-call workspaces of its layout copies (bitmaps, partial slabs, statistics): ONE call at a time per plan —
 *     calls on the same plan from several streams or threads must be serialised by the caller (different plans are independent).
 *     splahip_malloc / splahip_free are ordered on the DEFAULT stream: a buffer that kernels on another (non-blocking) stream still use must
 *     not be freed before that stream has been synchronised.
 *   - memory: idle device memory the library caches (returned mxm results, mxm workspaces, the stream-ordered pool) is handed back to the driver
 *     when one of its allocations reports out-of-memory, and that allocation is retried once.
 */
#ifndef SPLA_HIP_H
#define SPLA_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "spla_hip_ops.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum splahip_status {
    SPLAHIP_OK               = 0,
    SPLAHIP_ERROR            = 1, /* HIP runtime error, see splahip_last_error()              */
    SPLAHIP_NO_DEVICE        = 2, /* no usable gfx950 device                                   */
    SPLAHIP_INVALID_ARGUMENT = 6, /* bad pointer / size / id (same code as spla Status)        */
    SPLAHIP_NOT_IMPLEMENTED  = 1024
} splahip_status;

/* ---- runtime -------------------------------------------------------------------------------------------- */
int         splahip_device_count(void);
int         splahip_init(int device);             /* select device, check it is usable; idempotent */
int         splahip_device_info(char* buffer, int length);
const char* splahip_last_error(void);
int         splahip_version(void);
/* SPLAHIP_* environment knobs (A/B switches for tests and experiments) are read once and cached; a process that changes one afterwards calls this. */
int         splahip_env_refresh(void);
/* Deterministic fp32 sums for the calls the CALLING HOST THREAD makes from now on (on != 0) / back to the default (0); returns the previous setting (not a
 * status).  Default: an fp32 PLUS fold over the column-ordered blocks (mxv / vxm) or the LDS row classes (mxm) merges partial sums in arrival order — every
 * result within the stated 1e-5 of the reference's sequential fold, but the last bit may differ from run to run.  Deterministic: those calls run on the
 * fixed-order kernels (CSR wave-tile / row kernels, the sort form of mxm): two runs are bit-identical, at the CSR kernels' speed.  Integer and MIN / MAX
 * folds are exact in every form and unaffected.  The operator API sets it per call from Descriptor::set_deterministic (include/spla.hpp). */
int         splahip_set_deterministic(int on);
/* Measurement helper (bench.py's roofline.measured_read_stream_gbs): the rate of a read-only stream of `bytes` bytes read the way this library's kernels read their
 * streams — 16 bytes per lane, non-temporal, one 1024-thread workgroup per CU on a contiguous span, four loads in flight per wavefront — best of `reps` launches,
 * in GB/s.  Allocates and frees its own buffer; blocks until done. */
int         splahip_read_stream_gbs(size_t bytes, int reps, double* gbs, void* stream);
/* Reference fold order for the calls the CALLING HOST THREAD makes from now on (on != 0) / back to the default (0); returns the previous setting (not a
 * status).  An fp32 PLUS fold of splahip_mxv_masked / splahip_vxm_pull then adds every row's products strictly left to right from `init` in entry order
 * — the loop of the reference CPU backend, src/cpu/cpu_mxv.hpp:94-99, src/cpu/cpu_vxm.hpp:101-113 — so the result equals the reference's BIT FOR BIT
 * (short rows: one lane per row out of LDS; rows beyond 508 entries: one wavefront per row, a chain of dependent additions — slow on purpose: a few
 * times the default call).  Implies the deterministic setting for every other operator.  Every other fold (integers, MIN / MAX, ...) is exact in every
 * form and unaffected.  The operator API sets it per call from Descriptor::set_reference_order (include/spla.hpp). */
int         splahip_set_reference_order(int on);

int splahip_malloc(void** d_ptr, size_t bytes);
int splahip_free(void* d_ptr);
int splahip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream); /* async on stream      */
int splahip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream); /* blocks until landed  */
int splahip_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream); /* async on stream      */
int splahip_memset32(void* d_dst, uint32_t word, size_t count, void* stream);       /* fill with a 4-byte word */
int splahip_stream_sync(void* stream);

/* ---- device CSR launch plan ---------------------------------------------------------------------------------
 * A plan is the work decomposition of one device-resident CSR matrix (it is part of the "AccCsr" decoration:
 * built once per matrix on the device from d_Ap, reused by every call on it).  The nnz stream is cut into work
 * items of <= ~1 Ki entries, one per wavefront: runs of consecutive short rows, and segments of long rows
 * (DESIGN.md "wave tiles").                                                                                   */
typedef struct splahip_plan_s* splahip_plan;

int splahip_csr_plan_create(splahip_plan* plan, uint32_t n_rows, uint32_t nnz, const uint32_t* d_Ap, void* stream);
/* Optional: a sliced-ELL copy of the matrix with hot-column encoding, kept with the plan (+8 bytes per entry; DESIGN.md 4.1,
 * spla_amd/csrc/kernels/mxv_sell.hpp).  With it splahip_mxv_masked serves every associative op_add (no early exit) from the
 * copy: rows are cut into chunks of <= 256 entries, chunks sorted by length and laid out 64 to a slice, one lane per chunk,
 * the 40,000 most referenced columns of v served from LDS.  The copy holds the VALUES too: rebuild it when Aj or Ax change.
 * Blocks.  sell_info: slices, stored entries incl. padding, LDS-resident columns, rows longer than one chunk.            */
int splahip_csr_plan_attach_sell(splahip_plan plan, uint32_t n_cols, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, void* stream);
int splahip_csr_plan_sell_info(splahip_plan plan, uint32_t* n_slices, uint32_t* n_slots, uint32_t* n_hot, uint32_t* n_long_rows);
/* sell_values: *uniform = 1 when every stored value of the matrix is the same 4-byte word (*bits), as in a graph's adjacency matrix of
 * ones; the sparse-vector (bit) form then streams the 4-byte column words only — a structure-only pass, 4 B per entry instead of 8.   */
int splahip_csr_plan_sell_values(splahip_plan plan, int* uniform, uint32_t* bits);
/* Optional: a "column-ordered blocks" copy of the matrix, kept with the plan (+8 bytes per entry + 2 bytes per row; DESIGN.md 4.9,
 * spla_amd/csrc/kernels/mxv_cob.hpp).  Rows are grouped into ranges of at most R rows whose result slice lives in LDS — the rows of highest
 * degree, sorted by degree, form "hub" ranges; the other rows that hold an entry fill "natural" ranges in row order — and a range's entries are
 * stored sorted by column, so the gathers of v ascend and one 128-byte request serves several products; products enter the slice with the LDS
 * atomic of op_add.  With it splahip_mxv_masked serves every call without early exit whose op_add has such an atomic (PLUS; MIN / MAX / BOR /
 * BAND / BXOR on int and uint; LOR / LAND; fp32 MIN / MAX over op_mult = PLUS when the matrix holds no -0.0f): int / uint results and fp32 MIN /
 * MAX stay bit-identical to the reference; fp32 PLUS folds in column-block order, in double, rounded once per row and piece (1e-5 relative) and
 * is NOT bit-reproducible from run to run (the arrival order of the atomics differs) — the CSR kernels are: SPLAHIP_COB=0 keeps every call on them.
 * row_bits: 0 = the library's R (20224 slots: a double slice of that many rows fills a CU's LDS), else R = 2^row_bits, 8..14; tile_chunks: 0 = a
 * range dearer than one workgroup slot's share of the launch's memory requests is cut into pieces of three size levels that meet in a shared
 * accumulator (memory-side atomics; a natural range's last piece to arrive finishes its rows, the hub ranges' rows are finished by a small second
 * launch); > 0 = pieces of at most that many 256-entry chunks (tests: many pieces on small inputs).  One 16-wave workgroup per CU.  The copy holds
 * the VALUES: rebuild it when Aj or Ax change.  A hypersparse matrix (more (range, column window) groups than nnz / 512) is left without a copy.
 * One call at a time per plan: the shared accumulators, tickets and the per-call bitmap of v belong to the plan.
 * Blocks.  cob_info: workgroups, chunks, ranges cut into several pieces, slots per range (all 0 when no copy was built);
 * cob_layout: ranges, hub ranges, hub rows, device bytes held by the copy.                                                          */
int splahip_csr_plan_attach_cob(splahip_plan plan, uint32_t n_cols, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, uint32_t row_bits,
                                uint32_t tile_chunks, void* stream);
int splahip_csr_plan_cob_info(splahip_plan plan, uint32_t* n_tiles, uint32_t* n_chunks, uint32_t* n_multi, uint32_t* slots_per_range);
int splahip_csr_plan_cob_layout(splahip_plan plan, uint32_t* n_ranges, uint32_t* n_hub_ranges, uint32_t* n_hub_rows, uint64_t* bytes);
/* slots (rows) of a hub range and of a natural range of the copy (0 / 0 without one) */
int splahip_csr_plan_cob_slots(splahip_plan plan, uint32_t* hub_slots, uint32_t* natural_slots);
/* *hot_columns = K when the copy relabels the K most referenced columns (a matrix whose column labels carry no locality: the hot columns take the ids 0 .. K-1
 * inside the copy and every call gathers from [v[hot columns] | v], one extra pass over v per call); 0 when the labels already cluster them or no copy exists. */
int splahip_csr_plan_cob_relabel(splahip_plan plan, uint32_t* hot_columns);
/* What the last MASKED call over the blocks (op_select other than ALWAYS) found when it probed its mask on the device — a blocking read-back for tests and
 * benches: *selected_entries ~ entries of the selected rows (a 1-in-16 sample of the rows, scaled; exact below 2^18 rows), *served_by_rows = 1 when the
 * selected-rows kernels served the call (the selection holds too few entries for the blocks, which stream the whole matrix whatever the mask).  Both zero when
 * no probe has run.  Reference behaviour served: src/cpu/cpu_mxv.hpp:89-92 (unselected rows cost nothing). */
int splahip_csr_plan_cob_probe(splahip_plan plan, uint64_t* selected_entries, int* served_by_rows);
/* footprint: device bytes held for the matrix besides its CSR arrays — plan (work items + call workspace), sliced-ELL copy, column-ordered blocks (0: not built).
 * prefers_sell: 1 when splahip_mxv_masked runs this semiring on the sliced-ELL copy even beside the column-ordered blocks (the zero-family sparse-vector
 * forms, a BFS step): a host that builds copies on demand needs the +12 B per entry of the sliced-ELL copy only for these.                          */
int splahip_csr_plan_footprint(splahip_plan plan, uint64_t* plan_bytes, uint64_t* sell_bytes, uint64_t* cob_bytes);
int splahip_mxv_prefers_sell(int type, int op_mult, int op_add);
int splahip_csr_plan_destroy(splahip_plan plan);
/* n_items: work items; n_long: rows cut into segment items */
/* What the OWNER of a plan's value array says about it: every stored value is the word value_bits (is_uniform != 0) or not.  A plan is built from Ap alone and
 * does not see the values; the BFS-like fast paths (splahip_csr_plan_first_hit below, and the pull form of vxm) need to know that every product of a frontier
 * is one word.  The operator API calls it once per upload (splahip_words_range over the value array); attached layout copies know it from their own pass. */
int splahip_csr_plan_set_uniform_values(splahip_plan plan, int is_uniform, uint32_t value_bits);
/* smallest / largest 4-byte word of a device array (blocking): equal <=> one word */
int splahip_words_range(const void* d_words, size_t n, uint32_t* lo, uint32_t* hi, void* stream);
/* The products of the NEXT splahip_vxm_pull over this (transposed) plan, as counted by the splahip_vxm_probe / splahip_vxm_push_small its caller ran (one
 * shot).  With it, a traversal's pull level — BAND / BOR ..., a frontier of equal stored values, a matrix of one value word — whose frontier rows hold >= 1/6 of
 * the entries is settled output entry by output entry at the first stored in-neighbour (mxv_first_hit.hpp) instead of folding every row completely. */
int splahip_csr_plan_hint_products(splahip_plan plan, uint64_t products);
/* BFS-like calls (BAND / MULT / LAND products folded by BOR or uint MAX) over a matrix whose stored values are all one word (known from an attached layout
 * copy): splahip_mxv_masked passes over v once per call (bitmap; is every non-zero entry the same word?; a 2048-row sample of how far a first-hit walk would
 * go) and, when the LATEST FINISHED pass saw a frontier whose walk mostly ends within a row's first four entries, settles every row by its first hit over plain
 * CSR — the traversal step of reference src/algorithm.cpp:97-99 whatever early_exit says; results are the reference's for any vector and mask (a stale hint
 * only costs time).  A call with early_exit != 0 takes the kernel without a hint (a vector of several words is then folded with the reference's early-exit rule).  Reports what the plan's last such call did: *taken = 1 if it ran on the first-hit kernel; *uniform = the last finished pass' verdict
 * (-1: none yet); *nonzeros = its non-zero count; *tail_pct = share (%) of the sampled selected entries in rows the walk would enter past their fourth entry. */
int splahip_csr_plan_first_hit(splahip_plan plan, int* taken, int* uniform, uint32_t* nonzeros, uint32_t* tail_pct);
int splahip_csr_plan_info(splahip_plan plan, uint32_t* n_items, uint32_t* n_long);

/* ---- masked SpMV (pull): r[i] = select(mask[i]) ? fold_k add(., mult(Ax[k], v[Aj[k]])) from init : init -----
 * r, mask: n_rows words; v: n_cols words; all n_rows outputs are written (reference cpu_mxv.hpp:88-103).
 * early_exit != 0: a row's fold stops at the first partial result != init (reference cpu_mxv.hpp:98).
 * The matrix may be a row block of a larger matrix (1-D row decomposition): Aj holds GLOBAL column ids and
 * v is the full gathered vector.                                                                             */
int splahip_mxv_masked(splahip_plan plan, int type, uint32_t n_rows, uint32_t n_cols,
                       const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax,
                       const void* d_mask, const void* d_v, void* d_r,
                       int op_mult, int op_add, int op_select, uint32_t init_bits, int early_exit,
                       void* stream);

/* ---- user-defined ops: run-time compilation (spla_amd/csrc/kernels/jit.hip) ---------------------------------------------------------------
 * The same masked SpMV with ops given as SOURCE TEXT — the body a reference user passes to OpBinary::make_* / OpSelect::make_*, e.g.
 * "(float a, float b) { return 0.25f * a + 0.75f * b; }" (reference src/op.cpp:294-317, tests/test_op.cpp:56-59) — compiled with hiprtc, cached by
 * text.  *_src == NULL means the built-in op with the given id.  One lane per row, strict left fold with the reference's early-exit rule: exact for
 * any op.  Returns 8 (the reference's Status::CompilationError) with the compiler's log in splahip_last_error() when the text does not compile. */
int splahip_jit_available(void);
int splahip_mxv_masked_jit(int type, uint32_t n_rows, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, const void* d_mask, const void* d_v, void* d_r,
                           int op_mult, const char* mult_src, int op_add, const char* add_src, int op_select, const char* select_src, uint32_t init_bits,
                           int early_exit, void* stream);
/* the other three operators with user-defined ops: each the reference's own fold order, one run-time-compiled kernel inside the operator's usual driver
 * (spla_amd/csrc/kernels/jit.hip, jit_hooks.hpp).  *_src as above.  vxm: plan_T / d_T* = the plan and CSR of the TRANSPOSED matrix, n_in = rows of M (length of
 * v), n_out = columns of M; sorted = the indices of v ascend (splahip_vxm_probe); output as splahip_vxm_pull.  mxmT / mxm: arguments and hand-off of the
 * result arrays as splahip_mxmT_masked / splahip_mxm.  Blocking.                                                                                           */
int splahip_vxm_masked_jit(splahip_plan plan_T, int type, uint32_t n_in, uint32_t n_out, const uint32_t* d_Tp, const uint32_t* d_Tj, const void* d_Tx, const void* d_mask,
                           uint32_t nvals, const uint32_t* d_vi, const void* d_vx, int sorted, uint32_t* d_ri, void* d_rx, uint32_t* h_count, int op_mult, const char* mult_src,
                           int op_add, const char* add_src, int op_select, const char* select_src, void* stream);
int splahip_mxmT_masked_jit(int type, uint32_t n_rows, uint32_t mask_nnz, const uint32_t* d_Mp, const uint32_t* d_Mj, const void* d_Mx, const uint32_t* d_Ap,
                            const uint32_t* d_Aj, const void* d_Ax, const uint32_t* d_Bp, const uint32_t* d_Bj, const void* d_Bx, uint32_t* d_Rp, uint32_t** d_Rj, void** d_Rx,
                            uint32_t* h_nnz, int op_mult, const char* mult_src, int op_add, const char* add_src, int op_select, const char* select_src, uint32_t init_bits,
                            void* stream);
int splahip_mxm_jit(int type, uint32_t a_rows, uint32_t b_cols, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, const uint32_t* d_Bp, const uint32_t* d_Bj,
                    const void* d_Bx, uint32_t* d_Rp, uint32_t** d_Rj, void** d_Rx, uint32_t* h_nnz, int op_mult, const char* mult_src, int op_add, const char* add_src,
                    uint32_t init_bits, void* stream);

/* ---- 1-D row-block decomposition across the GPUs of one node (spla_amd/csrc/kernels/dist.hip) -------------------------------
 * One process per GPU.  Rank p owns a contiguous block of rows (its CSR slice with the column ids remapped to the PADDED layout
 * owner * max_rows + offset, the matching slices of mask and r) and slice p of the input vector; one exchange per product: the all-gather of v.
 *   splahip_comm : an RCCL communicator (RCCL is loaded at run time; splahip_dist_available() == 0 when it cannot be).  Rank 0 makes the 128-byte
 *                  id with splahip_dist_get_unique_id and hands it to the other ranks by any means (the host's launcher, torch.distributed, MPI).
 *   all_gather   : every rank contributes `words` 4-byte words, d_recv receives world x words in rank order; direct != 0 replaces ncclAllGather
 *                  by a grouped send / receive pair per peer (every slice over its own xGMI link instead of a ring).
 *   splahip_shard: the row block cut by column owner into a local part (needs only the rank's own slice of v) and a remote part (needs the
 *                  gathered vector), each with its plan and column-ordered-blocks copy; the caller's CSR arrays must outlive the shard.
 *   splahip_dist_mxv_masked: exchange on the communicator's stream || local part on `stream`, then the remote part folds onto r:
 *                  r[i] = select(mask[i]) ? add(add(init, local fold), remote fold) : init  (reference cpu_mxv.hpp:88-103 for an associative and
 *                  commutative op_add: bit-identical for int / uint, rounding-only for fp32 PLUS).  Calls the two-part form cannot serve (no LDS
 *                  atomic for op_add, early exit) run the unsplit block after the exchange (*overlapped = 0).  d_v_local: max_rows words (the
 *                  slice, padded); d_v_full: world x max_rows words of workspace.  The phases are also exported one by one (mxv_local / mxv_remote /
 *                  mxv_whole) for hosts that bring their own exchange.                                                                        */
typedef struct splahip_comm_s*  splahip_comm;
typedef struct splahip_shard_s* splahip_shard;
int splahip_dist_available(void);
int splahip_dist_get_unique_id(void* id128);
int splahip_dist_comm_create(splahip_comm* comm, int world, int rank, const void* id128);
int splahip_dist_comm_destroy(splahip_comm comm);
int splahip_dist_comm_info(splahip_comm comm, int* world, int* rank);
int splahip_dist_all_gather(splahip_comm comm, const void* d_send, void* d_recv, size_t words, int direct, void* stream);
int splahip_shard_create(splahip_shard* shard, int world, int rank, uint32_t n_local, uint32_t max_rows, uint32_t nnz, const uint32_t* d_Ap,
                         const uint32_t* d_Aj_padded, const void* d_Ax, void* stream);
int splahip_shard_destroy(splahip_shard shard);
int splahip_shard_info(splahip_shard shard, uint32_t* nnz_local_cols, uint32_t* nnz_remote_cols, int* split_ok);
/* 1: the two-part (overlapped) form serves this call.  splahip_shard_mxv_remote additionally REQUIRES d_r and d_mask to be 16-byte aligned (its row pass moves them
 * in 16-byte loads) and returns SPLAHIP_INVALID_ARGUMENT before touching r otherwise; splahip_dist_mxv_masked checks and runs the unsplit block for such a call. */
int splahip_shard_split_serves(splahip_shard shard, int type, int op_add, int early_exit);
/* 1: splahip_dist_mxv_masked runs the UNSPLIT block for this call — the exchange first, then one product over the block's own column-ordered blocks — although the
 * two-part form could serve it (SPLAHIP_DIST_FORM=whole; measured slower than the two-part form at the sizes tried, so not the default) */
int splahip_shard_prefers_whole(splahip_shard shard, int type, int op_add);
int splahip_shard_mxv_local(splahip_shard shard, int type, const void* d_mask, const void* d_v_local, void* d_r, int op_mult, int op_add, int op_select,
                            uint32_t init_bits, void* stream);
int splahip_shard_mxv_remote(splahip_shard shard, int type, const void* d_mask, const void* d_v_full, void* d_r, int op_mult, int op_add, int op_select, void* stream);
int splahip_shard_mxv_whole(splahip_shard shard, int type, const void* d_mask, const void* d_v_full, void* d_r, int op_mult, int op_add, int op_select,
                            uint32_t init_bits, int early_exit, void* stream);
int splahip_dist_mxv_masked(splahip_comm comm, splahip_shard shard, int type, const void* d_mask, const void* d_v_local, void* d_v_full, void* d_r, int op_mult,
                            int op_add, int op_select, uint32_t init_bits, int early_exit, int direct, int* overlapped, void* stream);
/* Sparsity-aware exchange of the input vector (a BFS frontier, the distances an SSSP step changed): only the entries that differ from the vector's fill
 * value travel — (position in the padded layout[, value]) — and land in d_v_full, which then equals what the dense all-gather would hold.  capacity = most
 * entries one rank may contribute to one exchange; m (0 = capacity) = the bound every rank knows for THIS exchange (e.g. the frontier size the previous
 * step counted): a rank's message is 1 + m (positions) [+ m (values)] words, so no count exchange and no host synchronisation is needed.  with_values = 0:
 * every live entry is written as value_bits (a 0/1 frontier).  The previous exchange's positions are returned to the fill value first.  pack / buffers /
 * scatter are the phases for a host that brings its own transport; exchange_sparse runs them around splahip_dist_all_gather.  overflowed: some rank met
 * more entries than its bound (the exchange was incomplete — redo it dense).                                                                          */
typedef struct splahip_sx_s* splahip_sx;
int splahip_dist_sx_create(splahip_sx* sx, int world, int rank, uint32_t max_rows, uint32_t capacity);
int splahip_dist_sx_destroy(splahip_sx sx);
int splahip_dist_sx_pack(splahip_sx sx, const void* d_v_local, uint32_t n_local, uint32_t fill_bits, uint32_t m, int with_values, void* stream);
int splahip_dist_sx_buffers(splahip_sx sx, uint32_t m, int with_values, void** d_send, void** d_recv, size_t* words_per_rank);
int splahip_dist_sx_scatter(splahip_sx sx, void* d_v_full, uint32_t fill_bits, uint32_t m, int with_values, uint32_t value_bits, void* stream);
int splahip_dist_sx_overflowed(splahip_sx sx, int* overflowed);
int splahip_dist_exchange_sparse(splahip_comm comm, splahip_sx sx, const void* d_v_local, uint32_t n_local, uint32_t fill_bits, uint32_t m, int with_values,
                                 uint32_t value_bits, void* d_v_full, int direct, void* stream);

/* ---- masked vxm: r = v^T * M on the columns j with select(mask[j]) (reference cpu_vxm.hpp:92-125) ----------------
 * v is sparse: nvals (index, value) pairs IN STORED ORDER; r is index-sorted COO written to d_ri / d_rx (room for
 * n_cols entries), its entry count is returned on the host (these calls block).  `init` is not an input: the
 * reference never uses it for vxm.
 *   probe : number of products sum_i nnz(M[v_i,:]) and whether v's indices are strictly ascending — what the host
 *           needs to choose the form (pull needs ascending indices, an associative + commutative op_add, and pays
 *           nnz(M) whatever the frontier; push pays ~9x the product count).
 *   pull  : masked SpMV over the CSR of M^T (d_Tp/d_Tj/d_Tx + its plan, see splahip_csr_transpose); v_is_identity != 0
 *           says v stores every index 0..n_rows-1 in order, so d_vx is used as the dense vector directly.
 *   push  : expand / stable sort by column / in-order fold of every column's run / compact.  Exact for every op.      */
int splahip_vxm_probe(uint32_t nvals, const uint32_t* d_vi, const uint32_t* d_Ap, uint64_t* h_products, int* h_sorted, void* stream);
int splahip_vxm_pull(splahip_plan plan_T, int type, uint32_t n_rows, uint32_t n_cols,
                     const uint32_t* d_Tp, const uint32_t* d_Tj, const void* d_Tx, const void* d_mask,
                     uint32_t nvals, const uint32_t* d_vi, const void* d_vx, int v_is_identity,
                     uint32_t* d_ri, void* d_rx, uint32_t* h_count,
                     int op_mult, int op_add, int op_select, void* stream);
int splahip_vxm_push(int type, uint32_t n_rows, uint32_t n_cols,
                     const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, const void* d_mask,
                     uint32_t nvals, const uint32_t* d_vi, const void* d_vx,
                     uint32_t* d_ri, void* d_rx, uint32_t* h_count,
                     int op_mult, int op_add, int op_select, void* stream);
/* same, with the product count splahip_vxm_probe already returned (0 = unknown): saves the call one read-back */
int splahip_vxm_push_n(int type, uint32_t n_rows, uint32_t n_cols,
                       const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, const void* d_mask,
                       uint32_t nvals, const uint32_t* d_vi, const void* d_vx,
                       uint32_t* d_ri, void* d_rx, uint32_t* h_count,
                       int op_mult, int op_add, int op_select, uint64_t n_products, void* stream);
/* small frontiers: probe, expansion and compaction back to back with ONE host synchronisation.  The product count is taken on the
 * device; when it is <= max_products the call completes the push (order-free op_add only: associative, commutative and not an fp32
 * PLUS / MULT) and *h_done = 1; otherwise *h_done = 0, nothing was written, and *h_products / *h_sorted are what splahip_vxm_probe
 * would have returned, so the caller goes on to splahip_vxm_pull / splahip_vxm_push_n without probing again. */
int splahip_vxm_push_small(int type, uint32_t n_rows, uint32_t n_cols,
                           const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, const void* d_mask,
                           uint32_t nvals, const uint32_t* d_vi, const void* d_vx,
                           uint32_t* d_ri, void* d_rx, uint32_t max_products, uint32_t* h_count,
                           uint64_t* h_products, int* h_sorted, int* h_done,
                           int op_mult, int op_add, int op_select, void* stream);
/* CSR of the transpose (n_cols rows): entries of a column keep ascending row order (stable).  d_Tp: n_cols+1 words,
 * d_Tj / d_Tx: nnz words.  Blocks; = splahip_m_transpose with the identity.  (reference: the CSC slot FormatMatrix::AccCsc, include/spla/config.hpp:102-122) */
int splahip_csr_transpose(uint32_t n_rows, uint32_t n_cols, uint32_t nnz, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax,
                          uint32_t* d_Tp, uint32_t* d_Tj, void* d_Tx, void* stream);

/* ---- SpGEMM ------------------------------------------------------------------------------------------------------
 * Outputs: d_Rp (rows+1 words) is caller-allocated; the column / value arrays are allocated by the library once the
 * result size is known and handed to the caller (*d_Rj, *d_Rx; release with splahip_free); *h_nnz = entry count.
 * These calls block.
 *
 * mxmT_masked: R = (A * B^T) .mask — for every stored mask entry (i, j, m) with select(m): the column-sorted rows A[i,:]
 *   and B[j,:] are intersected, r = fold from init of add(r, mult(a, b)) in ascending column order; entries with
 *   r == init are dropped (reference cpu_mxmT_masked.hpp:87-125).  mask/A/R have n_rows rows; B has one row per
 *   column of R.
 * mxm: R = A * B — R[i,j] folds mult(a_ik, b_kj) over k in the order of A[i,:], starting at init; entries == init are
 *   dropped; columns ascending (reference cpu_mxm.hpp:85-114).  An op_add with an LDS atomic (PLUS; MIN / MAX / BOR / BAND / BXOR on int and
 *   uint; LOR / LAND) runs by row classes (mxm_spa.hip): rows of more than 4096 products keep one result row's accumulator block in LDS
 *   (Gustavson), lighter rows and every other op_add run ESC over row blocks of at most `block_products` expanded products (0 = default);
 *   fp32 MULT / PLUS: a dense leading corner of A and B is multiplied on the matrix cores.  *h_products (optional) = total number of products. */
int splahip_mxmT_masked(int type, uint32_t n_rows, uint32_t mask_nnz,
                        const uint32_t* d_Mp, const uint32_t* d_Mj, const void* d_Mx,
                        const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax,
                        const uint32_t* d_Bp, const uint32_t* d_Bj, const void* d_Bx,
                        uint32_t* d_Rp, uint32_t** d_Rj, void** d_Rx, uint32_t* h_nnz,
                        int op_mult, int op_add, int op_select, uint32_t init_bits, void* stream);
int splahip_mxm(int type, uint32_t a_rows, uint32_t b_cols,
                const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax,
                const uint32_t* d_Bp, const uint32_t* d_Bj, const void* d_Bx,
                uint32_t* d_Rp, uint32_t** d_Rj, void** d_Rx, uint32_t* h_nnz, uint64_t* h_products,
                int op_mult, int op_add, uint32_t init_bits, uint64_t block_products, void* stream);
/* which row classes the last splahip_mxm of this host thread used (mxm_spa.hip): rows served by the LDS-accumulator form, entries the light rows'
 * ESC call produced, rows of the dense window computed on the matrix cores (0 = none; splahip_mxm_last_dense: where it lies).  All 0: the whole call ran the ESC form. */
int splahip_mxm_last_info(uint32_t* n_heavy_rows, uint32_t* light_nnz, uint32_t* dense_h);
/* Measurement helper (bench.py --workload mxm-dense): the time in ms of the two MFMA GEMMs (values and product counts) of the dense window of the calling thread's
 * last splahip_mxm call, measured with events when SPLAHIP_MXM_DENSE_TIME=1 was set for that call (0: no window, or not timed). */
int splahip_mxm_last_dense_ms(float* gemm_ms);
/* the dense window of that call: six words i0, k0, j0, rows, inner, columns — A[i0 : i0 + rows, k0 : k0 + inner] x B[k0 : k0 + inner, j0 : j0 + columns], blocks of
 * half-full 32 x 32 tiles found ANYWHERE in the operands (fp32 MULT / PLUS), went through the matrix cores (LDS-staged v_mfma_f32_32x32x2_f32); rows == 0: none.
 * (reference: the accumulator row of Algo_mxm_cpu, src/cpu/cpu_mxm.hpp:93-113, restricted to the window's cells) */
int splahip_mxm_last_dense(uint32_t* window6);
/* mxm / mxmT keep device memory between calls, per host thread: mxm's GB-sized ESC workspaces (grow-only), the scratch arena of the blocking calls and the cached
 * result arrays that were handed back with splahip_free.  This returns the calling thread's share to the driver; any allocation of the library
 * that runs out of memory does the same before it retries. */
int splahip_mxm_trim(void);

/* ---- vector operations between two hot-path calls of a traversal --------------------------------
 * v_assign_masked: r[i] = assign(r[i], value) where select(mask[i]); dense mask (n words) or sparse mask (nvals pairs)
 *                  (reference src/cpu/cpu_v_assign.hpp:68-132).
 * v_count_mf     : number of entries != fill of a dense vector, returned on the host (reference cpu_v_count_mf.hpp:99-108).
 * v_eadd_fdb     : r[i] = op(r[i], v[i]); the feedback vector receives the entries that changed: dense form fdb[i] = changed ?
 *                  r[i] : fdb_fill; sparse form (v = nvals pairs, each index at most once) appends (i, r[i]) in v's stored
 *                  order, count returned on the host (reference src/cpu/cpu_v_eadd_fdb.hpp:71-139).                       */
int splahip_v_assign_masked_dense(int type, uint32_t n, void* d_r, const void* d_mask, uint32_t value_bits, int op_assign, int op_select, void* stream);
int splahip_v_assign_masked_sparse(int type, uint32_t nvals, void* d_r, const uint32_t* d_mi, const void* d_mx, uint32_t value_bits, int op_assign,
                                   int op_select, void* stream);
int splahip_v_count_mf_dense(int type, uint32_t n, const void* d_v, uint32_t fill_bits, uint32_t* h_count, void* stream);
int splahip_v_eadd_fdb_dense(int type, uint32_t n, void* d_r, const void* d_v, void* d_fdb, uint32_t fdb_fill_bits, int op, void* stream);
int splahip_v_eadd_fdb_sparse(int type, uint32_t nvals, void* d_r, const uint32_t* d_vi, const void* d_vx, uint32_t* d_fi, void* d_fx,
                              uint32_t* h_count, int op, void* stream);

/* ---- the PageRank step remainder and the triangle-count tail -------------------------------
 * v_eadd: r = op(u, v) element-wise.  dense: every index (reference src/cpu/cpu_v_eadd.hpp:133-155).  sparse: the sorted merge
 * of two index-sorted COO vectors without duplicate indices; an index held by one side only meets the other side's fill
 * value (cpu_v_eadd.hpp:96-127); d_ri / d_rx need room for nu + nv entries; the count returns on the host (blocks).
 * reduce: *h_result_bits = fold of d_x[0..n) from s with op — exec_v_reduce over a vector's stored values
 * (src/cpu/cpu_v_reduce.hpp:69-117) and exec_m_reduce over a CSR's Ax (src/cpu/cpu_m_reduce.hpp:134-156).  Associative and
 * commutative ops use a fixed reduction tree (fp32 PLUS / MULT then differ from the sequential fold by rounding only);
 * every other op is folded strictly left to right.  Blocks until the result is on the host.                          */
int splahip_v_eadd_dense(int type, uint32_t n, void* d_r, const void* d_u, const void* d_v, int op, void* stream);
int splahip_v_eadd_sparse(int type, uint32_t nu, const uint32_t* d_ui, const void* d_ux, uint32_t u_fill_bits, uint32_t nv, const uint32_t* d_vi,
                          const void* d_vx, uint32_t v_fill_bits, uint32_t* d_ri, void* d_rx, uint32_t* h_count, int op, void* stream);
int splahip_reduce(int type, uint32_t n, const void* d_x, uint32_t s_bits, uint32_t* h_result_bits, int op, void* stream);

/* ---- transpose and element-wise map -----------------------------------------------------------
 * m_transpose: R = op_apply(M)^T, both CSR (reference exec_m_transpose, include/spla/exec.hpp:318-323; semantics of
 * Algo_m_transpose_cpu::execute_csr, src/cpu/cpu_m_transpose.hpp:141-185): d_Rp has n_cols + 1 entries, d_Rj / d_Rx nnz; inside a
 * row of R the entries are in ascending source-row order.  op_apply: a splahip_unop id.  Asynchronous on `stream`.
 * map: d_out[i] = op(d_x[i]), i < n (reference exec_v_map, exec.hpp:468-473; src/cpu/cpu_v_map.hpp:70-121); d_out may equal d_x. */
int splahip_m_transpose(int type, uint32_t n_rows, uint32_t n_cols, uint32_t nnz, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax,
                        uint32_t* d_Rp, uint32_t* d_Rj, void* d_Rx, int op_apply, void* stream);
int splahip_map(int type, uint32_t n, const void* d_x, void* d_out, int op, void* stream);
/* The vector operations with USER-DEFINED ops (OpBinary::make_* / OpSelect::make_* / OpUnary::make_*, reference src/op.cpp:294-317): *_src is the op's body text
 * "(T a, T b) { ... }" / "(T a) { ... }" with T spelled int / uint / float, compiled at run time (hiprtc) into the element-wise kernel of the operation; a NULL
 * source where an id is also taken means the built-in op with that id.  Same semantics, formats and outputs as the built-in entry points above; the reduce folds
 * strictly left to right from s (a user op need not be associative).  Status 8 = the body does not compile (the reference's Status::CompilationError).
 * splahip_v_assign_masked_jit: sparse != 0: d_mi / d_mask = the mask's stored indices / values (n of them); else d_mask = its dense image (d_mi unused). */
int splahip_v_assign_masked_jit(int type, int sparse, uint32_t n, void* d_r, const uint32_t* d_mi, const void* d_mask, uint32_t value_bits, int op_assign,
                                const char* assign_src, int op_select, const char* select_src, void* stream);
int splahip_v_eadd_fdb_dense_jit(int type, uint32_t n, void* d_r, const void* d_v, void* d_fdb, uint32_t fdb_fill_bits, const char* op_src, void* stream);
int splahip_v_eadd_fdb_sparse_jit(int type, uint32_t nvals, void* d_r, const uint32_t* d_vi, const void* d_vx, uint32_t* d_fi, void* d_fx, uint32_t* h_count,
                                  const char* op_src, void* stream);
int splahip_v_eadd_dense_jit(int type, uint32_t n, void* d_r, const void* d_u, const void* d_v, const char* op_src, void* stream);
int splahip_v_eadd_sparse_jit(int type, uint32_t nu, const uint32_t* d_ui, const void* d_ux, uint32_t u_fill_bits, uint32_t nv, const uint32_t* d_vi, const void* d_vx,
                              uint32_t v_fill_bits, uint32_t* d_ri, void* d_rx, uint32_t* h_count, const char* op_src, void* stream);
int splahip_reduce_jit(int type, uint32_t n, const void* d_x, uint32_t s_bits, uint32_t* h_result_bits, const char* op_src, void* stream);
int splahip_map_jit(int type, uint32_t n, const void* d_x, void* d_out, const char* op_src, void* stream);

/* ---- accelerator vector formats ------------------------------------------------------------------------------
 * AccDense <-> AccCoo (reference src/opencl/cl_format_dense_vec.hpp:90-143, cl_format_coo_vec.hpp:149-158).
 * dense_to_coo keeps the entries != fill in ascending index order (d_idx / d_val need room for n entries) and
 * returns the count on the host (blocks).  coo_to_dense fills with `fill` then scatters (last write wins).      */
int splahip_vec_dense_to_coo(int type, uint32_t n, const void* d_dense, uint32_t fill_bits,
                             uint32_t* d_idx, void* d_val, uint32_t* h_count, void* stream);
/* the same with the entries written only if there are at most `limit` of them (the count is returned either way: the caller of a v_count_mf that hands the vector
 * to a push step next — reference src/algorithm.cpp:88-99 — gets the sparse form for the price of the count when the frontier is small, the count alone when it is not) */
int splahip_vec_dense_to_coo_limited(int type, uint32_t n, const void* d_dense, uint32_t fill_bits,
                                     uint32_t* d_idx, void* d_val, uint32_t limit, uint32_t* h_count, void* stream);
int splahip_vec_coo_to_dense(uint32_t n, uint32_t nvals, const uint32_t* d_idx, const void* d_val, uint32_t fill_bits,
                             void* d_dense, void* stream);

/* name of the kernel variant the call above would launch for these arguments (for profiles / DESIGN.md) */
const char* splahip_mxv_variant(int type, int op_mult, int op_add, int early_exit);
const char* splahip_mxv_variant_of(splahip_plan plan, int type, int op_mult, int op_add, int early_exit); /* with the plan's sliced-ELL copy taken into account */

#ifdef __cplusplus
}
#endif

#endif /* SPLA_HIP_H */
