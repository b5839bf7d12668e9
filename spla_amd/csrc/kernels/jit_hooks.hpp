// jit_hooks.hpp — where a run-time-compiled kernel (jit.hip: user-defined ops) plugs into the operators' own drivers: the op-dependent kernel of
// each operator is replaced by the hiprtc-built one, everything around it (workspaces, scans, compaction, hand-off of the result arrays) is the
// driver the built-in ops use.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.hpp"

namespace splahip {

    // mxm.hip — exec_mxmT_masked: `entry` computes val[e] / keep[e] for every stored mask entry (reference cpu_mxmT_masked.hpp:87-125):
    //   (uint n_rows, uint n_entries, Mp, Mj, Mx, Ap, Aj, Ax, Bp, Bj, Bx, T* val, uint* keep, T init)
    int mxmT_jit(uint32_t n_rows, uint32_t n_entries, const uint32_t* Mp, const uint32_t* Mj, const void* Mx, const uint32_t* Ap, const uint32_t* Aj, const void* Ax,
                 const uint32_t* Bp, const uint32_t* Bj, const void* Bx, uint32_t* Rp, uint32_t** Rj, void** Rx, uint32_t* h_nnz, uint32_t init_bits, hipFunction_t entry,
                 hipStream_t s);
    // mxm.hip — exec_mxm, expand / stable sort / in-order compress (reference cpu_mxm.hpp:85-114): `expand` writes (64-bit key, product) per product,
    //   (uint n_products, uint e0, uint n_entries, offs, uint n_rows, Ap, Aj, Ax, Bp, Bj, Bx, uint r0, uint col_bits, u64* key, T* val)
    // `run` folds every run of equal keys in order from init,
    //   (const u64* key, const T* val, uint n, head, run_rank, u64* run_key, T* run_val, uint* run_keep, T init)
    int mxm_jit(uint32_t a_rows, uint32_t b_cols, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const uint32_t* Bp, const uint32_t* Bj, const void* Bx, uint32_t* Rp,
                uint32_t** Rj, void** Rx, uint32_t* h_nnz, uint32_t init_bits, hipFunction_t expand, hipFunction_t run, hipStream_t s);
    // vxm.hip — exec_vxm_masked, pull form over the transposed matrix (reference cpu_vxm.hpp:92-125): `column` folds one output index,
    //   (uint n_out, Tp, Tj, Tx, mask, const T* v_dense, const uint* present, const uint* vpos, int sorted, T* outv, uint* touched, uint long_from)
    // `column_long` (may be null; index-sorted v only) folds the listed output indices of more than long_from entries, a wavefront each,
    //   (const uint* rows, const uint* n_listed, Tp, Tj, Tx, mask, const T* v_dense, const uint* present, T* outv, uint* touched)
    int vxm_pull_jit(splahip_plan plan_T, uint32_t n_out, uint32_t n_in, const uint32_t* Tp, const uint32_t* Tj, const void* Tx, const void* mask, uint32_t nvals,
                     const uint32_t* vi, const void* vx, int sorted, uint32_t* ri, void* rx, uint32_t* h_count, hipFunction_t column, hipFunction_t column_long, hipStream_t s);

    // vector_ops.hip — v_eadd_fdb, sparse operand (reference cpu_v_eadd_fdb.hpp:71-139): `apply` is the pass that applies the op and flags the changed entries,
    //   (uint nvals, T* r, const uint* vi, const T* vx, uint* changed, T* newval); the order-preserving compaction of the feedback is the built-in drivers'
    int eadd_fdb_sparse_jit(uint32_t nvals, void* r, const uint32_t* vi, const void* vx, uint32_t* fi, void* fx, uint32_t* h_count, hipFunction_t apply, hipStream_t s);
    // vector_ops.hip — v_eadd, sparse (+) sparse (reference cpu_v_eadd.hpp:96-127): the built-in merge runs twice, with FIRST and with SECOND, so that the two
    // operands of every output entry (a stored value or the other side's fill value) sit side by side; `combine` then applies the op,
    //   (uint n, T* r, const T* a, const T* b)
    int eadd_sparse_jit(int type, uint32_t nu, const uint32_t* ui, const void* ux, uint32_t u_fill, uint32_t nv, const uint32_t* vi, const void* vx, uint32_t v_fill,
                        uint32_t* ri, void* rx, uint32_t* h_count, hipFunction_t combine, hipStream_t s);

}// namespace splahip
