/*
 * oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded restatement of the reference's four sequential CPU algorithms over plain CSR /
 * dense / COO arrays.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed baseline — the product (spla_amd) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks every function here against the genuine
 * reference CPU backend (oracle/_ref/libspla_ref.so, compiled by oracle/Makefile from /root/reference) on the
 * reference's own unit-test cases (tests/test_mxv.cpp, test_vxm.cpp, test_mxm.cpp, test_mxmT.cpp) and on
 * randomised inputs over every (type x op) cell, and against the golden vectors under tests/golden/ that
 * were captured from that same reference build (tests/golden/make_golden.py).
 *
 * Every element is a 4-byte word (int32 / uint32 / float, reference src/type.cpp:32-35) passed as uint32_t bit
 * patterns; `type` says how to interpret them.  Operation ids are those of include/spla_hip_ops.h.
 */
#ifndef SPLA_ORACLE_H
#define SPLA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* scalar ops, exposed so tests can probe single (op, type) cells.  follows reference src/op.cpp:194-266 */
uint32_t oracle_binop(int type, int op, uint32_t a, uint32_t b);
int      oracle_selop(int type, int op, uint32_t a);

/*
 * r[i] = select(mask[i]) ? fold_{k in row i, ascending k} add(sum, mult(Ax[k], v[Aj[k]])) from init : init
 * with the optional early exit "stop as soon as sum != init".
 * follows reference src/cpu/cpu_mxv.hpp:88-103 (fold order = position in the row; mult(M value, v value)).
 */
void oracle_mxv_masked(int type, uint32_t n_rows,
                       const uint32_t* Ap, const uint32_t* Aj, const uint32_t* Ax,
                       const uint32_t* mask, const uint32_t* v, uint32_t* r,
                       int op_mult, int op_add, int op_select, uint32_t init, int early_exit);

/*
 * r = v^T * M restricted to columns j with select(mask[j]); v sparse (vi, vx) in stored order; first hit of a
 * column stores mult(v_i, M_ij), later hits add(r_j, mult(v_i, M_ij)); init is unused; output sorted by index
 * with explicit zeros kept.  Returns the number of output entries written to (ri, rx) (capacity n_cols).
 * follows reference src/cpu/cpu_vxm.hpp:92-125.
 */
uint32_t oracle_vxm_masked(int type, uint32_t n_rows, uint32_t n_cols,
                           const uint32_t* Ap, const uint32_t* Aj, const uint32_t* Ax,
                           const uint32_t* mask, uint32_t nvals, const uint32_t* vi, const uint32_t* vx,
                           uint32_t* ri, uint32_t* rx, int op_mult, int op_add, int op_select);

/*
 * R = A * B, Gustavson row by row with an n_cols-wide accumulator pre-filled with init; entries whose final
 * value equals init are dropped; columns ascending.  Rp must hold A_rows+1 words; *Rj / *Rx are malloc'ed
 * (free with oracle_free).  follows reference src/cpu/cpu_mxm.hpp:85-114.
 */
void oracle_mxm(int type, uint32_t a_rows, uint32_t b_cols,
                const uint32_t* Ap, const uint32_t* Aj, const uint32_t* Ax,
                const uint32_t* Bp, const uint32_t* Bj, const uint32_t* Bx,
                uint32_t* Rp, uint32_t** Rj, uint32_t** Rx,
                int op_mult, int op_add, uint32_t init);

/*
 * R = (A * B^T) .mask : for every stored mask entry (i, j, m) in row order, r = init; if select(m) two-pointer
 * merge of A[i,:] and B[j,:] (both column-sorted), r = add(r, mult(a, b)) on equal columns; emitted iff r != init.
 * follows reference src/cpu/cpu_mxmT_masked.hpp:87-125.
 */
void oracle_mxmT_masked(int type, uint32_t n_rows,
                        const uint32_t* Mp, const uint32_t* Mj, const uint32_t* Mx,
                        const uint32_t* Ap, const uint32_t* Aj, const uint32_t* Ax,
                        const uint32_t* Bp, const uint32_t* Bj, const uint32_t* Bx,
                        uint32_t* Rp, uint32_t** Rj, uint32_t** Rx,
                        int op_mult, int op_add, int op_select, uint32_t init);

/* ---- the vector operations of a traversal step ---- */
/* r[i] = assign(r[i], value) where select(mask[i]): dense mask / sparse mask in stored order.  reference src/cpu/cpu_v_assign.hpp:68-132 */
void oracle_v_assign_dense(int type, uint32_t n, uint32_t* r, const uint32_t* mask, uint32_t value, int op_assign, int op_select);
void oracle_v_assign_sparse(int type, uint32_t nvals, uint32_t* r, const uint32_t* mi, const uint32_t* mx, uint32_t value, int op_assign, int op_select);
/* r[i] = op(r[i], v[i]) with feedback of the changed entries.  reference src/cpu/cpu_v_eadd_fdb.hpp:71-139 */
void     oracle_v_eadd_fdb_dense(int type, uint32_t n, uint32_t* r, const uint32_t* v, uint32_t* fdb, uint32_t fdb_fill, int op);
uint32_t oracle_v_eadd_fdb_sparse(int type, uint32_t nvals, uint32_t* r, const uint32_t* vi, const uint32_t* vx, uint32_t* fi, uint32_t* fx, int op);
/* entries != fill of a dense vector.  reference src/cpu/cpu_v_count_mf.hpp:99-108 */
uint32_t oracle_v_count_mf_dense(int type, uint32_t n, const uint32_t* v, uint32_t fill);

/* ---- PageRank step remainder and triangle-count tail ---- */
/* r = op(u, v): dense, and the merge of two index-sorted sparse vectors.  reference src/cpu/cpu_v_eadd.hpp:70-155 */
void     oracle_v_eadd_dense(int type, uint32_t n, uint32_t* r, const uint32_t* u, const uint32_t* v, int op);
uint32_t oracle_v_eadd_sparse(int type, uint32_t nu, const uint32_t* ui, const uint32_t* ux, uint32_t u_fill, uint32_t nv, const uint32_t* vi,
                              const uint32_t* vx, uint32_t v_fill, uint32_t* ri, uint32_t* rx, int op);
/* strict left fold of x[0..n) from s.  reference src/cpu/cpu_v_reduce.hpp:69-117, src/cpu/cpu_m_reduce.hpp:134-156 */
uint32_t oracle_reduce(int type, uint32_t n, const uint32_t* x, uint32_t s, int op);

/* ---- transpose and element-wise map ---- */
uint32_t oracle_unop(int type, int op, uint32_t a); /* reference src/op.cpp:99-133 */
/* R = op(M)^T, CSR to CSR; Rp: n_cols + 1 words, Rj / Rx: nnz words.  reference src/cpu/cpu_m_transpose.hpp:141-185 */
void oracle_m_transpose(int type, uint32_t n_rows, uint32_t n_cols, const uint32_t* Ap, const uint32_t* Aj, const uint32_t* Ax, uint32_t* Rp,
                        uint32_t* Rj, uint32_t* Rx, int op_apply);
/* out[i] = op(x[i]).  reference src/cpu/cpu_v_map.hpp:70-121 */
void oracle_map(int type, uint32_t n, const uint32_t* x, uint32_t* out, int op);

void oracle_free(void* p);

#ifdef __cplusplus
}
#endif

#endif /* SPLA_ORACLE_H */
