/*
 * oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See oracle.h for the contract and the pinning status.
 *
 * Plain C99, single thread, one indirect call per scalar op (the reference pays a std::function call per op,
 * reference src/core/top.hpp:173-192), so its timing is a fair "port" of the reference CPU path.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/spla_hip_ops.h"

typedef uint32_t word;
typedef word (*binfn)(word, word);
typedef int (*selfn)(word);

static inline int32_t as_i(word w) { int32_t x; memcpy(&x, &w, 4); return x; }
static inline float   as_f(word w) { float x;   memcpy(&x, &w, 4); return x; }
static inline word    from_i(int32_t x) { word w; memcpy(&w, &x, 4); return w; }
static inline word    from_f(float x)   { word w; memcpy(&w, &x, 4); return w; }

/* ---- binary ops, reference src/op.cpp:194-241.  int32 +,-,* use wrap-around (the reference has signed
 * overflow UB there; fixtures stay clear of it) ---- */
#define BIN_I(name, expr) static word name##_i(word wa, word wb) { int32_t a = as_i(wa), b = as_i(wb); (void)a; (void)b; return from_i((int32_t)(expr)); }
#define BIN_U(name, expr) static word name##_u(word a, word b) { (void)a; (void)b; return (word)(expr); }
#define BIN_F(name, expr) static word name##_f(word wa, word wb) { float a = as_f(wa), b = as_f(wb); (void)a; (void)b; return from_f((float)(expr)); }

BIN_I(plus, (uint32_t)a + (uint32_t)b)   BIN_U(plus, a + b)   BIN_F(plus, a + b)
BIN_I(minus, (uint32_t)a - (uint32_t)b)  BIN_U(minus, a - b)  BIN_F(minus, a - b)
BIN_I(mult, (uint32_t)a * (uint32_t)b)   BIN_U(mult, a * b)   BIN_F(mult, a * b)
BIN_I(divi, b == 0 ? 0 : (a == INT32_MIN && b == -1 ? a : a / b))
BIN_U(divi, b == 0 ? 0 : a / b)
BIN_F(divi, a / b)
BIN_I(mpow2, ((uint32_t)a - (uint32_t)b) * ((uint32_t)a - (uint32_t)b))
BIN_U(mpow2, (a - b) * (a - b))
BIN_F(mpow2, (a - b) * (a - b))
BIN_I(first, a)   BIN_U(first, a)   BIN_F(first, a)
BIN_I(second, b)  BIN_U(second, b)  BIN_F(second, b)
BIN_I(bone, 1)    BIN_U(bone, 1)    BIN_F(bone, 1.0f)
/* std::min(a,b) = (b < a) ? b : a ; std::max(a,b) = (a < b) ? b : a   (reference src/op.cpp:151-155) */
BIN_I(min, (b < a) ? b : a)  BIN_U(min, (b < a) ? b : a)  BIN_F(min, (b < a) ? b : a)
BIN_I(max, (a < b) ? b : a)  BIN_U(max, (a < b) ? b : a)  BIN_F(max, (a < b) ? b : a)
BIN_I(lor, (a || b) ? 1 : 0)   BIN_U(lor, (a || b) ? 1u : 0u)   BIN_F(lor, (a || b) ? 1.0f : 0.0f)
BIN_I(land, (a && b) ? 1 : 0)  BIN_U(land, (a && b) ? 1u : 0u)  BIN_F(land, (a && b) ? 1.0f : 0.0f)
BIN_I(bor, a | b)   BIN_U(bor, a | b)
BIN_I(band, a & b)  BIN_U(band, a & b)
BIN_I(bxor, a ^ b)  BIN_U(bxor, a ^ b)

static const binfn k_bin[3][SPLAHIP_BINOP_COUNT] = {
        {plus_i, minus_i, mult_i, divi_i, mpow2_i, first_i, second_i, bone_i, min_i, max_i, lor_i, land_i, bor_i, band_i, bxor_i},
        {plus_u, minus_u, mult_u, divi_u, mpow2_u, first_u, second_u, bone_u, min_u, max_u, lor_u, land_u, bor_u, band_u, bxor_u},
        {plus_f, minus_f, mult_f, divi_f, mpow2_f, first_f, second_f, bone_f, min_f, max_f, lor_f, land_f, NULL, NULL, NULL}};

/* ---- select ops, reference src/op.cpp:243-266 ---- */
#define SEL3(name, op)                                                    \
    static int name##_i(word w) { int32_t a = as_i(w); return a op 0; }   \
    static int name##_u(word a) { return a op 0u; }                       \
    static int name##_f(word w) { float a = as_f(w); return a op 0; }
SEL3(eqz, ==) SEL3(nqz, !=) SEL3(gtz, >) SEL3(gez, >=) SEL3(ltz, <) SEL3(lez, <=)
static int always_x(word a) { (void)a; return 1; }
static int never_x(word a)  { (void)a; return 0; }

static const selfn k_sel[3][SPLAHIP_SELOP_COUNT] = {
        {eqz_i, nqz_i, gtz_i, gez_i, ltz_i, lez_i, always_x, never_x},
        {eqz_u, nqz_u, gtz_u, gez_u, ltz_u, lez_u, always_x, never_x},
        {eqz_f, nqz_f, gtz_f, gez_f, ltz_f, lez_f, always_x, never_x}};

/* "!=" on the element type (float compare for F, so -0.0f == 0.0f and NaN != NaN as in the reference's
 * `sum != sum_init`, reference src/cpu/cpu_mxv.hpp:98) */
static inline int neq(int type, word a, word b) {
    return type == SPLAHIP_FLOAT ? (as_f(a) != as_f(b)) : (a != b);
}

/* ---- unary ops, reference src/op.cpp:99-133.  The reference's integer `1 / a` with a == 0 and abs(INT_MIN) are undefined; the
 * restatement (and the device) use 1/0 = 0 and a wrapping ABS ---- */
typedef word (*unfn)(word);
#define UN_I(name, expr) static word name##_i(word wa) { int32_t a = as_i(wa); (void)a; return from_i((int32_t)(expr)); }
#define UN_U(name, expr) static word name##_u(word a) { (void)a; return (word)(expr); }
#define UN_F(name, expr) static word name##_f(word wa) { float a = as_f(wa); (void)a; return from_f((float)(expr)); }
UN_I(ident, a)                       UN_U(ident, a)                   UN_F(ident, a)
UN_I(ainv, 0u - (uint32_t)a)         UN_U(ainv, 0u - a)               UN_F(ainv, -a)
UN_I(minv, a == 0 ? 0 : 1 / a)       UN_U(minv, a == 0 ? 0u : 1u / a) UN_F(minv, 1.0f / a)
UN_I(lnot, !(a != 0))                UN_U(lnot, !(a != 0))            UN_F(lnot, !(a != 0))
UN_I(uone, 1)                        UN_U(uone, 1u)                   UN_F(uone, 1.0f)
UN_I(absv, a < 0 ? 0u - (uint32_t)a : (uint32_t)a)  UN_U(absv, a)     UN_F(absv, fabs((double) a))
UN_I(bnot, ~a)                       UN_U(bnot, ~a)
/* the reference calls the unqualified C functions (`sin(a)`, src/op.cpp:121-133): evaluated in DOUBLE, then rounded to float
 * (pinned against the reference build: 200 000 random arguments per op, bit for bit) */
UN_F(sqrtv, sqrt((double) a)) UN_F(logv, log((double) a)) UN_F(expv, exp((double) a)) UN_F(sinv, sin((double) a)) UN_F(cosv, cos((double) a))
UN_F(tanv, tan((double) a)) UN_F(asinv, asin((double) a)) UN_F(acosv, acos((double) a)) UN_F(atanv, atan((double) a))
UN_F(ceilv, ceil((double) a)) UN_F(floorv, floor((double) a)) UN_F(roundv, round((double) a)) UN_F(truncv, trunc((double) a))
static const unfn k_un[3][SPLAHIP_UNOP_COUNT] = {
        {ident_i, ainv_i, minv_i, lnot_i, uone_i, absv_i, bnot_i, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
        {ident_u, ainv_u, minv_u, lnot_u, uone_u, absv_u, bnot_u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
        {ident_f, ainv_f, minv_f, lnot_f, uone_f, absv_f, NULL, sqrtv_f, logv_f, expv_f, sinv_f, cosv_f, tanv_f, asinv_f, acosv_f, atanv_f,
         ceilv_f, floorv_f, roundv_f, truncv_f}};

uint32_t oracle_unop(int type, int op, uint32_t a) { return k_un[type][op](a); }
uint32_t oracle_binop(int type, int op, uint32_t a, uint32_t b) { return k_bin[type][op](a, b); }
int      oracle_selop(int type, int op, uint32_t a) { return k_sel[type][op](a); }
void     oracle_free(void* p) { free(p); }

/* reference src/cpu/cpu_mxv.hpp:88-103 */
void oracle_mxv_masked(int type, uint32_t n_rows,
                       const uint32_t* Ap, const uint32_t* Aj, const uint32_t* Ax,
                       const uint32_t* mask, const uint32_t* v, uint32_t* r,
                       int op_mult, int op_add, int op_select, uint32_t init, int early_exit) {
    const binfn mult = k_bin[type][op_mult];
    const binfn add  = k_bin[type][op_add];
    const selfn sel  = k_sel[type][op_select];

    for (uint32_t i = 0; i < n_rows; ++i) {
        word sum = init;
        if (sel(mask[i])) {
            for (uint32_t k = Ap[i]; k < Ap[i + 1]; ++k) {
                sum = add(sum, mult(Ax[k], v[Aj[k]]));
                if (early_exit && neq(type, sum, init)) break;
            }
        }
        r[i] = sum;
    }
}

/* reference src/cpu/cpu_vxm.hpp:92-125.  The reference accumulates in a hash map and sorts the keys; a dense
 * accumulator + touched flags walked in ascending index order yields the same (index, value) list. */
uint32_t oracle_vxm_masked(int type, uint32_t n_rows, uint32_t n_cols,
                           const uint32_t* Ap, const uint32_t* Aj, const uint32_t* Ax,
                           const uint32_t* mask, uint32_t nvals, const uint32_t* vi, const uint32_t* vx,
                           uint32_t* ri, uint32_t* rx, int op_mult, int op_add, int op_select) {
    (void) n_rows;
    const binfn mult = k_bin[type][op_mult];
    const binfn add  = k_bin[type][op_add];
    const selfn sel  = k_sel[type][op_select];

    word*    acc     = (word*) malloc(sizeof(word) * (size_t) (n_cols ? n_cols : 1));
    uint8_t* touched = (uint8_t*) calloc(n_cols ? n_cols : 1, 1);

    for (uint32_t idx = 0; idx < nvals; ++idx) {
        const uint32_t i   = vi[idx];
        const word     v_x = vx[idx];
        for (uint32_t k = Ap[i]; k < Ap[i + 1]; ++k) {
            const uint32_t j = Aj[k];
            if (sel(mask[j])) {
                if (touched[j]) {
                    acc[j] = add(acc[j], mult(v_x, Ax[k]));
                } else {
                    acc[j]     = mult(v_x, Ax[k]);
                    touched[j] = 1;
                }
            }
        }
    }

    uint32_t count = 0;
    for (uint32_t j = 0; j < n_cols; ++j) {
        if (touched[j]) {
            ri[count] = j;
            rx[count] = acc[j];
            ++count;
        }
    }
    free(acc);
    free(touched);
    return count;
}

static int cmp_u32(const void* a, const void* b) {
    const uint32_t x = *(const uint32_t*) a, y = *(const uint32_t*) b;
    return (x > y) - (x < y);
}

typedef struct {
    uint32_t* j;
    uint32_t* x;
    size_t    size, cap;
} outbuf;

static void out_push(outbuf* o, uint32_t j, uint32_t x) {
    if (o->size == o->cap) {
        o->cap = o->cap ? o->cap * 2 : 1024;
        o->j   = (uint32_t*) realloc(o->j, o->cap * sizeof(uint32_t));
        o->x   = (uint32_t*) realloc(o->x, o->cap * sizeof(uint32_t));
    }
    o->j[o->size] = j;
    o->x[o->size] = x;
    o->size++;
}

/* reference src/cpu/cpu_mxm.hpp:85-114.  The reference refills and rescans the whole n_cols-wide accumulator per
 * row; only columns touched by a product can end up != init, so tracking + sorting the touched columns emits
 * the identical ascending (col, value) list without the O(n_rows * n_cols) scan. */
void oracle_mxm(int type, uint32_t a_rows, uint32_t b_cols,
                const uint32_t* Ap, const uint32_t* Aj, const uint32_t* Ax,
                const uint32_t* Bp, const uint32_t* Bj, const uint32_t* Bx,
                uint32_t* Rp, uint32_t** Rj, uint32_t** Rx,
                int op_mult, int op_add, uint32_t init) {
    const binfn mult = k_bin[type][op_mult];
    const binfn add  = k_bin[type][op_add];

    const size_t nc      = b_cols ? b_cols : 1;
    word*        acc     = (word*) malloc(sizeof(word) * nc);
    uint8_t*     touched = (uint8_t*) calloc(nc, 1);
    uint32_t*    list    = (uint32_t*) malloc(sizeof(uint32_t) * nc);
    outbuf       out     = {0, 0, 0, 0};

    for (size_t j = 0; j < nc; ++j) acc[j] = init;

    Rp[0] = 0;
    for (uint32_t i = 0; i < a_rows; ++i) {
        uint32_t n_touched = 0;
        for (uint32_t ka = Ap[i]; ka < Ap[i + 1]; ++ka) {
            const uint32_t k = Aj[ka];
            const word     a = Ax[ka];
            for (uint32_t kb = Bp[k]; kb < Bp[k + 1]; ++kb) {
                const uint32_t j = Bj[kb];
                acc[j]           = add(acc[j], mult(a, Bx[kb]));
                if (!touched[j]) {
                    touched[j]        = 1;
                    list[n_touched++] = j;
                }
            }
        }
        qsort(list, n_touched, sizeof(uint32_t), cmp_u32);
        if (neq(type, init, init)) {
            /* a NaN init (fp32): `R_tmp[col] != I` (cpu_mxm.hpp:109-113) holds for EVERY column, touched or not — the reference emits a dense row */
            for (uint32_t j = 0; j < b_cols; ++j) out_push(&out, j, acc[j]);
        } else {
            for (uint32_t t = 0; t < n_touched; ++t) {
                const uint32_t j = list[t];
                if (neq(type, acc[j], init)) out_push(&out, j, acc[j]);
            }
        }
        for (uint32_t t = 0; t < n_touched; ++t) {
            acc[list[t]]     = init;
            touched[list[t]] = 0;
        }
        Rp[i + 1] = (uint32_t) out.size;
    }

    free(acc);
    free(touched);
    free(list);
    *Rj = out.j;
    *Rx = out.x;
}

/* reference src/cpu/cpu_mxmT_masked.hpp:87-125 */
void oracle_mxmT_masked(int type, uint32_t n_rows,
                        const uint32_t* Mp, const uint32_t* Mj, const uint32_t* Mx,
                        const uint32_t* Ap, const uint32_t* Aj, const uint32_t* Ax,
                        const uint32_t* Bp, const uint32_t* Bj, const uint32_t* Bx,
                        uint32_t* Rp, uint32_t** Rj, uint32_t** Rx,
                        int op_mult, int op_add, int op_select, uint32_t init) {
    const binfn mult = k_bin[type][op_mult];
    const binfn add  = k_bin[type][op_add];
    const selfn sel  = k_sel[type][op_select];
    outbuf      out  = {0, 0, 0, 0};

    Rp[0] = 0;
    for (uint32_t i = 0; i < n_rows; ++i) {
        for (uint32_t km = Mp[i]; km < Mp[i + 1]; ++km) {
            const uint32_t j = Mj[km];
            word           r = init;
            if (sel(Mx[km])) {
                uint32_t ka = Ap[i], ea = Ap[i + 1];
                uint32_t kb = Bp[j], eb = Bp[j + 1];
                while (ka < ea && kb < eb) {
                    if (Aj[ka] == Bj[kb]) {
                        r = add(r, mult(Ax[ka], Bx[kb]));
                        ++ka;
                        ++kb;
                    } else if (Aj[ka] < Bj[kb]) {
                        ++ka;
                    } else {
                        ++kb;
                    }
                }
            }
            if (neq(type, r, init)) out_push(&out, j, r);
        }
        Rp[i + 1] = (uint32_t) out.size;
    }
    *Rj = out.j;
    *Rx = out.x;
}

/* reference src/cpu/cpu_v_assign.hpp:104-132 (dense mask) and :68-102 (sparse mask, stored order) */
void oracle_v_assign_dense(int type, uint32_t n, uint32_t* r, const uint32_t* mask, uint32_t value, int op_assign, int op_select) {
    const binfn assign = k_bin[type][op_assign];
    const selfn sel    = k_sel[type][op_select];
    for (uint32_t i = 0; i < n; ++i)
        if (sel(mask[i])) r[i] = assign(r[i], value);
}
void oracle_v_assign_sparse(int type, uint32_t nvals, uint32_t* r, const uint32_t* mi, const uint32_t* mx, uint32_t value, int op_assign, int op_select) {
    const binfn assign = k_bin[type][op_assign];
    const selfn sel    = k_sel[type][op_select];
    for (uint32_t k = 0; k < nvals; ++k)
        if (sel(mx[k])) r[mi[k]] = assign(r[mi[k]], value);
}

/* reference src/cpu/cpu_v_eadd_fdb.hpp:107-139 (dense) and :71-105 (sparse) */
void oracle_v_eadd_fdb_dense(int type, uint32_t n, uint32_t* r, const uint32_t* v, uint32_t* fdb, uint32_t fdb_fill, int op) {
    const binfn f = k_bin[type][op];
    for (uint32_t i = 0; i < n; ++i) {
        const word prev = r[i];
        r[i]            = f(prev, v[i]);
        fdb[i]          = neq(type, prev, r[i]) ? r[i] : fdb_fill;
    }
}
uint32_t oracle_v_eadd_fdb_sparse(int type, uint32_t nvals, uint32_t* r, const uint32_t* vi, const uint32_t* vx, uint32_t* fi, uint32_t* fx, int op) {
    const binfn f     = k_bin[type][op];
    uint32_t    count = 0;
    for (uint32_t k = 0; k < nvals; ++k) {
        const uint32_t i    = vi[k];
        const word     prev = r[i];
        r[i]                = f(prev, vx[k]);
        if (neq(type, prev, r[i])) {
            fi[count] = i;
            fx[count] = r[i];
            ++count;
        }
    }
    return count;
}

/* reference src/cpu/cpu_v_count_mf.hpp:99-108 */
uint32_t oracle_v_count_mf_dense(int type, uint32_t n, const uint32_t* v, uint32_t fill) {
    uint32_t count = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (neq(type, v[i], fill)) ++count;
    return count;
}

/* reference src/cpu/cpu_v_eadd.hpp:133-155 (dense x dense) */
void oracle_v_eadd_dense(int type, uint32_t n, uint32_t* r, const uint32_t* u, const uint32_t* v, int op) {
    const binfn f = k_bin[type][op];
    for (uint32_t i = 0; i < n; ++i) r[i] = f(u[i], v[i]);
}
/* reference src/cpu/cpu_v_eadd.hpp:70-131 (sparse x sparse): the two-pointer merge of two index-sorted lists; an index held by
 * one side only meets the other side's fill value.  ri / rx need room for nu + nv entries; returns the entry count. */
uint32_t oracle_v_eadd_sparse(int type, uint32_t nu, const uint32_t* ui, const uint32_t* ux, uint32_t u_fill, uint32_t nv, const uint32_t* vi,
                              const uint32_t* vx, uint32_t v_fill, uint32_t* ri, uint32_t* rx, int op) {
    const binfn f = k_bin[type][op];
    uint32_t    a = 0, b = 0, n = 0;
    while (a < nu && b < nv) {
        if (ui[a] < vi[b]) {
            ri[n] = ui[a];
            rx[n] = f(ux[a], v_fill);
            ++a;
        } else if (vi[b] < ui[a]) {
            ri[n] = vi[b];
            rx[n] = f(u_fill, vx[b]);
            ++b;
        } else {
            ri[n] = ui[a];
            rx[n] = f(ux[a], vx[b]);
            ++a;
            ++b;
        }
        ++n;
    }
    for (; a < nu; ++a, ++n) {
        ri[n] = ui[a];
        rx[n] = f(ux[a], v_fill);
    }
    for (; b < nv; ++b, ++n) {
        ri[n] = vi[b];
        rx[n] = f(u_fill, vx[b]);
    }
    return n;
}
/* reference src/cpu/cpu_v_reduce.hpp:69-117 and src/cpu/cpu_m_reduce.hpp:134-156: strict left fold of the stored values from s */
uint32_t oracle_reduce(int type, uint32_t n, const uint32_t* x, uint32_t s, int op) {
    const binfn f   = k_bin[type][op];
    word        sum = s;
    for (uint32_t i = 0; i < n; ++i) sum = f(sum, x[i]);
    return sum;
}

/* reference src/cpu/cpu_m_transpose.hpp:141-185 (execute_csr): count per column, exclusive scan, then a row-major walk that drops
 * every entry at its column's next free slot — so a row of R lists its entries by ascending source row */
void oracle_m_transpose(int type, uint32_t n_rows, uint32_t n_cols, const uint32_t* Ap, const uint32_t* Aj, const uint32_t* Ax, uint32_t* Rp,
                        uint32_t* Rj, uint32_t* Rx, int op_apply) {
    const unfn f      = k_un[type][op_apply];
    uint32_t*  offset = (uint32_t*) calloc((size_t) n_cols + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n_rows; ++i)
        for (uint32_t k = Ap[i]; k < Ap[i + 1]; ++k) offset[Aj[k]] += 1;
    uint32_t sum = 0;
    for (uint32_t j = 0; j < n_cols; ++j) {
        Rp[j] = sum;
        sum += offset[j];
        offset[j] = 0;
    }
    Rp[n_cols] = sum;
    for (uint32_t i = 0; i < n_rows; ++i)
        for (uint32_t k = Ap[i]; k < Ap[i + 1]; ++k) {
            const uint32_t j = Aj[k], at = Rp[j] + offset[j]++;
            Rj[at] = i;
            Rx[at] = f(Ax[k]);
        }
    free(offset);
}
/* reference src/cpu/cpu_v_map.hpp:70-121: out[i] = op(x[i]) over the stored values (sparse) or every index (dense) */
void oracle_map(int type, uint32_t n, const uint32_t* x, uint32_t* out, int op) {
    const unfn f = k_un[type][op];
    for (uint32_t i = 0; i < n; ++i) out[i] = f(x[i]);
}
