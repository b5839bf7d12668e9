/*
 * spla.h — C ABI of the MI355X backend build: the hot-path subset of the reference's C bindings (reference
 * include/spla.h; implementation src/binding/c_*.cpp), same symbol names, argument order and status codes, so a
 * binding written against the reference for this path (e.g. python/pyspla/bridge.py:162-171 via ctypes) loads this
 * library instead.  Written from scratch; citations give the reference declaration each group replaces.
 *
 * Ownership (reference src/binding/c_matrix.cpp:30-34, c_config.hpp:42-45): every *_make returns a handle that owns
 * ONE reference — release it with spla_RefCnt_unref; arguments are borrowed for the duration of the call; type and op
 * getters return library-owned objects (never unref them).  spla_*_read returns non-owning views into the object,
 * valid until the object is next mutated.  With `task != NULL` an spla_Exec_* call only creates the task.
 *
 * Not in the reference (marked "extension"): SPLA_ACCELERATOR_TYPE_HIP, and the three spla_Descriptor_* calls — the
 * reference's C ABI has no way to build a Descriptor (reference include/spla.h:119), so early_exit was unreachable from C.
 */
#ifndef SPLA_AMD_SPLA_C_H
#define SPLA_AMD_SPLA_C_H

#include <stddef.h>
#include <stdint.h>

#define SPLA_API

#if defined(__cplusplus)
extern "C" {
#endif

/* ---- status and enums (reference include/spla.h:71-111; values are ABI) ---- */
typedef enum spla_Status {
    SPLA_STATUS_OK                 = 0,
    SPLA_STATUS_ERROR              = 1,
    SPLA_STATUS_NO_ACCELERATION    = 2,
    SPLA_STATUS_PLATFORM_NOT_FOUND = 3,
    SPLA_STATUS_DEVICE_NOT_FOUND   = 4,
    SPLA_STATUS_INVALID_STATE      = 5,
    SPLA_STATUS_INVALID_ARGUMENT   = 6,
    SPLA_STATUS_NO_VALUE           = 7,
    SPLA_STATUS_NOT_IMPLEMENTED    = 1024
} spla_Status;

typedef enum spla_AcceleratorType {
    SPLA_ACCELERATOR_TYPE_NONE   = 0,
    SPLA_ACCELERATOR_TYPE_OPENCL = 1, /* reference value; unavailable here */
    SPLA_ACCELERATOR_TYPE_HIP    = 2  /* extension: the MI355X backend */
} spla_AcceleratorType;

typedef enum spla_FormatMatrix {
    SPLA_FORMAT_MATRIX_CPU_LIL = 0, SPLA_FORMAT_MATRIX_CPU_DOK = 1, SPLA_FORMAT_MATRIX_CPU_COO = 2, SPLA_FORMAT_MATRIX_CPU_CSR = 3,
    SPLA_FORMAT_MATRIX_CPU_CSC = 4, SPLA_FORMAT_MATRIX_ACC_COO = 5, SPLA_FORMAT_MATRIX_ACC_CSR = 6, SPLA_FORMAT_MATRIX_ACC_CSC = 7,
    SPLA_FORMAT_MATRIX_COUNT   = 8
} spla_FormatMatrix;

typedef enum spla_FormatVector {
    SPLA_FORMAT_VECTOR_CPU_DOK = 0, SPLA_FORMAT_VECTOR_CPU_DENSE = 1, SPLA_FORMAT_VECTOR_CPU_COO = 2,
    SPLA_FORMAT_VECTOR_ACC_DENSE = 3, SPLA_FORMAT_VECTOR_ACC_COO = 4, SPLA_FORMAT_VECTOR_COUNT = 5
} spla_FormatVector;

#define SPLA_NULL NULL

typedef int32_t  spla_bool;
typedef uint32_t spla_uint;
typedef size_t   spla_size_t;

/* opaque handles: raw pointers to intrusively counted objects (reference include/spla.h:119-132) */
#define SPLA_C_HANDLES(X) X(RefCnt) X(MemView) X(Object) X(Type) X(Descriptor) X(Matrix) X(Vector) X(Scalar) X(ScheduleTask) X(OpBinary) X(OpSelect) X(OpUnary) X(MtxLoader)
#define SPLA_C_DECL_HANDLE(Name) typedef struct spla_##Name##_t* spla_##Name;
SPLA_C_HANDLES(SPLA_C_DECL_HANDLE)
#undef SPLA_C_DECL_HANDLE

typedef void(spla_MessageCallback)(spla_Status status, const char* text, const char* file, const char* function, int line, void* user);

/* the three element types of the path, as (accessor suffix, C type): spla_*_set_<suffix> / spla_*_get_<suffix> */
#define SPLA_C_ELEMENT_TYPES(X, Obj, ...) X(Obj, int, int, __VA_ARGS__) X(Obj, uint, unsigned int, __VA_ARGS__) X(Obj, float, float, __VA_ARGS__)

/* ---- library (reference include/spla.h:136-144) ---- */
SPLA_API void        spla_Library_finalize(void);
SPLA_API spla_Status spla_Library_initialize(void);
SPLA_API spla_Status spla_Library_set_accelerator(spla_AcceleratorType kind);
SPLA_API spla_Status spla_Library_set_platform(int platform_index);
SPLA_API spla_Status spla_Library_set_device(int device_index);
SPLA_API spla_Status spla_Library_set_queues_count(int n_queues);
SPLA_API spla_Status spla_Library_set_message_callback(spla_MessageCallback on_message, void* user);
SPLA_API spla_Status spla_Library_set_default_callback(void);
SPLA_API spla_Status spla_Library_get_accelerator_info(char* text, int capacity);

/* ---- element types (reference include/spla.h:150-153) ---- */
#define SPLA_C_DECL_TYPE(NAME) SPLA_API spla_Type spla_Type_##NAME(void);
SPLA_C_DECL_TYPE(BOOL) SPLA_C_DECL_TYPE(INT) SPLA_C_DECL_TYPE(UINT) SPLA_C_DECL_TYPE(FLOAT)
#undef SPLA_C_DECL_TYPE

/* ---- built-in ops (reference include/spla.h:197-267): spla_OpBinary_<OP>_{INT,UINT,FLOAT} (the bitwise ones: INT and UINT only)
 *      and spla_OpSelect_<OP>_{INT,UINT,FLOAT}; library-owned objects ---- */
#define SPLA_C_BINARY_OPS_ARITH(X) X(PLUS) X(MINUS) X(MULT) X(DIV) X(MINUS_POW2) X(FIRST) X(SECOND) X(BONE) X(MIN) X(MAX) X(LOR) X(LAND)
#define SPLA_C_BINARY_OPS_BITWISE(X) X(BOR) X(BAND) X(BXOR)
#define SPLA_C_SELECT_OPS(X) X(EQZERO) X(NQZERO) X(GTZERO) X(GEZERO) X(LTZERO) X(LEZERO) X(ALWAYS) X(NEVER)
#define SPLA_C_GETTER(Kind, OP, T) SPLA_API spla_Op##Kind spla_Op##Kind##_##OP##_##T(void);
#define SPLA_C_DECL_BIN3(OP) SPLA_C_GETTER(Binary, OP, INT) SPLA_C_GETTER(Binary, OP, UINT) SPLA_C_GETTER(Binary, OP, FLOAT)
#define SPLA_C_DECL_BIN2(OP) SPLA_C_GETTER(Binary, OP, INT) SPLA_C_GETTER(Binary, OP, UINT)
#define SPLA_C_DECL_SEL3(OP) SPLA_C_GETTER(Select, OP, INT) SPLA_C_GETTER(Select, OP, UINT) SPLA_C_GETTER(Select, OP, FLOAT)
/* unary ops (reference include/spla.h:159-191): spla_OpUnary_<OP>_{INT,UINT,FLOAT}; BNOT has no FLOAT form, SQRT .. TRUNC only that */
#define SPLA_C_UNARY_OPS_ALL(X) X(IDENTITY) X(AINV) X(MINV) X(LNOT) X(UONE) X(ABS)
#define SPLA_C_UNARY_OPS_FLOAT(X) X(SQRT) X(LOG) X(EXP) X(SIN) X(COS) X(TAN) X(ASIN) X(ACOS) X(ATAN) X(CEIL) X(FLOOR) X(ROUND) X(TRUNC)
#define SPLA_C_DECL_UN3(OP) SPLA_C_GETTER(Unary, OP, INT) SPLA_C_GETTER(Unary, OP, UINT) SPLA_C_GETTER(Unary, OP, FLOAT)
#define SPLA_C_DECL_UN1(OP) SPLA_C_GETTER(Unary, OP, FLOAT)
SPLA_C_UNARY_OPS_ALL(SPLA_C_DECL_UN3)
SPLA_C_GETTER(Unary, BNOT, INT) SPLA_C_GETTER(Unary, BNOT, UINT)
SPLA_C_UNARY_OPS_FLOAT(SPLA_C_DECL_UN1)
#undef SPLA_C_DECL_UN3
#undef SPLA_C_DECL_UN1
SPLA_C_BINARY_OPS_ARITH(SPLA_C_DECL_BIN3)
SPLA_C_BINARY_OPS_BITWISE(SPLA_C_DECL_BIN2)
SPLA_C_SELECT_OPS(SPLA_C_DECL_SEL3)
#undef SPLA_C_DECL_BIN3
#undef SPLA_C_DECL_BIN2
#undef SPLA_C_DECL_SEL3
#undef SPLA_C_GETTER

/* ---- reference counting (reference include/spla.h:273-274) ---- */
SPLA_API spla_Status spla_RefCnt_ref(spla_RefCnt handle);
SPLA_API spla_Status spla_RefCnt_unref(spla_RefCnt handle);

/* ---- memory views over caller memory (reference include/spla.h:284-289) ---- */
SPLA_API spla_Status spla_MemView_make(spla_MemView* out, void* memory, spla_size_t n_bytes, spla_bool writable);
SPLA_API spla_Status spla_MemView_read(spla_MemView view, spla_size_t at, spla_size_t n_bytes, void* into);
SPLA_API spla_Status spla_MemView_write(spla_MemView view, spla_size_t at, spla_size_t n_bytes, const void* from);
SPLA_API spla_Status spla_MemView_get_buffer(spla_MemView view, void** memory);
SPLA_API spla_Status spla_MemView_get_size(spla_MemView view, spla_size_t* n_bytes);
SPLA_API spla_Status spla_MemView_is_mutable(spla_MemView view, spla_bool* writable);

/* ---- scalars (reference include/spla.h:295-301) ---- */
SPLA_API spla_Status spla_Scalar_make(spla_Scalar* out, spla_Type type);
#define SPLA_C_SCALAR_ACCESS(Obj, sfx, ctype, ...)                          \
    SPLA_API spla_Status spla_Scalar_set_##sfx(spla_Scalar scalar, ctype x); \
    SPLA_API spla_Status spla_Scalar_get_##sfx(spla_Scalar scalar, ctype* x);
SPLA_C_ELEMENT_TYPES(SPLA_C_SCALAR_ACCESS, Scalar, 0)
#undef SPLA_C_SCALAR_ACCESS

/* ---- vectors (reference include/spla.h:324-336) ---- */
SPLA_API spla_Status spla_Vector_make(spla_Vector* out, spla_uint n_rows, spla_Type type);
SPLA_API spla_Status spla_Vector_set_format(spla_Vector vec, int format /* spla_FormatVector */);
SPLA_API spla_Status spla_Vector_set_fill_value(spla_Vector vec, spla_Scalar fill);
SPLA_API spla_Status spla_Vector_set_reduce(spla_Vector vec, spla_OpBinary on_duplicate);
#define SPLA_C_VECTOR_ACCESS(Obj, sfx, ctype, ...)                                     \
    SPLA_API spla_Status spla_Vector_set_##sfx(spla_Vector vec, spla_uint i, ctype x);  \
    SPLA_API spla_Status spla_Vector_get_##sfx(spla_Vector vec, spla_uint i, ctype* x);
SPLA_C_ELEMENT_TYPES(SPLA_C_VECTOR_ACCESS, Vector, 0)
#undef SPLA_C_VECTOR_ACCESS
SPLA_API spla_Status spla_Vector_build(spla_Vector vec, spla_MemView indices, spla_MemView values);
SPLA_API spla_Status spla_Vector_read(spla_Vector vec, spla_MemView* indices, spla_MemView* values);
SPLA_API spla_Status spla_Vector_clear(spla_Vector vec);

/* ---- matrices (reference include/spla.h:342-354) ---- */
SPLA_API spla_Status spla_Matrix_make(spla_Matrix* out, spla_uint n_rows, spla_uint n_cols, spla_Type type);
SPLA_API spla_Status spla_Matrix_set_format(spla_Matrix mat, int format /* spla_FormatMatrix */);
SPLA_API spla_Status spla_Matrix_set_fill_value(spla_Matrix mat, spla_Scalar fill);
SPLA_API spla_Status spla_Matrix_set_reduce(spla_Matrix mat, spla_OpBinary on_duplicate);
#define SPLA_C_MATRIX_ACCESS(Obj, sfx, ctype, ...)                                                  \
    SPLA_API spla_Status spla_Matrix_set_##sfx(spla_Matrix mat, spla_uint i, spla_uint j, ctype x);  \
    SPLA_API spla_Status spla_Matrix_get_##sfx(spla_Matrix mat, spla_uint i, spla_uint j, ctype* x);
SPLA_C_ELEMENT_TYPES(SPLA_C_MATRIX_ACCESS, Matrix, 0)
#undef SPLA_C_MATRIX_ACCESS
SPLA_API spla_Status spla_Matrix_build(spla_Matrix mat, spla_MemView rows, spla_MemView cols, spla_MemView values);
SPLA_API spla_Status spla_Matrix_read(spla_Matrix mat, spla_MemView* rows, spla_MemView* cols, spla_MemView* values);
SPLA_API spla_Status spla_Matrix_clear(spla_Matrix mat);

/* Every spla_Exec_* call ends with the same two arguments (reference include/spla.h:369-388): an optional descriptor and, when
 * `task_out` is not NULL, the request to only create the task object and hand it back. */
#define SPLA_C_EXEC_TAIL spla_Descriptor options, spla_ScheduleTask* task_out

/* ---- the hot path (reference include/spla.h:369-373) ---- */
SPLA_API spla_Status spla_Exec_mxm(spla_Matrix out, spla_Matrix lhs, spla_Matrix rhs,
                                   spla_OpBinary mul, spla_OpBinary add, spla_Scalar init, SPLA_C_EXEC_TAIL);
SPLA_API spla_Status spla_Exec_mxmT_masked(spla_Matrix out, spla_Matrix mask, spla_Matrix lhs, spla_Matrix rhs_transposed,
                                           spla_OpBinary mul, spla_OpBinary add, spla_OpSelect keep, spla_Scalar init, SPLA_C_EXEC_TAIL);
SPLA_API spla_Status spla_Exec_mxv_masked(spla_Vector out, spla_Vector mask, spla_Matrix mat, spla_Vector vec,
                                          spla_OpBinary mul, spla_OpBinary add, spla_OpSelect keep, spla_Scalar init, SPLA_C_EXEC_TAIL);
SPLA_API spla_Status spla_Exec_vxm_masked(spla_Vector out, spla_Vector mask, spla_Vector vec, spla_Matrix mat,
                                          spla_OpBinary mul, spla_OpBinary add, spla_OpSelect keep, spla_Scalar init, SPLA_C_EXEC_TAIL);

/* ---- the vector operations of a traversal step (reference include/spla.h:384,385,388) ---- */
SPLA_API spla_Status spla_Exec_v_eadd_fdb(spla_Vector inout, spla_Vector operand, spla_Vector changed, spla_OpBinary op, SPLA_C_EXEC_TAIL);
SPLA_API spla_Status spla_Exec_v_assign_masked(spla_Vector inout, spla_Vector mask, spla_Scalar x, spla_OpBinary assign, spla_OpSelect keep,
                                               SPLA_C_EXEC_TAIL);
SPLA_API spla_Status spla_Exec_v_count_mf(spla_Scalar count, spla_Vector vec, SPLA_C_EXEC_TAIL);

/* ---- PageRank step remainder and triangle-count tail (reference include/spla.h:378,382,387) ---- */
SPLA_API spla_Status spla_Exec_m_reduce(spla_Scalar out, spla_Scalar start, spla_Matrix mat, spla_OpBinary op, SPLA_C_EXEC_TAIL);
SPLA_API spla_Status spla_Exec_v_eadd(spla_Vector out, spla_Vector lhs, spla_Vector rhs, spla_OpBinary op, SPLA_C_EXEC_TAIL);
SPLA_API spla_Status spla_Exec_v_reduce(spla_Scalar out, spla_Scalar start, spla_Vector vec, spla_OpBinary op, SPLA_C_EXEC_TAIL);

/* ---- transpose and element-wise map (reference include/spla.h:379,386) ---- */
SPLA_API spla_Status spla_Exec_m_transpose(spla_Matrix out, spla_Matrix mat, spla_OpUnary apply, SPLA_C_EXEC_TAIL);
SPLA_API spla_Status spla_Exec_v_map(spla_Vector out, spla_Vector vec, spla_OpUnary op, SPLA_C_EXEC_TAIL);

/* ---- graph algorithms (reference include/spla.h:360-363); options may be NULL = defaults (push-pull, front factor 0.1) ---- */
SPLA_API spla_Status spla_Algorithm_bfs(spla_Vector levels, spla_Matrix adjacency, spla_uint source, spla_Descriptor options);
SPLA_API spla_Status spla_Algorithm_sssp(spla_Vector distances, spla_Matrix weights, spla_uint source, spla_Descriptor options);
SPLA_API spla_Status spla_Algorithm_pr(spla_Vector* ranks_inout, spla_Matrix scaled_adjacency, float alpha, float eps, spla_Descriptor options);
SPLA_API spla_Status spla_Algorithm_tc(int* n_triangles, spla_Matrix lower, spla_Matrix products_out, spla_Descriptor options);

/* ---- extension: the MatrixMarket loader from C (the reference has MtxLoader in C++ only, include/spla/io.hpp:47-85) ----
 * load: parse + symmetrise / drop loops / shift indices as asked, sort, de-duplicate; get: sizes and the two index arrays (owned by
 * the loader, valid until delete). */
SPLA_API spla_Status spla_MtxLoader_load(spla_MtxLoader* out, const char* path, spla_bool to_zero_based, spla_bool add_reverse_edges, spla_bool drop_self_loops);
SPLA_API spla_Status spla_MtxLoader_get(spla_MtxLoader loader, spla_uint* n_rows, spla_uint* n_cols, spla_size_t* n_edges, const spla_uint** rows, const spla_uint** cols);
SPLA_API spla_Status spla_MtxLoader_delete(spla_MtxLoader loader);

/* ---- extension: descriptors from C ---- */
SPLA_API spla_Status spla_Descriptor_make(spla_Descriptor* out);
SPLA_API spla_Status spla_Descriptor_set_early_exit(spla_Descriptor options, spla_bool on);
SPLA_API spla_Status spla_Descriptor_set_struct_only(spla_Descriptor options, spla_bool on);
/* an addition of this build: fp32 sums of the call on the fixed-order kernels, bit-identical from run to run (spla_hip.h: splahip_set_deterministic) */
SPLA_API spla_Status spla_Descriptor_set_deterministic(spla_Descriptor options, spla_bool on);
/* an addition of this build: fp32 sums of the call folded in the reference CPU backend's order, bit-identical to it (spla_hip.h: splahip_set_reference_order) */
SPLA_API spla_Status spla_Descriptor_set_reference_order(spla_Descriptor options, spla_bool on);
SPLA_API spla_Status spla_Descriptor_set_traversal_mode(spla_Descriptor options, int mode); /* 0 push, 1 pull, 2 push-pull */
SPLA_API spla_Status spla_Descriptor_set_front_factor(spla_Descriptor options, float fraction);

#if defined(__cplusplus)
}
#endif

#endif /* SPLA_AMD_SPLA_C_H */
