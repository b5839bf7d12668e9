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
typedef struct spla_RefCnt_t*       spla_RefCnt;
typedef struct spla_MemView_t*      spla_MemView;
typedef struct spla_Object_t*       spla_Object;
typedef struct spla_Type_t*         spla_Type;
typedef struct spla_Descriptor_t*   spla_Descriptor;
typedef struct spla_Matrix_t*       spla_Matrix;
typedef struct spla_Vector_t*       spla_Vector;
typedef struct spla_Scalar_t*       spla_Scalar;
typedef struct spla_ScheduleTask_t* spla_ScheduleTask;
typedef struct spla_OpBinary_t*     spla_OpBinary;
typedef struct spla_OpSelect_t*     spla_OpSelect;

typedef void(spla_MessageCallback)(spla_Status, const char* message, const char* file, const char* function, int line, void* p_user_data);

/* ---- library (reference include/spla.h:136-144) ---- */
SPLA_API void        spla_Library_finalize(void);
SPLA_API spla_Status spla_Library_initialize(void);
SPLA_API spla_Status spla_Library_set_accelerator(spla_AcceleratorType accelerator);
SPLA_API spla_Status spla_Library_set_platform(int index);
SPLA_API spla_Status spla_Library_set_device(int index);
SPLA_API spla_Status spla_Library_set_queues_count(int count);
SPLA_API spla_Status spla_Library_set_message_callback(spla_MessageCallback callback, void* p_user_data);
SPLA_API spla_Status spla_Library_set_default_callback(void);
SPLA_API spla_Status spla_Library_get_accelerator_info(char* buffer, int length);

/* ---- element types (reference include/spla.h:150-153) ---- */
SPLA_API spla_Type spla_Type_BOOL(void);
SPLA_API spla_Type spla_Type_INT(void);
SPLA_API spla_Type spla_Type_UINT(void);
SPLA_API spla_Type spla_Type_FLOAT(void);

/* ---- built-in binary ops (reference include/spla.h:197-241): spla_OpBinary_<OP>_{INT,UINT,FLOAT};
 *      the bitwise ops exist for INT and UINT only ---- */
#define SPLA_C_BINARY_OPS_ARITH(X) X(PLUS) X(MINUS) X(MULT) X(DIV) X(MINUS_POW2) X(FIRST) X(SECOND) X(BONE) X(MIN) X(MAX) X(LOR) X(LAND)
#define SPLA_C_BINARY_OPS_BITWISE(X) X(BOR) X(BAND) X(BXOR)
#define SPLA_C_SELECT_OPS(X) X(EQZERO) X(NQZERO) X(GTZERO) X(GEZERO) X(LTZERO) X(LEZERO) X(ALWAYS) X(NEVER)

#define SPLA_C_DECL_BIN3(OP)                                \
    SPLA_API spla_OpBinary spla_OpBinary_##OP##_INT(void);  \
    SPLA_API spla_OpBinary spla_OpBinary_##OP##_UINT(void); \
    SPLA_API spla_OpBinary spla_OpBinary_##OP##_FLOAT(void);
#define SPLA_C_DECL_BIN2(OP)                                \
    SPLA_API spla_OpBinary spla_OpBinary_##OP##_INT(void);  \
    SPLA_API spla_OpBinary spla_OpBinary_##OP##_UINT(void);
#define SPLA_C_DECL_SEL3(OP)                                \
    SPLA_API spla_OpSelect spla_OpSelect_##OP##_INT(void);  \
    SPLA_API spla_OpSelect spla_OpSelect_##OP##_UINT(void); \
    SPLA_API spla_OpSelect spla_OpSelect_##OP##_FLOAT(void);
SPLA_C_BINARY_OPS_ARITH(SPLA_C_DECL_BIN3)
SPLA_C_BINARY_OPS_BITWISE(SPLA_C_DECL_BIN2)
/* ---- built-in select ops (reference include/spla.h:244-267) ---- */
SPLA_C_SELECT_OPS(SPLA_C_DECL_SEL3)
#undef SPLA_C_DECL_BIN3
#undef SPLA_C_DECL_BIN2
#undef SPLA_C_DECL_SEL3

/* ---- reference counting (reference include/spla.h:273-274) ---- */
SPLA_API spla_Status spla_RefCnt_ref(spla_RefCnt object);
SPLA_API spla_Status spla_RefCnt_unref(spla_RefCnt object);

/* ---- memory views (reference include/spla.h:284-289) ---- */
SPLA_API spla_Status spla_MemView_make(spla_MemView* view, void* buffer, spla_size_t size, spla_bool is_mutable);
SPLA_API spla_Status spla_MemView_read(spla_MemView view, spla_size_t offset, spla_size_t size, void* dst);
SPLA_API spla_Status spla_MemView_write(spla_MemView view, spla_size_t offset, spla_size_t size, const void* src);
SPLA_API spla_Status spla_MemView_get_buffer(spla_MemView view, void** buffer);
SPLA_API spla_Status spla_MemView_get_size(spla_MemView view, spla_size_t* size);
SPLA_API spla_Status spla_MemView_is_mutable(spla_MemView view, spla_bool* is_mutable);

/* ---- scalars (reference include/spla.h:295-301) ---- */
SPLA_API spla_Status spla_Scalar_make(spla_Scalar* scalar, spla_Type type);
SPLA_API spla_Status spla_Scalar_set_int(spla_Scalar s, int value);
SPLA_API spla_Status spla_Scalar_set_uint(spla_Scalar s, unsigned int value);
SPLA_API spla_Status spla_Scalar_set_float(spla_Scalar s, float value);
SPLA_API spla_Status spla_Scalar_get_int(spla_Scalar s, int* value);
SPLA_API spla_Status spla_Scalar_get_uint(spla_Scalar s, unsigned int* value);
SPLA_API spla_Status spla_Scalar_get_float(spla_Scalar s, float* value);

/* ---- vectors (reference include/spla.h:324-336) ---- */
SPLA_API spla_Status spla_Vector_make(spla_Vector* v, spla_uint n_rows, spla_Type type);
SPLA_API spla_Status spla_Vector_set_format(spla_Vector v, int format);
SPLA_API spla_Status spla_Vector_set_fill_value(spla_Vector v, spla_Scalar value);
SPLA_API spla_Status spla_Vector_set_reduce(spla_Vector v, spla_OpBinary reduce);
SPLA_API spla_Status spla_Vector_set_int(spla_Vector v, spla_uint row_id, int value);
SPLA_API spla_Status spla_Vector_set_uint(spla_Vector v, spla_uint row_id, unsigned int value);
SPLA_API spla_Status spla_Vector_set_float(spla_Vector v, spla_uint row_id, float value);
SPLA_API spla_Status spla_Vector_get_int(spla_Vector v, spla_uint row_id, int* value);
SPLA_API spla_Status spla_Vector_get_uint(spla_Vector v, spla_uint row_id, unsigned int* value);
SPLA_API spla_Status spla_Vector_get_float(spla_Vector v, spla_uint row_id, float* value);
SPLA_API spla_Status spla_Vector_build(spla_Vector v, spla_MemView keys, spla_MemView values);
SPLA_API spla_Status spla_Vector_read(spla_Vector v, spla_MemView* keys, spla_MemView* values);
SPLA_API spla_Status spla_Vector_clear(spla_Vector v);

/* ---- matrices (reference include/spla.h:342-354) ---- */
SPLA_API spla_Status spla_Matrix_make(spla_Matrix* M, spla_uint n_rows, spla_uint n_cols, spla_Type type);
SPLA_API spla_Status spla_Matrix_set_format(spla_Matrix M, int format);
SPLA_API spla_Status spla_Matrix_set_fill_value(spla_Matrix M, spla_Scalar value);
SPLA_API spla_Status spla_Matrix_set_reduce(spla_Matrix M, spla_OpBinary reduce);
SPLA_API spla_Status spla_Matrix_set_int(spla_Matrix M, spla_uint row_id, spla_uint col_id, int value);
SPLA_API spla_Status spla_Matrix_set_uint(spla_Matrix M, spla_uint row_id, spla_uint col_id, unsigned int value);
SPLA_API spla_Status spla_Matrix_set_float(spla_Matrix M, spla_uint row_id, spla_uint col_id, float value);
SPLA_API spla_Status spla_Matrix_get_int(spla_Matrix M, spla_uint row_id, spla_uint col_id, int* value);
SPLA_API spla_Status spla_Matrix_get_uint(spla_Matrix M, spla_uint row_id, spla_uint col_id, unsigned int* value);
SPLA_API spla_Status spla_Matrix_get_float(spla_Matrix M, spla_uint row_id, spla_uint col_id, float* value);
SPLA_API spla_Status spla_Matrix_build(spla_Matrix M, spla_MemView keys1, spla_MemView keys2, spla_MemView values);
SPLA_API spla_Status spla_Matrix_read(spla_Matrix M, spla_MemView* keys1, spla_MemView* keys2, spla_MemView* values);
SPLA_API spla_Status spla_Matrix_clear(spla_Matrix M);

/* ---- the hot path (reference include/spla.h:369-373) ---- */
SPLA_API spla_Status spla_Exec_mxm(spla_Matrix R, spla_Matrix A, spla_Matrix B, spla_OpBinary op_multiply, spla_OpBinary op_add, spla_Scalar init, spla_Descriptor desc, spla_ScheduleTask* task);
SPLA_API spla_Status spla_Exec_mxmT_masked(spla_Matrix R, spla_Matrix mask, spla_Matrix A, spla_Matrix B, spla_OpBinary op_multiply, spla_OpBinary op_add, spla_OpSelect op_select, spla_Scalar init, spla_Descriptor desc, spla_ScheduleTask* task);
SPLA_API spla_Status spla_Exec_mxv_masked(spla_Vector r, spla_Vector mask, spla_Matrix M, spla_Vector v, spla_OpBinary op_multiply, spla_OpBinary op_add, spla_OpSelect op_select, spla_Scalar init, spla_Descriptor desc, spla_ScheduleTask* task);
SPLA_API spla_Status spla_Exec_vxm_masked(spla_Vector r, spla_Vector mask, spla_Vector v, spla_Matrix M, spla_OpBinary op_multiply, spla_OpBinary op_add, spla_OpSelect op_select, spla_Scalar init, spla_Descriptor desc, spla_ScheduleTask* task);

/* ---- the vector operations of a traversal step (reference include/spla.h: spla_Exec_v_eadd_fdb, spla_Exec_v_assign_masked,
 *      spla_Exec_v_count_mf) ---- */
SPLA_API spla_Status spla_Exec_v_eadd_fdb(spla_Vector r, spla_Vector v, spla_Vector fdb, spla_OpBinary op, spla_Descriptor desc, spla_ScheduleTask* task);
SPLA_API spla_Status spla_Exec_v_assign_masked(spla_Vector r, spla_Vector mask, spla_Scalar value, spla_OpBinary op_assign, spla_OpSelect op_select, spla_Descriptor desc, spla_ScheduleTask* task);
SPLA_API spla_Status spla_Exec_v_count_mf(spla_Scalar r, spla_Vector v, spla_Descriptor desc, spla_ScheduleTask* task);

/* ---- PageRank step remainder and triangle-count tail (reference include/spla.h:378,382,387) ---- */
SPLA_API spla_Status spla_Exec_m_reduce(spla_Scalar r, spla_Scalar s, spla_Matrix M, spla_OpBinary op_reduce, spla_Descriptor desc, spla_ScheduleTask* task);
SPLA_API spla_Status spla_Exec_v_eadd(spla_Vector r, spla_Vector u, spla_Vector v, spla_OpBinary op, spla_Descriptor desc, spla_ScheduleTask* task);
SPLA_API spla_Status spla_Exec_v_reduce(spla_Scalar r, spla_Scalar s, spla_Vector v, spla_OpBinary op_reduce, spla_Descriptor desc, spla_ScheduleTask* task);

/* ---- graph algorithms (reference include/spla.h:360-363); descriptor may be NULL = defaults (push-pull, front factor 0.1) ---- */
SPLA_API spla_Status spla_Algorithm_bfs(spla_Vector v, spla_Matrix A, spla_uint s, spla_Descriptor descriptor);
SPLA_API spla_Status spla_Algorithm_sssp(spla_Vector v, spla_Matrix A, spla_uint s, spla_Descriptor descriptor);
SPLA_API spla_Status spla_Algorithm_pr(spla_Vector* p, spla_Matrix A, float alpha, float eps, spla_Descriptor descriptor);
SPLA_API spla_Status spla_Algorithm_tc(int* ntrins, spla_Matrix A, spla_Matrix B, spla_Descriptor descriptor);

/* ---- extension: the MatrixMarket loader from C (the reference has MtxLoader in C++ only, include/spla/io.hpp:47-85) ----
 * load: parse + symmetrise / drop loops / shift indices as asked, sort, de-duplicate; get: sizes and the two index arrays (owned by
 * the loader, valid until delete). */
typedef struct spla_MtxLoader_t* spla_MtxLoader;
SPLA_API spla_Status spla_MtxLoader_load(spla_MtxLoader* loader, const char* file_path, spla_bool offset_indices, spla_bool make_undirected, spla_bool remove_loops);
SPLA_API spla_Status spla_MtxLoader_get(spla_MtxLoader loader, spla_uint* n_rows, spla_uint* n_cols, spla_size_t* n_values, const spla_uint** Ai, const spla_uint** Aj);
SPLA_API spla_Status spla_MtxLoader_delete(spla_MtxLoader loader);

/* ---- extension: descriptors from C ---- */
SPLA_API spla_Status spla_Descriptor_make(spla_Descriptor* desc);
SPLA_API spla_Status spla_Descriptor_set_early_exit(spla_Descriptor desc, spla_bool value);
SPLA_API spla_Status spla_Descriptor_set_struct_only(spla_Descriptor desc, spla_bool value);
SPLA_API spla_Status spla_Descriptor_set_traversal_mode(spla_Descriptor desc, int mode); /* 0 push, 1 pull, 2 push-pull */
SPLA_API spla_Status spla_Descriptor_set_front_factor(spla_Descriptor desc, float value);

#if defined(__cplusplus)
}
#endif

#endif /* SPLA_AMD_SPLA_C_H */
