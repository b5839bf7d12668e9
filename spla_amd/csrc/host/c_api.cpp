// c_api.cpp — extern "C" bindings of include/spla.h over the C++ API (include/spla.hpp).
// Handle = raw object pointer; *_make hands the caller one reference; arguments are re-wrapped in temporary ref_ptrs
// for the duration of a call (the convention of reference src/binding/c_config.hpp:42-45, c_matrix.cpp:30-34).
#include <spla.h>
#include <spla.hpp>

#include <cstring>

using namespace spla;

namespace {
    template<typename T, typename H>
    ref_ptr<T> as_ref(H handle) { return ref_ptr<T>(reinterpret_cast<T*>(handle)); }
    template<typename T, typename H>
    T* as_ptr(H handle) { return reinterpret_cast<T*>(handle); }
    inline spla_Status to_c(Status s) { return static_cast<spla_Status>(s); }

    template<typename H, typename T>
    spla_Status give(H* out, ref_ptr<T> obj) {
        if (!out) return SPLA_STATUS_INVALID_ARGUMENT;
        *out = reinterpret_cast<H>(obj.release());// the caller now owns this reference (null handle if creation failed)
        return SPLA_STATUS_OK;
    }
}// namespace

extern "C" {

void        spla_Library_finalize(void) { Library::get()->finalize(); }
spla_Status spla_Library_initialize(void) { return Library::get() ? SPLA_STATUS_OK : SPLA_STATUS_ERROR; }
spla_Status spla_Library_set_accelerator(spla_AcceleratorType accelerator) { return to_c(Library::get()->set_accelerator(static_cast<AcceleratorType>(accelerator))); }
spla_Status spla_Library_set_platform(int index) { return to_c(Library::get()->set_platform(index)); }
spla_Status spla_Library_set_device(int index) { return to_c(Library::get()->set_device(index)); }
spla_Status spla_Library_set_queues_count(int count) { return to_c(Library::get()->set_queues_count(count)); }
spla_Status spla_Library_set_message_callback(spla_MessageCallback callback, void* p_user_data) {
    if (!callback) return to_c(Library::get()->set_message_callback(MessageCallback()));
    return to_c(Library::get()->set_message_callback([=](Status st, const std::string& msg, const std::string& file, const std::string& function, int line) {
        callback(to_c(st), msg.c_str(), file.c_str(), function.c_str(), line, p_user_data);
    }));
}
spla_Status spla_Library_set_default_callback(void) { return to_c(Library::get()->set_default_callback()); }
spla_Status spla_Library_get_accelerator_info(char* buffer, int length) {
    if (!buffer || length <= 0) return SPLA_STATUS_INVALID_ARGUMENT;
    std::string  info;
    const Status st = Library::get()->get_accelerator_info(info);
    std::strncpy(buffer, info.c_str(), static_cast<std::size_t>(length));
    buffer[length - 1] = '\0';
    return to_c(st);
}

spla_Type spla_Type_BOOL(void) { return reinterpret_cast<spla_Type>(BOOL.get()); }
spla_Type spla_Type_INT(void) { return reinterpret_cast<spla_Type>(INT.get()); }
spla_Type spla_Type_UINT(void) { return reinterpret_cast<spla_Type>(UINT.get()); }
spla_Type spla_Type_FLOAT(void) { return reinterpret_cast<spla_Type>(FLOAT.get()); }

// op getters: library-owned globals, no reference is transferred
#define SPLA_C_DEF_BIN(OP, TY) \
    spla_OpBinary spla_OpBinary_##OP##_##TY(void) { return (Library::get(), reinterpret_cast<spla_OpBinary>(OP##_##TY.get())); }
#define SPLA_C_DEF_BIN3(OP) SPLA_C_DEF_BIN(OP, INT) SPLA_C_DEF_BIN(OP, UINT) SPLA_C_DEF_BIN(OP, FLOAT)
#define SPLA_C_DEF_BIN2(OP) SPLA_C_DEF_BIN(OP, INT) SPLA_C_DEF_BIN(OP, UINT)
#define SPLA_C_DEF_SEL(OP, TY) \
    spla_OpSelect spla_OpSelect_##OP##_##TY(void) { return (Library::get(), reinterpret_cast<spla_OpSelect>(OP##_##TY.get())); }
#define SPLA_C_DEF_SEL3(OP) SPLA_C_DEF_SEL(OP, INT) SPLA_C_DEF_SEL(OP, UINT) SPLA_C_DEF_SEL(OP, FLOAT)
#define SPLA_C_DEF_UN(OP, TY) \
    spla_OpUnary spla_OpUnary_##OP##_##TY(void) { return (Library::get(), reinterpret_cast<spla_OpUnary>(OP##_##TY.get())); }
#define SPLA_C_DEF_UN3(OP) SPLA_C_DEF_UN(OP, INT) SPLA_C_DEF_UN(OP, UINT) SPLA_C_DEF_UN(OP, FLOAT)
#define SPLA_C_DEF_UN1(OP) SPLA_C_DEF_UN(OP, FLOAT)
SPLA_C_UNARY_OPS_ALL(SPLA_C_DEF_UN3)
SPLA_C_DEF_UN(BNOT, INT) SPLA_C_DEF_UN(BNOT, UINT)
SPLA_C_UNARY_OPS_FLOAT(SPLA_C_DEF_UN1)
SPLA_C_BINARY_OPS_ARITH(SPLA_C_DEF_BIN3)
SPLA_C_BINARY_OPS_BITWISE(SPLA_C_DEF_BIN2)
SPLA_C_SELECT_OPS(SPLA_C_DEF_SEL3)

spla_Status spla_RefCnt_ref(spla_RefCnt object) {
    if (!object) return SPLA_STATUS_INVALID_ARGUMENT;
    as_ptr<RefCnt>(object)->add_ref();
    return SPLA_STATUS_OK;
}
spla_Status spla_RefCnt_unref(spla_RefCnt object) {
    if (!object) return SPLA_STATUS_INVALID_ARGUMENT;
    as_ptr<RefCnt>(object)->rel_ref();
    return SPLA_STATUS_OK;
}

spla_Status spla_MemView_make(spla_MemView* view, void* buffer, spla_size_t size, spla_bool is_mutable) { return give(view, MemView::make(buffer, size, is_mutable != 0)); }
spla_Status spla_MemView_read(spla_MemView view, spla_size_t offset, spla_size_t size, void* dst) { return to_c(as_ptr<MemView>(view)->read(offset, size, dst)); }
spla_Status spla_MemView_write(spla_MemView view, spla_size_t offset, spla_size_t size, const void* src) { return to_c(as_ptr<MemView>(view)->write(offset, size, src)); }
spla_Status spla_MemView_get_buffer(spla_MemView view, void** buffer) { return (*buffer = as_ptr<MemView>(view)->get_buffer(), SPLA_STATUS_OK); }
spla_Status spla_MemView_get_size(spla_MemView view, spla_size_t* size) { return (*size = as_ptr<MemView>(view)->get_size(), SPLA_STATUS_OK); }
spla_Status spla_MemView_is_mutable(spla_MemView view, spla_bool* is_mutable) { return (*is_mutable = as_ptr<MemView>(view)->is_mutable() ? 1 : 0, SPLA_STATUS_OK); }

spla_Status spla_Scalar_make(spla_Scalar* scalar, spla_Type type) { return give(scalar, Scalar::make(as_ref<Type>(type))); }
spla_Status spla_Scalar_set_int(spla_Scalar s, int value) { return to_c(as_ptr<Scalar>(s)->set_int(value)); }
spla_Status spla_Scalar_set_uint(spla_Scalar s, unsigned int value) { return to_c(as_ptr<Scalar>(s)->set_uint(value)); }
spla_Status spla_Scalar_set_float(spla_Scalar s, float value) { return to_c(as_ptr<Scalar>(s)->set_float(value)); }
spla_Status spla_Scalar_get_int(spla_Scalar s, int* value) { return to_c(as_ptr<Scalar>(s)->get_int(*value)); }
spla_Status spla_Scalar_get_uint(spla_Scalar s, unsigned int* value) { return to_c(as_ptr<Scalar>(s)->get_uint(*value)); }
spla_Status spla_Scalar_get_float(spla_Scalar s, float* value) { return to_c(as_ptr<Scalar>(s)->get_float(*value)); }

spla_Status spla_Vector_make(spla_Vector* v, spla_uint n_rows, spla_Type type) { return give(v, Vector::make(n_rows, as_ref<Type>(type))); }
spla_Status spla_Vector_set_format(spla_Vector v, int format) { return to_c(as_ptr<Vector>(v)->set_format(static_cast<FormatVector>(format))); }
spla_Status spla_Vector_set_fill_value(spla_Vector v, spla_Scalar value) { return to_c(as_ptr<Vector>(v)->set_fill_value(as_ref<Scalar>(value))); }
spla_Status spla_Vector_set_reduce(spla_Vector v, spla_OpBinary reduce) { return to_c(as_ptr<Vector>(v)->set_reduce(as_ref<OpBinary>(reduce))); }
spla_Status spla_Vector_set_int(spla_Vector v, spla_uint row_id, int value) { return to_c(as_ptr<Vector>(v)->set_int(row_id, value)); }
spla_Status spla_Vector_set_uint(spla_Vector v, spla_uint row_id, unsigned int value) { return to_c(as_ptr<Vector>(v)->set_uint(row_id, value)); }
spla_Status spla_Vector_set_float(spla_Vector v, spla_uint row_id, float value) { return to_c(as_ptr<Vector>(v)->set_float(row_id, value)); }
spla_Status spla_Vector_get_int(spla_Vector v, spla_uint row_id, int* value) { return to_c(as_ptr<Vector>(v)->get_int(row_id, *value)); }
spla_Status spla_Vector_get_uint(spla_Vector v, spla_uint row_id, unsigned int* value) { return to_c(as_ptr<Vector>(v)->get_uint(row_id, *value)); }
spla_Status spla_Vector_get_float(spla_Vector v, spla_uint row_id, float* value) { return to_c(as_ptr<Vector>(v)->get_float(row_id, *value)); }
spla_Status spla_Vector_build(spla_Vector v, spla_MemView keys, spla_MemView values) { return to_c(as_ptr<Vector>(v)->build(as_ref<MemView>(keys), as_ref<MemView>(values))); }
spla_Status spla_Vector_read(spla_Vector v, spla_MemView* keys, spla_MemView* values) {
    ref_ptr<MemView> k, x;
    const Status     st = as_ptr<Vector>(v)->read(k, x);
    if (st == Status::Ok) {
        *keys   = reinterpret_cast<spla_MemView>(k.release());
        *values = reinterpret_cast<spla_MemView>(x.release());
    }
    return to_c(st);
}
spla_Status spla_Vector_clear(spla_Vector v) { return to_c(as_ptr<Vector>(v)->clear()); }

spla_Status spla_Matrix_make(spla_Matrix* M, spla_uint n_rows, spla_uint n_cols, spla_Type type) { return give(M, Matrix::make(n_rows, n_cols, as_ref<Type>(type))); }
spla_Status spla_Matrix_set_format(spla_Matrix M, int format) { return to_c(as_ptr<Matrix>(M)->set_format(static_cast<FormatMatrix>(format))); }
spla_Status spla_Matrix_set_fill_value(spla_Matrix M, spla_Scalar value) { return to_c(as_ptr<Matrix>(M)->set_fill_value(as_ref<Scalar>(value))); }
spla_Status spla_Matrix_set_reduce(spla_Matrix M, spla_OpBinary reduce) { return to_c(as_ptr<Matrix>(M)->set_reduce(as_ref<OpBinary>(reduce))); }
spla_Status spla_Matrix_set_int(spla_Matrix M, spla_uint row_id, spla_uint col_id, int value) { return to_c(as_ptr<Matrix>(M)->set_int(row_id, col_id, value)); }
spla_Status spla_Matrix_set_uint(spla_Matrix M, spla_uint row_id, spla_uint col_id, unsigned int value) { return to_c(as_ptr<Matrix>(M)->set_uint(row_id, col_id, value)); }
spla_Status spla_Matrix_set_float(spla_Matrix M, spla_uint row_id, spla_uint col_id, float value) { return to_c(as_ptr<Matrix>(M)->set_float(row_id, col_id, value)); }
spla_Status spla_Matrix_get_int(spla_Matrix M, spla_uint row_id, spla_uint col_id, int* value) { return to_c(as_ptr<Matrix>(M)->get_int(row_id, col_id, *value)); }
spla_Status spla_Matrix_get_uint(spla_Matrix M, spla_uint row_id, spla_uint col_id, unsigned int* value) { return to_c(as_ptr<Matrix>(M)->get_uint(row_id, col_id, *value)); }
spla_Status spla_Matrix_get_float(spla_Matrix M, spla_uint row_id, spla_uint col_id, float* value) { return to_c(as_ptr<Matrix>(M)->get_float(row_id, col_id, *value)); }
spla_Status spla_Matrix_build(spla_Matrix M, spla_MemView keys1, spla_MemView keys2, spla_MemView values) { return to_c(as_ptr<Matrix>(M)->build(as_ref<MemView>(keys1), as_ref<MemView>(keys2), as_ref<MemView>(values))); }
spla_Status spla_Matrix_read(spla_Matrix M, spla_MemView* keys1, spla_MemView* keys2, spla_MemView* values) {
    ref_ptr<MemView> k1, k2, x;
    const Status     st = as_ptr<Matrix>(M)->read(k1, k2, x);
    if (st == Status::Ok) {
        *keys1  = reinterpret_cast<spla_MemView>(k1.release());
        *keys2  = reinterpret_cast<spla_MemView>(k2.release());
        *values = reinterpret_cast<spla_MemView>(x.release());
    }
    return to_c(st);
}
spla_Status spla_Matrix_clear(spla_Matrix M) { return to_c(as_ptr<Matrix>(M)->clear()); }

// exec wrappers: with task != NULL only the task object is created and returned (owned by the caller)
#define SPLA_C_EXEC(call)                                                              \
    ref_ptr<ScheduleTask> hnd;                                                         \
    ref_ptr<ScheduleTask>* p_hnd = task ? &hnd : nullptr;                              \
    const Status st = call;                                                            \
    if (task) *task = reinterpret_cast<spla_ScheduleTask>(hnd.release());              \
    return to_c(st);

spla_Status spla_Exec_mxm(spla_Matrix R, spla_Matrix A, spla_Matrix B, spla_OpBinary op_multiply, spla_OpBinary op_add, spla_Scalar init, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_mxm(as_ref<Matrix>(R), as_ref<Matrix>(A), as_ref<Matrix>(B), as_ref<OpBinary>(op_multiply), as_ref<OpBinary>(op_add), as_ref<Scalar>(init), as_ref<Descriptor>(desc), p_hnd))
}
spla_Status spla_Exec_mxmT_masked(spla_Matrix R, spla_Matrix mask, spla_Matrix A, spla_Matrix B, spla_OpBinary op_multiply, spla_OpBinary op_add, spla_OpSelect op_select, spla_Scalar init, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_mxmT_masked(as_ref<Matrix>(R), as_ref<Matrix>(mask), as_ref<Matrix>(A), as_ref<Matrix>(B), as_ref<OpBinary>(op_multiply), as_ref<OpBinary>(op_add), as_ref<OpSelect>(op_select), as_ref<Scalar>(init), as_ref<Descriptor>(desc), p_hnd))
}
spla_Status spla_Exec_mxv_masked(spla_Vector r, spla_Vector mask, spla_Matrix M, spla_Vector v, spla_OpBinary op_multiply, spla_OpBinary op_add, spla_OpSelect op_select, spla_Scalar init, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_mxv_masked(as_ref<Vector>(r), as_ref<Vector>(mask), as_ref<Matrix>(M), as_ref<Vector>(v), as_ref<OpBinary>(op_multiply), as_ref<OpBinary>(op_add), as_ref<OpSelect>(op_select), as_ref<Scalar>(init), as_ref<Descriptor>(desc), p_hnd))
}
spla_Status spla_Exec_vxm_masked(spla_Vector r, spla_Vector mask, spla_Vector v, spla_Matrix M, spla_OpBinary op_multiply, spla_OpBinary op_add, spla_OpSelect op_select, spla_Scalar init, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_vxm_masked(as_ref<Vector>(r), as_ref<Vector>(mask), as_ref<Vector>(v), as_ref<Matrix>(M), as_ref<OpBinary>(op_multiply), as_ref<OpBinary>(op_add), as_ref<OpSelect>(op_select), as_ref<Scalar>(init), as_ref<Descriptor>(desc), p_hnd))
}

spla_Status spla_Exec_v_eadd_fdb(spla_Vector r, spla_Vector v, spla_Vector fdb, spla_OpBinary op, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_v_eadd_fdb(as_ref<Vector>(r), as_ref<Vector>(v), as_ref<Vector>(fdb), as_ref<OpBinary>(op), as_ref<Descriptor>(desc), p_hnd))
}
spla_Status spla_Exec_v_assign_masked(spla_Vector r, spla_Vector mask, spla_Scalar value, spla_OpBinary op_assign, spla_OpSelect op_select, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_v_assign_masked(as_ref<Vector>(r), as_ref<Vector>(mask), as_ref<Scalar>(value), as_ref<OpBinary>(op_assign), as_ref<OpSelect>(op_select), as_ref<Descriptor>(desc), p_hnd))
}
spla_Status spla_Exec_v_count_mf(spla_Scalar r, spla_Vector v, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_v_count_mf(as_ref<Scalar>(r), as_ref<Vector>(v), as_ref<Descriptor>(desc), p_hnd))
}

spla_Status spla_Exec_m_reduce(spla_Scalar r, spla_Scalar s, spla_Matrix M, spla_OpBinary op_reduce, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_m_reduce(as_ref<Scalar>(r), as_ref<Scalar>(s), as_ref<Matrix>(M), as_ref<OpBinary>(op_reduce), as_ref<Descriptor>(desc), p_hnd))
}
spla_Status spla_Exec_m_transpose(spla_Matrix R, spla_Matrix M, spla_OpUnary op_apply, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_m_transpose(as_ref<Matrix>(R), as_ref<Matrix>(M), as_ref<OpUnary>(op_apply), as_ref<Descriptor>(desc), p_hnd))
}
spla_Status spla_Exec_v_map(spla_Vector r, spla_Vector v, spla_OpUnary op, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_v_map(as_ref<Vector>(r), as_ref<Vector>(v), as_ref<OpUnary>(op), as_ref<Descriptor>(desc), p_hnd))
}
spla_Status spla_Exec_v_eadd(spla_Vector r, spla_Vector u, spla_Vector v, spla_OpBinary op, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_v_eadd(as_ref<Vector>(r), as_ref<Vector>(u), as_ref<Vector>(v), as_ref<OpBinary>(op), as_ref<Descriptor>(desc), p_hnd))
}
spla_Status spla_Exec_v_reduce(spla_Scalar r, spla_Scalar s, spla_Vector v, spla_OpBinary op_reduce, spla_Descriptor desc, spla_ScheduleTask* task) {
    SPLA_C_EXEC(exec_v_reduce(as_ref<Scalar>(r), as_ref<Scalar>(s), as_ref<Vector>(v), as_ref<OpBinary>(op_reduce), as_ref<Descriptor>(desc), p_hnd))
}

spla_Status spla_Algorithm_bfs(spla_Vector v, spla_Matrix A, spla_uint s, spla_Descriptor descriptor) {
    return to_c(bfs(as_ref<Vector>(v), as_ref<Matrix>(A), s, as_ref<Descriptor>(descriptor)));
}
spla_Status spla_Algorithm_sssp(spla_Vector v, spla_Matrix A, spla_uint s, spla_Descriptor descriptor) {
    return to_c(sssp(as_ref<Vector>(v), as_ref<Matrix>(A), s, as_ref<Descriptor>(descriptor)));
}
// *p is consumed and replaced: the result may live in a different Vector object than the one passed in (reference src/binding/c_algorithm.cpp:36-48)
spla_Status spla_Algorithm_pr(spla_Vector* p, spla_Matrix A, float alpha, float eps, spla_Descriptor descriptor) {
    if (!p || !*p) return SPLA_STATUS_INVALID_ARGUMENT;
    ref_ptr<Vector> inout = as_ref<Vector>(*p);
    const Status    st    = pr(inout, as_ref<Matrix>(A), alpha, eps, as_ref<Descriptor>(descriptor));
    if (st != Status::Ok) return to_c(st);
    Vector* out = inout.release();// one reference handed to the caller ...
    as_ptr<Vector>(*p)->rel_ref();   // ... in exchange for the caller's reference to the vector passed in
    *p = reinterpret_cast<spla_Vector>(out);
    return SPLA_STATUS_OK;
}
spla_Status spla_Algorithm_tc(int* ntrins, spla_Matrix A, spla_Matrix B, spla_Descriptor descriptor) {
    if (!ntrins) return SPLA_STATUS_INVALID_ARGUMENT;
    return to_c(tc(*ntrins, as_ref<Matrix>(A), as_ref<Matrix>(B), as_ref<Descriptor>(descriptor)));
}

spla_Status spla_MtxLoader_load(spla_MtxLoader* loader, const char* file_path, spla_bool offset_indices, spla_bool make_undirected, spla_bool remove_loops) {
    if (!loader || !file_path) return SPLA_STATUS_INVALID_ARGUMENT;
    auto* l = new MtxLoader();
    if (!l->load(file_path, offset_indices != 0, make_undirected != 0, remove_loops != 0)) {
        delete l;
        *loader = nullptr;
        return SPLA_STATUS_ERROR;
    }
    *loader = reinterpret_cast<spla_MtxLoader>(l);
    return SPLA_STATUS_OK;
}
spla_Status spla_MtxLoader_get(spla_MtxLoader loader, spla_uint* n_rows, spla_uint* n_cols, spla_size_t* n_values, const spla_uint** Ai, const spla_uint** Aj) {
    if (!loader) return SPLA_STATUS_INVALID_ARGUMENT;
    auto* l = reinterpret_cast<MtxLoader*>(loader);
    if (n_rows) *n_rows = l->get_n_rows();
    if (n_cols) *n_cols = l->get_n_cols();
    if (n_values) *n_values = l->get_n_values();
    if (Ai) *Ai = l->get_Ai().data();
    if (Aj) *Aj = l->get_Aj().data();
    return SPLA_STATUS_OK;
}
spla_Status spla_MtxLoader_delete(spla_MtxLoader loader) {
    delete reinterpret_cast<MtxLoader*>(loader);
    return SPLA_STATUS_OK;
}

spla_Status spla_Descriptor_make(spla_Descriptor* desc) { return give(desc, Descriptor::make()); }
spla_Status spla_Descriptor_set_early_exit(spla_Descriptor desc, spla_bool value) {
    if (!desc) return SPLA_STATUS_INVALID_ARGUMENT;
    as_ptr<Descriptor>(desc)->set_early_exit(value != 0);
    return SPLA_STATUS_OK;
}
spla_Status spla_Descriptor_set_struct_only(spla_Descriptor desc, spla_bool value) {
    if (!desc) return SPLA_STATUS_INVALID_ARGUMENT;
    as_ptr<Descriptor>(desc)->set_struct_only(value != 0);
    return SPLA_STATUS_OK;
}
spla_Status spla_Descriptor_set_deterministic(spla_Descriptor desc, spla_bool value) {
    if (!desc) return SPLA_STATUS_INVALID_ARGUMENT;
    as_ptr<Descriptor>(desc)->set_deterministic(value != 0);
    return SPLA_STATUS_OK;
}
spla_Status spla_Descriptor_set_reference_order(spla_Descriptor desc, spla_bool value) {
    if (!desc) return SPLA_STATUS_INVALID_ARGUMENT;
    as_ptr<Descriptor>(desc)->set_reference_order(value != 0);
    return SPLA_STATUS_OK;
}
spla_Status spla_Descriptor_set_traversal_mode(spla_Descriptor desc, int mode) {
    if (!desc || mode < 0 || mode > 2) return SPLA_STATUS_INVALID_ARGUMENT;
    as_ptr<Descriptor>(desc)->set_traversal_mode(static_cast<Descriptor::TraversalMode>(mode));
    return SPLA_STATUS_OK;
}
spla_Status spla_Descriptor_set_front_factor(spla_Descriptor desc, float value) {
    if (!desc) return SPLA_STATUS_INVALID_ARGUMENT;
    as_ptr<Descriptor>(desc)->set_front_factor(value);
    return SPLA_STATUS_OK;
}
}
