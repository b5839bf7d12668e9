// ref_shim.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A few extern "C" helpers compiled INTO oracle/_ref/libspla_ref.so next to the genuine reference objects
// (oracle/Makefile), against the reference's own headers where they lie (-I/root/reference/{include,src}).
// They exist because the reference's C ABI cannot express everything the parity tests need:
//   * it has no Descriptor constructor / setter (include/spla.h:119 is the only mention), so early_exit=true
//     cannot be requested through spla.h alone;
//   * Matrix::read() on an mxm / mxmT result walks CpuLil::values, which those algorithms never update
//     (src/cpu/cpu_mxm.hpp:109-113, src/cpu/cpu_mxmT_masked.hpp:121-123) and crashes — results are read here
//     straight from the LIL rows instead;
//   * bulk reads of dense results (spla_Vector_get_* is one hash lookup per element).
// Nothing here re-implements a reference algorithm: every function only constructs objects or copies bytes
// out of the reference's own storage decorations.
#include <spla.hpp>

#include <core/tmatrix.hpp>
#include <core/tvector.hpp>

#include <cstdint>
#include <cstring>

using namespace spla;

namespace {
    template<typename T>
    uint64_t lil_count(Matrix* m) {
        auto* t = dynamic_cast<TMatrix<T>*>(m);
        if (!t) return UINT64_MAX;
        t->validate_rw(FormatMatrix::CpuLil);
        uint64_t n = 0;
        for (const auto& row : t->template get<CpuLil<T>>()->Ar) n += row.size();
        return n;
    }
    template<typename T>
    bool lil_read(Matrix* m, uint32_t* Rp, uint32_t* Rj, uint32_t* Rx) {
        auto* t = dynamic_cast<TMatrix<T>*>(m);
        if (!t) return false;
        t->validate_rw(FormatMatrix::CpuLil);
        const auto& Ar = t->template get<CpuLil<T>>()->Ar;
        uint32_t    k  = 0;
        Rp[0]          = 0;
        for (std::size_t i = 0; i < Ar.size(); ++i) {
            for (const auto& e : Ar[i]) {
                Rj[k] = e.first;
                std::memcpy(&Rx[k], &e.second, 4);
                ++k;
            }
            Rp[i + 1] = k;
        }
        return true;
    }
    template<typename T>
    bool dense_read(Vector* v, uint32_t* out) {
        auto* t = dynamic_cast<TVector<T>*>(v);
        if (!t) return false;
        t->validate_rw(FormatVector::CpuDense);
        const auto& Ax = t->template get<CpuDenseVec<T>>()->Ax;
        std::memcpy(out, Ax.data(), Ax.size() * 4);
        return true;
    }
    template<typename T>
    bool dense_write(Vector* v, const uint32_t* in) {
        auto* t = dynamic_cast<TVector<T>*>(v);
        if (!t) return false;
        t->validate_wd(FormatVector::CpuDense);
        auto& Ax = t->template get<CpuDenseVec<T>>()->Ax;
        std::memcpy(Ax.data(), in, Ax.size() * 4);
        return true;
    }
}// namespace

extern "C" {

// Descriptor with early_exit set; caller owns one reference (release with spla_RefCnt_unref).
void* refshim_descriptor_make(int early_exit, int struct_only) {
    auto d = Descriptor::make();
    d->set_early_exit(early_exit != 0);
    d->set_struct_only(struct_only != 0);
    return d.release();
}

// Descriptor for the graph algorithms: traversal mode (0 push, 1 pull, 2 push-pull) and the push/pull switch density.
void* refshim_descriptor_make_traversal(int mode, float front_factor) {
    auto d = Descriptor::make();
    d->set_traversal_mode(static_cast<Descriptor::TraversalMode>(mode));
    d->set_front_factor(front_factor);
    return d.release();
}

// The reference's MatrixMarket loader (C++ only): load, hand out sizes + index arrays, free.
void* refshim_mtx_load(const char* path, int offset_indices, int make_undirected, int remove_loops) {
    auto* l = new MtxLoader();
    if (!l->load(path, offset_indices != 0, make_undirected != 0, remove_loops != 0)) {
        delete l;
        return nullptr;
    }
    return l;
}
void refshim_mtx_get(void* loader, uint32_t* n_rows, uint32_t* n_cols, uint64_t* n_values, const uint32_t** Ai, const uint32_t** Aj) {
    auto* l   = static_cast<MtxLoader*>(loader);
    *n_rows   = l->get_n_rows();
    *n_cols   = l->get_n_cols();
    *n_values = l->get_n_values();
    *Ai       = l->get_Ai().data();
    *Aj       = l->get_Aj().data();
}
void refshim_mtx_free(void* loader) { delete static_cast<MtxLoader*>(loader); }

// Stored entries of a matrix as seen through its LIL rows (robust to the `values` counter bug above).
uint64_t refshim_matrix_lil_count(void* m) {
    auto*    M = static_cast<Matrix*>(m);
    uint64_t n = lil_count<T_INT>(M);
    if (n == UINT64_MAX) n = lil_count<T_UINT>(M);
    if (n == UINT64_MAX) n = lil_count<T_FLOAT>(M);
    return n;
}
int refshim_matrix_lil_read(void* m, uint32_t* Rp, uint32_t* Rj, uint32_t* Rx) {
    auto* M = static_cast<Matrix*>(m);
    return (lil_read<T_INT>(M, Rp, Rj, Rx) || lil_read<T_UINT>(M, Rp, Rj, Rx) || lil_read<T_FLOAT>(M, Rp, Rj, Rx)) ? 0 : 1;
}

// Whole dense image of a vector (n_rows words), absent entries = fill value, exactly what the CPU algorithms see.
int refshim_vector_dense_read(void* v, uint32_t* out) {
    auto* V = static_cast<Vector*>(v);
    return (dense_read<T_INT>(V, out) || dense_read<T_UINT>(V, out) || dense_read<T_FLOAT>(V, out)) ? 0 : 1;
}
// Replace a vector's content with a dense image (bulk alternative to n_rows set_* calls).
int refshim_vector_dense_write(void* v, const uint32_t* in) {
    auto* V = static_cast<Vector*>(v);
    return (dense_write<T_INT>(V, in) || dense_write<T_UINT>(V, in) || dense_write<T_FLOAT>(V, in)) ? 0 : 1;
}

// Force the CPU backend (the build has no accelerator anyway) and silence the logger.
void refshim_force_cpu() {
    Library::get()->set_force_no_acceleration(true);
}
}
