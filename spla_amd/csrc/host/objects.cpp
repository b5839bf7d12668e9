// objects.cpp — Vector::make / Matrix::make and the registered formats + converters of the two format graphs.
// Converter semantics follow the reference's (file:line at each edge); the accelerator edges move data through the
// kernel-level C ABI (include/spla_hip.h) on the accelerator's in-order default stream.
#include "storage.hpp"

namespace spla {

    namespace {
        constexpr int V(FormatVector f) { return static_cast<int>(f); }
        constexpr int M(FormatMatrix f) { return static_cast<int>(f); }

        void require_acc() {
            if (!get_acc_hip()) throw std::runtime_error("the HIP accelerator is not active: accelerator formats are unavailable");
        }

        template<typename T>
        constexpr int type_id() { return TypeTraits<T>::id; }

        // ---------------------------------------------------------------------------------------------------- vectors
        template<typename T>
        FormatGraph<T, kVecFormats> make_vector_graph() {
            using S = Storage<T, kVecFormats>;
            FormatGraph<T, kVecFormats> g;
            auto dok   = [](S& s) { return s.template get<CpuDokVec<T>>(V(FormatVector::CpuDok)); };
            auto dense = [](S& s) { return s.template get<CpuDenseVec<T>>(V(FormatVector::CpuDense)); };
            auto coo   = [](S& s) { return s.template get<CpuCooVec<T>>(V(FormatVector::CpuCoo)); };
            auto adense = [](S& s) { return s.template get<HipDenseVec<T>>(V(FormatVector::AccDense)); };
            auto acoo   = [](S& s) { return s.template get<HipCooVec<T>>(V(FormatVector::AccCoo)); };

            g.set_constructor(V(FormatVector::CpuDok), [](S& s) { s.slots[V(FormatVector::CpuDok)] = ref_ptr<Decoration<T>>(new CpuDokVec<T>()); });
            g.set_constructor(V(FormatVector::CpuCoo), [](S& s) { s.slots[V(FormatVector::CpuCoo)] = ref_ptr<Decoration<T>>(new CpuCooVec<T>()); });
            g.set_constructor(V(FormatVector::CpuDense), [](S& s) {
                auto* d = new CpuDenseVec<T>();
                d->Ax.resize(s.n_rows);
                d->values                        = s.n_rows;
                s.slots[V(FormatVector::CpuDense)] = ref_ptr<Decoration<T>>(d);
            });
            g.set_constructor(V(FormatVector::AccDense), [](S& s) {
                require_acc();
                auto* d = new HipDenseVec<T>();
                d->Ax.reserve(s.n_rows);
                d->values                        = s.n_rows;
                s.slots[V(FormatVector::AccDense)] = ref_ptr<Decoration<T>>(d);
            });
            g.set_constructor(V(FormatVector::AccCoo), [](S& s) {
                require_acc();
                s.slots[V(FormatVector::AccCoo)] = ref_ptr<Decoration<T>>(new HipCooVec<T>());
            });

            // an object that holds nothing yet: sparse formats are empty, dense formats are all `fill`
            // (reference src/storage/storage_manager_vector.hpp:68-76,108-115)
            auto clear_dok = [dok](S& s) { dok(s)->Ax.clear(), dok(s)->values = 0; };
            auto clear_coo = [coo](S& s) { coo(s)->Ai.clear(), coo(s)->Ax.clear(), coo(s)->values = 0; };
            g.set_validator(V(FormatVector::CpuDok), clear_dok);
            g.set_discard(V(FormatVector::CpuDok), clear_dok);
            g.set_validator(V(FormatVector::CpuCoo), clear_coo);
            g.set_discard(V(FormatVector::CpuCoo), clear_coo);
            g.set_validator(V(FormatVector::CpuDense), [dense](S& s) { std::fill(dense(s)->Ax.begin(), dense(s)->Ax.end(), s.fill); });
            g.set_validator(V(FormatVector::AccDense), [adense](S& s) {
                hip_check(splahip_memset32(adense(s)->Ax.ptr(), word_of(s.fill), s.n_rows, nullptr), "fill AccDense");
            });
            auto clear_acoo = [acoo](S& s) { acoo(s)->values = 0; };
            g.set_validator(V(FormatVector::AccCoo), clear_acoo);
            g.set_discard(V(FormatVector::AccCoo), clear_acoo);

            // Dok -> Coo: sorted by index (reference src/cpu/cpu_format_dok_vec.hpp:42-69)
            g.add_converter(V(FormatVector::CpuDok), V(FormatVector::CpuCoo), [dok, coo](S& s) {
                auto*                           in  = dok(s);
                auto*                           out = coo(s);
                std::vector<std::pair<uint, T>> tmp(in->Ax.begin(), in->Ax.end());
                std::sort(tmp.begin(), tmp.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
                out->Ai.resize(tmp.size());
                out->Ax.resize(tmp.size());
                for (std::size_t k = 0; k < tmp.size(); ++k) out->Ai[k] = tmp[k].first, out->Ax[k] = tmp[k].second;
                out->values = static_cast<uint>(tmp.size());
            });
            // Dok -> Dense: fill then scatter (reference src/storage/storage_manager_vector.hpp:81-86)
            g.add_converter(V(FormatVector::CpuDok), V(FormatVector::CpuDense), [dok, dense](S& s) {
                auto& Ax = dense(s)->Ax;
                std::fill(Ax.begin(), Ax.end(), s.fill);
                for (const auto& e : dok(s)->Ax) Ax[e.first] = e.second;
            });
            // Coo -> Dok: later duplicates overwrite (reference src/cpu/cpu_format_coo_vec.hpp:77-90)
            g.add_converter(V(FormatVector::CpuCoo), V(FormatVector::CpuDok), [dok, coo](S& s) {
                auto* out = dok(s);
                out->Ax.clear();
                out->values = 0;
                for (std::size_t k = 0; k < coo(s)->Ai.size(); ++k) {
                    out->Ax[coo(s)->Ai[k]] = coo(s)->Ax[k];
                }
                out->values = static_cast<uint>(out->Ax.size());
            });
            // Dense -> Dok: entries equal to the fill value vanish (reference src/cpu/cpu_format_dense_vec.hpp:53-67)
            g.add_converter(V(FormatVector::CpuDense), V(FormatVector::CpuDok), [dok, dense](S& s) {
                auto* out = dok(s);
                out->Ax.clear();
                const auto& Ax = dense(s)->Ax;
                for (uint i = 0; i < s.n_rows; ++i)
                    if (Ax[i] != s.fill) out->Ax.emplace(i, Ax[i]);
                out->values = static_cast<uint>(out->Ax.size());
            });

            // Dok / Coo -> AccDense in one hop: the stored entries travel (a few words), the fill happens on the device.  The shortest path used to be
            // Dok -> CpuDense -> AccDense: a host fill and a 4 n-byte upload from pageable memory per vector — spla_Algorithm_sssp / pr create four such vectors per
            // run (the empty mask, the all-`inf` distance vector ...): ~0.4 ms each at n = 2^20
            auto scatter_to_device = [adense](S& s, const std::vector<uint>& idx, const std::vector<T>& val) {
                DeviceBuffer   di, dx;
                const uint32_t n = static_cast<uint32_t>(idx.size());
                if (n) di.upload(idx.data(), n), dx.upload(val.data(), n);
                hip_check(splahip_vec_coo_to_dense(s.n_rows, n, di.template as<uint32_t>(), dx.ptr(), word_of(s.fill), adense(s)->Ax.ptr(), nullptr), "host entries -> AccDense");
            };
            g.add_converter(V(FormatVector::CpuDok), V(FormatVector::AccDense), [dok, scatter_to_device](S& s) {
                std::vector<uint> idx;
                std::vector<T>    val;
                idx.reserve(dok(s)->Ax.size()), val.reserve(dok(s)->Ax.size());
                for (const auto& e : dok(s)->Ax) idx.push_back(e.first), val.push_back(e.second);
                scatter_to_device(s, idx, val);
            });
            g.add_converter(V(FormatVector::CpuCoo), V(FormatVector::AccDense), [coo, scatter_to_device](S& s) {
                // (later duplicates overwrite, as in Coo -> Dok: the scatter of equal indices is not ordered on the device, so duplicates are resolved here)
                std::unordered_map<uint, T> last;
                bool                        dup = false;
                for (std::size_t k = 0; k < coo(s)->Ai.size() && !dup; ++k) dup = k > 0 && coo(s)->Ai[k] <= coo(s)->Ai[k - 1];
                if (!dup) return scatter_to_device(s, coo(s)->Ai, coo(s)->Ax);
                for (std::size_t k = 0; k < coo(s)->Ai.size(); ++k) last[coo(s)->Ai[k]] = coo(s)->Ax[k];
                std::vector<uint> idx;
                std::vector<T>    val;
                for (const auto& e : last) idx.push_back(e.first), val.push_back(e.second);
                scatter_to_device(s, idx, val);
            });
            // host <-> device (reference src/storage/storage_manager_vector.hpp:117-152)
            g.add_converter(V(FormatVector::CpuDense), V(FormatVector::AccDense), [dense, adense](S& s) { adense(s)->Ax.upload(dense(s)->Ax.data(), s.n_rows); });
            g.add_converter(V(FormatVector::AccDense), V(FormatVector::CpuDense), [dense, adense](S& s) {
                if (s.n_rows) adense(s)->Ax.download(dense(s)->Ax.data(), s.n_rows);
            });
            g.add_converter(V(FormatVector::CpuCoo), V(FormatVector::AccCoo), [coo, acoo](S& s) {
                const uint n = coo(s)->values;
                acoo(s)->Ai.upload(coo(s)->Ai.data(), n);
                acoo(s)->Ax.upload(coo(s)->Ax.data(), n);
                acoo(s)->values = n;
            });
            g.add_converter(V(FormatVector::AccCoo), V(FormatVector::CpuCoo), [coo, acoo](S& s) {
                const uint n = acoo(s)->values;
                coo(s)->Ai.resize(n);
                coo(s)->Ax.resize(n);
                coo(s)->values = n;
                if (n) {
                    acoo(s)->Ai.download(coo(s)->Ai.data(), n);
                    acoo(s)->Ax.download(coo(s)->Ax.data(), n);
                }
            });
            // device <-> device (reference src/storage/storage_manager_vector.hpp:153-166); dense -> coo is order-preserving here
            g.add_converter(V(FormatVector::AccCoo), V(FormatVector::AccDense), [adense, acoo](S& s) {
                hip_check(splahip_vec_coo_to_dense(s.n_rows, acoo(s)->values, acoo(s)->Ai.template as<uint32_t>(), acoo(s)->Ax.ptr(),
                                                   word_of(s.fill), adense(s)->Ax.ptr(), nullptr),
                          "AccCoo -> AccDense");
            });
            g.add_converter(V(FormatVector::AccDense), V(FormatVector::AccCoo), [adense, acoo](S& s) {
                acoo(s)->Ai.reserve(s.n_rows);
                acoo(s)->Ax.reserve(s.n_rows);
                uint32_t count = 0;
                hip_check(splahip_vec_dense_to_coo(type_id<T>(), s.n_rows, adense(s)->Ax.ptr(), word_of(s.fill),
                                                   acoo(s)->Ai.template as<uint32_t>(), acoo(s)->Ax.ptr(), &count, nullptr),
                          "AccDense -> AccCoo");
                acoo(s)->values = count;
            });
            return g;
        }

        // --------------------------------------------------------------------------------------------------- matrices
        template<typename T>
        void csr_from_rows(uint n_rows, const std::vector<uint>& row_of, CpuCsr<T>& out) {
            out.Ap.assign(n_rows + 1, 0);
            for (uint i : row_of) out.Ap[i + 1] += 1;
            std::partial_sum(out.Ap.begin(), out.Ap.end(), out.Ap.begin());
        }

        template<typename T>
        FormatGraph<T, kMatFormats> make_matrix_graph() {
            using S = Storage<T, kMatFormats>;
            FormatGraph<T, kMatFormats> g;
            auto lil  = [](S& s) { return s.template get<CpuLil<T>>(M(FormatMatrix::CpuLil)); };
            auto dok  = [](S& s) { return s.template get<CpuDok<T>>(M(FormatMatrix::CpuDok)); };
            auto coo  = [](S& s) { return s.template get<CpuCoo<T>>(M(FormatMatrix::CpuCoo)); };
            auto csr  = [](S& s) { return s.template get<CpuCsr<T>>(M(FormatMatrix::CpuCsr)); };
            auto acsr = [](S& s) { return s.template get<HipCsr<T>>(M(FormatMatrix::AccCsr)); };

            g.set_constructor(M(FormatMatrix::CpuLil), [](S& s) {
                auto* l = new CpuLil<T>();
                l->Ar.resize(s.n_rows);
                s.slots[M(FormatMatrix::CpuLil)] = ref_ptr<Decoration<T>>(l);
            });
            g.set_constructor(M(FormatMatrix::CpuDok), [](S& s) { s.slots[M(FormatMatrix::CpuDok)] = ref_ptr<Decoration<T>>(new CpuDok<T>()); });
            g.set_constructor(M(FormatMatrix::CpuCoo), [](S& s) { s.slots[M(FormatMatrix::CpuCoo)] = ref_ptr<Decoration<T>>(new CpuCoo<T>()); });
            g.set_constructor(M(FormatMatrix::CpuCsr), [](S& s) {
                auto* c = new CpuCsr<T>();
                c->Ap.assign(s.n_rows + 1, 0);
                s.slots[M(FormatMatrix::CpuCsr)] = ref_ptr<Decoration<T>>(c);
            });
            g.set_constructor(M(FormatMatrix::AccCsr), [](S& s) {
                require_acc();
                s.slots[M(FormatMatrix::AccCsr)] = ref_ptr<Decoration<T>>(new HipCsr<T>());
            });
            g.set_constructor(M(FormatMatrix::AccCsc), [](S& s) {
                require_acc();
                s.slots[M(FormatMatrix::AccCsc)] = ref_ptr<Decoration<T>>(new HipCsr<T>());
            });

            auto clear_lil = [lil](S& s) {
                for (auto& row : lil(s)->Ar) row.clear();
                lil(s)->Ar.resize(s.n_rows);
                lil(s)->values = 0;
            };
            auto clear_dok = [dok](S& s) { dok(s)->Ax.clear(), dok(s)->values = 0; };
            auto clear_coo = [coo](S& s) { coo(s)->Ai.clear(), coo(s)->Aj.clear(), coo(s)->Ax.clear(), coo(s)->values = 0; };
            auto clear_csr = [csr](S& s) { csr(s)->Ap.assign(s.n_rows + 1, 0), csr(s)->Aj.clear(), csr(s)->Ax.clear(), csr(s)->values = 0; };
            g.set_validator(M(FormatMatrix::CpuLil), clear_lil), g.set_discard(M(FormatMatrix::CpuLil), clear_lil);
            g.set_validator(M(FormatMatrix::CpuDok), clear_dok), g.set_discard(M(FormatMatrix::CpuDok), clear_dok);
            g.set_validator(M(FormatMatrix::CpuCoo), clear_coo), g.set_discard(M(FormatMatrix::CpuCoo), clear_coo);
            g.set_validator(M(FormatMatrix::CpuCsr), clear_csr), g.set_discard(M(FormatMatrix::CpuCsr), clear_csr);
            auto clear_acsr = [acsr](S& s) {// an empty device CSR: Ap = zeros, plan over zero entries
                auto* a = acsr(s);
                a->drop_plan();
                a->values = 0;
                a->Ap.reserve(s.n_rows + 1);
                hip_check(splahip_memset32(a->Ap.ptr(), 0, s.n_rows + 1, nullptr), "clear AccCsr");
            };
            g.set_validator(M(FormatMatrix::AccCsr), clear_acsr), g.set_discard(M(FormatMatrix::AccCsr), clear_acsr);

            // Lil -> Dok / Coo / Csr (reference src/cpu/cpu_format_lil.hpp:76-147)
            g.add_converter(M(FormatMatrix::CpuLil), M(FormatMatrix::CpuDok), [lil, dok](S& s) {
                auto* out = dok(s);
                out->Ax.clear();
                for (uint i = 0; i < s.n_rows; ++i)
                    for (const auto& e : lil(s)->Ar[i]) out->Ax[CpuDok<T>::key(i, e.first)] = e.second;
                out->values = static_cast<uint>(out->Ax.size());
            });
            g.add_converter(M(FormatMatrix::CpuLil), M(FormatMatrix::CpuCoo), [lil, coo](S& s) {
                auto* out = coo(s);
                out->Ai.clear(), out->Aj.clear(), out->Ax.clear();
                for (uint i = 0; i < s.n_rows; ++i)
                    for (const auto& e : lil(s)->Ar[i]) out->Ai.push_back(i), out->Aj.push_back(e.first), out->Ax.push_back(e.second);
                out->values = static_cast<uint>(out->Ax.size());
            });
            g.add_converter(M(FormatMatrix::CpuLil), M(FormatMatrix::CpuCsr), [lil, csr](S& s) {
                auto* out = csr(s);
                out->Ap.assign(s.n_rows + 1, 0);
                out->Aj.clear(), out->Ax.clear();
                for (uint i = 0; i < s.n_rows; ++i) {
                    for (const auto& e : lil(s)->Ar[i]) out->Aj.push_back(e.first), out->Ax.push_back(e.second);
                    out->Ap[i + 1] = static_cast<uint>(out->Aj.size());
                }
                out->values = static_cast<uint>(out->Aj.size());
            });
            // Coo -> Lil: rows appended in COO order (reference src/cpu/cpu_format_coo.hpp:58-76)
            g.add_converter(M(FormatMatrix::CpuCoo), M(FormatMatrix::CpuLil), [lil, coo](S& s) {
                auto* out = lil(s);
                for (auto& row : out->Ar) row.clear();
                out->Ar.resize(s.n_rows);
                const auto* in = coo(s);
                for (std::size_t k = 0; k < in->Ai.size(); ++k) out->Ar[in->Ai[k]].emplace_back(in->Aj[k], in->Ax[k]);
                out->values = static_cast<uint>(in->Ai.size());
            });
            g.add_converter(M(FormatMatrix::CpuCoo), M(FormatMatrix::CpuDok), [dok, coo](S& s) {
                auto* out = dok(s);
                out->Ax.clear();
                const auto* in = coo(s);
                for (std::size_t k = 0; k < in->Ai.size(); ++k) out->Ax[CpuDok<T>::key(in->Ai[k], in->Aj[k])] = in->Ax[k];
                out->values = static_cast<uint>(out->Ax.size());
            });
            // Coo -> Csr: counts per row, entries copied in COO order — the COO is trusted to be row-sorted
            // (reference src/cpu/cpu_format_coo.hpp:96-123)
            g.add_converter(M(FormatMatrix::CpuCoo), M(FormatMatrix::CpuCsr), [csr, coo](S& s) {
                auto*       out = csr(s);
                const auto* in  = coo(s);
                csr_from_rows<T>(s.n_rows, in->Ai, *out);
                out->Aj     = in->Aj;
                out->Ax     = in->Ax;
                out->values = static_cast<uint>(in->Ai.size());
            });
            // Csr -> Dok / Coo (reference src/cpu/cpu_format_csr.hpp)
            g.add_converter(M(FormatMatrix::CpuCsr), M(FormatMatrix::CpuDok), [dok, csr](S& s) {
                auto* out = dok(s);
                out->Ax.clear();
                const auto* in = csr(s);
                for (uint i = 0; i < s.n_rows; ++i)
                    for (uint k = in->Ap[i]; k < in->Ap[i + 1]; ++k) out->Ax[CpuDok<T>::key(i, in->Aj[k])] = in->Ax[k];
                out->values = static_cast<uint>(out->Ax.size());
            });
            g.add_converter(M(FormatMatrix::CpuCsr), M(FormatMatrix::CpuCoo), [coo, csr](S& s) {
                auto*       out = coo(s);
                const auto* in  = csr(s);
                out->Ai.resize(in->Aj.size());
                for (uint i = 0; i < s.n_rows; ++i)
                    for (uint k = in->Ap[i]; k < in->Ap[i + 1]; ++k) out->Ai[k] = i;
                out->Aj     = in->Aj;
                out->Ax     = in->Ax;
                out->values = static_cast<uint>(in->Aj.size());
            });
            // CpuCsr <-> AccCsr: three buffers + the work decomposition, built once per upload
            // (reference src/opencl/cl_format_csr.hpp:40-59 uploads the three arrays)
            g.add_converter(M(FormatMatrix::CpuCsr), M(FormatMatrix::AccCsr), [csr, acsr](S& s) {
                auto*       a  = acsr(s);
                const auto* in = csr(s);
                const uint  nnz = static_cast<uint>(in->Aj.size());
                a->drop_plan();
                a->Ap.upload(in->Ap.data(), s.n_rows + 1);
                a->Aj.upload(in->Aj.data(), nnz);
                a->Ax.upload(in->Ax.data(), nnz);
                a->values = nnz;
                hip_check(splahip_csr_plan_create(&a->plan, s.n_rows, nnz, a->Ap.template as<uint32_t>(), nullptr), "csr plan");
            });
            g.add_converter(M(FormatMatrix::AccCsr), M(FormatMatrix::CpuCsr), [csr, acsr](S& s) {
                auto*      out = csr(s);
                auto*      a   = acsr(s);
                const uint nnz = a->values;
                out->Ap.resize(s.n_rows + 1);
                out->Aj.resize(nnz);
                out->Ax.resize(nnz);
                a->Ap.download(out->Ap.data(), s.n_rows + 1);
                if (nnz) {
                    a->Aj.download(out->Aj.data(), nnz);
                    a->Ax.download(out->Ax.data(), nnz);
                }
                out->values = nnz;
            });
            return g;
        }
    }// namespace

    template<typename T>
    const FormatGraph<T, kVecFormats>& vector_formats() {
        static const FormatGraph<T, kVecFormats> graph = make_vector_graph<T>();
        return graph;
    }
    template<typename T>
    const FormatGraph<T, kMatFormats>& matrix_formats() {
        static const FormatGraph<T, kMatFormats> graph = make_matrix_graph<T>();
        return graph;
    }
    template const FormatGraph<T_INT, kVecFormats>&   vector_formats<T_INT>();
    template const FormatGraph<T_UINT, kVecFormats>&  vector_formats<T_UINT>();
    template const FormatGraph<T_FLOAT, kVecFormats>& vector_formats<T_FLOAT>();
    template const FormatGraph<T_INT, kMatFormats>&   matrix_formats<T_INT>();
    template const FormatGraph<T_UINT, kMatFormats>&  matrix_formats<T_UINT>();
    template const FormatGraph<T_FLOAT, kMatFormats>& matrix_formats<T_FLOAT>();

    // a null handle (not a status) for a zero dimension or a missing type (reference src/matrix.cpp:34-41)
    ref_ptr<Vector> Vector::make(uint n_rows, const ref_ptr<Type>& type) {
        if (n_rows == 0 || !type) return ref_ptr<Vector>();
        Library::get();
        if (type == INT) return ref_ptr<Vector>(new TVector<T_INT>(n_rows));
        if (type == UINT) return ref_ptr<Vector>(new TVector<T_UINT>(n_rows));
        if (type == FLOAT) return ref_ptr<Vector>(new TVector<T_FLOAT>(n_rows));
        SPLA_LOG(Status::NotImplemented, "no vector container for type " << type->get_name());
        return ref_ptr<Vector>();
    }

    ref_ptr<Matrix> Matrix::make(uint n_rows, uint n_cols, const ref_ptr<Type>& type) {
        if (n_rows == 0 || n_cols == 0 || !type) return ref_ptr<Matrix>();
        Library::get();
        if (type == INT) return ref_ptr<Matrix>(new TMatrix<T_INT>(n_rows, n_cols));
        if (type == UINT) return ref_ptr<Matrix>(new TMatrix<T_UINT>(n_rows, n_cols));
        if (type == FLOAT) return ref_ptr<Matrix>(new TMatrix<T_FLOAT>(n_rows, n_cols));
        SPLA_LOG(Status::NotImplemented, "no matrix container for type " << type->get_name());
        return ref_ptr<Matrix>();
    }

}// namespace spla
