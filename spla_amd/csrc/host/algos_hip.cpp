// algos_hip.cpp — the HIP accelerator's RegistryAlgo set for the hot path, registered under "<op>_<I|U|F>__hip".
// Each execute() mirrors the shape of the reference's accelerated algorithms (validate the operands into accelerator
// formats, then launch): reference src/opencl/cl_mxv.hpp:65-121, cl_vxm.hpp:60-173, cl_mxmT_masked.hpp:60-127 — with
// the launches going through the kernel-level C ABI (include/spla_hip.h) instead of JIT-built OpenCL programs.
#include "storage.hpp"

#include <cstdlib>
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace spla {

    namespace {

        template<typename T>
        int device_binop(const ref_ptr<OpBinary>& op) {
            auto t = op.template cast_safe<TOpBinary<T>>();// type mismatch throws -> Status::Error (reference ref.hpp:220-229)
            return t->device_id;
        }
        template<typename T>
        int device_selop(const ref_ptr<OpSelect>& op) {
            auto t = op.template cast_safe<TOpSelect<T>>();
            return t->device_id;
        }

        template<typename T>
        int device_unop(const ref_ptr<OpUnary>& op) {
            auto t = op.template cast_safe<TOpUnary<T>>();
            return t->device_id;
        }

        // the plan is told once per upload whether every stored value is one word (an adjacency matrix): the BFS-like steps then settle a row / an output entry
        // at its first hit (spla_hip.h: splahip_csr_plan_set_uniform_values)
        template<typename T>
        void tell_value_range(HipCsr<T>* a) {
            uint32_t lo = 1, hi = 0;
            if (a->values > 0 && splahip_words_range(a->Ax.ptr(), a->values, &lo, &hi, nullptr) == SPLAHIP_OK)
                (void) splahip_csr_plan_set_uniform_values(a->plan, lo == hi ? 1 : 0, lo);
        }
        template<typename T>
        void ensure_plan(HipCsr<T>* a, uint n_rows) {
            if (!a->plan) {
                hip_check(splahip_csr_plan_create(&a->plan, n_rows, a->values, a->Ap.template as<uint32_t>(), nullptr), "csr plan");
                tell_value_range(a);
            }
        }
        unsigned env_calls(const char* name, unsigned dflt) {
            const char* e = std::getenv(name);
            return e ? static_cast<unsigned>(std::strtoul(e, nullptr, 10)) : dflt;
        }
        // A matrix that keeps being multiplied gets its layout copies — rent or buy.  The column-ordered-blocks copy (+8 B per entry + 2 B per row, ~10 ms
        // per 65 M entries: what a few dozen SpMV calls save) is built when the second call arrives: SPLA_HIP_COB_AFTER=<calls> (0 = never).  The
        // sliced-ELL copy (+12 B per entry, tens of ms) serves only the calls the blocks do not take from it — the zero-family sparse-vector forms, a BFS
        // step (splahip_mxv_prefers_sell) — so only THOSE calls count towards it: once 256 of them have been served (SPLA_HIP_SELL_AFTER), or at the first
        // one for a matrix prepared with Matrix::set_format(AccCsr) (which builds the blocks at once, storage.hpp).  A matrix that never sees such a call
        // keeps 16 B per entry on the device (CSR 8 + blocks 8) instead of 28.  A copy that cannot be built is not an error: the CSR kernels stay in use.
        template<typename T>
        void maybe_build_copies(HipCsr<T>* a, uint n_rows, uint n_cols, bool wants_sell = false) {
            ++a->mxv_calls;
            if (wants_sell) ++a->sell_calls;
            const unsigned cob_after = env_calls("SPLA_HIP_COB_AFTER", 2u), sell_after = env_calls("SPLA_HIP_SELL_AFTER", 256u);
            if (!a->cob_tried && cob_after != 0 && (a->prepared || a->mxv_calls >= cob_after)) {
                a->cob_tried = true;
                if (splahip_csr_plan_attach_cob(a->plan, n_cols, a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(), a->Ax.ptr(), cob_row_bits<T>(n_rows), 0, nullptr) != SPLAHIP_OK)
                    SPLA_LOG(Status::Error, "column-ordered-blocks copy not built (" << splahip_last_error() << "); the CSR kernels stay in use");
            }
            if (!a->sell_tried && sell_after != 0 && wants_sell && (a->prepared || a->sell_calls >= sell_after)) {
                a->sell_tried = true;
                if (splahip_csr_plan_attach_sell(a->plan, n_cols, a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(), a->Ax.ptr(), nullptr) != SPLAHIP_OK)
                    SPLA_LOG(Status::Error, "sliced-ELL copy not built (" << splahip_last_error() << "); the CSR kernels stay in use");
            }
        }

        // a user-defined op (OpBinary::make_* / OpSelect::make_*) carries its body as text (reference src/op.cpp:294-317): text for the run-time compiler, or
        // null for a built-in op
        struct UserSrc {
            std::string mult, add, sel;
            const char *pm = nullptr, *pa = nullptr, *ps = nullptr;
            template<class Mul, class Add, class Sel>
            UserSrc(int op_mult, const Mul& m, int op_add, const Add& a, int op_sel, const Sel* sl) {
                if (op_mult < 0) mult = m->get_source_cl(), pm = mult.c_str();
                if (op_add < 0) add = a->get_source_cl(), pa = add.c_str();
                if (sl && op_sel < 0) sel = (*sl)->get_source_cl(), ps = sel.c_str();
            }
        };
        inline Status jit_status(int st, const char* what) {
            if (st == 8) {
                SPLA_LOG(Status::CompilationError, splahip_last_error());
                return Status::CompilationError;
            }
            if (st == SPLAHIP_NOT_IMPLEMENTED) return Status::NotImplemented;
            hip_check(st, what);
            return Status::Ok;
        }

        template<typename T>
        class Algo_mxv_masked_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "mxv_masked"; }
            std::string get_description() override { return "masked matrix-vector product, wave-tile CSR kernels on gfx950"; }

            Status execute(const DispatchContext& ctx) override {
                auto t    = ctx.task.template cast_safe<ScheduleTask_mxv_masked>();
                auto r    = t->r.template cast_safe<TVector<T>>();
                auto mask = t->mask.template cast_safe<TVector<T>>();
                auto M    = t->M.template cast_safe<TMatrix<T>>();
                auto v    = t->v.template cast_safe<TVector<T>>();
                auto init = t->init.template cast_safe<TScalar<T>>();

                const int op_mult = device_binop<T>(t->op_multiply);
                const int op_add  = device_binop<T>(t->op_add);
                const int op_sel  = device_selop<T>(t->op_select);
                if (r->get_n_rows() != M->get_n_rows() || mask->get_n_rows() != M->get_n_rows() || v->get_n_rows() != M->get_n_cols())
                    return Status::InvalidArgument;
                if (op_mult < 0 || op_add < 0 || op_sel < 0) {
                    // a user-defined op (OpBinary::make_* / OpSelect::make_*): its body text is compiled for the device at run time (jit.hip) — what the
                    // reference's OpenCL backend does with the same text (cl_program_builder.cpp:65-120); there is no CPU path to fall back to
                    r->validate_wd(FormatVector::AccDense);
                    mask->validate_rw(FormatVector::AccDense);
                    v->validate_rw(FormatVector::AccDense);
                    M->validate_rw(FormatMatrix::AccCsr);
                    auto*             a  = M->acc_csr();
                    const std::string sm = op_mult < 0 ? t->op_multiply->get_source_cl() : std::string();
                    const std::string sa = op_add < 0 ? t->op_add->get_source_cl() : std::string();
                    const std::string ss = op_sel < 0 ? t->op_select->get_source_cl() : std::string();
                    const int         st = splahip_mxv_masked_jit(TypeTraits<T>::id, M->get_n_rows(), a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(), a->Ax.ptr(),
                                                                  mask->acc_dense()->Ax.ptr(), v->acc_dense()->Ax.ptr(), r->acc_dense()->Ax.ptr(), op_mult,
                                                                  op_mult < 0 ? sm.c_str() : nullptr, op_add, op_add < 0 ? sa.c_str() : nullptr, op_sel,
                                                                  op_sel < 0 ? ss.c_str() : nullptr, word_of(init->get_value()),
                                                                  t->get_desc_or_default()->get_early_exit() ? 1 : 0, nullptr);
                    if (st == 8) {
                        SPLA_LOG(Status::CompilationError, splahip_last_error());
                        return Status::CompilationError;
                    }
                    if (st == SPLAHIP_NOT_IMPLEMENTED) return Status::NotImplemented;
                    hip_check(st, "splahip_mxv_masked_jit");
                    return Status::Ok;
                }

                // same format requests as the reference's accelerated mxv (cl_mxv.hpp:91-94); r is write-discard:
                // every row is overwritten (cpu_mxv.hpp:88-103)
                r->validate_wd(FormatVector::AccDense);
                mask->validate_rw(FormatVector::AccDense);
                v->validate_rw(FormatVector::AccDense);
                M->validate_rw(FormatMatrix::AccCsr);

                auto* a = M->acc_csr();
                ensure_plan(a, M->get_n_rows());
                const bool early = t->get_desc_or_default()->get_early_exit();
                if (!early && a->values > 0) maybe_build_copies(a, M->get_n_rows(), M->get_n_cols(), splahip_mxv_prefers_sell(TypeTraits<T>::id, op_mult, op_add) != 0);
                hip_check(splahip_mxv_masked(a->plan, TypeTraits<T>::id, M->get_n_rows(), M->get_n_cols(),
                                             a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(), a->Ax.ptr(),
                                             mask->acc_dense()->Ax.ptr(), v->acc_dense()->Ax.ptr(), r->acc_dense()->Ax.ptr(),
                                             op_mult, op_add, op_sel, word_of(init->get_value()), early ? 1 : 0, nullptr),
                          "splahip_mxv_masked");
                return Status::Ok;
            }
        };

        inline bool op_assoc_comm(int op) {
            switch (op) {
                case SPLAHIP_MINUS:
                case SPLAHIP_DIV:
                case SPLAHIP_MINUS_POW2:
                case SPLAHIP_FIRST:
                case SPLAHIP_SECOND: return false;
                default: return true;
            }
        }

        // CSR of M^T as the AccCsc decoration: derived on the device from AccCsr, cached until M changes
        template<typename T>
        HipCsr<T>* ensure_transpose(TMatrix<T>* M) {
            constexpr int slot = static_cast<int>(FormatMatrix::AccCsc);
            M->validate_ctor(FormatMatrix::AccCsc);
            auto* t = M->acc_csc();
            if (!M->is_valid(FormatMatrix::AccCsc)) {
                auto*      a   = M->acc_csr();
                const uint nnz = a->values;
                t->drop_plan();
                t->Ap.reserve(M->get_n_cols() + 1);
                t->Aj.reserve(nnz);
                t->Ax.reserve(nnz);
                hip_check(splahip_csr_transpose(M->get_n_rows(), M->get_n_cols(), nnz, a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(),
                                                a->Ax.ptr(), t->Ap.template as<uint32_t>(), t->Aj.template as<uint32_t>(), t->Ax.ptr(), nullptr),
                          "splahip_csr_transpose");
                t->values = nnz;
                hip_check(splahip_csr_plan_create(&t->plan, M->get_n_cols(), nnz, t->Ap.template as<uint32_t>(), nullptr), "csr plan (transpose)");
                tell_value_range(t);
                M->storage().valid.set(slot);
            }
            return t;
        }

        template<typename T>
        class Algo_vxm_masked_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "vxm_masked"; }
            std::string get_description() override { return "masked vector-matrix product: pull over the transposed CSR or push (expand / stable sort / fold)"; }

            Status execute(const DispatchContext& ctx) override {
                auto t    = ctx.task.template cast_safe<ScheduleTask_vxm_masked>();
                auto r    = t->r.template cast_safe<TVector<T>>();
                auto mask = t->mask.template cast_safe<TVector<T>>();
                auto v    = t->v.template cast_safe<TVector<T>>();
                auto M    = t->M.template cast_safe<TMatrix<T>>();
                (void) t->init.template cast_safe<TScalar<T>>();// read but never used (reference cpu_vxm.hpp:72)

                const int op_mult = device_binop<T>(t->op_multiply);
                const int op_add  = device_binop<T>(t->op_add);
                const int op_sel  = device_selop<T>(t->op_select);
                if (v->get_n_rows() != M->get_n_rows() || mask->get_n_rows() != M->get_n_cols() || r->get_n_rows() != M->get_n_cols())
                    return Status::InvalidArgument;
                if (op_mult < 0 || op_add < 0 || op_sel < 0) {
                    // user-defined ops: the pull form over the transposed matrix with the per-column fold compiled at run time (jit.hip), products in the
                    // stored order of v like the reference's loop (cpu_vxm.hpp:92-125)
                    r->validate_wd(FormatVector::AccCoo);
                    mask->validate_rw(FormatVector::AccDense);
                    v->validate_rw(FormatVector::AccCoo);
                    M->validate_rw(FormatMatrix::AccCsr);
                    auto* vin = v->acc_coo();
                    auto* out = r->acc_coo();
                    out->Ai.reserve(M->get_n_cols());
                    out->Ax.reserve(M->get_n_cols());
                    out->values = 0;
                    if (vin->values == 0 || M->acc_csr()->values == 0) return Status::Ok;
                    uint64_t products = 0;
                    int      sorted   = 1;
                    hip_check(splahip_vxm_probe(vin->values, vin->Ai.template as<uint32_t>(), M->acc_csr()->Ap.template as<uint32_t>(), &products, &sorted, nullptr), "splahip_vxm_probe");
                    auto*         tr = ensure_transpose<T>(M.get());
                    const UserSrc src(op_mult, t->op_multiply, op_add, t->op_add, op_sel, &t->op_select);
                    uint32_t      count = 0;
                    const Status  js    = jit_status(splahip_vxm_masked_jit(tr->plan, TypeTraits<T>::id, M->get_n_rows(), M->get_n_cols(), tr->Ap.template as<uint32_t>(),
                                                                            tr->Aj.template as<uint32_t>(), tr->Ax.ptr(), mask->acc_dense()->Ax.ptr(), vin->values,
                                                                            vin->Ai.template as<uint32_t>(), vin->Ax.ptr(), sorted, out->Ai.template as<uint32_t>(), out->Ax.ptr(),
                                                                            &count, op_mult, src.pm, op_add, src.pa, op_sel, src.ps, nullptr),
                                                     "splahip_vxm_masked_jit");
                    if (js == Status::Ok) out->values = count;
                    return js;
                }

                // sparse in / sparse out, dense mask (reference cpu_vxm.hpp:74-77, cl_vxm.hpp:85-88)
                r->validate_wd(FormatVector::AccCoo);
                mask->validate_rw(FormatVector::AccDense);
                v->validate_rw(FormatVector::AccCoo);
                M->validate_rw(FormatMatrix::AccCsr);

                auto*      a     = M->acc_csr();
                auto*      vin   = v->acc_coo();
                auto*      out   = r->acc_coo();
                const uint nvals = vin->values;
                out->Ai.reserve(M->get_n_cols());
                out->Ax.reserve(M->get_n_cols());
                out->values = 0;
                if (nvals == 0 || a->values == 0) return Status::Ok;

                uint64_t products = 0;
                int      sorted   = 1;
                uint32_t count    = 0;
                // small frontiers (the many cheap levels of a traversal): probe, expansion and compaction in one go with a single host
                // synchronisation, for an op_add that does not depend on the fold order; the device-side product count decides whether
                // the speculative push ran (<= 4 M products and below the pull threshold) or the general forms below take over
                const bool order_free = op_assoc_comm(op_add) && !(TypeTraits<T>::id == SPLAHIP_FLOAT && (op_add == SPLAHIP_PLUS || op_add == SPLAHIP_MULT));
                if (order_free && nvals <= (1u << 16)) {
                    const uint64_t pull_at      = a->values / 8ull;// products * 8 >= nnz(M) -> pull
                    const uint32_t max_products = static_cast<uint32_t>(std::min<uint64_t>(1ull << 22, pull_at ? pull_at - 1 : 0));
                    int            done         = 0;
                    hip_check(splahip_vxm_push_small(TypeTraits<T>::id, M->get_n_rows(), M->get_n_cols(), a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(),
                                                     a->Ax.ptr(), mask->acc_dense()->Ax.ptr(), nvals, vin->Ai.template as<uint32_t>(), vin->Ax.ptr(),
                                                     out->Ai.template as<uint32_t>(), out->Ax.ptr(), max_products, &count, &products, &sorted, &done, op_mult, op_add,
                                                     op_sel, nullptr),
                              "splahip_vxm_push_small");
                    if (done) {
                        out->values = count;
                        return Status::Ok;
                    }
                } else {
                    hip_check(splahip_vxm_probe(nvals, vin->Ai.template as<uint32_t>(), a->Ap.template as<uint32_t>(), &products, &sorted, nullptr), "splahip_vxm_probe");
                }
                if (products == 0) return Status::Ok;

                const bool can_pull = sorted && op_assoc_comm(op_add) && !(TypeTraits<T>::id == SPLAHIP_FLOAT && op_add == SPLAHIP_MULT);// (fp32 products: left fold only)
                const bool pull     = can_pull && products * 8ull >= a->values;// push costs ~9x the products, pull costs nnz(M)
                if (pull) {
                    auto*     tr       = ensure_transpose<T>(M.get());
                    const int identity = nvals == M->get_n_rows() ? 1 : 0;
                    // a dense v (a PageRank step) over a transpose that keeps being used: the layout copies of M^T, same rent-or-buy rule as mxv
                    if (identity && tr->values > 0) {
                        tr->prepared = tr->prepared || M->acc_csr()->prepared;
                        maybe_build_copies(tr, M->get_n_cols(), M->get_n_rows());
                    }
                    (void) splahip_csr_plan_hint_products(tr->plan, products);// (the probe's count: a frontier whose rows hold much of the matrix is settled at the first hit)
                    hip_check(splahip_vxm_pull(tr->plan, TypeTraits<T>::id, M->get_n_rows(), M->get_n_cols(), tr->Ap.template as<uint32_t>(),
                                               tr->Aj.template as<uint32_t>(), tr->Ax.ptr(), mask->acc_dense()->Ax.ptr(), nvals,
                                               vin->Ai.template as<uint32_t>(), vin->Ax.ptr(), identity, out->Ai.template as<uint32_t>(),
                                               out->Ax.ptr(), &count, op_mult, op_add, op_sel, nullptr),
                              "splahip_vxm_pull");
                } else {
                    if (products > 0xFFFFFFF0ull) return Status::InvalidArgument;// uint32 product indices
                    hip_check(splahip_vxm_push_n(TypeTraits<T>::id, M->get_n_rows(), M->get_n_cols(), a->Ap.template as<uint32_t>(),
                                                 a->Aj.template as<uint32_t>(), a->Ax.ptr(), mask->acc_dense()->Ax.ptr(), nvals,
                                                 vin->Ai.template as<uint32_t>(), vin->Ax.ptr(), out->Ai.template as<uint32_t>(), out->Ax.ptr(),
                                                 &count, op_mult, op_add, op_sel, products, nullptr),
                              "splahip_vxm_push");
                }
                out->values = count;
                return Status::Ok;
            }
        };

        template<typename T>
        class Algo_mxmT_masked_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "mxmT_masked"; }
            std::string get_description() override { return "masked matrix x matrix-transposed product: sorted-row intersection per mask entry"; }

            Status execute(const DispatchContext& ctx) override {
                auto t    = ctx.task.template cast_safe<ScheduleTask_mxmT_masked>();
                auto R    = t->R.template cast_safe<TMatrix<T>>();
                auto mask = t->mask.template cast_safe<TMatrix<T>>();
                auto A    = t->A.template cast_safe<TMatrix<T>>();
                auto B    = t->B.template cast_safe<TMatrix<T>>();
                auto init = t->init.template cast_safe<TScalar<T>>();

                const int op_mult = device_binop<T>(t->op_multiply);
                const int op_add  = device_binop<T>(t->op_add);
                const int op_sel  = device_selop<T>(t->op_select);
                if (R->get_n_rows() != A->get_n_rows() || R->get_n_cols() != B->get_n_rows() || A->get_n_cols() != B->get_n_cols() ||
                    mask->get_n_rows() != R->get_n_rows() || mask->get_n_cols() != R->get_n_cols())
                    return Status::InvalidArgument;
                const bool user_ops = op_mult < 0 || op_add < 0 || op_sel < 0;// compiled at run time (jit.hip): a lane per mask entry, the reference's merge order

                R->validate_wd(FormatMatrix::AccCsr);
                mask->validate_rw(FormatMatrix::AccCsr);
                A->validate_rw(FormatMatrix::AccCsr);
                B->validate_rw(FormatMatrix::AccCsr);

                auto *    r = R->acc_csr(), *m = mask->acc_csr(), *a = A->acc_csr(), *b = B->acc_csr();
                uint32_t* Rj  = nullptr;
                void*     Rx  = nullptr;
                uint32_t  nnz = 0;
                r->Ap.reserve(R->get_n_rows() + 1);
                if (user_ops) {
                    const UserSrc src(op_mult, t->op_multiply, op_add, t->op_add, op_sel, &t->op_select);
                    const Status  js = jit_status(splahip_mxmT_masked_jit(TypeTraits<T>::id, R->get_n_rows(), m->values, m->Ap.template as<uint32_t>(), m->Aj.template as<uint32_t>(),
                                                                          m->Ax.ptr(), a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(), a->Ax.ptr(),
                                                                          b->Ap.template as<uint32_t>(), b->Aj.template as<uint32_t>(), b->Ax.ptr(), r->Ap.template as<uint32_t>(),
                                                                          &Rj, &Rx, &nnz, op_mult, src.pm, op_add, src.pa, op_sel, src.ps, word_of(init->get_value()), nullptr),
                                                  "splahip_mxmT_masked_jit");
                    if (js != Status::Ok) return js;
                } else
                    hip_check(splahip_mxmT_masked(TypeTraits<T>::id, R->get_n_rows(), m->values, m->Ap.template as<uint32_t>(), m->Aj.template as<uint32_t>(),
                                                  m->Ax.ptr(), a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(), a->Ax.ptr(),
                                                  b->Ap.template as<uint32_t>(), b->Aj.template as<uint32_t>(), b->Ax.ptr(), r->Ap.template as<uint32_t>(),
                                                  &Rj, &Rx, &nnz, op_mult, op_add, op_sel, word_of(init->get_value()), nullptr),
                              "splahip_mxmT_masked");
                r->drop_plan();
                r->Aj.adopt(Rj, nnz);
                r->Ax.adopt(Rx, nnz);
                r->values = nnz;
                return Status::Ok;
            }
        };

        template<typename T>
        class Algo_mxm_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "mxm"; }
            std::string get_description() override { return "matrix-matrix product: expand / stable sort / in-order compress over row blocks"; }

            Status execute(const DispatchContext& ctx) override {
                auto t    = ctx.task.template cast_safe<ScheduleTask_mxm>();
                auto R    = t->R.template cast_safe<TMatrix<T>>();
                auto A    = t->A.template cast_safe<TMatrix<T>>();
                auto B    = t->B.template cast_safe<TMatrix<T>>();
                auto init = t->init.template cast_safe<TScalar<T>>();

                const int op_mult = device_binop<T>(t->op_multiply);
                const int op_add  = device_binop<T>(t->op_add);
                if (R->get_n_rows() != A->get_n_rows() || R->get_n_cols() != B->get_n_cols() || A->get_n_cols() != B->get_n_rows())
                    return Status::InvalidArgument;
                const bool user_ops = op_mult < 0 || op_add < 0;// compiled at run time (jit.hip): expand / stable sort / in-order compress, the reference's fold order

                R->validate_wd(FormatMatrix::AccCsr);
                A->validate_rw(FormatMatrix::AccCsr);
                B->validate_rw(FormatMatrix::AccCsr);

                auto *    r = R->acc_csr(), *a = A->acc_csr(), *b = B->acc_csr();
                uint32_t* Rj  = nullptr;
                void*     Rx  = nullptr;
                uint32_t  nnz = 0;
                r->Ap.reserve(R->get_n_rows() + 1);
                if (user_ops) {
                    const UserSrc src(op_mult, t->op_multiply, op_add, t->op_add, 0, static_cast<const ref_ptr<OpSelect>*>(nullptr));
                    const Status  js = jit_status(splahip_mxm_jit(TypeTraits<T>::id, A->get_n_rows(), B->get_n_cols(), a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(),
                                                                  a->Ax.ptr(), b->Ap.template as<uint32_t>(), b->Aj.template as<uint32_t>(), b->Ax.ptr(),
                                                                  r->Ap.template as<uint32_t>(), &Rj, &Rx, &nnz, op_mult, src.pm, op_add, src.pa, word_of(init->get_value()), nullptr),
                                                  "splahip_mxm_jit");
                    if (js != Status::Ok) return js;
                } else
                    hip_check(splahip_mxm(TypeTraits<T>::id, A->get_n_rows(), B->get_n_cols(), a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(),
                                          a->Ax.ptr(), b->Ap.template as<uint32_t>(), b->Aj.template as<uint32_t>(), b->Ax.ptr(),
                                          r->Ap.template as<uint32_t>(), &Rj, &Rx, &nnz, nullptr, op_mult, op_add, word_of(init->get_value()), 0, nullptr),
                              "splahip_mxm");
                r->drop_plan();
                r->Aj.adopt(Rj, nnz);
                r->Ax.adopt(Rx, nnz);
                r->values = nnz;
                return Status::Ok;
            }
        };

        // ---- vector operations of a traversal step.  Format choice follows the reference's CPU algorithms: a sparse operand is used
        //      as sparse when it is the valid format, else dense (cpu_v_assign.hpp:59-64, cpu_v_eadd_fdb.hpp:60-67, cpu_v_count_mf.hpp:59-66)
        template<typename T>
        bool coo_valid(TVector<T>* v) { return v->is_valid(FormatVector::CpuCoo) || v->is_valid(FormatVector::AccCoo); }
        template<typename T>
        bool dense_valid(TVector<T>* v) { return v->is_valid(FormatVector::CpuDense) || v->is_valid(FormatVector::AccDense); }

        template<typename T>
        class Algo_v_assign_masked_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "v_assign_masked"; }
            std::string get_description() override { return "masked scalar assign into a dense vector"; }
            Status      execute(const DispatchContext& ctx) override {
                auto t     = ctx.task.template cast_safe<ScheduleTask_v_assign_masked>();
                auto r     = t->r.template cast_safe<TVector<T>>();
                auto mask  = t->mask.template cast_safe<TVector<T>>();
                auto value = t->value.template cast_safe<TScalar<T>>();
                const int op_assign = device_binop<T>(t->op_assign);
                const int op_sel    = device_selop<T>(t->op_select);
                if (r->get_n_rows() != mask->get_n_rows()) return Status::InvalidArgument;
                // a user-defined op (OpBinary::make_* / OpSelect::make_*): its body text is compiled for the device at run time (jit.hip), like the hot-path operators'
                const bool        user = op_assign < 0 || op_sel < 0;
                const std::string sa = op_assign < 0 ? t->op_assign->get_source_cl() : std::string(), ss = op_sel < 0 ? t->op_select->get_source_cl() : std::string();
                // sparse mask when its COO is valid, dense when only its dense image is, sparse otherwise (cpu_v_assign.hpp:59-64):
                // the two forms differ on absent entries (the dense form applies op_select to the fill value)
                const bool sparse = coo_valid(mask.get()) || !dense_valid(mask.get());
                r->validate_rwd(FormatVector::AccDense);
                if (sparse) {
                    mask->validate_rw(FormatVector::AccCoo);
                    auto* m = mask->acc_coo();
                    if (user)
                        return jit_status(splahip_v_assign_masked_jit(TypeTraits<T>::id, 1, m->values, r->acc_dense()->Ax.ptr(), m->Ai.template as<uint32_t>(), m->Ax.ptr(),
                                                                      word_of(value->get_value()), op_assign, op_assign < 0 ? sa.c_str() : nullptr, op_sel,
                                                                      op_sel < 0 ? ss.c_str() : nullptr, nullptr),
                                          "splahip_v_assign_masked_jit");
                    hip_check(splahip_v_assign_masked_sparse(TypeTraits<T>::id, m->values, r->acc_dense()->Ax.ptr(), m->Ai.template as<uint32_t>(), m->Ax.ptr(),
                                                             word_of(value->get_value()), op_assign, op_sel, nullptr),
                              "splahip_v_assign_masked_sparse");
                } else {
                    mask->validate_rw(FormatVector::AccDense);
                    if (user)
                        return jit_status(splahip_v_assign_masked_jit(TypeTraits<T>::id, 0, r->get_n_rows(), r->acc_dense()->Ax.ptr(), nullptr, mask->acc_dense()->Ax.ptr(),
                                                                      word_of(value->get_value()), op_assign, op_assign < 0 ? sa.c_str() : nullptr, op_sel,
                                                                      op_sel < 0 ? ss.c_str() : nullptr, nullptr),
                                          "splahip_v_assign_masked_jit");
                    hip_check(splahip_v_assign_masked_dense(TypeTraits<T>::id, r->get_n_rows(), r->acc_dense()->Ax.ptr(), mask->acc_dense()->Ax.ptr(),
                                                            word_of(value->get_value()), op_assign, op_sel, nullptr),
                              "splahip_v_assign_masked_dense");
                }
                return Status::Ok;
            }
        };

        template<typename T>
        class Algo_v_eadd_fdb_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "v_eadd_fdb"; }
            std::string get_description() override { return "element-wise add into a dense vector with feedback of the changed entries"; }
            Status      execute(const DispatchContext& ctx) override {
                auto t   = ctx.task.template cast_safe<ScheduleTask_v_eadd_fdb>();
                auto r   = t->r.template cast_safe<TVector<T>>();
                auto v   = t->v.template cast_safe<TVector<T>>();
                auto fdb = t->fdb.template cast_safe<TVector<T>>();
                const int op = device_binop<T>(t->op);
                const std::string src = op < 0 ? t->op->get_source_cl() : std::string();// a user-defined op: compiled for the device at run time (jit.hip)
                if (r->get_n_rows() != v->get_n_rows() || r->get_n_rows() != fdb->get_n_rows()) return Status::InvalidArgument;
                // sp2dn when r is dense and v's COO is valid, dn2dn when both are dense, sp2dn otherwise (cpu_v_eadd_fdb.hpp:60-67)
                const bool r_dense = dense_valid(r.get());
                const bool sparse  = (r_dense && coo_valid(v.get())) || !(r_dense && dense_valid(v.get()));
                r->validate_rwd(FormatVector::AccDense);
                if (sparse) {
                    v->validate_rw(FormatVector::AccCoo);
                    fdb->validate_wd(FormatVector::AccCoo);
                    auto *   in = v->acc_coo(), *out = fdb->acc_coo();
                    uint32_t count = 0;
                    out->Ai.reserve(in->values);
                    out->Ax.reserve(in->values);
                    if (op < 0) {
                        const Status js = jit_status(splahip_v_eadd_fdb_sparse_jit(TypeTraits<T>::id, in->values, r->acc_dense()->Ax.ptr(), in->Ai.template as<uint32_t>(), in->Ax.ptr(),
                                                                                   out->Ai.template as<uint32_t>(), out->Ax.ptr(), &count, src.c_str(), nullptr),
                                                     "splahip_v_eadd_fdb_sparse_jit");
                        if (js == Status::Ok) out->values = count;
                        return js;
                    }
                    hip_check(splahip_v_eadd_fdb_sparse(TypeTraits<T>::id, in->values, r->acc_dense()->Ax.ptr(), in->Ai.template as<uint32_t>(), in->Ax.ptr(),
                                                        out->Ai.template as<uint32_t>(), out->Ax.ptr(), &count, op, nullptr),
                              "splahip_v_eadd_fdb_sparse");
                    out->values = count;
                } else {
                    v->validate_rw(FormatVector::AccDense);
                    fdb->validate_wd(FormatVector::AccDense);
                    if (op < 0)
                        return jit_status(splahip_v_eadd_fdb_dense_jit(TypeTraits<T>::id, r->get_n_rows(), r->acc_dense()->Ax.ptr(), v->acc_dense()->Ax.ptr(),
                                                                       fdb->acc_dense()->Ax.ptr(), word_of(fdb->get_fill_value()), src.c_str(), nullptr),
                                          "splahip_v_eadd_fdb_dense_jit");
                    hip_check(splahip_v_eadd_fdb_dense(TypeTraits<T>::id, r->get_n_rows(), r->acc_dense()->Ax.ptr(), v->acc_dense()->Ax.ptr(),
                                                       fdb->acc_dense()->Ax.ptr(), word_of(fdb->get_fill_value()), op, nullptr),
                              "splahip_v_eadd_fdb_dense");
                }
                return Status::Ok;
            }
        };

        template<typename T>
        class Algo_v_count_mf_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "v_count_mf"; }
            std::string get_description() override { return "count of stored / non-fill entries"; }
            Status      execute(const DispatchContext& ctx) override {
                auto t = ctx.task.template cast_safe<ScheduleTask_v_count_mf>();
                auto v = t->v.template cast_safe<TVector<T>>();
                // a sparse format that is valid answers with its stored count, explicit zeros included (cpu_v_count_mf.hpp:77,88)
                if (v->is_valid(FormatVector::CpuDok)) return t->r->set_uint(v->dok()->values);
                if (v->is_valid(FormatVector::CpuCoo)) return t->r->set_uint(v->coo()->values);
                if (v->is_valid(FormatVector::AccCoo)) return t->r->set_uint(v->acc_coo()->values);
                v->validate_rw(FormatVector::AccDense);
                // count by COMPACTING: the dense -> sparse conversion counts the same entries (`!= fill`) on its way, costs no more kernel time than the counting kernel
                // (two streaming passes of 4 B per row + 8 B per stored entry) and the same single host round trip — and the traversal loops that ask for this count (bfs:
                // the next frontier, sssp: the changed entries) then hand the vector to a push step, which would convert it anyway and wait for the count a second time.
                // SPLA_HIP_COUNT_BY_COMPACTION=0: the counting kernel
                static const bool by_compaction = [] {
                    const char* e = std::getenv("SPLA_HIP_COUNT_BY_COMPACTION");
                    return !(e && e[0] == '0');
                }();
                if (by_compaction) {
                    // ... the entries are written only if at most a quarter of the rows hold one (a push step is chosen for sparser vectors than that; a denser one
                    // stays dense-only and costs the two counting passes)
                    v->validate_ctor(FormatVector::AccCoo);
                    auto&          st  = v->storage();
                    auto*          out = st.template get<HipCooVec<T>>(static_cast<int>(FormatVector::AccCoo));
                    const uint32_t n   = v->get_n_rows();
                    out->Ai.reserve(n);
                    out->Ax.reserve(n);
                    uint32_t count = 0;
                    hip_check(splahip_vec_dense_to_coo_limited(TypeTraits<T>::id, n, v->acc_dense()->Ax.ptr(), word_of(v->get_fill_value()), out->Ai.template as<uint32_t>(),
                                                               out->Ax.ptr(), n / 4u, &count, nullptr),
                              "splahip_vec_dense_to_coo_limited");
                    if (count <= n / 4u) {
                        out->values = count;
                        st.valid.set(static_cast<int>(FormatVector::AccCoo));
                    }
                    return t->r->set_uint(count);
                }
                uint32_t count = 0;
                hip_check(splahip_v_count_mf_dense(TypeTraits<T>::id, v->get_n_rows(), v->acc_dense()->Ax.ptr(), word_of(v->get_fill_value()), &count, nullptr),
                          "splahip_v_count_mf_dense");
                return t->r->set_uint(count);
            }
        };

        // ---- PageRank step remainder and triangle-count tail
        template<typename T>
        T value_of_word(uint32_t w) {
            T v;
            std::memcpy(&v, &w, sizeof(T));
            return v;
        }

        template<typename T>
        class Algo_v_eadd_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "v_eadd"; }
            std::string get_description() override { return "element-wise add of two vectors (dense x dense, or the sorted merge of two sparse vectors)"; }
            Status      execute(const DispatchContext& ctx) override {
                auto t = ctx.task.template cast_safe<ScheduleTask_v_eadd>();
                auto r = t->r.template cast_safe<TVector<T>>();
                auto u = t->u.template cast_safe<TVector<T>>();
                auto v = t->v.template cast_safe<TVector<T>>();
                const int op = device_binop<T>(t->op);
                const std::string src = op < 0 ? t->op->get_source_cl() : std::string();// a user-defined op: compiled for the device at run time (jit.hip)
                if (r->get_n_rows() != u->get_n_rows() || r->get_n_rows() != v->get_n_rows()) return Status::InvalidArgument;
                // sparse x sparse when both COO images are valid, dense x dense when both dense images are, sparse otherwise (cpu_v_eadd.hpp:59-66)
                const bool dense = !(coo_valid(u.get()) && coo_valid(v.get())) && dense_valid(u.get()) && dense_valid(v.get());
                if (dense) {
                    u->validate_rw(FormatVector::AccDense);
                    v->validate_rw(FormatVector::AccDense);
                    if (r.get() != u.get() && r.get() != v.get()) r->validate_wd(FormatVector::AccDense);
                    else
                        r->validate_rwd(FormatVector::AccDense);// in place: keep the operand's image, drop its other formats
                    if (op < 0)
                        return jit_status(splahip_v_eadd_dense_jit(TypeTraits<T>::id, r->get_n_rows(), r->acc_dense()->Ax.ptr(), u->acc_dense()->Ax.ptr(), v->acc_dense()->Ax.ptr(),
                                                                   src.c_str(), nullptr),
                                          "splahip_v_eadd_dense_jit");
                    hip_check(splahip_v_eadd_dense(TypeTraits<T>::id, r->get_n_rows(), r->acc_dense()->Ax.ptr(), u->acc_dense()->Ax.ptr(), v->acc_dense()->Ax.ptr(), op, nullptr),
                              "splahip_v_eadd_dense");
                    return Status::Ok;
                }
                if (r.get() == u.get() || r.get() == v.get()) return Status::InvalidArgument;// the merge cannot run in place
                u->validate_rw(FormatVector::AccCoo);
                v->validate_rw(FormatVector::AccCoo);
                r->validate_wd(FormatVector::AccCoo);
                auto *   a = u->acc_coo(), *b = v->acc_coo(), *out = r->acc_coo();
                uint32_t count = 0;
                out->Ai.reserve(a->values + b->values);
                out->Ax.reserve(a->values + b->values);
                if (op < 0) {
                    const Status js = jit_status(splahip_v_eadd_sparse_jit(TypeTraits<T>::id, a->values, a->Ai.template as<uint32_t>(), a->Ax.ptr(), word_of(u->get_fill_value()),
                                                                           b->values, b->Ai.template as<uint32_t>(), b->Ax.ptr(), word_of(v->get_fill_value()),
                                                                           out->Ai.template as<uint32_t>(), out->Ax.ptr(), &count, src.c_str(), nullptr),
                                                 "splahip_v_eadd_sparse_jit");
                    if (js == Status::Ok) out->values = count;
                    return js;
                }
                hip_check(splahip_v_eadd_sparse(TypeTraits<T>::id, a->values, a->Ai.template as<uint32_t>(), a->Ax.ptr(), word_of(u->get_fill_value()), b->values,
                                                b->Ai.template as<uint32_t>(), b->Ax.ptr(), word_of(v->get_fill_value()), out->Ai.template as<uint32_t>(),
                                                out->Ax.ptr(), &count, op, nullptr),
                          "splahip_v_eadd_sparse");
                out->values = count;
                return Status::Ok;
            }
        };

        template<typename T>
        class Algo_v_reduce_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "v_reduce"; }
            std::string get_description() override { return "fold of a vector's stored values into a scalar"; }
            Status      execute(const DispatchContext& ctx) override {
                auto t = ctx.task.template cast_safe<ScheduleTask_v_reduce>();
                auto r = t->r.template cast_safe<TScalar<T>>();
                auto s = t->s.template cast_safe<TScalar<T>>();
                auto v = t->v.template cast_safe<TVector<T>>();
                const int op = device_binop<T>(t->op_reduce);
                const std::string src = op < 0 ? t->op_reduce->get_source_cl() : std::string();// a user-defined op: compiled for the device at run time (jit.hip)
                // the stored values of the sparse image when it is valid, every entry of the dense image when only that is (cpu_v_reduce.hpp:58-67)
                const bool  sparse = coo_valid(v.get()) || !dense_valid(v.get());
                uint32_t    bits   = 0, n;
                const void* x;
                if (sparse) {
                    v->validate_rw(FormatVector::AccCoo);
                    n = v->acc_coo()->values;
                    x = v->acc_coo()->Ax.ptr();
                } else {
                    v->validate_rw(FormatVector::AccDense);
                    n = v->get_n_rows();
                    x = v->acc_dense()->Ax.ptr();
                }
                if (op < 0) {
                    const Status js = jit_status(splahip_reduce_jit(TypeTraits<T>::id, n, x, word_of(s->get_value()), &bits, src.c_str(), nullptr), "splahip_reduce_jit");
                    if (js == Status::Ok) r->set_value(value_of_word<T>(bits));
                    return js;
                }
                hip_check(splahip_reduce(TypeTraits<T>::id, n, x, word_of(s->get_value()), &bits, op, nullptr), "splahip_reduce");
                r->set_value(value_of_word<T>(bits));
                return Status::Ok;
            }
        };

        template<typename T>
        class Algo_m_reduce_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "m_reduce"; }
            std::string get_description() override { return "fold of a matrix's stored values into a scalar"; }
            Status      execute(const DispatchContext& ctx) override {
                auto t = ctx.task.template cast_safe<ScheduleTask_m_reduce>();
                auto r = t->r.template cast_safe<TScalar<T>>();
                auto s = t->s.template cast_safe<TScalar<T>>();
                auto M = t->M.template cast_safe<TMatrix<T>>();
                const int op = device_binop<T>(t->op_reduce);
                M->validate_rw(FormatMatrix::AccCsr);// values in row-major order (cpu_m_reduce.hpp:134-156)
                uint32_t bits = 0;
                if (op < 0) {// a user-defined op: compiled for the device at run time (jit.hip)
                    const std::string src = t->op_reduce->get_source_cl();
                    const Status      js  = jit_status(splahip_reduce_jit(TypeTraits<T>::id, M->acc_csr()->values, M->acc_csr()->Ax.ptr(), word_of(s->get_value()), &bits, src.c_str(), nullptr),
                                                       "splahip_reduce_jit");
                    if (js == Status::Ok) r->set_value(value_of_word<T>(bits));
                    return js;
                }
                hip_check(splahip_reduce(TypeTraits<T>::id, M->acc_csr()->values, M->acc_csr()->Ax.ptr(), word_of(s->get_value()), &bits, op, nullptr), "splahip_reduce");
                r->set_value(value_of_word<T>(bits));
                return Status::Ok;
            }
        };

        // R = op_apply(M)^T through the accelerator CSR (the reference's CPU algorithm picks csr / lil / dok by what is valid and
        // produces the same matrix in each, cpu_m_transpose.hpp:57-75; here every format converges on AccCsr first)
        template<typename T>
        class Algo_m_transpose_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "m_transpose"; }
            std::string get_description() override { return "transpose with a unary op: stable radix sort of the entries by column"; }
            Status      execute(const DispatchContext& ctx) override {
                auto t = ctx.task.template cast_safe<ScheduleTask_m_transpose>();
                auto R = t->R.template cast_safe<TMatrix<T>>();
                auto M = t->M.template cast_safe<TMatrix<T>>();
                const int op = device_unop<T>(t->op_apply);
                if (R->get_n_rows() != M->get_n_cols() || R->get_n_cols() != M->get_n_rows() || R.get() == M.get()) return Status::InvalidArgument;
                R->validate_wd(FormatMatrix::AccCsr);
                M->validate_rw(FormatMatrix::AccCsr);
                auto *r = R->acc_csr(), *m = M->acc_csr();
                r->drop_plan();
                r->Ap.reserve(R->get_n_rows() + 1);
                r->Aj.reserve(m->values);
                r->Ax.reserve(m->values);
                hip_check(splahip_m_transpose(TypeTraits<T>::id, M->get_n_rows(), M->get_n_cols(), m->values, m->Ap.template as<uint32_t>(),
                                              m->Aj.template as<uint32_t>(), m->Ax.ptr(), r->Ap.template as<uint32_t>(), r->Aj.template as<uint32_t>(),
                                              r->Ax.ptr(), op < 0 ? SPLAHIP_IDENTITY : op, nullptr),
                          "splahip_m_transpose");
                r->values = m->values;
                if (op < 0) {// a user-defined op: the transposed values go through the run-time-compiled map (jit.hip)
                    const std::string src = t->op_apply->get_source_cl();
                    return jit_status(splahip_map_jit(TypeTraits<T>::id, r->values, r->Ax.ptr(), r->Ax.ptr(), src.c_str(), nullptr), "splahip_map_jit");
                }
                return Status::Ok;
            }
        };

        // r = op(v): over the stored entries when the sparse image is the valid one, else over every index (cpu_v_map.hpp:57-68)
        template<typename T>
        class Algo_v_map_hip final : public RegistryAlgo {
        public:
            std::string get_name() override { return "v_map"; }
            std::string get_description() override { return "element-wise unary op over a vector"; }
            Status      execute(const DispatchContext& ctx) override {
                auto t = ctx.task.template cast_safe<ScheduleTask_v_map>();
                auto r = t->r.template cast_safe<TVector<T>>();
                auto v = t->v.template cast_safe<TVector<T>>();
                const int op = device_unop<T>(t->op);
                const std::string src = op < 0 ? t->op->get_source_cl() : std::string();// a user-defined op: compiled for the device at run time (jit.hip)
                if (r->get_n_rows() != v->get_n_rows()) return Status::InvalidArgument;
                const bool sparse = coo_valid(v.get()) || !dense_valid(v.get());
                if (sparse) {
                    v->validate_rw(FormatVector::AccCoo);
                    const uint n = v->acc_coo()->values;
                    if (r.get() != v.get()) r->validate_wd(FormatVector::AccCoo);
                    else
                        r->validate_rwd(FormatVector::AccCoo);
                    auto *in = v->acc_coo(), *out = r->acc_coo();
                    if (in != out) {
                        out->Ai.reserve(n);
                        out->Ax.reserve(n);
                        if (n) hip_check(splahip_memcpy_d2d(out->Ai.ptr(), in->Ai.ptr(), std::size_t(n) * 4, nullptr), "index copy");
                    }
                    out->values = n;
                    if (op < 0) return jit_status(splahip_map_jit(TypeTraits<T>::id, n, in->Ax.ptr(), out->Ax.ptr(), src.c_str(), nullptr), "splahip_map_jit");
                    hip_check(splahip_map(TypeTraits<T>::id, n, in->Ax.ptr(), out->Ax.ptr(), op, nullptr), "splahip_map");
                } else {
                    v->validate_rw(FormatVector::AccDense);
                    if (r.get() != v.get()) r->validate_wd(FormatVector::AccDense);
                    else
                        r->validate_rwd(FormatVector::AccDense);
                    if (op < 0)
                        return jit_status(splahip_map_jit(TypeTraits<T>::id, v->get_n_rows(), v->acc_dense()->Ax.ptr(), r->acc_dense()->Ax.ptr(), src.c_str(), nullptr), "splahip_map_jit");
                    hip_check(splahip_map(TypeTraits<T>::id, v->get_n_rows(), v->acc_dense()->Ax.ptr(), r->acc_dense()->Ax.ptr(), op, nullptr), "splahip_map");
                }
                return Status::Ok;
            }
        };

        template<template<typename> class Algo>
        void add3(Registry* g, const std::string& name) {
            g->add(name + "_I__hip", std::make_shared<Algo<T_INT>>());
            g->add(name + "_U__hip", std::make_shared<Algo<T_UINT>>());
            g->add(name + "_F__hip", std::make_shared<Algo<T_FLOAT>>());
        }
    }// namespace

    void register_algo_hip(Registry* g) {
        add3<Algo_mxv_masked_hip>(g, "mxv_masked");
        add3<Algo_vxm_masked_hip>(g, "vxm_masked");
        add3<Algo_mxmT_masked_hip>(g, "mxmT_masked");
        add3<Algo_mxm_hip>(g, "mxm");
        add3<Algo_v_assign_masked_hip>(g, "v_assign_masked");
        add3<Algo_v_eadd_fdb_hip>(g, "v_eadd_fdb");
        add3<Algo_v_count_mf_hip>(g, "v_count_mf");
        add3<Algo_v_eadd_hip>(g, "v_eadd");
        add3<Algo_v_reduce_hip>(g, "v_reduce");
        add3<Algo_m_reduce_hip>(g, "m_reduce");
        add3<Algo_m_transpose_hip>(g, "m_transpose");
        add3<Algo_v_map_hip>(g, "v_map");
    }

}// namespace spla
