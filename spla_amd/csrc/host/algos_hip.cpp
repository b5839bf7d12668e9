// algos_hip.cpp — the HIP accelerator's RegistryAlgo set for the hot path, registered under "<op>_<I|U|F>__hip".
// Each execute() mirrors the shape of the reference's accelerated algorithms (validate the operands into accelerator
// formats, then launch): reference src/opencl/cl_mxv.hpp:65-121, cl_vxm.hpp:60-173, cl_mxmT_masked.hpp:60-127 — with
// the launches going through the kernel-level C ABI (include/spla_hip.h) instead of JIT-built OpenCL programs.
#include "storage.hpp"

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
        void ensure_plan(HipCsr<T>* a, uint n_rows) {
            if (!a->plan) hip_check(splahip_csr_plan_create(&a->plan, n_rows, a->values, a->Ap.template as<uint32_t>(), nullptr), "csr plan");
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
                if (op_mult < 0 || op_add < 0 || op_sel < 0) {
                    SPLA_LOG(Status::NotImplemented, "user-defined ops have no device code in this build");
                    return Status::NotImplemented;
                }
                if (r->get_n_rows() != M->get_n_rows() || mask->get_n_rows() != M->get_n_rows() || v->get_n_rows() != M->get_n_cols())
                    return Status::InvalidArgument;

                // same format requests as the reference's accelerated mxv (cl_mxv.hpp:91-94); r is write-discard:
                // every row is overwritten (cpu_mxv.hpp:88-103)
                r->validate_wd(FormatVector::AccDense);
                mask->validate_rw(FormatVector::AccDense);
                v->validate_rw(FormatVector::AccDense);
                M->validate_rw(FormatMatrix::AccCsr);

                auto* a = M->acc_csr();
                ensure_plan(a, M->get_n_rows());
                const bool early = t->get_desc_or_default()->get_early_exit();
                hip_check(splahip_mxv_masked(a->plan, TypeTraits<T>::id, M->get_n_rows(), M->get_n_cols(),
                                             a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(), a->Ax.ptr(),
                                             mask->acc_dense()->Ax.ptr(), v->acc_dense()->Ax.ptr(), r->acc_dense()->Ax.ptr(),
                                             op_mult, op_add, op_sel, word_of(init->get_value()), early ? 1 : 0, nullptr),
                          "splahip_mxv_masked");
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
    }

}// namespace spla
