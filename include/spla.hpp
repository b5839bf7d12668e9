// spla.hpp — C++ operator API of the MI355X backend build: the hot-path subset of the reference's public API
// (reference include/spla.hpp and include/spla/*.hpp), written from scratch with the same names, argument meaning
// and status behaviour, so code written against the reference for this path compiles and behaves the same:
//
//   exec_mxv_masked / exec_vxm_masked / exec_mxm / exec_mxmT_masked      reference include/spla/exec.hpp:75-197
//   Matrix / Vector / Scalar / MemView / Descriptor / Type / Op*         reference include/spla/{matrix,vector,scalar,memview,descriptor,type,op}.hpp
//   Library (AcceleratorType gains `Hip`)                                 reference include/spla/library.hpp:66-132, config.hpp:89-94
//   ref_ptr / RefCnt (intrusive, count starts at 0)                       reference include/spla/ref.hpp:55-243
//   ScheduleTask / Schedule (ordered, serial)                             reference include/spla/schedule.hpp
//
// What differs, on purpose: this build carries ONE backend, the HIP accelerator.  There is no CPU compute path
// behind exec_* (the reference's sequential CPU algorithms live only in oracle/ as the parity checker); without a
// usable MI355X every exec_* call returns Status::NoAcceleration.  Host-side object operations (build / read /
// set / get / format conversion) are plain C++ and work without a GPU.
#pragma once

#include <atomic>
#include <cassert>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <filesystem>
#include <functional>
#include <iostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace spla {

    using uint = std::uint32_t;

    // ---- status / enums: numeric values are part of the C ABI (reference include/spla/config.hpp:62-142) -----------
    enum class Status : uint {
        Ok               = 0,
        Error            = 1,
        NoAcceleration   = 2,
        PlatformNotFound = 3,
        DeviceNotFound   = 4,
        InvalidState     = 5,
        InvalidArgument  = 6,
        NoValue          = 7,
        CompilationError = 8,
        NotImplemented   = 1024
    };

    enum class AcceleratorType : uint {
        None   = 0,
        OpenCL = 1,// reference value; not available in this build
        Hip    = 2 // the MI355X (gfx950) backend
    };

    enum class FormatMatrix : uint { CpuLil = 0,
                                     CpuDok = 1,
                                     CpuCoo = 2,
                                     CpuCsr = 3,
                                     CpuCsc = 4,
                                     AccCoo = 5,
                                     AccCsr = 6,
                                     AccCsc = 7,
                                     Count  = 8 };

    enum class FormatVector : uint { CpuDok   = 0,
                                     CpuDense = 1,
                                     CpuCoo   = 2,
                                     AccDense = 3,
                                     AccCoo   = 4,
                                     Count    = 5 };

    using MessageCallback  = std::function<void(Status, const std::string& msg, const std::string& file, const std::string& function, int line)>;
    using ScheduleCallback = std::function<void()>;

    using T_BOOL  = bool;
    using T_INT   = std::int32_t;
    using T_UINT  = std::uint32_t;
    using T_FLOAT = float;

    // ---- intrusive reference counting ---------------------------------------------------------------------------------
    class RefCnt {
    public:
        virtual ~RefCnt() = default;
        std::int32_t add_ref() const { return m_refs.fetch_add(1, std::memory_order_relaxed) + 1; }
        std::int32_t rel_ref() const {
            const std::int32_t left = m_refs.fetch_sub(1, std::memory_order_acq_rel) - 1;
            if (left == 0) delete this;
            return left;
        }
        std::int32_t get_refs() const { return m_refs.load(std::memory_order_relaxed); }
        bool         is_unique() const { return get_refs() == 1; }

    private:
        mutable std::atomic<std::int32_t> m_refs{0};
    };

    template<typename T>
    class ref_ptr {
    public:
        ref_ptr() = default;
        explicit ref_ptr(T* p) : m_p(p) { grab(); }
        ref_ptr(const ref_ptr& o) : m_p(o.m_p) { grab(); }
        ref_ptr(ref_ptr&& o) noexcept : m_p(o.m_p) { o.m_p = nullptr; }
        template<typename G>
        ref_ptr(const ref_ptr<G>& o) : m_p(o.get()) { grab(); }
        ~ref_ptr() { drop(); }

        ref_ptr& operator=(const ref_ptr& o) {
            if (m_p != o.m_p) {
                T* old = m_p;
                m_p    = o.m_p;
                grab();
                if (old) old->rel_ref();
            }
            return *this;
        }
        ref_ptr& operator=(ref_ptr&& o) noexcept {
            if (this != &o) {
                drop();
                m_p   = o.m_p;
                o.m_p = nullptr;
            }
            return *this;
        }

        bool operator==(const ref_ptr& o) const { return m_p == o.m_p; }
        bool operator!=(const ref_ptr& o) const { return m_p != o.m_p; }
        explicit operator bool() const { return m_p != nullptr; }
        [[nodiscard]] bool is_null() const { return m_p == nullptr; }
        [[nodiscard]] bool is_not_null() const { return m_p != nullptr; }
        T*                 operator->() const { return m_p; }
        T&                 operator*() const { return *m_p; }
        T*                 get() const { return m_p; }

        void reset(T* p = nullptr) {
            drop();
            m_p = p;
            grab();
        }
        // hand the pointer (with its reference) to the caller: how the C ABI returns owned handles
        T* release() {
            T* p = m_p;
            m_p  = nullptr;
            return p;
        }
        T* ref_and_get() {
            grab();
            return m_p;
        }

        template<class G>
        [[nodiscard]] bool is() const { return !m_p || dynamic_cast<G*>(m_p); }
        template<class G>
        ref_ptr<G> as() const { return ref_ptr<G>(static_cast<G*>(m_p)); }
        template<class G>
        ref_ptr<G> cast() const { return ref_ptr<G>(dynamic_cast<G*>(m_p)); }
        template<class G>
        ref_ptr<G> cast_safe() const {// type mismatch -> exception -> Status::Error in the dispatcher
            G* g = dynamic_cast<G*>(m_p);
            if (m_p && !g) throw std::runtime_error("failed to do dynamic cast");
            return ref_ptr<G>(g);
        }

    private:
        void grab() {
            if (m_p) m_p->add_ref();
        }
        void drop() {
            if (m_p) m_p->rel_ref();
            m_p = nullptr;
        }
        T* m_p = nullptr;
    };

    template<typename T, typename... A>
    ref_ptr<T> make_ref(A&&... args) { return ref_ptr<T>(new T(std::forward<A>(args)...)); }

    // ---- objects --------------------------------------------------------------------------------------------------------
    class Object : public RefCnt {
    public:
        ~Object() override                                     = default;
        virtual void               set_label(std::string label) = 0;
        virtual const std::string& get_label() const            = 0;
    };

    class Type : public Object {
    public:
        virtual const std::string& get_name()        = 0;// "INT" / "UINT" / "FLOAT" / "BOOL"
        virtual const std::string& get_code()        = 0;// "I" / "U" / "F" / "B"
        virtual const std::string& get_cpp()         = 0;
        virtual const std::string& get_description() = 0;
        virtual int                get_size()        = 0;
        virtual int                get_id()          = 0;
    };
    extern ref_ptr<Type> BOOL;
    extern ref_ptr<Type> INT;
    extern ref_ptr<Type> UINT;
    extern ref_ptr<Type> FLOAT;

    class MemView final : public RefCnt {
    public:
        Status      read(std::size_t offset, std::size_t size, void* dst);
        Status      write(std::size_t offset, std::size_t size, const void* src);
        void*       get_buffer() const { return m_buffer; }
        std::size_t get_size() const { return m_size; }
        bool        is_mutable() const { return m_mutable; }

        static ref_ptr<MemView> make(void* buffer, std::size_t size, bool is_mutable = false);
        static ref_ptr<MemView> make();

    private:
        void*       m_buffer  = nullptr;
        std::size_t m_size    = 0;
        bool        m_mutable = false;
    };

    class Scalar : public Object {
    public:
        virtual ref_ptr<Type> get_type()                     = 0;
        virtual Status        set_int(std::int32_t value)    = 0;
        virtual Status        set_uint(std::uint32_t value)  = 0;
        virtual Status        set_float(float value)         = 0;
        virtual Status        get_int(std::int32_t& value)   = 0;
        virtual Status        get_uint(std::uint32_t& value) = 0;
        virtual Status        get_float(float& value)        = 0;
        virtual T_INT         as_int()                       = 0;
        virtual T_UINT        as_uint()                      = 0;
        virtual T_FLOAT       as_float()                     = 0;

        static ref_ptr<Scalar> make(const ref_ptr<Type>& type);
        static ref_ptr<Scalar> make_int(std::int32_t value);
        static ref_ptr<Scalar> make_uint(std::uint32_t value);
        static ref_ptr<Scalar> make_float(float value);
    };

    // per-call options (reference include/spla/descriptor.hpp:46-82); only early_exit is honoured, by mxv
    class Descriptor final : public Object {
    public:
        enum class TraversalMode { Push,
                                   Pull,
                                   PushPull };
        void  set_traversal_mode(TraversalMode value) { m_mode = value; }
        void  set_front_factor(float value) { m_front_factor = value; }
        void  set_early_exit(bool value) { m_early_exit = value; }
        void  set_struct_only(bool value) { m_struct_only = value; }
        // (an addition of this build) fp32 sums of this call run on the fixed-order kernels: bit-identical from run to run (include/spla_hip.h: splahip_set_deterministic)
        void  set_deterministic(bool value) { m_deterministic = value; }
        bool  get_deterministic() const { return m_deterministic; }
        // (an addition of this build) fp32 sums of this call are folded in the reference CPU backend's order — strictly left to right from init, src/cpu/cpu_mxv.hpp:94-99 —
        // and equal its results bit for bit, at a few times the default call's time (include/spla_hip.h: splahip_set_reference_order)
        void  set_reference_order(bool value) { m_reference_order = value; }
        bool  get_reference_order() const { return m_reference_order; }
        bool  get_push_only() const { return m_mode == TraversalMode::Push; }
        bool  get_pull_only() const { return m_mode == TraversalMode::Pull; }
        bool  get_push_pull() const { return m_mode == TraversalMode::PushPull; }
        float get_front_factor() const { return m_front_factor; }
        bool  get_early_exit() const { return m_early_exit; }
        bool  get_struct_only() const { return m_struct_only; }

        void               set_label(std::string label) override { m_label = std::move(label); }
        const std::string& get_label() const override { return m_label; }

        static ref_ptr<Descriptor> make();

    private:
        std::string   m_label;
        TraversalMode m_mode         = TraversalMode::PushPull;
        float         m_front_factor = 0.1f;
        bool          m_early_exit   = false;
        bool          m_struct_only  = false;
        bool          m_deterministic = false;
        bool          m_reference_order = false;
    };

    // ---- operations -----------------------------------------------------------------------------------------------------
    class Op : public Object {
    public:
        virtual ref_ptr<Type> get_type_res()  = 0;
        virtual std::string   get_name()      = 0;// e.g. "PLUS_INT"
        virtual std::string   get_key()       = 0;// e.g. "PLUS_III"
        virtual std::string   get_source_cl() = 0;// body text "(T a, T b) { ... }"; valid HIP device code as well
    };
    class OpBinary : public Op {
    public:
        virtual ref_ptr<Type> get_type_arg_0() = 0;
        virtual ref_ptr<Type> get_type_arg_1() = 0;
        static ref_ptr<OpBinary> make_int(std::string name, std::string code, std::function<T_INT(T_INT, T_INT)> function);
        static ref_ptr<OpBinary> make_uint(std::string name, std::string code, std::function<T_UINT(T_UINT, T_UINT)> function);
        static ref_ptr<OpBinary> make_float(std::string name, std::string code, std::function<T_FLOAT(T_FLOAT, T_FLOAT)> function);
    };
    class OpUnary : public Op {
    public:
        virtual ref_ptr<Type> get_type_arg_0() = 0;
        static ref_ptr<OpUnary> make_int(std::string name, std::string code, std::function<T_INT(T_INT)> function);
        static ref_ptr<OpUnary> make_uint(std::string name, std::string code, std::function<T_UINT(T_UINT)> function);
        static ref_ptr<OpUnary> make_float(std::string name, std::string code, std::function<T_FLOAT(T_FLOAT)> function);
    };
    class OpSelect : public Op {
    public:
        virtual ref_ptr<Type> get_type_arg_0() = 0;
        static ref_ptr<OpSelect> make_int(std::string name, std::string code, std::function<bool(T_INT)> function);
        static ref_ptr<OpSelect> make_uint(std::string name, std::string code, std::function<bool(T_UINT)> function);
        static ref_ptr<OpSelect> make_float(std::string name, std::string code, std::function<bool(T_FLOAT)> function);
    };

    // built-in binary ops (reference src/op.cpp:194-241) and select ops (:243-266); X(name, valid for FLOAT)
#define SPLA_BINARY_OPS(X) \
    X(PLUS, 1) X(MINUS, 1) X(MULT, 1) X(DIV, 1) X(MINUS_POW2, 1) X(FIRST, 1) X(SECOND, 1) X(BONE, 1) \
    X(MIN, 1) X(MAX, 1) X(LOR, 1) X(LAND, 1) X(BOR, 0) X(BAND, 0) X(BXOR, 0)
#define SPLA_SELECT_OPS(X) \
    X(EQZERO) X(NQZERO) X(GTZERO) X(GEZERO) X(LTZERO) X(LEZERO) X(ALWAYS) X(NEVER)

    // built-in unary ops (reference src/op.cpp:99-133); X(name, for INT/UINT, for FLOAT)
#define SPLA_UNARY_OPS(X) \
    X(IDENTITY, 1, 1) X(AINV, 1, 1) X(MINV, 1, 1) X(LNOT, 1, 1) X(UONE, 1, 1) X(ABS, 1, 1) X(BNOT, 1, 0) \
    X(SQRT, 0, 1) X(LOG, 0, 1) X(EXP, 0, 1) X(SIN, 0, 1) X(COS, 0, 1) X(TAN, 0, 1) X(ASIN, 0, 1) X(ACOS, 0, 1) X(ATAN, 0, 1) \
    X(CEIL, 0, 1) X(FLOOR, 0, 1) X(ROUND, 0, 1) X(TRUNC, 0, 1)
#define SPLA_DECL_UN(NAME, HASI, HASF)  \
    extern ref_ptr<OpUnary> NAME##_INT; \
    extern ref_ptr<OpUnary> NAME##_UINT;\
    extern ref_ptr<OpUnary> NAME##_FLOAT;/* null where the reference has no such op */
    SPLA_UNARY_OPS(SPLA_DECL_UN)
#undef SPLA_DECL_UN

#define SPLA_DECL_BIN(NAME, HASF)             \
    extern ref_ptr<OpBinary> NAME##_INT;      \
    extern ref_ptr<OpBinary> NAME##_UINT;     \
    extern ref_ptr<OpBinary> NAME##_FLOAT;/* null for the bitwise ops */
    SPLA_BINARY_OPS(SPLA_DECL_BIN)
#undef SPLA_DECL_BIN
#define SPLA_DECL_SEL(NAME)               \
    extern ref_ptr<OpSelect> NAME##_INT;  \
    extern ref_ptr<OpSelect> NAME##_UINT; \
    extern ref_ptr<OpSelect> NAME##_FLOAT;
    SPLA_SELECT_OPS(SPLA_DECL_SEL)
#undef SPLA_DECL_SEL

    // ---- containers -----------------------------------------------------------------------------------------------------
    class Vector : public Object {
    public:
        virtual uint          get_n_rows()                                                        = 0;
        virtual ref_ptr<Type> get_type()                                                          = 0;
        virtual Status        set_format(FormatVector format)                                     = 0;
        virtual Status        set_fill_value(const ref_ptr<Scalar>& value)                        = 0;
        virtual Status        set_reduce(ref_ptr<OpBinary> resolve_duplicates)                    = 0;
        virtual Status        set_int(uint row_id, T_INT value)                                   = 0;
        virtual Status        set_uint(uint row_id, T_UINT value)                                 = 0;
        virtual Status        set_float(uint row_id, T_FLOAT value)                               = 0;
        virtual Status        get_int(uint row_id, T_INT& value)                                  = 0;
        virtual Status        get_uint(uint row_id, T_UINT& value)                                = 0;
        virtual Status        get_float(uint row_id, float& value)                                = 0;
        virtual Status        fill_noize(uint seed)                                               = 0;
        virtual Status        fill_with(const ref_ptr<Scalar>& value)                             = 0;
        virtual Status        build(const ref_ptr<MemView>& keys, const ref_ptr<MemView>& values) = 0;
        virtual Status        read(ref_ptr<MemView>& keys, ref_ptr<MemView>& values)              = 0;
        virtual Status        clear()                                                             = 0;

        static ref_ptr<Vector> make(uint n_rows, const ref_ptr<Type>& type);
    };

    class Matrix : public Object {
    public:
        virtual uint          get_n_rows()                                                                                        = 0;
        virtual uint          get_n_cols()                                                                                        = 0;
        virtual ref_ptr<Type> get_type()                                                                                          = 0;
        virtual Status        set_format(FormatMatrix format)                                                                     = 0;
        virtual Status        set_fill_value(const ref_ptr<Scalar>& value)                                                        = 0;
        virtual Status        set_reduce(ref_ptr<OpBinary> resolve_duplicates)                                                    = 0;
        virtual Status        set_int(uint row_id, uint col_id, std::int32_t value)                                               = 0;
        virtual Status        set_uint(uint row_id, uint col_id, std::uint32_t value)                                             = 0;
        virtual Status        set_float(uint row_id, uint col_id, float value)                                                    = 0;
        virtual Status        get_int(uint row_id, uint col_id, std::int32_t& value)                                              = 0;
        virtual Status        get_uint(uint row_id, uint col_id, std::uint32_t& value)                                            = 0;
        virtual Status        get_float(uint row_id, uint col_id, float& value)                                                   = 0;
        virtual Status        build(const ref_ptr<MemView>& keys1, const ref_ptr<MemView>& keys2, const ref_ptr<MemView>& values) = 0;
        virtual Status        read(ref_ptr<MemView>& keys1, ref_ptr<MemView>& keys2, ref_ptr<MemView>& values)                    = 0;
        virtual Status        clear()                                                                                             = 0;

        static ref_ptr<Matrix> make(uint n_rows, uint n_cols, const ref_ptr<Type>& type);
    };

    // ---- deferred execution -----------------------------------------------------------------------------------------------
    class ScheduleTask : public Object {
    public:
        virtual std::string                  get_name()            = 0;
        virtual std::string                  get_key()             = 0;// "<name>_<I|U|F>", from the RESULT object's type
        virtual std::string                  get_key_full()        = 0;
        virtual std::vector<ref_ptr<Object>> get_args()            = 0;
        virtual ref_ptr<Descriptor>          get_desc()            = 0;
        virtual ref_ptr<Descriptor>          get_desc_or_default() = 0;
    };
    class Schedule : public Object {
    public:
        virtual Status step_task(ref_ptr<ScheduleTask> task)                = 0;
        virtual Status step_tasks(std::vector<ref_ptr<ScheduleTask>> tasks) = 0;
        virtual Status submit()                                             = 0;
    };
    ref_ptr<Schedule> make_schedule();

    // ---- the hot path -----------------------------------------------------------------------------------------------------
    // With task_hnd == nullptr the operation runs before the call returns; otherwise only the task object is created
    // and handed back (reference src/exec.cpp:37-53).
    Status exec_mxm(ref_ptr<Matrix> R, ref_ptr<Matrix> A, ref_ptr<Matrix> B,
                    ref_ptr<OpBinary> op_multiply, ref_ptr<OpBinary> op_add, ref_ptr<Scalar> init,
                    ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);

    Status exec_mxmT_masked(ref_ptr<Matrix> R, ref_ptr<Matrix> mask, ref_ptr<Matrix> A, ref_ptr<Matrix> B,
                            ref_ptr<OpBinary> op_multiply, ref_ptr<OpBinary> op_add, ref_ptr<OpSelect> op_select, ref_ptr<Scalar> init,
                            ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);

    Status exec_mxv_masked(ref_ptr<Vector> r, ref_ptr<Vector> mask, ref_ptr<Matrix> M, ref_ptr<Vector> v,
                           ref_ptr<OpBinary> op_multiply, ref_ptr<OpBinary> op_add, ref_ptr<OpSelect> op_select, ref_ptr<Scalar> init,
                           ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);

    Status exec_vxm_masked(ref_ptr<Vector> r, ref_ptr<Vector> mask, ref_ptr<Vector> v, ref_ptr<Matrix> M,
                           ref_ptr<OpBinary> op_multiply, ref_ptr<OpBinary> op_add, ref_ptr<OpSelect> op_select, ref_ptr<Scalar> init,
                           ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);

    // ---- the vector operations of a traversal step (reference include/spla/exec.hpp:423-453,515-519) ----
    Status exec_v_eadd_fdb(ref_ptr<Vector> r, ref_ptr<Vector> v, ref_ptr<Vector> fdb, ref_ptr<OpBinary> op,
                           ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);
    Status exec_v_assign_masked(ref_ptr<Vector> r, ref_ptr<Vector> mask, ref_ptr<Scalar> value, ref_ptr<OpBinary> op_assign, ref_ptr<OpSelect> op_select,
                                ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);
    Status exec_v_count_mf(ref_ptr<Scalar> r, ref_ptr<Vector> v, ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);

    // ---- PageRank step remainder and triangle-count tail (reference include/spla/exec.hpp:297-303,381-387,489-495) ----
    Status exec_v_eadd(ref_ptr<Vector> r, ref_ptr<Vector> u, ref_ptr<Vector> v, ref_ptr<OpBinary> op,
                       ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);
    Status exec_v_reduce(ref_ptr<Scalar> r, ref_ptr<Scalar> s, ref_ptr<Vector> v, ref_ptr<OpBinary> op_reduce,
                         ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);
    Status exec_m_reduce(ref_ptr<Scalar> r, ref_ptr<Scalar> s, ref_ptr<Matrix> M, ref_ptr<OpBinary> op_reduce,
                         ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);

    // transpose with a unary op on the way, element-wise map (reference include/spla/exec.hpp:318-323,468-473)
    Status exec_m_transpose(ref_ptr<Matrix> R, ref_ptr<Matrix> M, ref_ptr<OpUnary> op_apply,
                            ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);
    Status exec_v_map(ref_ptr<Vector> r, ref_ptr<Vector> v, ref_ptr<OpUnary> op,
                      ref_ptr<Descriptor> desc = ref_ptr<Descriptor>(), ref_ptr<ScheduleTask>* task_hnd = nullptr);

    // ---- graph algorithms over exec_* (reference include/spla/algorithm.hpp:54-68,86-101,120-137,157-170) ----
    Status bfs(const ref_ptr<Vector>& v, const ref_ptr<Matrix>& A, uint s, const ref_ptr<Descriptor>& descriptor = ref_ptr<Descriptor>());
    Status sssp(const ref_ptr<Vector>& v, const ref_ptr<Matrix>& A, uint s, const ref_ptr<Descriptor>& descriptor = ref_ptr<Descriptor>());
    Status pr(ref_ptr<Vector>& p, const ref_ptr<Matrix>& A, float alpha = 0.85, float eps = 1e-6, const ref_ptr<Descriptor>& descriptor = ref_ptr<Descriptor>());
    Status tc(int& ntrins, const ref_ptr<Matrix>& A, const ref_ptr<Matrix>& B, const ref_ptr<Descriptor>& descriptor = ref_ptr<Descriptor>());
    // host-side checkers over std::vector adjacency lists (reference include/spla/algorithm.hpp:70-84,103-118,139-155,172-175): what the
    // reference's examples verify accelerator results with; never reached from exec_*
    Status bfs_naive(std::vector<int>& v, std::vector<std::vector<spla::uint>>& A, uint s, const ref_ptr<Descriptor>& descriptor = ref_ptr<Descriptor>());
    Status sssp_naive(std::vector<float>& v, std::vector<std::vector<uint>>& Ai, std::vector<std::vector<float>>& Ax, uint s,
                      const ref_ptr<Descriptor>& descriptor = ref_ptr<Descriptor>());
    Status pr_naive(std::vector<float>& p, std::vector<std::vector<uint>>& Ai, std::vector<std::vector<float>>& Ax, float alpha = 0.85, float eps = 1e-6,
                    const ref_ptr<Descriptor>& descriptor = ref_ptr<Descriptor>());
    Status tc_naive(int& ntrins, std::vector<std::vector<spla::uint>>& Ai, const ref_ptr<Descriptor>& descriptor = ref_ptr<Descriptor>());

    // ---- MatrixMarket coordinate loader + degree statistics (reference include/spla/io.hpp:47-85) ------------------
    class MtxLoader {
    public:
        explicit MtxLoader(std::string name = "");
        bool load(std::filesystem::path file_path, bool offset_indices = true, bool make_undirected = true, bool remove_loops = true);
        bool save(const std::filesystem::path& file_path, bool stats_only = false);
        void calc_stats();
        void output_stats();
        const std::vector<uint>& get_Ai() const;
        const std::vector<uint>& get_Aj() const;
        uint                     get_n_rows() const;
        uint                     get_n_cols() const;
        std::size_t              get_n_values() const;

    private:
        std::string           m_name;
        std::filesystem::path m_file_path;
        std::vector<uint>     m_Ai, m_Aj;
        bool                  m_base_is_zero = false;
        uint                  m_n_rows = 0, m_n_cols = 0;
        std::size_t           m_n_values = 0;
        double                m_deg_avg = -1.0, m_deg_sd = -1.0, m_deg_min = -1.0, m_deg_max = -1.0;
        std::vector<double>   m_deg_distribution;
        std::vector<uint>     m_deg_ranges;
    };

    // ---- wall-clock timer with laps (mirror of reference include/spla/timer.hpp:49-62; used by the reference's perf tests) -----
    class Timer {
    public:
        void start() { m_begin = m_end = clock::now(); }
        void stop() { m_end = clock::now(); }
        void lap_begin() { m_lap = clock::now(); }
        void lap_end() { m_laps.push_back(ms_between(m_lap, clock::now())); }
        void print(std::ostream& out = std::cout) const {
            for (double t : m_laps) out << t << ", ";
        }
        double                     get_elapsed_ms() const { return ms_between(m_begin, m_end); }
        double                     get_elapsed_sec() const { return get_elapsed_ms() * 1e-3; }
        const std::vector<double>& get_laps_ms() const { return m_laps; }

    private:
        using clock = std::chrono::steady_clock;
        static double ms_between(clock::time_point a, clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        }
        clock::time_point   m_begin{}, m_end{}, m_lap{};
        std::vector<double> m_laps;
    };

    // ---- library ------------------------------------------------------------------------------------------------------------
    class Library final {
    public:
        Library();
        ~Library();
        Library(const Library&) = delete;
        Library(Library&&)      = delete;

        void   finalize();
        Status set_accelerator(AcceleratorType accelerator);
        Status set_platform(int index);
        Status set_device(int index);
        Status set_queues_count(int count);
        Status set_message_callback(MessageCallback callback);
        Status set_default_callback();
        Status set_force_no_acceleration(bool value);
        bool   is_set_force_no_acceleration();
        Status get_accelerator_info(std::string& info);

        class Accelerator* get_accelerator();
        class Registry*    get_registry();
        class Dispatcher*  get_dispatcher();
        class Logger*      get_logger();

        static Library* get();

    private:
        struct Impl;
        Impl* m_impl;
    };

}// namespace spla
