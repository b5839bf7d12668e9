// internal.hpp — runtime pieces behind include/spla.hpp: logger, accelerator, registry, dispatcher, typed scalars /
// ops, schedule tasks.  Written from scratch; the seams mirror the reference's so the HIP backend plugs in where an
// accelerator plugs into the reference:
//   Accelerator interface      reference src/core/accelerator.hpp:57-68
//   Registry / RegistryAlgo    reference src/core/registry.hpp:41-89   (key = "<op>_<I|U|F>" + suffix)
//   Dispatcher                 reference src/core/dispatcher.cpp:42-84 (accelerator key first; exceptions -> Status::Error)
//   ScheduleTask_*             reference src/schedule/schedule_tasks.hpp:80-184, schedule_tasks.cpp:68-163
#pragma once

#include <spla.hpp>

#include <memory>
#include <cstring>
#include <mutex>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace spla {

    // ---- logging --------------------------------------------------------------------------------------------------
    class Logger {
    public:
        void log(Status status, const std::string& msg, const char* file, const char* function, int line) {
            std::lock_guard<std::mutex> guard(m_mutex);
            if (m_callback) m_callback(status, msg, file, function, line);
        }
        void set_callback(MessageCallback cb) {
            std::lock_guard<std::mutex> guard(m_mutex);
            m_callback = std::move(cb);
        }

    private:
        MessageCallback m_callback;
        std::mutex      m_mutex;
    };

#define SPLA_LOG(status, expr)                                                                      \
    do {                                                                                            \
        if (auto* _lg = ::spla::Library::get()->get_logger()) {                                     \
            std::ostringstream _ss;                                                                 \
            _ss << expr;                                                                            \
            _lg->log((status), _ss.str(), __FILE__, __func__, __LINE__);                            \
        }                                                                                           \
    } while (0)

    // ---- accelerator ------------------------------------------------------------------------------------------------
    class Accelerator {
    public:
        virtual ~Accelerator()                                = default;
        virtual Status             init()                     = 0;
        virtual Status             set_platform(int index)    = 0;
        virtual Status             set_device(int index)      = 0;
        virtual Status             set_queues_count(int count) = 0;
        virtual const std::string& get_name()                 = 0;
        virtual const std::string& get_description()          = 0;
        virtual const std::string& get_suffix()               = 0;// "__hip"
    };

    // the MI355X accelerator: one device, one stream (the "queue"), kernels behind include/spla_hip.h
    class HipAccelerator final : public Accelerator {
    public:
        Status             init() override;
        Status             set_platform(int index) override;
        Status             set_device(int index) override;
        Status             set_queues_count(int count) override;
        const std::string& get_name() override { return m_name; }
        const std::string& get_description() override { return m_description; }
        const std::string& get_suffix() override { return m_suffix; }
        void*              stream() const { return nullptr; }// default stream: in-order with every transfer

    private:
        std::string m_name        = "HIP";
        std::string m_description = "no device";
        std::string m_suffix      = "__hip";
        int         m_device      = 0;
    };

    HipAccelerator* get_acc_hip();// nullptr when the accelerator is not active

    // ---- registry / dispatcher ------------------------------------------------------------------------------------------
    struct DispatchContext {
        ref_ptr<ScheduleTask> task;
        int                   thread_id = 0;
        int                   step_id   = 0;
    };

    class RegistryAlgo {
    public:
        virtual ~RegistryAlgo()                                  = default;
        virtual std::string get_name()                           = 0;
        virtual std::string get_description()                    = 0;
        virtual Status      execute(const DispatchContext& ctx)  = 0;
    };

    class Registry {
    public:
        void          add(const std::string& key, std::shared_ptr<RegistryAlgo> algo) { m_algos[key] = std::move(algo); }
        bool          has(const std::string& key) const { return m_algos.count(key) != 0; }
        RegistryAlgo* find(const std::string& key) const {
            auto it = m_algos.find(key);
            return it == m_algos.end() ? nullptr : it->second.get();
        }
        std::vector<std::string> keys() const {
            std::vector<std::string> out;
            for (const auto& kv : m_algos) out.push_back(kv.first);
            return out;
        }

    private:
        std::unordered_map<std::string, std::shared_ptr<RegistryAlgo>> m_algos;
    };

    class Dispatcher {
    public:
        Status dispatch(const DispatchContext& ctx);
    };

    void register_algo_hip(Registry* registry);// host/algos_hip.cpp

    // ---- typed scalars, types, ops ----------------------------------------------------------------------------------------
    template<typename T>
    struct TypeTraits;
    template<>
    struct TypeTraits<T_INT> {
        static constexpr int         id = 0;
        static constexpr const char* code = "I";
        static ref_ptr<Type>         type() { return INT; }
    };
    template<>
    struct TypeTraits<T_UINT> {
        static constexpr int         id = 1;
        static constexpr const char* code = "U";
        static ref_ptr<Type>         type() { return UINT; }
    };
    template<>
    struct TypeTraits<T_FLOAT> {
        static constexpr int         id = 2;
        static constexpr const char* code = "F";
        static ref_ptr<Type>         type() { return FLOAT; }
    };

    template<typename T>
    inline std::uint32_t word_of(T value) {
        std::uint32_t w;
        static_assert(sizeof(T) == 4, "4-byte elements only");
        std::memcpy(&w, &value, 4);
        return w;
    }

    template<typename T>
    class TScalar final : public Scalar {
    public:
        TScalar() = default;
        explicit TScalar(T v) : m_value(v) {}
        void               set_label(std::string label) override { m_label = std::move(label); }
        const std::string& get_label() const override { return m_label; }
        ref_ptr<Type>      get_type() override { return TypeTraits<T>::type(); }
        Status             set_int(std::int32_t v) override { return (m_value = static_cast<T>(v), Status::Ok); }
        Status             set_uint(std::uint32_t v) override { return (m_value = static_cast<T>(v), Status::Ok); }
        Status             set_float(float v) override { return (m_value = static_cast<T>(v), Status::Ok); }
        Status             get_int(std::int32_t& v) override { return (v = static_cast<std::int32_t>(m_value), Status::Ok); }
        Status             get_uint(std::uint32_t& v) override { return (v = static_cast<std::uint32_t>(m_value), Status::Ok); }
        Status             get_float(float& v) override { return (v = static_cast<float>(m_value), Status::Ok); }
        T_INT              as_int() override { return static_cast<T_INT>(m_value); }
        T_UINT             as_uint() override { return static_cast<T_UINT>(m_value); }
        T_FLOAT            as_float() override { return static_cast<T_FLOAT>(m_value); }
        T                  get_value() const { return m_value; }
        void               set_value(T v) { m_value = v; }

    private:
        T           m_value = T();
        std::string m_label;
    };

    // device_id: splahip_binop / splahip_selop id of a built-in op, -1 for a user-defined op (host function only)
    template<typename T>
    class TOpBinary final : public OpBinary {
    public:
        void               set_label(std::string label) override { m_label = std::move(label); }
        const std::string& get_label() const override { return m_label; }
        ref_ptr<Type>      get_type_res() override { return TypeTraits<T>::type(); }
        ref_ptr<Type>      get_type_arg_0() override { return TypeTraits<T>::type(); }
        ref_ptr<Type>      get_type_arg_1() override { return TypeTraits<T>::type(); }
        std::string        get_name() override { return name; }
        std::string        get_key() override { return key; }
        std::string        get_source_cl() override { return source; }

        std::function<T(T, T)> function;
        std::string            name, key, source;
        int                    device_id = -1;

    private:
        std::string m_label;
    };

    template<typename T>
    class TOpSelect final : public OpSelect {
    public:
        void               set_label(std::string label) override { m_label = std::move(label); }
        const std::string& get_label() const override { return m_label; }
        ref_ptr<Type>      get_type_res() override { return BOOL; }
        ref_ptr<Type>      get_type_arg_0() override { return TypeTraits<T>::type(); }
        std::string        get_name() override { return name; }
        std::string        get_key() override { return key; }
        std::string        get_source_cl() override { return source; }

        std::function<bool(T)> function;
        std::string            name, key, source;
        int                    device_id = -1;

    private:
        std::string m_label;
    };

    template<typename T>
    class TOpUnary final : public OpUnary {
    public:
        void               set_label(std::string label) override { m_label = std::move(label); }
        const std::string& get_label() const override { return m_label; }
        ref_ptr<Type>      get_type_res() override { return TypeTraits<T>::type(); }
        ref_ptr<Type>      get_type_arg_0() override { return TypeTraits<T>::type(); }
        std::string        get_name() override { return name; }
        std::string        get_key() override { return key; }
        std::string        get_source_cl() override { return source; }

        std::function<T(T)> function;
        std::string         name, key, source;
        int                 device_id = -1;// splahip_unop

    private:
        std::string m_label;
    };

    void register_ops();// host/core.cpp: fills the global op handles declared in spla.hpp

    // ---- schedule tasks -------------------------------------------------------------------------------------------------------
    class TaskBase : public ScheduleTask {
    public:
        void                set_label(std::string label) override { m_label = std::move(label); }
        const std::string&  get_label() const override { return m_label; }
        std::string         get_key_full() override { return get_name() + "__full"; }
        ref_ptr<Descriptor> get_desc() override { return desc; }
        ref_ptr<Descriptor> get_desc_or_default() override {
            static ref_ptr<Descriptor> fallback = Descriptor::make();
            return desc ? desc : fallback;
        }
        ref_ptr<Descriptor> desc;

    protected:
        std::string m_label;
    };

    class ScheduleTask_mxm final : public TaskBase {
    public:
        std::string                  get_name() override { return "mxm"; }
        std::string                  get_key() override { return "mxm_" + R->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {R.as<Object>(), A.as<Object>(), B.as<Object>(), op_multiply.as<Object>(), op_add.as<Object>(), init.as<Object>()}; }
        ref_ptr<Matrix>              R, A, B;
        ref_ptr<OpBinary>            op_multiply, op_add;
        ref_ptr<Scalar>              init;
    };
    class ScheduleTask_mxmT_masked final : public TaskBase {
    public:
        std::string                  get_name() override { return "mxmT_masked"; }
        std::string                  get_key() override { return "mxmT_masked_" + R->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {R.as<Object>(), mask.as<Object>(), A.as<Object>(), B.as<Object>(), op_multiply.as<Object>(), op_add.as<Object>(), op_select.as<Object>(), init.as<Object>()}; }
        ref_ptr<Matrix>              R, mask, A, B;
        ref_ptr<OpBinary>            op_multiply, op_add;
        ref_ptr<OpSelect>            op_select;
        ref_ptr<Scalar>              init;
    };
    class ScheduleTask_mxv_masked final : public TaskBase {
    public:
        std::string                  get_name() override { return "mxv_masked"; }
        std::string                  get_key() override { return "mxv_masked_" + r->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {r.as<Object>(), mask.as<Object>(), M.as<Object>(), v.as<Object>(), op_multiply.as<Object>(), op_add.as<Object>(), op_select.as<Object>(), init.as<Object>()}; }
        ref_ptr<Vector>              r, mask, v;
        ref_ptr<Matrix>              M;
        ref_ptr<OpBinary>            op_multiply, op_add;
        ref_ptr<OpSelect>            op_select;
        ref_ptr<Scalar>              init;
    };
    class ScheduleTask_vxm_masked final : public TaskBase {
    public:
        std::string                  get_name() override { return "vxm_masked"; }
        std::string                  get_key() override { return "vxm_masked_" + r->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {r.as<Object>(), mask.as<Object>(), v.as<Object>(), M.as<Object>(), op_multiply.as<Object>(), op_add.as<Object>(), op_select.as<Object>(), init.as<Object>()}; }
        ref_ptr<Vector>              r, mask, v;
        ref_ptr<Matrix>              M;
        ref_ptr<OpBinary>            op_multiply, op_add;
        ref_ptr<OpSelect>            op_select;
        ref_ptr<Scalar>              init;
    };

    // vector operations of a traversal step (reference src/schedule/schedule_tasks.hpp: ScheduleTask_v_eadd_fdb / v_assign_masked / v_count_mf)
    class ScheduleTask_v_eadd_fdb final : public TaskBase {
    public:
        std::string                  get_name() override { return "v_eadd_fdb"; }
        std::string                  get_key() override { return "v_eadd_fdb_" + r->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {r.as<Object>(), v.as<Object>(), fdb.as<Object>(), op.as<Object>()}; }
        ref_ptr<Vector>              r, v, fdb;
        ref_ptr<OpBinary>            op;
    };
    class ScheduleTask_v_assign_masked final : public TaskBase {
    public:
        std::string                  get_name() override { return "v_assign_masked"; }
        std::string                  get_key() override { return "v_assign_masked_" + r->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {r.as<Object>(), mask.as<Object>(), value.as<Object>(), op_assign.as<Object>(), op_select.as<Object>()}; }
        ref_ptr<Vector>              r, mask;
        ref_ptr<Scalar>              value;
        ref_ptr<OpBinary>            op_assign;
        ref_ptr<OpSelect>            op_select;
    };
    class ScheduleTask_v_count_mf final : public TaskBase {
    public:
        std::string                  get_name() override { return "v_count_mf"; }
        std::string                  get_key() override { return "v_count_mf_" + v->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {r.as<Object>(), v.as<Object>()}; }
        ref_ptr<Scalar>              r;
        ref_ptr<Vector>              v;
    };
    // PageRank step remainder and triangle-count tail (reference src/schedule/schedule_tasks.hpp: ScheduleTask_v_eadd / v_reduce / m_reduce;
    // keys by the RESULT's type, schedule_tasks.cpp:264-270,348-354,454-460)
    class ScheduleTask_v_eadd final : public TaskBase {
    public:
        std::string                  get_name() override { return "v_eadd"; }
        std::string                  get_key() override { return "v_eadd_" + r->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {r.as<Object>(), u.as<Object>(), v.as<Object>(), op.as<Object>()}; }
        ref_ptr<Vector>              r, u, v;
        ref_ptr<OpBinary>            op;
    };
    class ScheduleTask_v_reduce final : public TaskBase {
    public:
        std::string                  get_name() override { return "v_reduce"; }
        std::string                  get_key() override { return "v_reduce_" + r->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {r.as<Object>(), s.as<Object>(), v.as<Object>(), op_reduce.as<Object>()}; }
        ref_ptr<Scalar>              r, s;
        ref_ptr<Vector>              v;
        ref_ptr<OpBinary>            op_reduce;
    };
    class ScheduleTask_m_reduce final : public TaskBase {
    public:
        std::string                  get_name() override { return "m_reduce"; }
        std::string                  get_key() override { return "m_reduce_" + r->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {r.as<Object>(), s.as<Object>(), M.as<Object>(), op_reduce.as<Object>()}; }
        ref_ptr<Scalar>              r, s;
        ref_ptr<Matrix>              M;
        ref_ptr<OpBinary>            op_reduce;
    };
    // reference src/schedule/schedule_tasks.hpp:281-300 (m_transpose), ScheduleTask_v_map; keys by the result's type
    class ScheduleTask_m_transpose final : public TaskBase {
    public:
        std::string                  get_name() override { return "m_transpose"; }
        std::string                  get_key() override { return "m_transpose_" + R->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {R.as<Object>(), M.as<Object>(), op_apply.as<Object>()}; }
        ref_ptr<Matrix>              R, M;
        ref_ptr<OpUnary>             op_apply;
    };
    class ScheduleTask_v_map final : public TaskBase {
    public:
        std::string                  get_name() override { return "v_map"; }
        std::string                  get_key() override { return "v_map_" + r->get_type()->get_code(); }
        std::vector<ref_ptr<Object>> get_args() override { return {r.as<Object>(), v.as<Object>(), op.as<Object>()}; }
        ref_ptr<Vector>              r, v;
        ref_ptr<OpUnary>             op;
    };

}// namespace spla
