// core.cpp — types, scalars, memory views, the built-in op table, the library singleton, the dispatcher and the
// exec_* front-ends of the hot path.  From-scratch implementation of the behaviour described in SURVEY.md 8a/8b:
//   exec front-end       reference src/exec.cpp:37-53,66-172      (task object; run now or hand back)
//   dispatcher           reference src/core/dispatcher.cpp:42-84  (key + accelerator suffix; exception -> Status::Error)
//   library              reference src/library.cpp:70-126,218-238 (lazy singleton; accelerator picked at first use)
//   built-in ops + keys  reference src/op.cpp:194-266, src/core/top.hpp:84-89,111-114
#include "internal.hpp"

#include <spla_hip.h>

#include <cmath>
#include <cstdio>
#include <cstring>

namespace spla {

    // ------------------------------------------------------------------------------------------------------------ types
    namespace {
        class TTypeImpl final : public Type {
        public:
            TTypeImpl(std::string name, std::string code, std::string cpp, std::string desc, int size, int id)
                : m_name(std::move(name)), m_code(std::move(code)), m_cpp(std::move(cpp)), m_desc(std::move(desc)), m_size(size), m_id(id) {}
            void               set_label(std::string label) override { m_label = std::move(label); }
            const std::string& get_label() const override { return m_label; }
            const std::string& get_name() override { return m_name; }
            const std::string& get_code() override { return m_code; }
            const std::string& get_cpp() override { return m_cpp; }
            const std::string& get_description() override { return m_desc; }
            int                get_size() override { return m_size; }
            int                get_id() override { return m_id; }

        private:
            std::string m_name, m_code, m_cpp, m_desc, m_label;
            int         m_size, m_id;
        };
    }// namespace

    // BOOL is a 4-byte result type of select ops only; containers exist for INT / UINT / FLOAT (reference src/type.cpp:32-35)
    ref_ptr<Type> BOOL(new TTypeImpl("BOOL", "B", "bool", "4-byte logical value", 4, 3));
    ref_ptr<Type> INT(new TTypeImpl("INT", "I", "int", "signed 32-bit integer", 4, 0));
    ref_ptr<Type> UINT(new TTypeImpl("UINT", "U", "uint", "unsigned 32-bit integer", 4, 1));
    ref_ptr<Type> FLOAT(new TTypeImpl("FLOAT", "F", "float", "32-bit floating point", 4, 2));

    // ---------------------------------------------------------------------------------------------------------- memview
    Status MemView::read(std::size_t offset, std::size_t size, void* dst) {
        if (offset + size > m_size) return Status::InvalidArgument;
        if (size) std::memcpy(dst, static_cast<const char*>(m_buffer) + offset, size);
        return Status::Ok;
    }
    Status MemView::write(std::size_t offset, std::size_t size, const void* src) {
        if (!m_mutable) return Status::InvalidState;
        if (offset + size > m_size) return Status::InvalidArgument;
        if (size) std::memcpy(static_cast<char*>(m_buffer) + offset, src, size);
        return Status::Ok;
    }
    ref_ptr<MemView> MemView::make(void* buffer, std::size_t size, bool is_mutable) {
        auto view       = ref_ptr<MemView>(new MemView());
        view->m_buffer  = buffer;
        view->m_size    = size;
        view->m_mutable = is_mutable;
        return view;
    }
    ref_ptr<MemView> MemView::make() { return ref_ptr<MemView>(new MemView()); }

    // ----------------------------------------------------------------------------------------------------------- scalar
    ref_ptr<Scalar> Scalar::make(const ref_ptr<Type>& type) {
        if (type == INT) return ref_ptr<Scalar>(new TScalar<T_INT>());
        if (type == UINT) return ref_ptr<Scalar>(new TScalar<T_UINT>());
        if (type == FLOAT) return ref_ptr<Scalar>(new TScalar<T_FLOAT>());
        return ref_ptr<Scalar>();
    }
    ref_ptr<Scalar> Scalar::make_int(std::int32_t value) { return ref_ptr<Scalar>(new TScalar<T_INT>(value)); }
    ref_ptr<Scalar> Scalar::make_uint(std::uint32_t value) { return ref_ptr<Scalar>(new TScalar<T_UINT>(value)); }
    ref_ptr<Scalar> Scalar::make_float(float value) { return ref_ptr<Scalar>(new TScalar<T_FLOAT>(value)); }

    ref_ptr<Descriptor> Descriptor::make() { return ref_ptr<Descriptor>(new Descriptor()); }

    // -------------------------------------------------------------------------------------------------------------- ops
#define SPLA_DEF_BIN(NAME, HASF)        \
    ref_ptr<OpBinary> NAME##_INT;       \
    ref_ptr<OpBinary> NAME##_UINT;      \
    ref_ptr<OpBinary> NAME##_FLOAT;
    SPLA_BINARY_OPS(SPLA_DEF_BIN)
#undef SPLA_DEF_BIN
#define SPLA_DEF_SEL(NAME)              \
    ref_ptr<OpSelect> NAME##_INT;       \
    ref_ptr<OpSelect> NAME##_UINT;      \
    ref_ptr<OpSelect> NAME##_FLOAT;
    SPLA_SELECT_OPS(SPLA_DEF_SEL)
#undef SPLA_DEF_SEL

#define SPLA_DEF_UN(NAME, HASI, HASF)   \
    ref_ptr<OpUnary> NAME##_INT;        \
    ref_ptr<OpUnary> NAME##_UINT;       \
    ref_ptr<OpUnary> NAME##_FLOAT;
    SPLA_UNARY_OPS(SPLA_DEF_UN)
#undef SPLA_DEF_UN

    namespace {
        template<typename T>
        const char* type_name() {
            if (std::is_same<T, T_INT>::value) return "INT";
            if (std::is_same<T, T_UINT>::value) return "UINT";
            return "FLOAT";
        }
        template<typename T>
        const char* type_cl() {
            if (std::is_same<T, T_INT>::value) return "int";
            if (std::is_same<T, T_UINT>::value) return "uint";
            return "float";
        }

        template<typename T>
        ref_ptr<OpBinary> bin(const char* prefix, int device_id, const char* body, std::function<T(T, T)> fn) {
            auto op       = ref_ptr<TOpBinary<T>>(new TOpBinary<T>());
            const std::string c = TypeTraits<T>::code;
            op->name      = std::string(prefix) + "_" + type_name<T>();
            op->key       = std::string(prefix) + "_" + c + c + c;
            op->source    = std::string("(") + type_cl<T>() + " a, " + type_cl<T>() + " b) { return " + body + "; }";
            op->function  = std::move(fn);
            op->device_id = device_id;
            return op.template as<OpBinary>();
        }
        template<typename T>
        ref_ptr<OpSelect> sel(const char* prefix, int device_id, const char* body, std::function<bool(T)> fn) {
            auto op       = ref_ptr<TOpSelect<T>>(new TOpSelect<T>());
            op->name      = std::string(prefix) + "_" + type_name<T>();
            op->key       = std::string(prefix) + "_" + TypeTraits<T>::code;
            op->source    = std::string("(") + type_cl<T>() + " a) { return " + body + "; }";
            op->function  = std::move(fn);
            op->device_id = device_id;
            return op.template as<OpSelect>();
        }

        template<typename T>
        ref_ptr<OpUnary> un(const char* prefix, int device_id, const char* body, std::function<T(T)> fn) {
            auto op       = ref_ptr<TOpUnary<T>>(new TOpUnary<T>());
            const std::string c = TypeTraits<T>::code;
            op->name      = std::string(prefix) + "_" + type_name<T>();
            op->key       = std::string(prefix) + "_" + c + c;
            op->source    = std::string("(") + type_cl<T>() + " a) { return " + body + "; }";
            op->function  = std::move(fn);
            op->device_id = device_id;
            return op.template as<OpUnary>();
        }

        // host-side scalar semantics; int32 +,-,* wrap like the device (the reference's signed overflow is UB)
        template<typename T>
        T w_add(T a, T b) {
            if constexpr (std::is_same<T, T_INT>::value) return static_cast<T_INT>(static_cast<T_UINT>(a) + static_cast<T_UINT>(b));
            else
                return a + b;
        }
        template<typename T>
        T w_sub(T a, T b) {
            if constexpr (std::is_same<T, T_INT>::value) return static_cast<T_INT>(static_cast<T_UINT>(a) - static_cast<T_UINT>(b));
            else
                return a - b;
        }
        template<typename T>
        T w_mul(T a, T b) {
            if constexpr (std::is_same<T, T_INT>::value) return static_cast<T_INT>(static_cast<T_UINT>(a) * static_cast<T_UINT>(b));
            else
                return a * b;
        }
        template<typename T>
        T w_div(T a, T b) {
            if constexpr (std::is_same<T, T_FLOAT>::value) return a / b;
            else if constexpr (std::is_same<T, T_INT>::value)
                return b == 0 ? 0 : ((a == INT32_MIN && b == -1) ? a : a / b);
            else
                return b == 0 ? 0u : a / b;
        }

        template<typename T>
        void register_arith(ref_ptr<OpBinary>& plus, ref_ptr<OpBinary>& minus, ref_ptr<OpBinary>& mult, ref_ptr<OpBinary>& div,
                            ref_ptr<OpBinary>& mpow2, ref_ptr<OpBinary>& first, ref_ptr<OpBinary>& second, ref_ptr<OpBinary>& bone,
                            ref_ptr<OpBinary>& mn, ref_ptr<OpBinary>& mx, ref_ptr<OpBinary>& lor, ref_ptr<OpBinary>& land) {
            plus   = bin<T>("PLUS", SPLAHIP_PLUS, "a + b", [](T a, T b) { return w_add(a, b); });
            minus  = bin<T>("MINUS", SPLAHIP_MINUS, "a - b", [](T a, T b) { return w_sub(a, b); });
            mult   = bin<T>("MULT", SPLAHIP_MULT, "a * b", [](T a, T b) { return w_mul(a, b); });
            div    = bin<T>("DIV", SPLAHIP_DIV, "a / b", [](T a, T b) { return w_div(a, b); });
            mpow2  = bin<T>("MINUS_POW2", SPLAHIP_MINUS_POW2, "(a - b) * (a - b)", [](T a, T b) { return w_mul(w_sub(a, b), w_sub(a, b)); });
            first  = bin<T>("FIRST", SPLAHIP_FIRST, "a", [](T a, T) { return a; });
            second = bin<T>("SECOND", SPLAHIP_SECOND, "b", [](T, T b) { return b; });
            bone   = bin<T>("BONE", SPLAHIP_BONE, "1", [](T, T) { return T(1); });
            mn     = bin<T>("MIN", SPLAHIP_MIN, "min(a, b)", [](T a, T b) { return (b < a) ? b : a; });
            mx     = bin<T>("MAX", SPLAHIP_MAX, "max(a, b)", [](T a, T b) { return (a < b) ? b : a; });
            lor    = bin<T>("LOR", SPLAHIP_LOR, "a || b", [](T a, T b) { return (a || b) ? T(1) : T(0); });
            land   = bin<T>("LAND", SPLAHIP_LAND, "a && b", [](T a, T b) { return (a && b) ? T(1) : T(0); });
        }
        template<typename T>
        void register_bitwise(ref_ptr<OpBinary>& bor, ref_ptr<OpBinary>& band, ref_ptr<OpBinary>& bxor) {
            bor  = bin<T>("BOR", SPLAHIP_BOR, "a | b", [](T a, T b) { return a | b; });
            band = bin<T>("BAND", SPLAHIP_BAND, "a & b", [](T a, T b) { return a & b; });
            bxor = bin<T>("BXOR", SPLAHIP_BXOR, "a ^ b", [](T a, T b) { return a ^ b; });
        }
        template<typename T>
        void register_select(ref_ptr<OpSelect>& eqz, ref_ptr<OpSelect>& nqz, ref_ptr<OpSelect>& gtz, ref_ptr<OpSelect>& gez,
                             ref_ptr<OpSelect>& ltz, ref_ptr<OpSelect>& lez, ref_ptr<OpSelect>& always, ref_ptr<OpSelect>& never) {
            eqz    = sel<T>("EQZERO", SPLAHIP_EQZERO, "a == 0", [](T a) { return a == T(0); });
            nqz    = sel<T>("NQZERO", SPLAHIP_NQZERO, "a != 0", [](T a) { return a != T(0); });
            gtz    = sel<T>("GTZERO", SPLAHIP_GTZERO, "a > 0", [](T a) { return a > T(0); });
            gez    = sel<T>("GEZERO", SPLAHIP_GEZERO, "a >= 0", [](T a) { return a >= T(0); });
            ltz    = sel<T>("LTZERO", SPLAHIP_LTZERO, "a < 0", [](T a) { return a < T(0); });
            lez    = sel<T>("LEZERO", SPLAHIP_LEZERO, "a <= 0", [](T a) { return a <= T(0); });
            always = sel<T>("ALWAYS", SPLAHIP_ALWAYS, "1", [](T) { return true; });
            never  = sel<T>("NEVER", SPLAHIP_NEVER, "0", [](T) { return false; });
        }
    }// namespace

    namespace {
        // unary ops every element type has (reference src/op.cpp:99-116); integer 1/0 = 0 and ABS(INT_MIN) wraps, as on the device
        template<typename T>
        void register_unary_common(ref_ptr<OpUnary>& identity, ref_ptr<OpUnary>& ainv, ref_ptr<OpUnary>& minv, ref_ptr<OpUnary>& lnot,
                                   ref_ptr<OpUnary>& uone, ref_ptr<OpUnary>& abs_) {
            identity = un<T>("IDENTITY", SPLAHIP_IDENTITY, "a", [](T a) { return a; });
            ainv     = un<T>("AINV", SPLAHIP_AINV, "-a", [](T a) { return w_sub(T(0), a); });
            minv     = un<T>("MINV", SPLAHIP_MINV, "1 / a", [](T a) { return w_div(T(1), a); });
            lnot     = un<T>("LNOT", SPLAHIP_LNOT, "!(a != 0)", [](T a) { return a != T(0) ? T(0) : T(1); });
            uone     = un<T>("UONE", SPLAHIP_UONE, "1", [](T) { return T(1); });
            abs_     = un<T>("ABS", SPLAHIP_ABS, "abs(a)", [](T a) { return a < T(0) ? w_sub(T(0), a) : a; });
        }
    }// namespace

    void register_ops() {
        if (PLUS_INT) return;
        register_unary_common<T_INT>(IDENTITY_INT, AINV_INT, MINV_INT, LNOT_INT, UONE_INT, ABS_INT);
        register_unary_common<T_UINT>(IDENTITY_UINT, AINV_UINT, MINV_UINT, LNOT_UINT, UONE_UINT, ABS_UINT);
        register_unary_common<T_FLOAT>(IDENTITY_FLOAT, AINV_FLOAT, MINV_FLOAT, LNOT_FLOAT, UONE_FLOAT, ABS_FLOAT);
        BNOT_INT    = un<T_INT>("BNOT", SPLAHIP_BNOT, "~a", [](T_INT a) { return ~a; });
        BNOT_UINT   = un<T_UINT>("BNOT", SPLAHIP_BNOT, "~a", [](T_UINT a) { return ~a; });
        SQRT_FLOAT  = un<T_FLOAT>("SQRT", SPLAHIP_SQRT, "sqrt(a)", [](T_FLOAT a) { return std::sqrt(a); });
        LOG_FLOAT   = un<T_FLOAT>("LOG", SPLAHIP_LOG, "log(a)", [](T_FLOAT a) { return static_cast<T_FLOAT>(std::log(static_cast<double>(a))); });
        EXP_FLOAT   = un<T_FLOAT>("EXP", SPLAHIP_EXP, "exp(a)", [](T_FLOAT a) { return static_cast<T_FLOAT>(std::exp(static_cast<double>(a))); });
        SIN_FLOAT   = un<T_FLOAT>("SIN", SPLAHIP_SIN, "sin(a)", [](T_FLOAT a) { return static_cast<T_FLOAT>(std::sin(static_cast<double>(a))); });
        COS_FLOAT   = un<T_FLOAT>("COS", SPLAHIP_COS, "cos(a)", [](T_FLOAT a) { return static_cast<T_FLOAT>(std::cos(static_cast<double>(a))); });
        TAN_FLOAT   = un<T_FLOAT>("TAN", SPLAHIP_TAN, "tan(a)", [](T_FLOAT a) { return static_cast<T_FLOAT>(std::tan(static_cast<double>(a))); });
        ASIN_FLOAT  = un<T_FLOAT>("ASIN", SPLAHIP_ASIN, "asin(a)", [](T_FLOAT a) { return static_cast<T_FLOAT>(std::asin(static_cast<double>(a))); });
        ACOS_FLOAT  = un<T_FLOAT>("ACOS", SPLAHIP_ACOS, "acos(a)", [](T_FLOAT a) { return static_cast<T_FLOAT>(std::acos(static_cast<double>(a))); });
        ATAN_FLOAT  = un<T_FLOAT>("ATAN", SPLAHIP_ATAN, "atan(a)", [](T_FLOAT a) { return static_cast<T_FLOAT>(std::atan(static_cast<double>(a))); });
        CEIL_FLOAT  = un<T_FLOAT>("CEIL", SPLAHIP_CEIL, "ceil(a)", [](T_FLOAT a) { return std::ceil(a); });
        FLOOR_FLOAT = un<T_FLOAT>("FLOOR", SPLAHIP_FLOOR, "floor(a)", [](T_FLOAT a) { return std::floor(a); });
        ROUND_FLOAT = un<T_FLOAT>("ROUND", SPLAHIP_ROUND, "round(a)", [](T_FLOAT a) { return std::round(a); });
        TRUNC_FLOAT = un<T_FLOAT>("TRUNC", SPLAHIP_TRUNC, "trunc(a)", [](T_FLOAT a) { return std::trunc(a); });
        register_arith<T_INT>(PLUS_INT, MINUS_INT, MULT_INT, DIV_INT, MINUS_POW2_INT, FIRST_INT, SECOND_INT, BONE_INT, MIN_INT, MAX_INT, LOR_INT, LAND_INT);
        register_arith<T_UINT>(PLUS_UINT, MINUS_UINT, MULT_UINT, DIV_UINT, MINUS_POW2_UINT, FIRST_UINT, SECOND_UINT, BONE_UINT, MIN_UINT, MAX_UINT, LOR_UINT, LAND_UINT);
        register_arith<T_FLOAT>(PLUS_FLOAT, MINUS_FLOAT, MULT_FLOAT, DIV_FLOAT, MINUS_POW2_FLOAT, FIRST_FLOAT, SECOND_FLOAT, BONE_FLOAT, MIN_FLOAT, MAX_FLOAT, LOR_FLOAT, LAND_FLOAT);
        register_bitwise<T_INT>(BOR_INT, BAND_INT, BXOR_INT);
        register_bitwise<T_UINT>(BOR_UINT, BAND_UINT, BXOR_UINT);
        register_select<T_INT>(EQZERO_INT, NQZERO_INT, GTZERO_INT, GEZERO_INT, LTZERO_INT, LEZERO_INT, ALWAYS_INT, NEVER_INT);
        register_select<T_UINT>(EQZERO_UINT, NQZERO_UINT, GTZERO_UINT, GEZERO_UINT, LTZERO_UINT, LEZERO_UINT, ALWAYS_UINT, NEVER_UINT);
        register_select<T_FLOAT>(EQZERO_FLOAT, NQZERO_FLOAT, GTZERO_FLOAT, GEZERO_FLOAT, LTZERO_FLOAT, LEZERO_FLOAT, ALWAYS_FLOAT, NEVER_FLOAT);
    }

    // user-defined ops carry a host function and the body text; they have no device id, so an exec_* call that uses
    // one is answered with Status::NotImplemented by the HIP algorithms (the reference would fall back to its CPU backend)
    namespace {
        template<typename T>
        ref_ptr<OpBinary> user_bin(std::string name, std::string code, std::function<T(T, T)> fn) {
            auto op       = ref_ptr<TOpBinary<T>>(new TOpBinary<T>());
            const std::string c = TypeTraits<T>::code;
            op->name      = name;
            op->key       = name + "_" + c + c + c;
            op->source    = std::move(code);
            op->function  = std::move(fn);
            return op.template as<OpBinary>();
        }
        template<typename T>
        ref_ptr<OpSelect> user_sel(std::string name, std::string code, std::function<bool(T)> fn) {
            auto op      = ref_ptr<TOpSelect<T>>(new TOpSelect<T>());
            op->name     = name;
            op->key      = name + "_" + TypeTraits<T>::code;
            op->source   = std::move(code);
            op->function = std::move(fn);
            return op.template as<OpSelect>();
        }
    }// namespace
    namespace {
        template<typename T>
        ref_ptr<OpUnary> user_un(std::string name, std::string code, std::function<T(T)> fn) {
            auto op      = ref_ptr<TOpUnary<T>>(new TOpUnary<T>());
            const std::string c = TypeTraits<T>::code;
            op->name     = name;
            op->key      = name + "_" + c + c;
            op->source   = std::move(code);
            op->function = std::move(fn);
            return op.template as<OpUnary>();
        }
    }// namespace
    ref_ptr<OpUnary> OpUnary::make_int(std::string n, std::string c, std::function<T_INT(T_INT)> f) { return user_un<T_INT>(std::move(n), std::move(c), std::move(f)); }
    ref_ptr<OpUnary> OpUnary::make_uint(std::string n, std::string c, std::function<T_UINT(T_UINT)> f) { return user_un<T_UINT>(std::move(n), std::move(c), std::move(f)); }
    ref_ptr<OpUnary> OpUnary::make_float(std::string n, std::string c, std::function<T_FLOAT(T_FLOAT)> f) { return user_un<T_FLOAT>(std::move(n), std::move(c), std::move(f)); }
    ref_ptr<OpBinary> OpBinary::make_int(std::string n, std::string c, std::function<T_INT(T_INT, T_INT)> f) { return user_bin<T_INT>(std::move(n), std::move(c), std::move(f)); }
    ref_ptr<OpBinary> OpBinary::make_uint(std::string n, std::string c, std::function<T_UINT(T_UINT, T_UINT)> f) { return user_bin<T_UINT>(std::move(n), std::move(c), std::move(f)); }
    ref_ptr<OpBinary> OpBinary::make_float(std::string n, std::string c, std::function<T_FLOAT(T_FLOAT, T_FLOAT)> f) { return user_bin<T_FLOAT>(std::move(n), std::move(c), std::move(f)); }
    ref_ptr<OpSelect> OpSelect::make_int(std::string n, std::string c, std::function<bool(T_INT)> f) { return user_sel<T_INT>(std::move(n), std::move(c), std::move(f)); }
    ref_ptr<OpSelect> OpSelect::make_uint(std::string n, std::string c, std::function<bool(T_UINT)> f) { return user_sel<T_UINT>(std::move(n), std::move(c), std::move(f)); }
    ref_ptr<OpSelect> OpSelect::make_float(std::string n, std::string c, std::function<bool(T_FLOAT)> f) { return user_sel<T_FLOAT>(std::move(n), std::move(c), std::move(f)); }

    // ------------------------------------------------------------------------------------------------------ accelerator
    Status HipAccelerator::init() {
        if (splahip_device_count() <= 0) return Status::NoAcceleration;
        if (splahip_init(m_device) != SPLAHIP_OK) return Status::DeviceNotFound;
        char info[256];
        splahip_device_info(info, sizeof(info));
        m_description = info;
        return Status::Ok;
    }
    Status HipAccelerator::set_platform(int) { return Status::Ok; }// one platform: ROCm
    Status HipAccelerator::set_device(int index) {
        if (index < 0 || index >= splahip_device_count()) return Status::DeviceNotFound;
        m_device = index;
        return init();
    }
    Status HipAccelerator::set_queues_count(int) { return Status::Ok; }// one in-order stream

    // ---------------------------------------------------------------------------------------------------------- library
    struct Library::Impl {
        std::unique_ptr<Accelerator> accelerator;
        std::unique_ptr<Registry>    registry;
        std::unique_ptr<Dispatcher>  dispatcher;
        std::unique_ptr<Logger>      logger;
        bool                         force_no_acc = false;
    };

    Library::Library() : m_impl(new Impl()) {
        m_impl->logger     = std::make_unique<Logger>();
        m_impl->registry   = std::make_unique<Registry>();
        m_impl->dispatcher = std::make_unique<Dispatcher>();
        register_ops();
        register_algo_hip(m_impl->registry.get());
    }
    Library::~Library() {
        delete m_impl;
        m_impl = nullptr;
    }
    void Library::finalize() {
        if (m_impl) m_impl->accelerator.reset();
    }
    Status Library::set_accelerator(AcceleratorType accelerator) {
        if (accelerator == AcceleratorType::None) {
            m_impl->accelerator.reset();
            return Status::Ok;
        }
        if (accelerator == AcceleratorType::Hip) {
            auto         acc = std::make_unique<HipAccelerator>();
            const Status st  = acc->init();
            if (st != Status::Ok) {
                SPLA_LOG(st, "HIP accelerator unavailable: " << splahip_last_error());
                m_impl->accelerator.reset();
                return Status::NoAcceleration;
            }
            m_impl->accelerator = std::move(acc);
            return Status::Ok;
        }
        return Status::NoAcceleration;// OpenCL is not part of this build
    }
    Status Library::set_platform(int index) { return m_impl->accelerator ? m_impl->accelerator->set_platform(index) : Status::NoAcceleration; }
    Status Library::set_device(int index) { return m_impl->accelerator ? m_impl->accelerator->set_device(index) : Status::NoAcceleration; }
    Status Library::set_queues_count(int count) { return m_impl->accelerator ? m_impl->accelerator->set_queues_count(count) : Status::NoAcceleration; }
    Status Library::set_message_callback(MessageCallback callback) {
        m_impl->logger->set_callback(std::move(callback));
        return Status::Ok;
    }
    Status Library::set_default_callback() {
        m_impl->logger->set_callback([](Status status, const std::string& msg, const std::string& file, const std::string& function, int line) {
            std::fprintf(stderr, "[%s:%d %s] status %u: %s\n", file.c_str(), line, function.c_str(), static_cast<unsigned>(status), msg.c_str());
        });
        return Status::Ok;
    }
    Status Library::set_force_no_acceleration(bool value) {
        m_impl->force_no_acc = value;
        return Status::Ok;
    }
    bool   Library::is_set_force_no_acceleration() { return m_impl->force_no_acc; }
    Status Library::get_accelerator_info(std::string& info) {
        if (m_impl->accelerator) {
            info = m_impl->accelerator->get_name() + ": " + m_impl->accelerator->get_description();
            return Status::Ok;
        }
        info = "no acceleration";
        return Status::NoAcceleration;
    }
    Accelerator* Library::get_accelerator() { return m_impl->accelerator.get(); }
    Registry*    Library::get_registry() { return m_impl->registry.get(); }
    Dispatcher*  Library::get_dispatcher() { return m_impl->dispatcher.get(); }
    Logger*      Library::get_logger() { return m_impl->logger.get(); }

    // lazily constructed process-global; the first use tries to bring up the accelerator and stays usable for host-side
    // work when that fails (the reference does the same with OpenCL, src/library.cpp:218-238)
    Library* Library::get() {
        static std::unique_ptr<Library> instance;
        static bool                     constructing = false;
        if (!instance && !constructing) {
            constructing = true;
            instance     = std::make_unique<Library>();
            constructing = false;
            instance->set_accelerator(AcceleratorType::Hip);
        }
        return instance.get();
    }

    HipAccelerator* get_acc_hip() {
        Library* lib = Library::get();
        if (!lib || lib->is_set_force_no_acceleration()) return nullptr;
        return dynamic_cast<HipAccelerator*>(lib->get_accelerator());
    }

    // ------------------------------------------------------------------------------------------------------- dispatcher
    Status Dispatcher::dispatch(const DispatchContext& ctx) {
        Library*     lib = Library::get();
        Accelerator* acc = lib->get_accelerator();
        const std::string key = ctx.task->get_key();
        if (!acc || lib->is_set_force_no_acceleration()) {
            // this build has no "__cpu" algorithms to fall back to (they would be the reference's own)
            SPLA_LOG(Status::NoAcceleration, "no accelerator for " << key << " and no CPU backend in this build");
            return Status::NoAcceleration;
        }
        RegistryAlgo* algo = lib->get_registry()->find(key + acc->get_suffix());
        if (!algo) {
            SPLA_LOG(Status::NotImplemented, "no algorithm registered for key " << key << acc->get_suffix());
            return Status::NotImplemented;
        }
        // the call's Descriptor may ask for run-to-run identical fp32 sums: a per-host-thread setting of the kernel library, held for this call only
        // (the scope only ever RAISES the setting: a host thread that switched determinism on through the kernel-level call keeps it inside an operator-API
        // call whose Descriptor does not ask for it)
        struct DeterministicScope {
            bool raised = false;
            explicit DeterministicScope(bool on) {
                if (on) raised = splahip_set_deterministic(1) == 0;
            }
            ~DeterministicScope() {
                if (raised) (void) splahip_set_deterministic(0);
            }
        } scope(ctx.task->get_desc_or_default()->get_deterministic());
        // the same for the reference-order setting (fp32 sums folded as the reference CPU backend folds them)
        struct ReferenceOrderScope {
            bool raised = false;
            explicit ReferenceOrderScope(bool on) {
                if (on) raised = splahip_set_reference_order(1) == 0;
            }
            ~ReferenceOrderScope() {
                if (raised) (void) splahip_set_reference_order(0);
            }
        } order_scope(ctx.task->get_desc_or_default()->get_reference_order());
        try {
            return algo->execute(ctx);
        } catch (const std::exception& e) {
            SPLA_LOG(Status::Error, "algorithm " << algo->get_name() << " failed: " << e.what());
            return Status::Error;
        } catch (...) {
            SPLA_LOG(Status::Error, "algorithm " << algo->get_name() << " failed with an unknown exception");
            return Status::Error;
        }
    }

    // ------------------------------------------------------------------------------------------------------ exec + tasks
    namespace {
        Status run_or_return(ref_ptr<ScheduleTask> task, ref_ptr<ScheduleTask>* task_hnd) {
            if (task_hnd) {
                *task_hnd = task;
                return Status::Ok;
            }
            DispatchContext ctx;
            ctx.task = std::move(task);
            return Library::get()->get_dispatcher()->dispatch(ctx);
        }

        class ScheduleSerial final : public Schedule {
        public:
            void               set_label(std::string label) override { m_label = std::move(label); }
            const std::string& get_label() const override { return m_label; }
            Status             step_task(ref_ptr<ScheduleTask> task) override {
                m_steps.push_back({std::move(task)});
                return Status::Ok;
            }
            Status step_tasks(std::vector<ref_ptr<ScheduleTask>> tasks) override {
                m_steps.push_back(std::move(tasks));
                return Status::Ok;
            }
            Status submit() override {// ordered, serial (reference src/schedule/schedule_st.cpp:46-69)
                int step = 0;
                for (auto& tasks : m_steps) {
                    for (auto& task : tasks) {
                        DispatchContext ctx;
                        ctx.task    = task;
                        ctx.step_id = step;
                        const Status st = Library::get()->get_dispatcher()->dispatch(ctx);
                        if (st != Status::Ok) return st;
                    }
                    ++step;
                }
                return Status::Ok;
            }

        private:
            std::vector<std::vector<ref_ptr<ScheduleTask>>> m_steps;
            std::string                                     m_label;
        };
    }// namespace

    ref_ptr<Schedule> make_schedule() {
        Library::get();
        return ref_ptr<Schedule>(new ScheduleSerial());
    }

    Status exec_mxm(ref_ptr<Matrix> R, ref_ptr<Matrix> A, ref_ptr<Matrix> B, ref_ptr<OpBinary> op_multiply, ref_ptr<OpBinary> op_add,
                    ref_ptr<Scalar> init, ref_ptr<Descriptor> desc, ref_ptr<ScheduleTask>* task_hnd) {
        if (!R || !A || !B || !op_multiply || !op_add || !init) return Status::InvalidArgument;
        auto task         = ref_ptr<ScheduleTask_mxm>(new ScheduleTask_mxm());
        task->R           = std::move(R);
        task->A           = std::move(A);
        task->B           = std::move(B);
        task->op_multiply = std::move(op_multiply);
        task->op_add      = std::move(op_add);
        task->init        = std::move(init);
        task->desc        = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

    Status exec_mxmT_masked(ref_ptr<Matrix> R, ref_ptr<Matrix> mask, ref_ptr<Matrix> A, ref_ptr<Matrix> B, ref_ptr<OpBinary> op_multiply,
                            ref_ptr<OpBinary> op_add, ref_ptr<OpSelect> op_select, ref_ptr<Scalar> init, ref_ptr<Descriptor> desc,
                            ref_ptr<ScheduleTask>* task_hnd) {
        if (!R || !mask || !A || !B || !op_multiply || !op_add || !op_select || !init) return Status::InvalidArgument;
        auto task         = ref_ptr<ScheduleTask_mxmT_masked>(new ScheduleTask_mxmT_masked());
        task->R           = std::move(R);
        task->mask        = std::move(mask);
        task->A           = std::move(A);
        task->B           = std::move(B);
        task->op_multiply = std::move(op_multiply);
        task->op_add      = std::move(op_add);
        task->op_select   = std::move(op_select);
        task->init        = std::move(init);
        task->desc        = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

    Status exec_mxv_masked(ref_ptr<Vector> r, ref_ptr<Vector> mask, ref_ptr<Matrix> M, ref_ptr<Vector> v, ref_ptr<OpBinary> op_multiply,
                           ref_ptr<OpBinary> op_add, ref_ptr<OpSelect> op_select, ref_ptr<Scalar> init, ref_ptr<Descriptor> desc,
                           ref_ptr<ScheduleTask>* task_hnd) {
        if (!r || !mask || !M || !v || !op_multiply || !op_add || !op_select || !init) return Status::InvalidArgument;
        auto task         = ref_ptr<ScheduleTask_mxv_masked>(new ScheduleTask_mxv_masked());
        task->r           = std::move(r);
        task->mask        = std::move(mask);
        task->M           = std::move(M);
        task->v           = std::move(v);
        task->op_multiply = std::move(op_multiply);
        task->op_add      = std::move(op_add);
        task->op_select   = std::move(op_select);
        task->init        = std::move(init);
        task->desc        = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

    Status exec_vxm_masked(ref_ptr<Vector> r, ref_ptr<Vector> mask, ref_ptr<Vector> v, ref_ptr<Matrix> M, ref_ptr<OpBinary> op_multiply,
                           ref_ptr<OpBinary> op_add, ref_ptr<OpSelect> op_select, ref_ptr<Scalar> init, ref_ptr<Descriptor> desc,
                           ref_ptr<ScheduleTask>* task_hnd) {
        if (!r || !mask || !v || !M || !op_multiply || !op_add || !op_select || !init) return Status::InvalidArgument;
        auto task         = ref_ptr<ScheduleTask_vxm_masked>(new ScheduleTask_vxm_masked());
        task->r           = std::move(r);
        task->mask        = std::move(mask);
        task->v           = std::move(v);
        task->M           = std::move(M);
        task->op_multiply = std::move(op_multiply);
        task->op_add      = std::move(op_add);
        task->op_select   = std::move(op_select);
        task->init        = std::move(init);
        task->desc        = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

    Status exec_v_eadd_fdb(ref_ptr<Vector> r, ref_ptr<Vector> v, ref_ptr<Vector> fdb, ref_ptr<OpBinary> op, ref_ptr<Descriptor> desc,
                           ref_ptr<ScheduleTask>* task_hnd) {
        if (!r || !v || !fdb || !op) return Status::InvalidArgument;
        auto task  = ref_ptr<ScheduleTask_v_eadd_fdb>(new ScheduleTask_v_eadd_fdb());
        task->r    = std::move(r);
        task->v    = std::move(v);
        task->fdb  = std::move(fdb);
        task->op   = std::move(op);
        task->desc = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

    Status exec_v_assign_masked(ref_ptr<Vector> r, ref_ptr<Vector> mask, ref_ptr<Scalar> value, ref_ptr<OpBinary> op_assign, ref_ptr<OpSelect> op_select,
                                ref_ptr<Descriptor> desc, ref_ptr<ScheduleTask>* task_hnd) {
        if (!r || !mask || !value || !op_assign || !op_select) return Status::InvalidArgument;
        auto task       = ref_ptr<ScheduleTask_v_assign_masked>(new ScheduleTask_v_assign_masked());
        task->r         = std::move(r);
        task->mask      = std::move(mask);
        task->value     = std::move(value);
        task->op_assign = std::move(op_assign);
        task->op_select = std::move(op_select);
        task->desc      = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

    Status exec_v_count_mf(ref_ptr<Scalar> r, ref_ptr<Vector> v, ref_ptr<Descriptor> desc, ref_ptr<ScheduleTask>* task_hnd) {
        if (!r || !v) return Status::InvalidArgument;
        auto task  = ref_ptr<ScheduleTask_v_count_mf>(new ScheduleTask_v_count_mf());
        task->r    = std::move(r);
        task->v    = std::move(v);
        task->desc = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

    Status exec_v_eadd(ref_ptr<Vector> r, ref_ptr<Vector> u, ref_ptr<Vector> v, ref_ptr<OpBinary> op, ref_ptr<Descriptor> desc, ref_ptr<ScheduleTask>* task_hnd) {
        if (!r || !u || !v || !op) return Status::InvalidArgument;
        auto task  = ref_ptr<ScheduleTask_v_eadd>(new ScheduleTask_v_eadd());
        task->r    = std::move(r);
        task->u    = std::move(u);
        task->v    = std::move(v);
        task->op   = std::move(op);
        task->desc = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

    Status exec_v_reduce(ref_ptr<Scalar> r, ref_ptr<Scalar> s, ref_ptr<Vector> v, ref_ptr<OpBinary> op_reduce, ref_ptr<Descriptor> desc,
                         ref_ptr<ScheduleTask>* task_hnd) {
        if (!r || !s || !v || !op_reduce) return Status::InvalidArgument;
        auto task       = ref_ptr<ScheduleTask_v_reduce>(new ScheduleTask_v_reduce());
        task->r         = std::move(r);
        task->s         = std::move(s);
        task->v         = std::move(v);
        task->op_reduce = std::move(op_reduce);
        task->desc      = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

    Status exec_m_reduce(ref_ptr<Scalar> r, ref_ptr<Scalar> s, ref_ptr<Matrix> M, ref_ptr<OpBinary> op_reduce, ref_ptr<Descriptor> desc,
                         ref_ptr<ScheduleTask>* task_hnd) {
        if (!r || !s || !M || !op_reduce) return Status::InvalidArgument;
        auto task       = ref_ptr<ScheduleTask_m_reduce>(new ScheduleTask_m_reduce());
        task->r         = std::move(r);
        task->s         = std::move(s);
        task->M         = std::move(M);
        task->op_reduce = std::move(op_reduce);
        task->desc      = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

    Status exec_m_transpose(ref_ptr<Matrix> R, ref_ptr<Matrix> M, ref_ptr<OpUnary> op_apply, ref_ptr<Descriptor> desc, ref_ptr<ScheduleTask>* task_hnd) {
        if (!R || !M || !op_apply) return Status::InvalidArgument;
        auto task      = ref_ptr<ScheduleTask_m_transpose>(new ScheduleTask_m_transpose());
        task->R        = std::move(R);
        task->M        = std::move(M);
        task->op_apply = std::move(op_apply);
        task->desc     = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

    Status exec_v_map(ref_ptr<Vector> r, ref_ptr<Vector> v, ref_ptr<OpUnary> op, ref_ptr<Descriptor> desc, ref_ptr<ScheduleTask>* task_hnd) {
        if (!r || !v || !op) return Status::InvalidArgument;
        auto task  = ref_ptr<ScheduleTask_v_map>(new ScheduleTask_v_map());
        task->r    = std::move(r);
        task->v    = std::move(v);
        task->op   = std::move(op);
        task->desc = std::move(desc);
        return run_or_return(task.as<ScheduleTask>(), task_hnd);
    }

}// namespace spla
