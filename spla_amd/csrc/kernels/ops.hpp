// ops.hpp — device-side scalar operations of the semirings, as compile-time functors plus a runtime-switched
// fallback.  Semantics follow reference src/op.cpp:194-266 one to one (ids: include/spla_hip_ops.h).
//
// The reference injects an op into its OpenCL kernels as source text and JIT-compiles one program per
// (kernel, type, ops) combination (reference src/opencl/cl_program_builder.cpp:65-120).  Here the common semirings
// are ahead-of-time template instantiations and every other combination runs through `DynBin`, a functor that
// switches on the op id (wave-uniform branch), so any (op_mult, op_add) pair is served on the device.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/spla_hip_ops.h"

namespace splahip {

    using uint = uint32_t;

    template<typename T>
    struct TypeId;
    template<>
    struct TypeId<int32_t> { static constexpr int value = SPLAHIP_INT; };
    template<>
    struct TypeId<uint32_t> { static constexpr int value = SPLAHIP_UINT; };
    template<>
    struct TypeId<float> { static constexpr int value = SPLAHIP_FLOAT; };

    template<typename T>
    __host__ __device__ inline T from_bits(uint32_t w) {
        union {
            uint32_t u;
            T        t;
        } c;
        c.u = w;
        return c.t;
    }
    template<typename T>
    __host__ __device__ inline uint32_t to_bits(T t) {
        union {
            uint32_t u;
            T        t;
        } c;
        c.t = t;
        return c.u;
    }

    // int32 +,-,* wrap (two's complement) instead of the CPU's signed-overflow UB
    template<typename T>
    __device__ __forceinline__ T wrap_add(T a, T b) {
        if constexpr (std::is_same<T, int32_t>::value) return (int32_t) ((uint32_t) a + (uint32_t) b);
        else
            return a + b;
    }
    template<typename T>
    __device__ __forceinline__ T wrap_sub(T a, T b) {
        if constexpr (std::is_same<T, int32_t>::value) return (int32_t) ((uint32_t) a - (uint32_t) b);
        else
            return a - b;
    }
    template<typename T>
    __device__ __forceinline__ T wrap_mul(T a, T b) {
        if constexpr (std::is_same<T, int32_t>::value) return (int32_t) ((uint32_t) a * (uint32_t) b);
        else
            return a * b;
    }
    template<typename T>
    __device__ __forceinline__ T safe_div(T a, T b) {
        if constexpr (std::is_same<T, float>::value) return a / b;
        else if constexpr (std::is_same<T, int32_t>::value)
            return b == 0 ? 0 : ((a == INT32_MIN && b == -1) ? a : a / b);// UB on the CPU; defined here
        else
            return b == 0 ? 0u : a / b;
    }

    // Evaluate binary op `OP` (a splahip_binop id, compile-time constant).
    template<int OP, typename T>
    __device__ __forceinline__ T eval_bin(T a, T b) {
        if constexpr (OP == SPLAHIP_PLUS) return wrap_add(a, b);
        else if constexpr (OP == SPLAHIP_MINUS)
            return wrap_sub(a, b);
        else if constexpr (OP == SPLAHIP_MULT)
            return wrap_mul(a, b);
        else if constexpr (OP == SPLAHIP_DIV)
            return safe_div(a, b);
        else if constexpr (OP == SPLAHIP_MINUS_POW2) {
            T d = wrap_sub(a, b);
            return wrap_mul(d, d);
        } else if constexpr (OP == SPLAHIP_FIRST)
            return a;
        else if constexpr (OP == SPLAHIP_SECOND)
            return b;
        else if constexpr (OP == SPLAHIP_BONE)
            return T(1);
        else if constexpr (OP == SPLAHIP_MIN)
            return (b < a) ? b : a;// std::min
        else if constexpr (OP == SPLAHIP_MAX)
            return (a < b) ? b : a;// std::max
        else if constexpr (OP == SPLAHIP_LOR)
            return (a || b) ? T(1) : T(0);
        else if constexpr (OP == SPLAHIP_LAND)
            return (a && b) ? T(1) : T(0);
        else if constexpr (OP == SPLAHIP_BOR || OP == SPLAHIP_BAND || OP == SPLAHIP_BXOR) {
            if constexpr (std::is_same<T, float>::value) return a;// not defined for F; never dispatched
            else if constexpr (OP == SPLAHIP_BOR)
                return a | b;
            else if constexpr (OP == SPLAHIP_BAND)
                return a & b;
            else
                return a ^ b;
        } else
            return a;
    }

    // compile-time functor
    template<int OP>
    struct StaBin {
        static constexpr bool is_static = true;
        static constexpr int  op_id     = OP;
        __host__ __device__ explicit StaBin(int = OP) {}
        template<typename T>
        __device__ __forceinline__ T operator()(T a, T b) const { return eval_bin<OP>(a, b); }
    };

    // runtime-switched functor (any op id)
    struct DynBin {
        static constexpr bool is_static = false;
        int                   op;
        __host__ __device__ explicit DynBin(int id) : op(id) {}
        template<typename T>
        __device__ __forceinline__ T operator()(T a, T b) const {
            switch (op) {
                case SPLAHIP_PLUS: return eval_bin<SPLAHIP_PLUS>(a, b);
                case SPLAHIP_MINUS: return eval_bin<SPLAHIP_MINUS>(a, b);
                case SPLAHIP_MULT: return eval_bin<SPLAHIP_MULT>(a, b);
                case SPLAHIP_DIV: return eval_bin<SPLAHIP_DIV>(a, b);
                case SPLAHIP_MINUS_POW2: return eval_bin<SPLAHIP_MINUS_POW2>(a, b);
                case SPLAHIP_FIRST: return eval_bin<SPLAHIP_FIRST>(a, b);
                case SPLAHIP_SECOND: return eval_bin<SPLAHIP_SECOND>(a, b);
                case SPLAHIP_BONE: return eval_bin<SPLAHIP_BONE>(a, b);
                case SPLAHIP_MIN: return eval_bin<SPLAHIP_MIN>(a, b);
                case SPLAHIP_MAX: return eval_bin<SPLAHIP_MAX>(a, b);
                case SPLAHIP_LOR: return eval_bin<SPLAHIP_LOR>(a, b);
                case SPLAHIP_LAND: return eval_bin<SPLAHIP_LAND>(a, b);
                case SPLAHIP_BOR: return eval_bin<SPLAHIP_BOR>(a, b);
                case SPLAHIP_BAND: return eval_bin<SPLAHIP_BAND>(a, b);
                case SPLAHIP_BXOR: return eval_bin<SPLAHIP_BXOR>(a, b);
                default: return a;
            }
        }
    };

    // Run `f` with a COMPILE-TIME functor where the op allows it: a loop that applies a runtime-switched op once per element pays its 15-way switch per element — the
    // chain of a long row in the ordered kernels ran 12 ms per R-MAT 23 call through DynBin against 2.3 ms through StaBin<PLUS> (round 6).  The ops listed are the ones
    // whose folds must keep the reference's order (everything else has faster kernels); any other op runs `f` with the dynamic functor itself.
    template<class F>
    __device__ __forceinline__ void with_static_op(const DynBin& op, F&& f) {
        switch (op.op) {
            case SPLAHIP_PLUS: f(StaBin<SPLAHIP_PLUS>()); break;
            case SPLAHIP_MINUS: f(StaBin<SPLAHIP_MINUS>()); break;
            case SPLAHIP_MULT: f(StaBin<SPLAHIP_MULT>()); break;
            case SPLAHIP_DIV: f(StaBin<SPLAHIP_DIV>()); break;
            case SPLAHIP_MINUS_POW2: f(StaBin<SPLAHIP_MINUS_POW2>()); break;
            case SPLAHIP_FIRST: f(StaBin<SPLAHIP_FIRST>()); break;
            case SPLAHIP_SECOND: f(StaBin<SPLAHIP_SECOND>()); break;
            default: f(op); break;
        }
    }
    template<int OP, class F>
    __device__ __forceinline__ void with_static_op(const StaBin<OP>& op, F&& f) {
        f(op);
    }

    __host__ __device__ inline int binop_id(const DynBin& f) { return f.op; }
    template<int OP>
    __host__ __device__ constexpr int binop_id(const StaBin<OP>&) { return OP; }

    // op_add as the kernels that RE-ASSOCIATE a fold use it (tree / lane-strided / segment partials that start at a product instead of
    // at `init`).  Identical to the op except for fp32 MIN / MAX, where the reference's left fold `b < a ? b : a` from a non-NaN `init`
    // (src/op.cpp:151-155,222-227, src/cpu/cpu_mxv.hpp:94-99) never picks a NaN second argument: a partial that STARTS at a NaN product
    // would keep it, so the join drops a NaN on either side.  With `init` folded in by the same functor the result is the left fold's,
    // bit for bit, for every input whose `init` is not NaN; a NaN `init` sticks in the reference and is served as op_add = FIRST
    // (canonical_add below).  Wrap at kernel entry: `const Reassoc<Add> add(add_raw);`.
    template<class Add>
    struct Reassoc {
        Add raw;
        __host__ __device__ explicit Reassoc(const Add& a) : raw(a) {}
        template<typename T>
        __device__ __forceinline__ T operator()(T x, T y) const {
            if constexpr (std::is_same<T, float>::value) {
                const int op = binop_id(raw);
                if (op == SPLAHIP_MIN || op == SPLAHIP_MAX) return y != y ? x : (x != x ? y : raw(x, y));
            }
            return raw(x, y);
        }
    };
    template<class Add>
    __host__ __device__ inline int binop_id(const Reassoc<Add>& f) { return binop_id(f.raw); }

    // select op, evaluated once per row: always runtime-switched
    template<typename T>
    __device__ __forceinline__ bool eval_sel(int op, T a) {
        switch (op) {
            case SPLAHIP_EQZERO: return a == T(0);
            case SPLAHIP_NQZERO: return a != T(0);
            case SPLAHIP_GTZERO: return a > T(0);
            case SPLAHIP_GEZERO: return a >= T(0);
            case SPLAHIP_LTZERO: return a < T(0);
            case SPLAHIP_LEZERO: return a <= T(0);
            case SPLAHIP_ALWAYS: return true;
            default: return false;// NEVER
        }
    }

    // unary op (m_transpose's op_apply, v_map): evaluated once per stored entry of a streaming pass, runtime-switched
    template<typename T>
    __device__ __forceinline__ T eval_un(int op, T a) {
        if constexpr (std::is_same<T, float>::value) {
            switch (op) {
                case SPLAHIP_AINV: return -a;
                case SPLAHIP_MINV: return 1.0f / a;
                case SPLAHIP_LNOT: return a != 0.0f ? 0.0f : 1.0f;
                case SPLAHIP_UONE: return 1.0f;
                case SPLAHIP_ABS: return fabsf(a);
                // the reference calls the unqualified C functions on a float (src/op.cpp:121-133): evaluated in double, rounded once
                case SPLAHIP_SQRT: return sqrtf(a);// correctly rounded either way
                case SPLAHIP_LOG: return (float) log((double) a);
                case SPLAHIP_EXP: return (float) exp((double) a);
                case SPLAHIP_SIN: return (float) sin((double) a);
                case SPLAHIP_COS: return (float) cos((double) a);
                case SPLAHIP_TAN: return (float) tan((double) a);
                case SPLAHIP_ASIN: return (float) asin((double) a);
                case SPLAHIP_ACOS: return (float) acos((double) a);
                case SPLAHIP_ATAN: return (float) atan((double) a);
                case SPLAHIP_CEIL: return ceilf(a);
                case SPLAHIP_FLOOR: return floorf(a);
                case SPLAHIP_ROUND: return roundf(a);
                case SPLAHIP_TRUNC: return truncf(a);
                default: return a;// IDENTITY
            }
        } else {
            using U = uint32_t;
            switch (op) {
                case SPLAHIP_AINV: return T(U(0) - U(a));
                case SPLAHIP_MINV: return a == T(0) ? T(0) : T(T(1) / a);
                case SPLAHIP_LNOT: return a != T(0) ? T(0) : T(1);
                case SPLAHIP_UONE: return T(1);
                case SPLAHIP_ABS:
                    if constexpr (std::is_same<T, int32_t>::value) return a < 0 ? T(U(0) - U(a)) : a;
                    else
                        return a;
                case SPLAHIP_BNOT: return T(~U(a));
                default: return a;// IDENTITY
            }
        }
    }
    inline bool unop_valid_for(int op, int type) {
        if (op < 0 || op >= SPLAHIP_UNOP_COUNT) return false;
        if (type == SPLAHIP_FLOAT) return op != SPLAHIP_BNOT;
        return op <= SPLAHIP_BNOT;
    }

    // Algebraic traits of an op used as the ADD of a semiring (host side; decide which kernel may be used).
    //   associative: a tree / segmented reduction gives the left fold's result (bit-exact for I/U; fp32 PLUS and
    //                MULT are treated as associative under the stated 1e-5 relative tolerance)
    //   commutative: partial results may be combined in any order
    inline bool binop_is_associative(int op) {
        switch (op) {
            case SPLAHIP_MINUS:
            case SPLAHIP_DIV:
            case SPLAHIP_MINUS_POW2: return false;
            default: return true;
        }
    }
    // may a kernel fold with this op_add in another association than the reference's left fold?  Associativity — except fp32 MULT: a
    // tree of products overflows to inf (or underflows to 0) where the sequential fold does not (a row of 600 values around 2 under
    // init 0: the reference keeps 0, a tree reaches 0 x inf = NaN), which no rounding tolerance covers.  fp32 PLUS differs by rounding
    // only (the stated 1e-5 bar).
    inline bool fold_is_reorderable(int op, int type) { return binop_is_associative(op) && !(type == SPLAHIP_FLOAT && op == SPLAHIP_MULT); }

    // fp32 MIN / MAX from a NaN `init`: `b < a` is false for every b, so the reference's fold returns `init` whatever the products are
    // (and its early exit `sum != init` fires on the first product with the sum still `init`).  That is exactly FIRST; every entry point
    // that folds from `init` (mxv, mxm, mxmT, reduce) replaces op_add by it, so no kernel ever sees a NaN `init` under MIN / MAX.
    inline int canonical_add(int type, int op_add, uint32_t init_bits) {
        if (type == SPLAHIP_FLOAT && (op_add == SPLAHIP_MIN || op_add == SPLAHIP_MAX) && (init_bits & 0x7fffffffu) > 0x7f800000u) return SPLAHIP_FIRST;
        return op_add;
    }

    inline bool binop_is_commutative(int op) {
        switch (op) {
            case SPLAHIP_MINUS:
            case SPLAHIP_DIV:
            case SPLAHIP_FIRST:
            case SPLAHIP_SECOND: return false;
            default: return true;
        }
    }
    inline bool binop_valid_for(int op, int type) {
        if (op < 0 || op >= SPLAHIP_BINOP_COUNT) return false;
        if (type == SPLAHIP_FLOAT && (op == SPLAHIP_BOR || op == SPLAHIP_BAND || op == SPLAHIP_BXOR)) return false;
        return true;
    }

}// namespace splahip
