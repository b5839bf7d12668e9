/*
 * spla_hip_ops.h — numeric identifiers shared by the HIP kernels, the host layer and the test oracle.
 *
 * The reference identifies an operation by a string key built from its name and type codes
 * (reference: src/core/top.hpp:84-89,111-114, e.g. "PLUS_FFF", "MIN_III", "EQZERO_I") and an element type by a
 * one-letter code (reference: src/type.cpp:32-35).  Across a plain C ABI those become small integers; the
 * host layer maps key -> id (spla_amd/csrc/host/ops.cpp) and an unknown key yields "not found" so a caller can
 * report NotImplemented exactly as the reference dispatcher does (reference: src/core/dispatcher.cpp:82-83).
 *
 * The order of the binary/select lists follows the registration order in reference src/op.cpp:194-266.
 */
#ifndef SPLA_HIP_OPS_H
#define SPLA_HIP_OPS_H

#ifdef __cplusplus
extern "C" {
#endif

/* element types: 4-byte scalars only (reference: src/type.cpp:32-35; containers exist for I/U/F only) */
typedef enum splahip_type {
    SPLAHIP_INT   = 0, /* int32_t  "I" */
    SPLAHIP_UINT  = 1, /* uint32_t "U" */
    SPLAHIP_FLOAT = 2  /* float    "F" */
} splahip_type;

/* binary ops T x T -> T (reference: src/op.cpp:194-241) */
typedef enum splahip_binop {
    SPLAHIP_PLUS       = 0,  /* a + b            */
    SPLAHIP_MINUS      = 1,  /* a - b            */
    SPLAHIP_MULT       = 2,  /* a * b            */
    SPLAHIP_DIV        = 3,  /* a / b            */
    SPLAHIP_MINUS_POW2 = 4,  /* (a-b)*(a-b)      */
    SPLAHIP_FIRST      = 5,  /* a                */
    SPLAHIP_SECOND     = 6,  /* b                */
    SPLAHIP_BONE       = 7,  /* 1                */
    SPLAHIP_MIN        = 8,  /* std::min(a,b): b<a ? b : a */
    SPLAHIP_MAX        = 9,  /* std::max(a,b): a<b ? b : a */
    SPLAHIP_LOR        = 10, /* (a || b) -> 0/1  */
    SPLAHIP_LAND       = 11, /* (a && b) -> 0/1  */
    SPLAHIP_BOR        = 12, /* a | b   (I,U)    */
    SPLAHIP_BAND       = 13, /* a & b   (I,U)    */
    SPLAHIP_BXOR       = 14, /* a ^ b   (I,U)    */
    SPLAHIP_BINOP_COUNT = 15
} splahip_binop;

/* select ops T -> bool (reference: src/op.cpp:243-266) */
typedef enum splahip_selop {
    SPLAHIP_EQZERO = 0,
    SPLAHIP_NQZERO = 1,
    SPLAHIP_GTZERO = 2,
    SPLAHIP_GEZERO = 3,
    SPLAHIP_LTZERO = 4,
    SPLAHIP_LEZERO = 5,
    SPLAHIP_ALWAYS = 6,
    SPLAHIP_NEVER  = 7,
    SPLAHIP_SELOP_COUNT = 8
} splahip_selop;

/* unary ops T -> T (reference: src/op.cpp:99-133; BNOT for I,U only, SQRT..TRUNC for F only).  Integer MINV by zero
 * and ABS(INT_MIN) are undefined in the reference; here 1/0 = 0 and ABS wraps. */
typedef enum splahip_unop {
    SPLAHIP_IDENTITY = 0,  /* a          */
    SPLAHIP_AINV     = 1,  /* -a         */
    SPLAHIP_MINV     = 2,  /* 1 / a      */
    SPLAHIP_LNOT     = 3,  /* !(a != 0) -> 0/1 */
    SPLAHIP_UONE     = 4,  /* 1          */
    SPLAHIP_ABS      = 5,  /* |a|        */
    SPLAHIP_BNOT     = 6,  /* ~a  (I,U)  */
    SPLAHIP_SQRT     = 7,  /* F only from here */
    SPLAHIP_LOG      = 8,
    SPLAHIP_EXP      = 9,
    SPLAHIP_SIN      = 10,
    SPLAHIP_COS      = 11,
    SPLAHIP_TAN      = 12,
    SPLAHIP_ASIN     = 13,
    SPLAHIP_ACOS     = 14,
    SPLAHIP_ATAN     = 15,
    SPLAHIP_CEIL     = 16,
    SPLAHIP_FLOOR    = 17,
    SPLAHIP_ROUND    = 18,
    SPLAHIP_TRUNC    = 19,
    SPLAHIP_UNOP_COUNT = 20
} splahip_unop;

#ifdef __cplusplus
}
#endif

#endif /* SPLA_HIP_OPS_H */
