"""Numeric ids of element types and semiring ops (mirror of include/spla_hip_ops.h).

Names follow the reference's op prefixes (reference src/op.cpp:194-266): e.g. ``BINOP["PLUS"]``; the reference
key of an op for a type is ``f"{name}_{code*3}"`` for binary ops (``PLUS_FFF``) and ``f"{name}_{code}"`` for
select ops (``EQZERO_I``), reference src/core/top.hpp:84-89,111-114.
"""
from __future__ import annotations

import numpy as np

TYPE = {"INT": 0, "UINT": 1, "FLOAT": 2}
TYPE_CODE = {0: "I", 1: "U", 2: "F"}
TYPE_NP = {0: np.int32, 1: np.uint32, 2: np.float32}

BINOP = {
    "PLUS": 0, "MINUS": 1, "MULT": 2, "DIV": 3, "MINUS_POW2": 4, "FIRST": 5, "SECOND": 6, "BONE": 7,
    "MIN": 8, "MAX": 9, "LOR": 10, "LAND": 11, "BOR": 12, "BAND": 13, "BXOR": 14,
}
SELOP = {"EQZERO": 0, "NQZERO": 1, "GTZERO": 2, "GEZERO": 3, "LTZERO": 4, "LEZERO": 5, "ALWAYS": 6, "NEVER": 7}
UNOP = {
    "IDENTITY": 0, "AINV": 1, "MINV": 2, "LNOT": 3, "UONE": 4, "ABS": 5, "BNOT": 6, "SQRT": 7, "LOG": 8, "EXP": 9, "SIN": 10,
    "COS": 11, "TAN": 12, "ASIN": 13, "ACOS": 14, "ATAN": 15, "CEIL": 16, "FLOOR": 17, "ROUND": 18, "TRUNC": 19,
}
FLOAT_ONLY_UNOPS = tuple(n for n, i in UNOP.items() if i >= 7)

BITWISE = ("BOR", "BAND", "BXOR")
NON_ASSOCIATIVE = ("MINUS", "DIV", "MINUS_POW2")
NON_COMMUTATIVE = ("MINUS", "DIV", "FIRST", "SECOND")


def type_id(t) -> int:
    if isinstance(t, str):
        return TYPE[t.upper()]
    return int(t)


def np_dtype(t):
    return TYPE_NP[type_id(t)]


def binop_id(name) -> int:
    return BINOP[name.upper()] if isinstance(name, str) else int(name)


def selop_id(name) -> int:
    return SELOP[name.upper()] if isinstance(name, str) else int(name)


def unop_id(name) -> int:
    return UNOP[name.upper()] if isinstance(name, str) else int(name)


def unop_valid(name: str, t) -> bool:
    if type_id(t) == TYPE["FLOAT"]:
        return name.upper() != "BNOT"
    return name.upper() not in FLOAT_ONLY_UNOPS


def binop_valid(name: str, t) -> bool:
    return not (type_id(t) == TYPE["FLOAT"] and name.upper() in BITWISE)


def scalar_bits(value, t) -> int:
    """uint32 bit pattern of a scalar of element type t (how `init` crosses the C ABI)."""
    return int(np.array([value]).astype(np_dtype(t)).view(np.uint32)[0])
