// mxv_cob.hpp — masked SpMV over the "column-ordered blocks" copy of a CSR matrix (cob.hip builds it).
//
// Semantics: reference Algo_mxv_masked_cpu<T>::execute, src/cpu/cpu_mxv.hpp:88-103, for an op_add that is associative AND
// commutative and has a fold identity (PLUS, MIN, MAX, BOR, BAND, BXOR on every type they exist for; LOR / LAND on the truth value):
//     r[i] = select(mask[i]) ? add(init, acc_i) : init,   acc_i = fold over row i's products in COLUMN-BLOCK order
// so int / uint results are bit-identical to the reference and fp32 PLUS differs by rounding only (the stated 1e-5 bar).
//
// Why this form exists (profiles/r01_notes.md): with one 4-byte gather of v per entry every L1-missing gather costs a
// 128-byte request, and the chip serves ~110 G such requests per second.  Here the rows are grouped into RANGES of at most R rows whose
// result slice lives in LDS, and a range's entries are stored sorted by column, so the 64 lanes of a wave gather ascending, mostly
// neighbouring columns of v: one request serves several products, and the products are added into the LDS slice with ds_add / ds_min /
// ds_or ... — no row walk, no segmented reduction, no padding by row length.
//
// Which rows share a range decides how many distinct lines of v a range touches (round 3, tools/exp/sim_ranges.py: R-MAT 22, 128-byte
// lines summed over ranges: 6.5 M for 2^14 consecutive rows; 3.0 M for the grouping below):
//   * HUB ranges: the rows of highest degree, sorted by degree, R per range — a few ranges that hold most of the entries, cut into
//     many pieces by column; slot s of hub range q is row hub_row[q R + s]; their results are scattered to r one word at a time
//   * NATURAL ranges: consecutive rows in their own order, but only the rows that hold an entry and are no hub take a slot, so a range
//     spans R such rows however many empty rows lie between them; slot16[row] = slot, 0xFFFF = no entry (r = init), 0xFFFE = hub row
//     (written by its hub range); a natural range's rows are written in order, 16 bytes per lane where no hub row intervenes
//
// Layout (all built on the device, cob.hip):
//   chunk  = 256 entry slots = one wave iteration; THREE packed streams, 7 bytes per slot (round 5; 8 until then): sl (16-bit slot in range), dl (8-bit column
//            DELTA to the slot before it in sorted order; 0 for the chunk's first slot), ax (value word), and hdr[chunk] = column of the first slot.  The
//            element at memory position 4 l + j of a stream is the chunk's sorted slot 64 j + l: a lane's loads (8 + 4 + 16 bytes) bring its four slots, and the
//            wave's j-th gather instruction covers 64 CONSECUTIVE sorted slots.  cob_decode turns a lane's packed words into the four words
//            slot << 16 | column - hdr[chunk] the loop works on (two wave scans over the deltas: a chunk's columns span < 2^16).
//            a gap of more than 255 columns between two entries is bridged by padding slots (one per 255 columns), built into the copy; a padding / bridging
//            slot carries a slot value >= R, one of the spare words behind the LDS slice (kCobSpare: one per lane), value 0, and gathers the chunk's base
//            column — no count, no predicate in the loop
//            a chunk never straddles a (range, column window of 2^16 columns) group; only a group's last chunk is padded
//   piece  = consecutive chunks of ONE range = one workgroup's work: zero the LDS slice, walk the chunks (wave w takes chunks
//            w, w + waves, ...), then either finish the rows (a range that is a single piece) or add the slice into the range's shared
//            accumulator in global memory (memory-side atomics of op_add) and take a ticket: the piece that arrives last reads the
//            accumulator back, re-arms it for the next call, and finishes the rows — one launch, no slabs, no combine pass
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_utils.hpp"
#include "ops.hpp"

namespace splahip {

#ifdef SPLAHIP_COB_TIMING
    // DIAGNOSTIC BUILD ONLY (tools/exp/exp_cob_timing.py): per workgroup of the last cob launch: [0] s_memrealtime at entry, [1] after the slice is zeroed,
    // [2] after the chunk loop, [3] at exit (100 MHz ticks), [4] XCC id, [5] HW_ID, [6] chunks of the piece, [7] slab id
    static __device__ unsigned long long g_cob_t[8][8192];
#endif

    struct CobTile {
        uint32_t first_chunk;
        uint32_t n_chunks;
        uint32_t range;
        uint32_t multi;// index of the range's shared accumulator, or 0xFFFFFFFF when the piece is the whole range (it finishes its rows itself)
    };
    struct CobRange {
        uint32_t row0, row1;// natural range: its rows [row0, row1); row0 is a multiple of 8
        uint32_t hub_base;  // hub range: first word of its slot -> row list in hub_row; 0xFFFFFFFF for a natural range
        uint32_t n_pieces;
    };
    constexpr uint32_t kCobNoLast    = 0xFFFFFFFFu;// CobRange::n_pieces: no piece of this range runs its row pass (cob_hub_finish_kernel does, after the launch)
    constexpr uint32_t kCobSlotEmpty = 0xFFFFu;// slot16: the row holds no entry
    constexpr uint32_t kCobSlotHub   = 0xFFFEu;// slot16: the row belongs to a hub range

    constexpr uint32_t kCobChunk  = 256;// slots per chunk
    // spare words behind a range's R slice words: the padding slot at memory position 4 l + j of a chunk carries the slot value R + l, so the 64 lanes of a
    // wave add their padding products into 64 DIFFERENT words.  (One shared spare word — rounds 2 and early 3 — made every padded chunk 64 same-address LDS
    // atomics per instruction: the partially filled last chunk of every (range, window) group and the all-padding steps behind every piece ran at
    // 350 - 700 ns per chunk against ~90; profiles/r03_notes.md.)  + 4 words: the last arriver's flag
    constexpr uint32_t kCobSpare  = 64;
    constexpr uint32_t kCobLdsExtra = kCobSpare + 4;
    constexpr uint32_t kCobNoSlab = 0xFFFFFFFFu;

    // how a product enters the LDS slice: the LDS atomic of op_add
    enum CobAcc : int { COB_ADD = 0,
                        COB_MIN,
                        COB_MAX,
                        COB_OR,
                        COB_AND,
                        COB_XOR,
                        COB_LOR, // slice holds the truth value 0 / 1 (identity 0)
                        COB_LAND,// slice holds the truth value 0 / 1 (identity 1)
                        COB_FADD,// fp32 sum in a FLOAT slice, added with a compare-and-swap loop: the natural ranges of an fp32 PLUS call when they hold more slots
                                 // than a double slice has room for (never picked by cob_acc_kind: the kernel of COB_ADD on float switches to it per piece)
                        COB_NONE };

    // which LDS atomic serves (op_add, type); COB_NONE: the cob form cannot serve it
    inline int cob_acc_kind(int op_add, int type) {
        switch (op_add) {
            case SPLAHIP_PLUS: return COB_ADD;
            case SPLAHIP_MIN: return type == SPLAHIP_FLOAT ? COB_NONE : COB_MIN;// fp32: ds_min_f32 orders -0 < +0 and NaNs differently from b < a ? b : a
            case SPLAHIP_MAX: return type == SPLAHIP_FLOAT ? COB_NONE : COB_MAX;
            case SPLAHIP_BOR: return type == SPLAHIP_FLOAT ? COB_NONE : COB_OR;
            case SPLAHIP_BAND: return type == SPLAHIP_FLOAT ? COB_NONE : COB_AND;
            case SPLAHIP_BXOR: return type == SPLAHIP_FLOAT ? COB_NONE : COB_XOR;
            case SPLAHIP_LOR: return COB_LOR;
            case SPLAHIP_LAND: return COB_LAND;
            default: return COB_NONE;
        }
    }

    // fp32 MIN / MAX (the reference's own SSSP semiring is fp32 min-plus, src/algorithm.cpp:208-213): the slice holds the ORDER-PRESERVING integer
    // key of a float, b ^ ((b >> 31) & 0x7fffffff) as int32, and products enter it with ds_min_i32 / ds_max_i32.  That is the reference's
    // b < a ? b : a fold (src/op.cpp:222-227) bit for bit provided (1) NaN products are dropped — std::min / std::max never pick a NaN second
    // argument, and a NaN accumulator can only come from init, which is folded in last by the reference's own functor — and (2) no product is
    // -0.0f (key order puts -0 below +0, the reference keeps whichever zero came first): the caller serves op_mult = PLUS only, where a -0
    // product needs a -0 matrix value, and checks the matrix for one (cob.hip: cob_ax_stats).  A slot still at the identity key held no product.
    __host__ __device__ inline int32_t cob_fkey(float f) {
        const uint32_t b = to_bits(f);
        return (int32_t) (b ^ ((uint32_t) ((int32_t) b >> 31) & 0x7FFFFFFFu));
    }
    __host__ __device__ inline float cob_funkey(int32_t k) { return from_bits<float>((uint32_t) k ^ ((uint32_t) (k >> 31) & 0x7FFFFFFFu)); }
    // which LDS atomic serves (op_add, type) in the masked SpMV, where the caller has checked what the keyed float forms need
    inline int cob_acc_kind_mxv(int op_add, int op_mult, int type) {
        if (type == SPLAHIP_FLOAT && op_mult == SPLAHIP_PLUS && (op_add == SPLAHIP_MIN || op_add == SPLAHIP_MAX)) return op_add == SPLAHIP_MIN ? COB_MIN : COB_MAX;
        return cob_acc_kind(op_add, type);
    }

    // bits of the fold identity the slice starts from
    template<typename T, int ACC>
    __host__ __device__ constexpr uint32_t cob_identity() {
        if constexpr (ACC == COB_ADD || ACC == COB_FADD) return std::is_same<T, float>::value ? 0x80000000u : 0u;// -0.0f: x + -0.0f == x for every x, +0.0f and -0.0f included
        else if constexpr (ACC == COB_MIN)
            return std::is_same<T, uint32_t>::value ? 0xFFFFFFFFu : 0x7FFFFFFFu;// float: the largest key (the bits of a NaN: no product has them)
        else if constexpr (ACC == COB_MAX)
            return std::is_same<T, uint32_t>::value ? 0u : 0x80000000u;
        else if constexpr (ACC == COB_AND)
            return 0xFFFFFFFFu;
        else if constexpr (ACC == COB_LAND)
            return 1u;
        else
            return 0u;// OR, XOR, LOR
    }

    template<typename T, int ACC>
    __device__ __forceinline__ void cob_accumulate(uint32_t* slot, T p) {
#if defined(SPLAHIP_COB_EXP) && (SPLAHIP_COB_EXP == 3 || SPLAHIP_COB_EXP == 4)
        *reinterpret_cast<volatile uint32_t*>(slot) = to_bits(p);// DIAGNOSTIC BUILD: a plain LDS store where the atomic would be
        return;
#endif
        if constexpr (ACC == COB_FADD) {// (one word at a time: the bit forms; the value form batches four, cob_accumulate4)
            uint32_t old = *reinterpret_cast<volatile uint32_t*>(slot);
            for (;;) {
                uint32_t seen = old;
                if (__hip_atomic_compare_exchange_strong(slot, &seen, to_bits(from_bits<float>(old) + (float) p), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                old = seen;
            }
        } else if constexpr (ACC == COB_ADD) {
            (void) __hip_atomic_fetch_add(slot, to_bits(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);// int / uint (fp32 sums live in a double slice: cob_accumulate4)
        } else if constexpr (ACC == COB_MIN) {
            if constexpr (std::is_same<T, int32_t>::value) (void) __hip_atomic_fetch_min(reinterpret_cast<int32_t*>(slot), p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (std::is_same<T, float>::value) {
                if (p == p) (void) __hip_atomic_fetch_min(reinterpret_cast<int32_t*>(slot), cob_fkey(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else
                (void) __hip_atomic_fetch_min(slot, to_bits(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if constexpr (ACC == COB_MAX) {
            if constexpr (std::is_same<T, int32_t>::value) (void) __hip_atomic_fetch_max(reinterpret_cast<int32_t*>(slot), p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (std::is_same<T, float>::value) {
                if (p == p) (void) __hip_atomic_fetch_max(reinterpret_cast<int32_t*>(slot), cob_fkey(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else
                (void) __hip_atomic_fetch_max(slot, to_bits(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if constexpr (ACC == COB_OR)
            (void) __hip_atomic_fetch_or(slot, to_bits(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if constexpr (ACC == COB_AND)
            (void) __hip_atomic_fetch_and(slot, to_bits(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if constexpr (ACC == COB_XOR)
            (void) __hip_atomic_fetch_xor(slot, to_bits(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if constexpr (ACC == COB_LOR) {
            if (p != T(0)) (void) __hip_atomic_fetch_or(slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if constexpr (ACC == COB_LAND) {
            if (p == T(0)) (void) __hip_atomic_fetch_and(slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }

    // fp32 PLUS does NOT use ds_add_f32: on gfx950 that instruction (and ds_pk_add_f16) takes ~169 cycles per wave instruction whatever the
    // addresses — 0.38 lane-ops per clock and CU against 6 - 7.5 for ds_add_u32 / ds_min_i32 (tools/exp/lds_atomic_bench2.hip,
    // profiles/r02_lds_atomic_bench.txt), 255 us on R-MAT scale 22 for the accumulation alone — while ds_add_f64 runs at 15 - 25 cycles.  So the
    // fp32 sum keeps its LDS slice in DOUBLE: every product is the reference's fp32 product, widened, added with ds_add_f64, and the row's sum
    // is rounded to fp32 once when the slice is flushed (closer to the exact sum than the reference's sequential fp32 fold; the 1e-5 bar).
    template<typename T, int ACC>
    struct CobSlice {
        using word = uint32_t;
    };
    template<>
    struct CobSlice<float, COB_ADD> {
        using word = double;
    };

    template<typename T, int ACC>
    __device__ __forceinline__ void cob_accumulate4(typename CobSlice<T, ACC>::word* s_acc, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, T p0, T p1, T p2, T p3) {
        if constexpr (ACC == COB_ADD && std::is_same<T, float>::value) {
#if defined(SPLAHIP_COB_EXP) && (SPLAHIP_COB_EXP == 3 || SPLAHIP_COB_EXP == 4)
            *reinterpret_cast<volatile double*>(s_acc + r0) = p0, *reinterpret_cast<volatile double*>(s_acc + r1) = p1;// DIAGNOSTIC BUILD
            *reinterpret_cast<volatile double*>(s_acc + r2) = p2, *reinterpret_cast<volatile double*>(s_acc + r3) = p3;
#else
            (void) __hip_atomic_fetch_add(s_acc + r0, (double) p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void) __hip_atomic_fetch_add(s_acc + r1, (double) p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void) __hip_atomic_fetch_add(s_acc + r2, (double) p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void) __hip_atomic_fetch_add(s_acc + r3, (double) p3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        } else if constexpr (ACC == COB_FADD) {
            // four compare-and-swap adds in flight (one read round trip, one swap round trip); a swap that lost — two lanes of the wave on one row, or another
            // wave in between — is retried on its own.  20 - 26 cycles per wave instruction on uniformly spread rows, which is what a natural range's
            // low-degree rows are (profiles/r02_lds_atomic_bench.txt; under a skewed row stream 110: the hub ranges keep their double slice and ds_add_f64)
            uint32_t* const q[4] = {s_acc + r0, s_acc + r1, s_acc + r2, s_acc + r3};
            const float     x[4] = {(float) p0, (float) p1, (float) p2, (float) p3};
            uint32_t        o[4], seen[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = *reinterpret_cast<volatile uint32_t*>(q[j]);
            bool again = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                seen[j] = o[j];
                if (!__hip_atomic_compare_exchange_strong(q[j], &seen[j], to_bits(from_bits<float>(o[j]) + x[j]), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) again = true;
            }
            if (again) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (seen[j] != o[j]) {
                        uint32_t old = seen[j];
                        for (;;) {
                            uint32_t sn = old;
                            if (__hip_atomic_compare_exchange_strong(q[j], &sn, to_bits(from_bits<float>(old) + x[j]), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                            old = sn;
                        }
                    }
            }
        } else {
            cob_accumulate<T, ACC>(s_acc + r0, p0);
            cob_accumulate<T, ACC>(s_acc + r1, p1);
            cob_accumulate<T, ACC>(s_acc + r2, p2);
            cob_accumulate<T, ACC>(s_acc + r3, p3);
        }
    }

    // a piece of a range that is cut into several pieces adds its slice word into the range's shared accumulator in global memory: the same
    // monoid as a memory-side atomic (agent scope, no return value)
    template<typename T, int ACC>
    __device__ __forceinline__ void cob_merge_global(uint32_t* acc, uint32_t w) {
        if constexpr (ACC == COB_ADD || ACC == COB_FADD) {
            if constexpr (std::is_same<T, float>::value) unsafeAtomicAdd(reinterpret_cast<float*>(acc), from_bits<float>(w));// global_atomic_add_f32
            else
                (void) __hip_atomic_fetch_add(acc, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if constexpr (ACC == COB_MIN) {
            if constexpr (std::is_same<T, uint32_t>::value) (void) __hip_atomic_fetch_min(acc, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                (void) __hip_atomic_fetch_min(reinterpret_cast<int32_t*>(acc), (int32_t) w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);// int32, or a float's key
        } else if constexpr (ACC == COB_MAX) {
            if constexpr (std::is_same<T, uint32_t>::value) (void) __hip_atomic_fetch_max(acc, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                (void) __hip_atomic_fetch_max(reinterpret_cast<int32_t*>(acc), (int32_t) w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if constexpr (ACC == COB_OR || ACC == COB_LOR)
            (void) __hip_atomic_fetch_or(acc, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if constexpr (ACC == COB_AND || ACC == COB_LAND)
            (void) __hip_atomic_fetch_and(acc, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            (void) __hip_atomic_fetch_xor(acc, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // op_add of an accumulation kind (cob_acc_kind maps one to one): the row pass folds init with a compile-time functor — a runtime-switched one costs a
    // 15-way switch per ROW, and a natural range's row pass is instruction-bound on ONE CU (160 K rows at ~20 instructions each ran 28 - 34 us)
    template<int ACC>
    constexpr int cob_op_of() {
        return (ACC == COB_ADD || ACC == COB_FADD) ? SPLAHIP_PLUS : ACC == COB_MIN ? SPLAHIP_MIN : ACC == COB_MAX ? SPLAHIP_MAX : ACC == COB_OR ? SPLAHIP_BOR : ACC == COB_AND ? SPLAHIP_BAND
               : ACC == COB_XOR ? SPLAHIP_BXOR : ACC == COB_LOR ? SPLAHIP_LOR : SPLAHIP_LAND;
    }
    // the row's result from its accumulated slice word: add(init, acc); LOR / LAND slices hold a truth value, fp32 MIN / MAX slices a key
    template<typename T, int ACC, class Add>
    __device__ __forceinline__ T cob_finish(uint32_t acc, T init, const Add&) {
        constexpr int OP = cob_op_of<ACC>();
        if constexpr (ACC == COB_LOR || ACC == COB_LAND) return eval_bin<OP>(init, acc ? T(1) : T(0));
        else if constexpr (std::is_same<T, float>::value && (ACC == COB_MIN || ACC == COB_MAX))
            return acc == cob_identity<T, ACC>() ? init : eval_bin<OP>(init, cob_funkey((int32_t) acc));
        else
            return eval_bin<OP>(init, from_bits<T>(acc));
    }
    // the fold itself as a value of T (vxm mode: an output is the fold, there is no init)
    template<typename T, int ACC>
    __device__ __forceinline__ T cob_value(uint32_t acc) {
        if constexpr (std::is_same<T, float>::value && (ACC == COB_MIN || ACC == COB_MAX)) return cob_funkey((int32_t) acc);
        else
            return from_bits<T>(acc);
    }

    // The packed streams of a chunk -> the four words slot << 16 | column - hdr[chunk] of a lane's slots (the form the loop below works on).
    // A chunk stores 2 + 1 + 4 bytes per slot: a 16-bit slot in range, an 8-bit column DELTA to the slot before it in sorted order (the chunk's first slot: 0;
    // a gap of more than 255 columns is bridged by padding slots when the copy is built) and the value word.  The lane's slots are the sorted slots l, 64 + l,
    // 128 + l, 192 + l: their columns are the chunk's running sums — two wave-wide inclusive scans over pairs of 16-bit partial sums (a group of 64 deltas
    // sums to < 2^14, the whole chunk to < 2^16) by DPP row shifts and broadcasts, plus the three group totals from lane 63.
    __device__ __forceinline__ uint32_t cob_scan16x2(uint32_t x) {
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x111, 0xF, 0xF, false);// row_shr:1
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x112, 0xF, 0xF, false);// row_shr:2
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x114, 0xF, 0xF, false);// row_shr:4
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x118, 0xF, 0xF, false);// row_shr:8
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x142, 0xA, 0xF, false);// row_bcast:15 into rows 1 and 3
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x143, 0xC, 0xF, false);// row_bcast:31 into rows 2 and 3
        return x;
    }
    typedef uint32_t cob_u4 __attribute__((ext_vector_type(4)));
    typedef uint32_t cob_u2 __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ cob_u4 cob_decode(cob_u2 s, uint32_t d) {
        const uint32_t a = cob_scan16x2((d & 0xFFu) | ((d & 0xFF00u) << 8));          // running sums of the deltas of groups 0 | 1
        const uint32_t b = cob_scan16x2(((d >> 16) & 0xFFu) | ((d >> 24) << 16));      //                                   2 | 3
        const uint32_t ta = (uint32_t) __builtin_amdgcn_readlane((int) a, 63), tb = (uint32_t) __builtin_amdgcn_readlane((int) b, 63);
        const uint32_t t0 = ta & 0xFFFFu, t01 = t0 + (ta >> 16), t012 = t01 + (tb & 0xFFFFu);
        cob_u4         e;
        e.x = (s.x << 16) | (a & 0xFFFFu);
        e.y = (s.x & 0xFFFF0000u) | (t0 + (a >> 16));
        e.z = (s.y << 16) | (t01 + (b & 0xFFFFu));
        e.w = (s.y & 0xFFFF0000u) | (t012 + (b >> 16));
        return e;
    }

    struct CobArgs {
        const CobTile*  tiles;
        const CobRange* ranges;
        const uint32_t* hdr;// per chunk: first column of its window
        const uint16_t* sl;// per slot: slot in range, 16 bits
        const uint8_t*  dl;// per slot: column delta to the slot before it in sorted order, 8 bits
        const uint32_t* ax;
        const uint16_t* slot16; // per row: its slot in its natural range, kCobSlotEmpty, or kCobSlotHub (padded to a multiple of 8 rows)
        const uint32_t* hub_row;// per hub slot: its row, 0xFFFFFFFF for the padding behind the last hub row
        uint32_t*       acc;    // shared accumulators of the ranges cut into several pieces (n_multi x R words, at the identity between calls)
        uint32_t*       ticket; // n_multi arrival counters, 0 between calls
        uint32_t*       fin;    // per hub range: finishers done (in-kernel finishers only), 0 between calls
        uint32_t*       touched;// VXM only: per output index, 1 = selected and at least one product
        uint32_t        n_rows;
        uint32_t        R;       // slots of a hub range (a multiple of 64)
        uint32_t        Rn;      // slots of a natural range (ranges >= n_hub_ranges); Rn > R: the natural ranges of an fp32 sum keep FLOAT words (COB_FADD)
        uint32_t        n_hub_ranges;
        uint32_t        acc_stride;// words between two shared accumulators: max(R, Rn)
        uint32_t        col_bits;
        uint32_t        pad_chunk;// two all-padding chunks behind the last real one: pad_chunk for a hub piece (slot values R + lane), pad_chunk + 1 for a natural one (Rn + lane)
        const uint32_t* vbits;     // MODE 1 / 2: bit c = v[c] != 0 (n_cols bits)
        uint32_t        x_uniform; // MODE 2: the one value word of the matrix
        uint32_t        accumulate;// 1: r already holds the fold of another column part of the same rows (1-D row decomposition, dist.hip): a selected
                                   // row that holds an entry here becomes add(r[row], acc); every other row is left as it is
    };

    // One workgroup per tile.  LDS: the range's R result words.  VXM: products are mult(v, x) (the pull form of vxm over the copy of the
    // transposed matrix, reference cpu_vxm.hpp:113) — everything else is the same fold.
    // MODE 0: every entry gathers its value of v.  MODE 1 / 2 — the sparse-input-vector form for semirings in which a zero of v cannot change a row's fold
    // (0 annihilates op_mult and is op_add's identity: BAND / BOR, LAND / LOR, MULT / PLUS ... on int / uint; sell.hip: sell_skip_family == 1): an entry first
    // gathers one BIT of a bitmap of v (a.vbits, written per call by cob_vbits_kernel: 128 bytes cover 1024 columns, so a wave's gather is one or two
    // requests), and only the entries whose bit is set gather the value and touch the LDS slice — a BFS step adds a few per cent of its products.
    // MODE 2: every stored value of the matrix is the same word (a.x_uniform: an adjacency matrix of ones): the value stream is not read at all.
    // waves per SIMD the register budget is cut for: the bit forms run two 16-wave workgroups per CU on 4-byte slices (8); the value form keeps the
    // compiler's own choice (a forced 8 spilled 20 words in the int instantiations)
    template<typename T, int ACC, int MODE>
    constexpr unsigned cob_min_waves() {
#ifdef SPLAHIP_COB_MIN_WAVES
        return SPLAHIP_COB_MIN_WAVES;
#else
        // (the bit forms were cut for 64 registers — two workgroups per CU — until round 5; with one workgroup per CU and the packed streams' decode they spill
        // 27 words at that cut and the min-plus calls ran 14 - 28 % slower: every form now gets the budget of 4 waves per SIMD)
        return 4u;
#endif
    }

    // The row pass of a range: results leave the LDS slice (4-byte words: a double slice has been rounded in place, a merged range's words read back).
    // One copy per (type, op_add family, mode) however many op_mult / input-vector forms of the kernel exist — a real call, not inlined.
    // TIGHT: the caller is one of the kernels cut for 64 registers (two 16-wave workgroups per CU): smaller batches, so that the pass fits them too.
    // REARM: the words are the range's shared accumulator in GLOBAL memory (the run-time schedule's finishing kernel, mxv_cob_dyn.hpp): a word that is read
    // is set back to the fold identity by its reader — every slot of a range belongs to exactly one row, so the accumulator rests at the identity afterwards
    template<typename T, int ACC, bool VXM, unsigned TPB, bool TIGHT, bool REARM = false>
    __device__ __noinline__ void cob_row_pass(const uint32_t* s_words, const CobRange rg, const uint16_t* __restrict__ slot16, const uint32_t* __restrict__ hub_row,
                                              uint32_t* __restrict__ touched, uint32_t n_rows, uint32_t R, uint32_t accumulate, const T* __restrict__ mask, T* __restrict__ r,
                                              DynBin add, int op_select, T init) {
        auto s_words_at = [&](uint32_t i) -> uint32_t {
            const uint32_t w = s_words[i];
            if constexpr (REARM) const_cast<uint32_t*>(s_words)[i] = cob_identity<T, ACC>();
            return w;
        };
        if (rg.hub_base != 0xFFFFFFFFu) {
            // hub range: slot s is row hub_row[hub_base + s]; one scattered word per row (rows, then masks, requested in batches)
            constexpr int KH = TIGHT ? 4 : 8;
            for (uint32_t base = threadIdx.x; base < R; base += TPB * KH) {
                uint32_t rows[KH];
#pragma unroll
                for (int k = 0; k < KH; ++k) {
                    const uint32_t sidx = base + k * TPB;
                    rows[k]             = sidx < R ? hub_row[rg.hub_base + sidx] : 0xFFFFFFFFu;
                }
                T    mk[KH];
                T    prev[KH];
#pragma unroll
                for (int k = 0; k < KH; ++k) {
                    mk[k]   = rows[k] != 0xFFFFFFFFu && op_select != SPLAHIP_ALWAYS ? mask[rows[k]] : T(0);
                    prev[k] = rows[k] != 0xFFFFFFFFu && !VXM && accumulate ? r[rows[k]] : init;
                }
#pragma unroll
                for (int k = 0; k < KH; ++k) {
                    const uint32_t row = rows[k];
                    if (row == 0xFFFFFFFFu) continue;
                    const uint32_t w   = s_words_at(base + k * TPB);
                    const bool     sel = op_select == SPLAHIP_ALWAYS || eval_sel<T>(op_select, mk[k]);
                    if constexpr (VXM) {
                        r[row]         = cob_value<T, ACC>(w);
                        touched[row] = sel ? 1u : 0u;
                    } else if (accumulate) {
                        if (sel) r[row] = cob_finish<T, ACC>(w, prev[k], add);
                    } else
                        r[row] = sel ? cob_finish<T, ACC>(w, init, add) : init;
                }
            }
        } else {
            // natural range: its rows in order, 8 per thread and step (one 16-byte load of their slots; the mask / the previous r as 16-byte loads too);
            // KN steps' loads are requested together — a row pass is a chain of memory round trips, one per batch — and rows are stored 16 bytes at a
            // time where all 4 of a group are written
            auto pass = [&](auto kn_tag, auto side_tag) {
                constexpr int  KN   = decltype(kn_tag)::value;
                constexpr bool SIDE = decltype(side_tag)::value;// the call reads mask (op_select != ALWAYS) or r (accumulate)
                const bool     need_mask = SIDE && op_select != SPLAHIP_ALWAYS, need_prev = SIDE && !VXM && accumulate;
                for (uint32_t row0 = rg.row0 + threadIdx.x * 8; row0 < rg.row1; row0 += TPB * 8 * KN) {
                    uint4 sl[KN], mk[SIDE ? KN : 1][2], pv[SIDE ? KN : 1][2];
#pragma unroll
                    for (int k = 0; k < KN; ++k) {
                        const uint32_t row = row0 + k * TPB * 8;
                        sl[k]              = row < rg.row1 ? *reinterpret_cast<const uint4*>(slot16 + row) : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                        if constexpr (SIDE) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const uint32_t rb = row + 4 * h;
                                const bool     in = row < rg.row1 && rb + 4 <= n_rows;// (the last group of the matrix may be cut short: its words are read one by one)
                                mk[k][h]          = need_mask && in ? *reinterpret_cast<const uint4*>(mask + rb) : make_uint4(0, 0, 0, 0);
                                pv[k][h]          = need_prev && in ? *reinterpret_cast<const uint4*>(r + rb) : make_uint4(0, 0, 0, 0);
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KN; ++k) {
                        const uint32_t row = row0 + k * TPB * 8;
                        if (row >= rg.row1) break;
                        if constexpr (!SIDE && !VXM) {// eight rows without an entry (most of a cold range's span): two 16-byte stores of init and on
                            if ((sl[k].x & sl[k].y & sl[k].z & sl[k].w) == 0xFFFFFFFFu && row + 8 <= n_rows) {
                                const uint4 iv = make_uint4(to_bits(init), to_bits(init), to_bits(init), to_bits(init));
                                *reinterpret_cast<uint4*>(r + row)     = iv;
                                *reinterpret_cast<uint4*>(r + row + 4) = iv;
                                continue;
                            }
                        }
                        const uint32_t sw[4] = {sl[k].x, sl[k].y, sl[k].z, sl[k].w};
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const uint32_t rb = row + 4 * h;
                            if (rb >= n_rows) break;
                            const bool     whole = rb + 4 <= n_rows;
                            const uint32_t mw[4] = {mk[SIDE ? k : 0][h].x, mk[SIDE ? k : 0][h].y, mk[SIDE ? k : 0][h].z, mk[SIDE ? k : 0][h].w};
                            const uint32_t pw[4] = {pv[SIDE ? k : 0][h].x, pv[SIDE ? k : 0][h].y, pv[SIDE ? k : 0][h].z, pv[SIDE ? k : 0][h].w};
                            T              out[4];
                            uint32_t       tch[4] = {0, 0, 0, 0};
                            bool           wr[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t rj = rb + j;
                                const uint32_t sj = (sw[2 * h + (j >> 1)] >> (16 * (j & 1))) & 0xFFFFu;
                                wr[j]             = false;
                                if (rj >= n_rows || sj == kCobSlotHub) continue;
                                const bool     has = sj != kCobSlotEmpty;
                                const uint32_t w   = has ? s_words_at(sj) : cob_identity<T, ACC>();
                                bool           sel = true;
                                if (need_mask) sel = eval_sel<T>(op_select, whole ? from_bits<T>(mw[j]) : mask[rj]);
                                if constexpr (VXM) {// no init: an output exists where the index is selected and holds a product; its value is the fold itself
                                    tch[j] = sel && has;
                                    out[j] = cob_value<T, ACC>(w);
                                    wr[j]  = true;
                                } else if (need_prev) {
                                    if (sel && has) {
                                        out[j] = cob_finish<T, ACC>(w, whole ? from_bits<T>(pw[j]) : r[rj], add);
                                        wr[j]  = true;
                                    }
                                } else {
                                    // (LOR / LAND: a row without an entry keeps init itself, not its truth value — `has` covers it)
                                    out[j] = sel && has ? cob_finish<T, ACC>(w, init, add) : init;
                                    wr[j]  = true;
                                }
                            }
                            if (wr[0] && wr[1] && wr[2] && wr[3]) {
                                *reinterpret_cast<uint4*>(r + rb) = make_uint4(to_bits(out[0]), to_bits(out[1]), to_bits(out[2]), to_bits(out[3]));
                                if constexpr (VXM) *reinterpret_cast<uint4*>(touched + rb) = make_uint4(tch[0], tch[1], tch[2], tch[3]);
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (wr[j]) {
                                        r[rb + j] = out[j];
                                        if constexpr (VXM) touched[rb + j] = tch[j];
                                    }
                            }
                        }
                    }
                }
            };
            if (op_select == SPLAHIP_ALWAYS && (VXM || !accumulate)) pass(std::integral_constant<int, TIGHT ? 2 : 8>{}, std::false_type{});
            else
                pass(std::integral_constant<int, TIGHT ? 1 : 2>{}, std::true_type{});
        }
    }

    // What a piece does behind its chunk loop, as ONE real call per (type, op_add family, mode): merge into the range's shared accumulator and take a
    // ticket (a range in several pieces), read the accumulator back as the last arriver, round a double slice, then the row pass above.  A hub range's
    // slot -> row words are requested BEFORE the merge (they depend on nothing), so the scattered stores behind it wait for no load of their own.
    template<typename T, int ACC, bool VXM, unsigned TPB, bool TIGHT>
    __device__ __noinline__ void cob_piece_end(unsigned char* s_raw, const CobTile t, const CobRange rg, const uint16_t* __restrict__ slot16, const uint32_t* __restrict__ hub_row,
                                               uint32_t* __restrict__ acc, uint32_t* __restrict__ ticket, uint32_t* __restrict__ touched, uint32_t n_rows, uint32_t R,
                                               uint32_t acc_stride, uint32_t accumulate, const T* __restrict__ mask, T* __restrict__ r, DynBin add, int op_select, T init) {
        using word = typename CobSlice<T, ACC>::word;
        constexpr bool kWide = sizeof(word) == 8;
        word* const    s_acc = reinterpret_cast<word*>(s_raw);
        // a hub range's slot -> row words: requested by the piece that will run the row pass as soon as it knows it (under the accumulator read-back), so the
        // scattered stores wait for no load of their own (not in the 64-register kernels: their row pass loads them in batches)
        constexpr uint32_t KP = TIGHT ? 1 : 20;// 20 x TPB >= R (20224 with 1024 threads, <= 10236 with 512)
        uint32_t           hrow[KP];
        const bool         hub_early = !TIGHT && rg.hub_base != 0xFFFFFFFFu;
        auto               request_hub_rows = [&]() {
            if (hub_early) {
#pragma unroll
                for (uint32_t k = 0; k < KP; ++k) {
                    const uint32_t sidx = threadIdx.x + k * TPB;
                    hrow[k]             = sidx < R ? hub_row[rg.hub_base + sidx] : 0xFFFFFFFFu;
                }
            }
        };
        if (t.multi == kCobNoSlab) request_hub_rows();
        // slice word i as the 4-byte result word (a double slice is rounded to fp32 here, once per row and piece)
        auto own_word = [&](uint32_t i) -> uint32_t {
            if constexpr (kWide) return to_bits((float) s_acc[i]);
            else
                return s_acc[i];
        };
        const uint32_t* s_words = reinterpret_cast<const uint32_t*>(s_raw);// the last arriver's view: the accumulated 4-byte words
        bool            merged  = false;
        if (t.multi != kCobNoSlab) {
            // add the slice into the range's shared accumulator (words still at the identity hold nothing), then take a ticket
            uint32_t* const gacc = acc + (size_t) t.multi * acc_stride;
            for (uint32_t i = threadIdx.x; i < R; i += TPB) {
                const uint32_t w = own_word(i);
                if (w != cob_identity<T, ACC>()) cob_merge_global<T, ACC>(gacc + i, w);
            }
            if (rg.n_pieces == kCobNoLast) return;// (the end of the launch makes the atomics visible to the finishing kernel)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");// every adding wave: its atomics have been performed
            __syncthreads();
            uint32_t* const s_flag = reinterpret_cast<uint32_t*>(s_raw + (size_t) (R + kCobSpare) * sizeof(word));
            if (threadIdx.x == 0) *s_flag = __hip_atomic_fetch_add(ticket + t.multi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == rg.n_pieces - 1 ? 1u : 0u;
            __syncthreads();
            if (*s_flag == 0) return;
            request_hub_rows();
            // the last piece of the range: every other piece's atomics were performed before its ticket add.  Read the accumulator back (agent-scope
            // loads: past this CU's L1), re-arm it and the ticket for the next call, and finish the rows from the LDS copy
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
            // 16 bytes per lane with the sc1 bit (agent-scope, like the atomic loads the compiler would emit one word at a time): R / 4 / TPB <= 5 loads per
            // thread, all requested before the one wait — one memory round trip for the whole accumulator.  The re-arming stores are plain 16-byte
            // stores: nobody reads these words again in this launch, and a launch boundary makes them visible to the next call's atomics.
            uint4* s_w4 = reinterpret_cast<uint4*>(s_raw);
            uint4* g4   = reinterpret_cast<uint4*>(gacc);
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            constexpr int KB = 5;
            for (uint32_t base = threadIdx.x; base < R / 4; base += TPB * KB) {
                u4v w[KB];
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    const uint32_t i = base + k * TPB;
                    const uint4*   q = g4 + (i < R / 4 ? i : 0);
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[k]) : "v"(q) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4])::"memory");
                const uint32_t id = cob_identity<T, ACC>();
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    const uint32_t i = base + k * TPB;
                    if (i < R / 4) {
                        s_w4[i] = make_uint4(w[k].x, w[k].y, w[k].z, w[k].w);
                        g4[i]   = make_uint4(id, id, id, id);
                    }
                }
            }
            if (threadIdx.x == 0) __hip_atomic_store(ticket + t.multi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            merged = true;
        }
        if constexpr (kWide) {
            if (!merged) {// round the double slice to fp32 words in place: every thread's doubles into registers, a barrier, then the words
                constexpr uint32_t KC = TPB >= 1024 ? 20 : 24;// (R <= 20224 with 1024 threads; R <= 10112 + with 512: 24 x 512 covers 12288)
                float              f[KC];
#pragma unroll
                for (uint32_t k = 0; k < KC; ++k) {
                    const uint32_t i = threadIdx.x + k * TPB;
                    f[k]             = i < R ? (float) s_acc[i] : 0.0f;
                }
                __syncthreads();
                float* s_f = reinterpret_cast<float*>(s_raw);
#pragma unroll
                for (uint32_t k = 0; k < KC; ++k) {
                    const uint32_t i = threadIdx.x + k * TPB;
                    if (i < R) s_f[i] = f[k];
                }
                __syncthreads();
            }
        }
        if (hub_early) {// slot s of the hub range is row hrow[.]: one scattered word per row, no load in front of the stores (mask / previous r: one batch)
            T mk[KP], prev[KP];
#pragma unroll
            for (uint32_t k = 0; k < KP; ++k) {
                mk[k]   = hrow[k] != 0xFFFFFFFFu && op_select != SPLAHIP_ALWAYS ? mask[hrow[k]] : T(0);
                prev[k] = hrow[k] != 0xFFFFFFFFu && !VXM && accumulate ? r[hrow[k]] : init;
            }
#pragma unroll
            for (uint32_t k = 0; k < KP; ++k) {
                const uint32_t row = hrow[k];
                if (row == 0xFFFFFFFFu) continue;
                const uint32_t w   = s_words[threadIdx.x + k * TPB];
                const bool     sel = op_select == SPLAHIP_ALWAYS || eval_sel<T>(op_select, mk[k]);
                if constexpr (VXM) {
                    r[row]       = cob_value<T, ACC>(w);
                    touched[row] = sel ? 1u : 0u;
                } else if (accumulate) {
                    if (sel) r[row] = cob_finish<T, ACC>(w, prev[k], add);
                } else
                    r[row] = sel ? cob_finish<T, ACC>(w, init, add) : init;
            }
            return;
        }
        cob_row_pass<T, ACC, VXM, TPB, TIGHT>(s_words, rg, slot16, hub_row, touched, n_rows, R, accumulate, mask, r, add, op_select, init);
    }

    // one row of a hub range from its merged accumulator word (cob_hub_finish_kernel)
    template<typename T, int ACC, bool VXM>
    __device__ __forceinline__ void cob_hub_row(uint32_t w, uint32_t row, const T* __restrict__ mask, T* __restrict__ r, uint32_t* __restrict__ touched, uint32_t accumulate,
                                                DynBin add, int op_select, T init) {
        const bool sel = op_select == SPLAHIP_ALWAYS || eval_sel<T>(op_select, mask[row]);
        if constexpr (VXM) {
            r[row]       = cob_value<T, ACC>(w);
            touched[row] = sel ? 1u : 0u;
        } else if (accumulate) {
            if (sel) r[row] = cob_finish<T, ACC>(w, r[row], add);
        } else
            r[row] = sel ? cob_finish<T, ACC>(w, init, add) : init;
    }

    template<typename T, class Mult, class Add, int ACC, bool VXM, unsigned TPB, int MODE = 0>
    __global__ __launch_bounds__(TPB, (cob_min_waves<T, ACC, MODE>())) void cob_mxv_kernel(CobArgs a, const T* __restrict__ v, const T* __restrict__ mask, T* __restrict__ r, Mult mult,
                                                          Add add, int op_select, T init) {
        using word = typename CobSlice<T, ACC>::word;
        constexpr bool kWide = sizeof(word) == 8;// fp32 PLUS: the slice holds doubles
        extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
        word* const        s_acc = reinterpret_cast<word*>(s_raw);
        constexpr unsigned NW    = TPB / kWave;
        const CobTile      t     = a.tiles[blockIdx.x];// one piece per workgroup, launched largest piece first
        const bool         nat   = t.range >= a.n_hub_ranges;
        const uint32_t     R     = nat ? a.Rn : a.R;
        // fp32 PLUS over a natural range that holds more slots than a double slice has room for: float words, compare-and-swap adds (COB_FADD)
        [[maybe_unused]] const bool narrow = kWide && nat && a.Rn > a.R;
        uint32_t* const    s_acc32 = reinterpret_cast<uint32_t*>(s_raw);
#ifdef SPLAHIP_COB_TIMING
        if (threadIdx.x == 0 && blockIdx.x < 8192) {
            uint32_t xcc, hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            g_cob_t[0][blockIdx.x] = __builtin_amdgcn_s_memrealtime();
            g_cob_t[4][blockIdx.x] = xcc;
            g_cob_t[5][blockIdx.x] = hwid;
            g_cob_t[6][blockIdx.x] = t.n_chunks;
            g_cob_t[7][blockIdx.x] = ((unsigned long long) t.range << 32) | t.multi;
        }
#endif
        {
            const uint32_t id = cob_identity<T, ACC>();
            uint4*         s4 = reinterpret_cast<uint4*>(s_raw);
            const uint4    z  = kWide && !narrow ? make_uint4(0u, id, 0u, id) : make_uint4(id, id, id, id);// a double's high word carries the sign of -0.0
            const uint32_t n16 = (R + kCobSpare) * (narrow ? 4u : (uint32_t) sizeof(word)) / 16;
            for (uint32_t i = threadIdx.x; i < n16; i += TPB) s4[i] = z;// + the spare words the padding slots add into
        }
        __syncthreads();
#ifdef SPLAHIP_COB_TIMING
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_cob_t[1][blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
        const unsigned lane  = lane_id();
        const uint32_t cmask = (1u << a.col_bits) - 1u;
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const cob_u2*   sl2 = reinterpret_cast<const cob_u2*>(a.sl) + lane;// a lane's four 16-bit slots of a chunk
        const uint32_t* dl1 = reinterpret_cast<const uint32_t*>(a.dl) + lane;// its four 8-bit column deltas
        const u4*       ax4 = reinterpret_cast<const u4*>(a.ax) + lane;
        const uint32_t* hb  = a.hdr;
        // Software pipeline, four register sets deep, no data-dependent exit: every wave runs the same number of steps (a multiple of 4);
        // a step beyond the tile's chunks walks the all-padding chunk behind the last real one (a.pad_chunk).  In step s a wave
        //     requests the gathers of step s + 1 (its streams were requested three steps ago),
        //     requests the streams of step s + 4 into the set step s just freed ... see below: of step s + 3,
        //     adds the products of step s (its gathers were requested one step ago) into the LDS slice,
        // so the LDS round trips of a step run under the memory latency of the next one, and ~6 KB of streams per wave are in flight.
        // The wave's chunk index stays in a vector register on purpose: the chunk's base column then arrives through the vector-memory
        // queue like the two streams (a scalar load would have to be waited for with lgkmcnt(0) right where it is issued).
        // A padding slot holds column offset 0 and the row field R: it gathers v[base], a valid address, and its product lands in the
        // spare word behind the slice.
        const uint32_t steps    = ((t.n_chunks + NW - 1) / NW + 3) & ~3u;
        const uint32_t c0       = threadIdx.x / kWave;
        const uint32_t pad_chunk = a.pad_chunk + (nat ? 1u : 0u);
        auto           chunk_of = [&](uint32_t ci) { return ci < t.n_chunks ? t.first_chunk + ci : pad_chunk; };
        u4             E[4], X[4];
        cob_u2         S[4];   // the packed slot words and column deltas of a step, as loaded: decoded into E by the step's first user (cob_decode)
        uint32_t       D[4];
        uint32_t       B[4];
        T              G[2][4];
        u4             W[2];   // MODE 1 / 2: the gathered bitmap words of a step
        uint32_t       M[2];   //             and, once they are in, the step's four bits
        auto load_set = [&](int q, uint32_t ci) {
            const uint32_t k = chunk_of(ci);
            S[q]             = __builtin_nontemporal_load(sl2 + (size_t) k * (kCobChunk / 4));
            D[q]             = __builtin_nontemporal_load(dl1 + (size_t) k * (kCobChunk / 4));
            if constexpr (MODE != 2) X[q] = __builtin_nontemporal_load(ax4 + (size_t) k * (kCobChunk / 4));
            B[q] = hb[k];
        };
        // a step's packed words -> E; a padding / bridging slot (slot field >= R) gets column offset 0: it reads the chunk's base column instead of a line of its own — the bridging
        // slots of the column gaps sit at columns nothing else refers to: 1.9 M extra lines of v per call at R-MAT 23 when they gathered there
        auto decode_set = [&](int q) {
            cob_u4 e = cob_decode(S[q], D[q]);
            e.x = (e.x >> 16) >= R ? e.x & 0xFFFF0000u : e.x, e.y = (e.y >> 16) >= R ? e.y & 0xFFFF0000u : e.y;
            e.z = (e.z >> 16) >= R ? e.z & 0xFFFF0000u : e.z, e.w = (e.w >> 16) >= R ? e.w & 0xFFFF0000u : e.w;
            E[q] = e;
        };
        auto gather_set = [&](int q, int gq) {
            if constexpr (MODE == 0) decode_set(q);
#if defined(SPLAHIP_COB_EXP) && (SPLAHIP_COB_EXP == 2 || SPLAHIP_COB_EXP == 4)
            G[gq][0] = from_bits<T>(B[q] + (E[q].x & cmask)), G[gq][1] = from_bits<T>(B[q] + (E[q].y & cmask)), G[gq][2] = from_bits<T>(B[q] + (E[q].z & cmask)),
            G[gq][3] = from_bits<T>(B[q] + (E[q].w & cmask));// DIAGNOSTIC BUILD: no gathers
#else
            if constexpr (MODE == 0) {
#if defined(SPLAHIP_COB_EXP) && SPLAHIP_COB_EXP >= 16// DIAGNOSTIC BUILD: every gather folded into the first 2^EXP words of v (same lines per instruction, all cache-resident)
                constexpr uint32_t wm = (1u << SPLAHIP_COB_EXP) - 1u;
                G[gq][0] = v[(B[q] + (E[q].x & cmask)) & wm];
                G[gq][1] = v[(B[q] + (E[q].y & cmask)) & wm];
                G[gq][2] = v[(B[q] + (E[q].z & cmask)) & wm];
                G[gq][3] = v[(B[q] + (E[q].w & cmask)) & wm];
#else
                G[gq][0] = v[B[q] + (E[q].x & cmask)];
                G[gq][1] = v[B[q] + (E[q].y & cmask)];
                G[gq][2] = v[B[q] + (E[q].z & cmask)];
                G[gq][3] = v[B[q] + (E[q].w & cmask)];
#endif
            } else {// the bits of this step arrived one step ago: extract them, then gather v only where a bit is set (every other lane re-reads v[base]: one line)
                const uint32_t c0 = B[q] + (E[q].x & cmask), c1 = B[q] + (E[q].y & cmask), c2 = B[q] + (E[q].z & cmask), c3 = B[q] + (E[q].w & cmask);
                const uint32_t m  = ((W[gq].x >> (c0 & 31)) & 1u) | (((W[gq].y >> (c1 & 31)) & 1u) << 1) | (((W[gq].z >> (c2 & 31)) & 1u) << 2) | (((W[gq].w >> (c3 & 31)) & 1u) << 3);
                M[gq]    = m;
                G[gq][0] = v[(m & 1u) ? c0 : B[q]];
                G[gq][1] = v[(m & 2u) ? c1 : B[q]];
                G[gq][2] = v[(m & 4u) ? c2 : B[q]];
                G[gq][3] = v[(m & 8u) ? c3 : B[q]];
            }
#endif
        };
        auto bits_set = [&](int q, int wq) {// MODE 1 / 2: one bitmap word per slot (a padding slot reads the word of the base column)
            decode_set(q);
            W[wq].x = a.vbits[(B[q] + (E[q].x & cmask)) >> 5];
            W[wq].y = a.vbits[(B[q] + (E[q].y & cmask)) >> 5];
            W[wq].z = a.vbits[(B[q] + (E[q].z & cmask)) >> 5];
            W[wq].w = a.vbits[(B[q] + (E[q].w & cmask)) >> 5];
        };
#if defined(SPLAHIP_COB_EXP) && SPLAHIP_COB_EXP == 1
        T dbg = T(0);
#endif
        auto add_set = [&](int q, int gq) {
            const u4 e = E[q];
            u4       x;
            if constexpr (MODE == 2) x = {a.x_uniform, a.x_uniform, a.x_uniform, a.x_uniform};
            else
                x = X[q];
            const T g0 = G[gq][0], g1 = G[gq][1], g2 = G[gq][2], g3 = G[gq][3];
#if defined(SPLAHIP_COB_EXP) && SPLAHIP_COB_EXP == 1
            dbg = dbg + mult(from_bits<T>(x.x), g0) + mult(from_bits<T>(x.y), g1) + mult(from_bits<T>(x.z), g2) + mult(from_bits<T>(x.w), g3) + T(e.x >> a.col_bits);// DIAGNOSTIC BUILD: no LDS accumulation
            return;
#endif
            if constexpr (MODE != 0) {// only the products of the set bits can change a row (a padding slot's bit may be set: its row field is the spare word)
                const uint32_t m = M[gq];
                if (m & 1u) cob_accumulate<T, ACC>(s_acc + (e.x >> a.col_bits), mult(from_bits<T>(x.x), g0));
                if (m & 2u) cob_accumulate<T, ACC>(s_acc + (e.y >> a.col_bits), mult(from_bits<T>(x.y), g1));
                if (m & 4u) cob_accumulate<T, ACC>(s_acc + (e.z >> a.col_bits), mult(from_bits<T>(x.z), g2));
                if (m & 8u) cob_accumulate<T, ACC>(s_acc + (e.w >> a.col_bits), mult(from_bits<T>(x.w), g3));
            } else if constexpr (VXM) {
                if constexpr (kWide) {
                    if (narrow) {
                        cob_accumulate4<T, COB_FADD>(s_acc32, e.x >> a.col_bits, e.y >> a.col_bits, e.z >> a.col_bits, e.w >> a.col_bits, mult(g0, from_bits<T>(x.x)),
                                                     mult(g1, from_bits<T>(x.y)), mult(g2, from_bits<T>(x.z)), mult(g3, from_bits<T>(x.w)));
                        return;
                    }
                }
                cob_accumulate4<T, ACC>(s_acc, e.x >> a.col_bits, e.y >> a.col_bits, e.z >> a.col_bits, e.w >> a.col_bits, mult(g0, from_bits<T>(x.x)),
                                        mult(g1, from_bits<T>(x.y)), mult(g2, from_bits<T>(x.z)), mult(g3, from_bits<T>(x.w)));
            }
            else {
                if constexpr (kWide) {
                    if (narrow) {
                        cob_accumulate4<T, COB_FADD>(s_acc32, e.x >> a.col_bits, e.y >> a.col_bits, e.z >> a.col_bits, e.w >> a.col_bits, mult(from_bits<T>(x.x), g0),
                                                     mult(from_bits<T>(x.y), g1), mult(from_bits<T>(x.z), g2), mult(from_bits<T>(x.w), g3));
                        return;
                    }
                }
                cob_accumulate4<T, ACC>(s_acc, e.x >> a.col_bits, e.y >> a.col_bits, e.z >> a.col_bits, e.w >> a.col_bits, mult(from_bits<T>(x.x), g0),
                                        mult(from_bits<T>(x.y), g1), mult(from_bits<T>(x.z), g2), mult(from_bits<T>(x.w), g3));
            }
        };
        // prologue: streams of steps 0, 1, 2; (bits of steps 0 and 1;) gathers of step 0
        load_set(0, c0);
        load_set(1, c0 + NW);
        load_set(2, c0 + 2 * NW);
        if constexpr (MODE != 0) {
            bits_set(0, 0);
            bits_set(1, 1);
        }
        gather_set(0, 0);
        for (uint32_t s = 0; s < steps; s += 4) {
            const uint32_t cs = c0 + s * NW;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // step s + k works on set k (k & 3): streams of step s + k + 3 go into set (k + 3) & 3, which step s + k - 1 has just finished with;
                // the value gathers run one step ahead (MODE 1 / 2: the bitmap words two steps ahead, in the word set the gathers of step s + k + 1 — issued
                // just before — have consumed... the other one: W[(k + 2) & 1] was read by gather_set of step s + k, one body earlier)
                load_set((k + 3) & 3, cs + (k + 3) * NW);
                gather_set((k + 1) & 3, (k + 1) & 1);
                if constexpr (MODE != 0) bits_set((k + 2) & 3, (k + 2) & 1);
                add_set(k, k & 1);
            }
        }
#if defined(SPLAHIP_COB_EXP) && SPLAHIP_COB_EXP == 1
        reinterpret_cast<uint32_t*>(s_raw)[threadIdx.x] = to_bits(dbg);
#endif
        __syncthreads();
#ifdef SPLAHIP_COB_TIMING
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_cob_t[2][blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
        if constexpr (kWide) {
            if (narrow) {
                cob_piece_end<T, COB_FADD, VXM, TPB, false>(s_raw, t, a.ranges[t.range], a.slot16, a.hub_row, a.acc, a.ticket, a.touched, a.n_rows, R, a.acc_stride, a.accumulate, mask,
                                                            r, add, op_select, init);
#ifdef SPLAHIP_COB_TIMING
                if (threadIdx.x == 0 && blockIdx.x < 8192) g_cob_t[3][blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
                return;
            }
        }
        cob_piece_end<T, ACC, VXM, TPB, (cob_min_waves<T, ACC, MODE>() > 4u)>(s_raw, t, a.ranges[t.range], a.slot16, a.hub_row, a.acc, a.ticket, a.touched, a.n_rows, R, a.acc_stride,
                                                                              a.accumulate, mask, r, add, op_select, init);
#ifdef SPLAHIP_COB_TIMING
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_cob_t[3][blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    }

    // Row pass of the hub ranges as a kernel of its own (SPLAHIP_COB_HUB_FINISH=1): a hub range's 20 K results are 20 K scattered 4-byte stores, ~19 us when
    // ONE workgroup — the range's last arriver — issues them at the very end of the launch, ~2 us when every CU takes a share.  Thread = hub slot: read the
    // merged accumulator word (plain load: the launch boundary is the release / acquire), re-arm it, finish the row.
    template<typename T, int ACC, bool VXM>
    __global__ __launch_bounds__(256) void cob_hub_finish_kernel(uint32_t n_slots, uint32_t R, uint32_t acc_stride, uint32_t* __restrict__ acc, const uint32_t* __restrict__ hub_row,
                                                                 const T* __restrict__ mask, T* __restrict__ r, uint32_t* __restrict__ touched, uint32_t accumulate, DynBin add,
                                                                 int op_select, T init) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i >= n_slots) return;
        const size_t   ia  = (size_t) (i / R) * acc_stride + i % R;// accumulator q belongs to hub range q
        const uint32_t w   = acc[ia];
        const uint32_t row = hub_row[i];
        acc[ia]            = cob_identity<T, ACC>();
        if (row == 0xFFFFFFFFu) return;
        cob_hub_row<T, ACC, VXM>(w, row, mask, r, touched, accumulate, add, op_select, init);
    }

}// namespace splahip
