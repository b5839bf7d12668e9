// mxv_cob.hpp — masked SpMV over the "column-ordered blocks" copy of a CSR matrix (cob.hip builds it).
//
// Semantics: reference Algo_mxv_masked_cpu<T>::execute, src/cpu/cpu_mxv.hpp:88-103, for an op_add that is associative AND
// commutative and has a fold identity (PLUS, MIN, MAX, BOR, BAND, BXOR on every type they exist for; LOR / LAND on the truth value):
//     r[i] = select(mask[i]) ? add(init, acc_i) : init,   acc_i = fold over row i's products in COLUMN-BLOCK order
// so int / uint results are bit-identical to the reference and fp32 PLUS differs by rounding only (the stated 1e-5 bar).
//
// Why this form exists (profiles/r01_notes.md): with one 4-byte gather of v per entry every L1-missing gather costs a
// 128-byte request, and the chip serves ~110 G such requests per second: R-MAT scale 22 needs 28.7 M of them beyond the 4.2 M lines
// of the entry stream.  Here the rows are cut into RANGES of R = 2^row_bits consecutive rows whose result slice lives in LDS, and a
// range's entries are stored sorted by column, so the 64 lanes of a wave gather ascending, mostly neighbouring columns of v: one request
// serves several products (7.4 M gather requests for the same matrix at R = 16 Ki, 5.5 M at 32 Ki; tools/exp/sim_requests.py), and the
// products are added into the LDS slice with ds_add / ds_min / ds_or ... — no row walk, no segmented reduction, no padding by row length.
//
// Layout (all built on the device, cob.hip):
//   chunk  = 256 entry slots = one wave iteration (a 16-byte load of 4 slots per lane for each of the two streams); the word at memory
//            position 4 l + j is the chunk's sorted entry 64 j + l, so the wave's j-th gather instruction covers 64 CONSECUTIVE sorted entries
//            cr[slot] = row_in_range << col_bits | column - base_col      ax[slot] = value bits      hdr[chunk] = base_col
//            col_bits = 31 - row_bits: the row field has one bit more than a row needs, and a padding slot carries the row value R,
//            the spare word behind the LDS slice — no count, no predicate in the loop
//            a chunk never straddles a (range, column window of 2^col_bits columns) group; only a group's last chunk is padded
//   tile   = up to T consecutive chunks of ONE range = one workgroup's work: zero the LDS slice, walk the chunks (wave w takes chunks
//            w, w + waves, ...), then either finish the rows (a range that is a single tile) or store the slice as a partial slab
//   cob_combine_kernel finishes the rows of the ranges that were cut into several tiles from their slabs, in tile order (deterministic)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_utils.hpp"
#include "ops.hpp"

namespace splahip {

    struct CobTile {
        uint32_t first_chunk;
        uint32_t n_chunks;
        uint32_t range;
        uint32_t slab;// partial-slab index, or 0xFFFFFFFF when the tile is the whole range (it finishes its rows itself)
    };
    struct CobMulti {// a range cut into several tiles
        uint32_t range;
        uint32_t first_slab;
        uint32_t n_slabs;
        uint32_t pad;
    };

    constexpr uint32_t kCobChunk  = 256;// slots per chunk
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

    // bits of the fold identity the slice starts from
    template<typename T, int ACC>
    __host__ __device__ constexpr uint32_t cob_identity() {
        if constexpr (ACC == COB_ADD) return std::is_same<T, float>::value ? 0x80000000u : 0u;// -0.0f: x + -0.0f == x for every x, +0.0f and -0.0f included
        else if constexpr (ACC == COB_MIN)
            return std::is_same<T, int32_t>::value ? 0x7FFFFFFFu : 0xFFFFFFFFu;
        else if constexpr (ACC == COB_MAX)
            return std::is_same<T, int32_t>::value ? 0x80000000u : 0u;
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
        if constexpr (ACC == COB_ADD) {
            (void) __hip_atomic_fetch_add(slot, to_bits(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);// int / uint (fp32 sums live in a double slice: cob_accumulate4)
        } else if constexpr (ACC == COB_MIN) {
            if constexpr (std::is_same<T, int32_t>::value) (void) __hip_atomic_fetch_min(reinterpret_cast<int32_t*>(slot), p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                (void) __hip_atomic_fetch_min(slot, to_bits(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if constexpr (ACC == COB_MAX) {
            if constexpr (std::is_same<T, int32_t>::value) (void) __hip_atomic_fetch_max(reinterpret_cast<int32_t*>(slot), p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
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
        } else {
            cob_accumulate<T, ACC>(s_acc + r0, p0);
            cob_accumulate<T, ACC>(s_acc + r1, p1);
            cob_accumulate<T, ACC>(s_acc + r2, p2);
            cob_accumulate<T, ACC>(s_acc + r3, p3);
        }
    }

    // two partial slices of the same rows (combine pass): the same monoid on plain values
    template<typename T, int ACC>
    __device__ __forceinline__ uint32_t cob_merge(uint32_t a, uint32_t b) {
        if constexpr (ACC == COB_ADD) return to_bits(wrap_add(from_bits<T>(a), from_bits<T>(b)));
        else if constexpr (ACC == COB_MIN)
            return to_bits(eval_bin<SPLAHIP_MIN>(from_bits<T>(a), from_bits<T>(b)));
        else if constexpr (ACC == COB_MAX)
            return to_bits(eval_bin<SPLAHIP_MAX>(from_bits<T>(a), from_bits<T>(b)));
        else if constexpr (ACC == COB_OR || ACC == COB_LOR)
            return a | b;
        else if constexpr (ACC == COB_AND || ACC == COB_LAND)
            return a & b;
        else
            return a ^ b;
    }

    // the row's result from its accumulated slice word: add(init, acc); LOR / LAND slices hold a truth value
    template<typename T, int ACC, class Add>
    __device__ __forceinline__ T cob_finish(uint32_t acc, T init, const Add& add) {
        if constexpr (ACC == COB_LOR || ACC == COB_LAND) return add(init, acc ? T(1) : T(0));
        else
            return add(init, from_bits<T>(acc));
    }

    struct CobArgs {
        const uint32_t* span_first;// first piece of every workgroup (n_spans + 1)
        const CobTile*  tiles;
        const uint32_t* hdr;// per chunk: first column of its window
        const uint32_t* cr;
        const uint32_t* ax;
        const uint32_t* Ap;// n_rows + 1 (rows without an entry keep init exactly; LOR / LAND only)
        uint32_t*       slabs;
        uint32_t*       touched;// VXM only: per output index, 1 = selected and at least one product
        uint32_t        n_rows;
        uint32_t        row_bits;
        uint32_t        col_bits;
        uint32_t        pad_chunk;// an all-padding chunk behind the last real one
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
        return (MODE != 0 && sizeof(typename CobSlice<T, ACC>::word) == 4) ? 8u : 4u;
#endif
    }

    template<typename T, class Mult, class Add, int ACC, bool VXM, unsigned TPB, int MODE = 0>
    __global__ __launch_bounds__(TPB, (cob_min_waves<T, ACC, MODE>())) void cob_mxv_kernel(CobArgs a, const T* __restrict__ v, const T* __restrict__ mask, T* __restrict__ r, Mult mult,
                                                          Add add, int op_select, T init) {
        using word = typename CobSlice<T, ACC>::word;
        constexpr bool kWide = sizeof(word) == 8;// fp32 PLUS: the slice holds doubles
        extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
        word* const        s_acc = reinterpret_cast<word*>(s_raw);
        constexpr unsigned NW    = TPB / kWave;
        const uint32_t     R     = 1u << a.row_bits;
        const uint32_t     t_end = a.span_first[blockIdx.x + 1];
        for (uint32_t ti = a.span_first[blockIdx.x]; ti < t_end; ++ti) {// the workgroup's pieces, one after the other
        const CobTile t = a.tiles[ti];
        {
            const uint32_t id = cob_identity<T, ACC>();
            uint4*         s4 = reinterpret_cast<uint4*>(s_raw);
            const uint4    z  = kWide ? make_uint4(0u, id, 0u, id) : make_uint4(id, id, id, id);// a double's high word carries the sign of -0.0
            for (uint32_t i = threadIdx.x; i < R * sizeof(word) / 16 + 1; i += TPB) s4[i] = z;// + the spare words the padding slots add into
        }
        __syncthreads();
        const unsigned lane  = lane_id();
        const uint32_t cmask = (1u << a.col_bits) - 1u;
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4*       cr4 = reinterpret_cast<const u4*>(a.cr) + lane;
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
        auto           chunk_of = [&](uint32_t ci) { return ci < t.n_chunks ? t.first_chunk + ci : a.pad_chunk; };
        u4             E[4], X[4];
        uint32_t       B[4];
        T              G[2][4];
        u4             W[2];   // MODE 1 / 2: the gathered bitmap words of a step
        uint32_t       M[2];   //             and, once they are in, the step's four bits
        auto load_set = [&](int q, uint32_t ci) {
            const uint32_t k = chunk_of(ci);
            E[q]             = __builtin_nontemporal_load(cr4 + (size_t) k * (kCobChunk / 4));
            if constexpr (MODE != 2) X[q] = __builtin_nontemporal_load(ax4 + (size_t) k * (kCobChunk / 4));
            B[q] = hb[k];
        };
        auto gather_set = [&](int q, int gq) {
#if defined(SPLAHIP_COB_EXP) && (SPLAHIP_COB_EXP == 2 || SPLAHIP_COB_EXP == 4)
            G[gq][0] = from_bits<T>(B[q] + (E[q].x & cmask)), G[gq][1] = from_bits<T>(B[q] + (E[q].y & cmask)), G[gq][2] = from_bits<T>(B[q] + (E[q].z & cmask)),
            G[gq][3] = from_bits<T>(B[q] + (E[q].w & cmask));// DIAGNOSTIC BUILD: no gathers
#else
            if constexpr (MODE == 0) {
                G[gq][0] = v[B[q] + (E[q].x & cmask)];
                G[gq][1] = v[B[q] + (E[q].y & cmask)];
                G[gq][2] = v[B[q] + (E[q].z & cmask)];
                G[gq][3] = v[B[q] + (E[q].w & cmask)];
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
            } else if constexpr (VXM)
                cob_accumulate4<T, ACC>(s_acc, e.x >> a.col_bits, e.y >> a.col_bits, e.z >> a.col_bits, e.w >> a.col_bits, mult(g0, from_bits<T>(x.x)),
                                        mult(g1, from_bits<T>(x.y)), mult(g2, from_bits<T>(x.z)), mult(g3, from_bits<T>(x.w)));
            else
                cob_accumulate4<T, ACC>(s_acc, e.x >> a.col_bits, e.y >> a.col_bits, e.z >> a.col_bits, e.w >> a.col_bits, mult(from_bits<T>(x.x), g0),
                                        mult(from_bits<T>(x.y), g1), mult(from_bits<T>(x.z), g2), mult(from_bits<T>(x.w), g3));
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
        const uint32_t row0 = t.range << a.row_bits;
        // four consecutive slice words as fp32 / int bits (a double slice is rounded to fp32 here, once per row and tile)
        auto slice4 = [&](uint32_t i) {
            if constexpr (kWide) {
                const double2 lo = *reinterpret_cast<const double2*>(s_acc + i), hi = *reinterpret_cast<const double2*>(s_acc + i + 2);
                return make_uint4(to_bits((float) lo.x), to_bits((float) lo.y), to_bits((float) hi.x), to_bits((float) hi.y));
            } else {
                return *reinterpret_cast<const uint4*>(s_acc + i);
            }
        };
        if (t.slab != kCobNoSlab) {
            uint4* out = reinterpret_cast<uint4*>(a.slabs + (size_t) t.slab * R);
            for (uint32_t i = threadIdx.x; i < R / 4; i += TPB) out[i] = slice4(i * 4);
        } else
        // the piece is the whole range: finish its rows (16 bytes of r per thread)
        for (uint32_t i = threadIdx.x * 4; i < R; i += TPB * 4) {
            const uint32_t row = row0 + i;
            if (row >= a.n_rows) break;
            const uint4 acc = slice4(i);
            const uint32_t aw[4] = {acc.x, acc.y, acc.z, acc.w};
            T              out[4];
            uint32_t       tch[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t rj  = row + j;
                bool           sel = rj < a.n_rows;
                if (sel && op_select != SPLAHIP_ALWAYS) sel = eval_sel<T>(op_select, mask[rj]);
                if constexpr (VXM) {// no init: an output exists where the index is selected and holds a product; its value is the fold itself
                    tch[j] = sel && a.Ap[rj + 1] != a.Ap[rj];
                    out[j] = from_bits<T>(aw[j]);
                } else if (a.accumulate) {
                    const T prev = rj < a.n_rows ? r[rj] : init;
                    if (sel) sel = a.Ap[rj + 1] != a.Ap[rj];
                    out[j] = sel ? cob_finish<T, ACC>(aw[j], prev, add) : prev;
                } else {
                    if constexpr (ACC == COB_LOR || ACC == COB_LAND) {
                        if (sel) sel = a.Ap[rj + 1] != a.Ap[rj];// a row without an entry keeps init itself, not its truth value
                    }
                    out[j] = sel ? cob_finish<T, ACC>(aw[j], init, add) : init;
                }
            }
            if (row + 3 < a.n_rows) {
                *reinterpret_cast<uint4*>(r + row) = make_uint4(to_bits(out[0]), to_bits(out[1]), to_bits(out[2]), to_bits(out[3]));
                if constexpr (VXM) *reinterpret_cast<uint4*>(a.touched + row) = make_uint4(tch[0], tch[1], tch[2], tch[3]);
            } else
                for (int j = 0; j < 4 && row + j < a.n_rows; ++j) {
                    r[row + j] = out[j];
                    if constexpr (VXM) a.touched[row + j] = tch[j];
                }
        }
        __syncthreads();// the slice is re-initialised for the next piece
        }
    }

    // rows of the ranges that were cut into several tiles: merge the range's slabs in tile order, then finish like the main kernel.
    // grid: (R / (TPB * 4) blocks per range) x n_multi
    template<typename T, class Add, int ACC, bool VXM, unsigned TPB>
    __global__ __launch_bounds__(TPB) void cob_combine_kernel(const CobMulti* __restrict__ multi, const uint32_t* __restrict__ slabs, const uint32_t* __restrict__ Ap,
                                                              const T* __restrict__ mask, T* __restrict__ r, uint32_t* __restrict__ touched, uint32_t n_rows,
                                                              uint32_t row_bits, Add add, int op_select, T init, uint32_t accumulate) {
        const uint32_t R   = 1u << row_bits;
        const CobMulti m   = multi[blockIdx.y];
        const uint32_t i   = (blockIdx.x * TPB + threadIdx.x) * 4;
        const uint32_t row = (m.range << row_bits) + i;
        if (i >= R || row >= n_rows) return;
        uint4 acc = *reinterpret_cast<const uint4*>(slabs + (size_t) m.first_slab * R + i);
        for (uint32_t s = 1; s < m.n_slabs; ++s) {
            const uint4 b = *reinterpret_cast<const uint4*>(slabs + (size_t) (m.first_slab + s) * R + i);
            acc.x         = cob_merge<T, ACC>(acc.x, b.x);
            acc.y         = cob_merge<T, ACC>(acc.y, b.y);
            acc.z         = cob_merge<T, ACC>(acc.z, b.z);
            acc.w         = cob_merge<T, ACC>(acc.w, b.w);
        }
        const uint32_t aw[4] = {acc.x, acc.y, acc.z, acc.w};
        T              out[4];
        uint32_t       tch[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t rj  = row + j;
            bool           sel = rj < n_rows;
            if (sel && op_select != SPLAHIP_ALWAYS) sel = eval_sel<T>(op_select, mask[rj]);
            if constexpr (VXM) {
                tch[j] = sel && Ap[rj + 1] != Ap[rj];
                out[j] = from_bits<T>(aw[j]);
            } else if (accumulate) {
                const T prev = rj < n_rows ? r[rj] : init;
                if (sel) sel = Ap[rj + 1] != Ap[rj];
                out[j] = sel ? cob_finish<T, ACC>(aw[j], prev, add) : prev;
            } else {
                if constexpr (ACC == COB_LOR || ACC == COB_LAND) {
                    if (sel) sel = Ap[rj + 1] != Ap[rj];
                }
                out[j] = sel ? cob_finish<T, ACC>(aw[j], init, add) : init;
            }
        }
        if (row + 3 < n_rows) {
            *reinterpret_cast<uint4*>(r + row) = make_uint4(to_bits(out[0]), to_bits(out[1]), to_bits(out[2]), to_bits(out[3]));
            if constexpr (VXM) *reinterpret_cast<uint4*>(touched + row) = make_uint4(tch[0], tch[1], tch[2], tch[3]);
        } else
            for (int j = 0; j < 4 && row + j < n_rows; ++j) {
                r[row + j] = out[j];
                if constexpr (VXM) touched[row + j] = tch[j];
            }
    }

}// namespace splahip
