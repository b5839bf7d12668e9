// mxv_sell.hpp — masked SpMV over a sliced-ELL copy of the matrix with an LDS-resident cache of the hottest columns of v.
//
// Why (measured on MI355X, profiles/r01_notes.md): a CSR SpMV on a scale-free matrix is not HBM-bound but bound by the
// divergent 4-byte gathers v[Aj[k]] — every gathered lane that misses the vector L1 moves a 128-B line over the CU's
// 64 B/clk L2->L1 path (~1.1 G gathers/s/CU), and even an L1 hit costs ~0.8 clk per lane.  The only gather path that is
// an order of magnitude cheaper is LDS (a random ds_read_b32 costs ~0.1 clk per lane), and LDS can only be spent on the
// vector if the matrix side needs none of it.  Hence this layout:
//
//   * virtual rows: every row is cut into chunks of at most C = 256 entries; rows of <= C entries are one chunk.
//   * the chunks are sorted by length (descending, stable) and grouped 64 at a time into SLICES; a slice stores its
//     entries column-major — entry e of lane l at off[s] + 64 e + l — padded to the slice's longest chunk (global sort:
//     0.1 % padding on R-MAT).  One wavefront owns a slice: lane l walks chunk l sequentially, so every load of Aj2 / Ax2
//     is a fully coalesced 256-B wave access, products are folded in a register in the row's own entry order (the
//     reference's strict left fold — bit-exact in fp32 for rows <= C), and neither LDS staging nor a cross-lane
//     reduction exists.
//   * the 40,000 most referenced columns get an LDS slot: their ids are re-encoded in Aj2 (bit 31 + slot), their values
//     are packed once per launch (hot_pack) and copied by every workgroup into its 156 KB of LDS.  On R-MAT scale 22 they
//     take ~2/3 of all gathers off the vector-memory pipeline.
//   * ONE 1024-thread workgroup per CU (LDS is the limiter, 16 waves); memory-level parallelism comes from the loop, not
//     from occupancy: each lane keeps a group of U = 8 (Aj, Ax) pairs and 8 gathers in flight and prefetches the next
//     group and the next slice's header, all with static load counts so hipcc can place exact s_waitcnt values.  A workgroup's waves take its slices widest
//     first from an LDS counter (dynamic: which slices a mask skips is not known to the plan).
//   * chunks of rows longer than C write partials; sell_combine folds them in chunk order (associative op_add only).
//   * rows without entries and unselected rows get `init` from a coalesced pass over the workgroup's share of rows.
//
//
// Sparse input vectors (BITS form).  When 0 annihilates op_mult and is the identity of op_add (int / uint BAND|MULT|LAND x
// BOR|PLUS|BXOR, uint MAX), an entry whose v value is 0 cannot change its row's fold, so all the kernel needs from v is ONE BIT
// per column — and 156 KB of LDS hold the bits of the 1.28 M most referenced columns (97 % of the entries of R-MAT scale 22)
// where they held 40,000 values.  The columns are stored a second time as RANKS (position in the descending reference-count
// order): rank < 1.28 M reads its bit from LDS; the few entries of colder columns (3 % on R-MAT scale 22) are taken out of the slices
// into a rank-sorted side list that sell_cold folds into r afterwards with integer atomics (every BITS op_add is commutative), so
// the main kernel's inner loop issues no vector-memory access besides its two coalesced streams — a conditional global load there
// forces s_waitcnt vmcnt(0) per entry and with it a full memory latency per entry.  A pass over v in column order
// (sell_vprep) scatters the bits (atomicOr), the non-zero values in rank order, and min / max / count of the non-zero values;
// one launch carries both forms and the device-side statistics decide which one a workgroup runs (no host round trip): BITS when all
// non-zero values are equal (a BFS frontier: the set bit IS the value) or when at most a quarter of v is non-zero (set bits
// gather their value from the rank-ordered copy).  With an LDS-resident bit per entry the kernel has no vector-memory gather
// left: it streams the selected rows' entries once.
//
// Semantics: reference Algo_mxv_masked_cpu<T>::execute, src/cpu/cpu_mxv.hpp:56-106.
#pragma once

#include "block_utils.hpp"
#include "ops.hpp"

#ifndef SPLAHIP_SELL_EXP
#define SPLAHIP_SELL_EXP 0
#endif

namespace splahip {

#ifdef SPLAHIP_SELL_TIMING
    // DIAGNOSTIC BUILD ONLY (tools/exp/exp_sell_timing.py): per wave of the launch (4096), s_memtime ticks summed over its slices:
    // [0] top of the loop .. loads issued, [1] .. first group landed, [2] .. fold + stores issued, [3] slices, [4] wave total, [5] prologue
    static __device__ unsigned long long g_sell_t[12][4096];// [6] live slices [7] group rounds [8] sum of widths [9] selected lanes [10] kernel entry .. first slice
#define SELL_STAMP(t)                                            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    t = __builtin_amdgcn_s_memtime();
#define SELL_NOW(t) t = __builtin_amdgcn_s_memtime();
#else
#define SELL_STAMP(t)
#define SELL_NOW(t)
#endif

    struct SellCfg {
        static constexpr uint kC       = 256;  // max entries per virtual row (chunk)
        static constexpr uint kXL      = kC;   // rows longer than this get ROW-EXCLUSIVE slices: 64 lanes x <= kC entries of one row, lane l
                                               // holding entries l, l + 64, ... of the slice's 16 Ki-entry block (= the row in CSR order).
                                               // Must equal kC: a "partial" lane then always belongs to a row-exclusive slice (the bit form
                                               // folds such a slice's lanes in the wave)
        static constexpr uint kHot     = 40000;// LDS-resident values of the gathered vector (156.25 KB of the 160 KB)
        static constexpr uint kTpb     = 1024; // one workgroup per CU
        static constexpr uint kWaves   = kTpb / kWave;
        static constexpr uint kU       = 8;    // entries per lane and group (16: same speed, 98 VGPRs)
        static constexpr uint kHotFlag = 0x80000000u;
        static constexpr uint kPartial = 0x80000000u;// in vrow: this chunk belongs to a row longer than kC
        // BITS form: ranks [0, kBitsLds) have an LDS-resident bit; entries of colder columns live in the cold list.  (80 KB of bits and two
        // workgroups per CU was tried: the 64-VGPR cap spills inside the slice loop, 165 -> 173 us on the BFS step.)
        static constexpr uint kBitWords = kHot;
        static constexpr uint kBitsLds  = kBitWords * 32 - 1;
        // What Aj2r stores for a column of rank r: bit b = r + 1 of the bitmap is the column's (bit 0 is never set), and the CODE of an
        // LDS-resident bit is (b >> 5) << 7 | (b & 31) — code >> 5 is the byte address of its LDS word, the low five bits are the shift.
        // Code 0 folds nothing: padding, colder columns (their entries are in the cold list), and what a bounds-checked buffer load
        // returns for a lane that asks beyond its slice.
        __host__ __device__ static constexpr uint bits_code(uint rank) { return rank < kBitsLds ? ((((rank + 1u) >> 5) << 7) | ((rank + 1u) & 31u)) : 0u; }
        __host__ __device__ static constexpr uint code_rank(uint code) { return (((code >> 7) << 5) | (code & 31u)) - 1u; }
        // statistics behind the bitmap, 4 words per sell_vprep workgroup: count of non-zero v entries, max of ~bits (= ~min), max of bits
        static constexpr uint kPrepBlocks = 4096;
        static constexpr uint kStatCount = 0, kStatNotMin = 1, kStatMax = 2;
        static constexpr uint kDecision = 4 * kPrepBlocks;// behind the slots: form (1 = BITS), is_const, const bits — written by the BITS launch
    };

    // which form runs, from the per-workgroup statistics sell_vprep left on the device (every wave of both kernels evaluates the
    // same predicate: kPrepBlocks x 3 words, no atomics and no host round trip); force: 0 = decide, 1 = BITS (tests / A-B runs)
    __device__ __forceinline__ bool sell_bits_form(const uint* __restrict__ stats, uint n_cols, int force, bool& is_const, uint& const_bits) {
        uint count = 0, notmin = 0, mx = 0;
        for (uint b = lane_id(); b < SellCfg::kPrepBlocks; b += kWave) {
            count += stats[4 * b + SellCfg::kStatCount];
            notmin = max(notmin, stats[4 * b + SellCfg::kStatNotMin]);
            mx     = max(mx, stats[4 * b + SellCfg::kStatMax]);
        }
        count      = wave_sum(count);
        notmin     = wave_max(notmin);
        mx         = wave_max(mx);
        is_const   = count == 0 || ~notmin == mx;
        const_bits = mx;
        if (force) return force == 1;
        return is_const || count <= n_cols / 4;
    }

    template<class Add>
    __device__ __forceinline__ int functor_op(const Add& add) {
        return binop_id(add);
    }
    // the atomic of a bit-form op_add: BOR, PLUS, BXOR, uint MAX (zero family) and MIN (min-plus family; fp32 through compare-and-swap with
    // the reference's `b < a ? b : a`, so a NaN never replaces a number)
    template<typename T>
    __device__ __forceinline__ void atomic_fold(T* dst, T value, int op_add) {
        uint*      d = reinterpret_cast<uint*>(dst);
        const uint p = to_bits(value);
        switch (op_add) {
            case SPLAHIP_BOR: atomicOr(d, p); break;
            case SPLAHIP_BXOR: atomicXor(d, p); break;
            case SPLAHIP_MAX: atomicMax(d, p); break;
            case SPLAHIP_MIN:
                if constexpr (std::is_same<T, int32_t>::value) atomicMin(reinterpret_cast<int*>(d), (int) p);
                else if constexpr (std::is_same<T, uint32_t>::value)
                    atomicMin(d, p);
                else {
                    uint seen = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (value < from_bits<T>(seen)) {
                        const uint prev = atomicCAS(d, seen, p);
                        if (prev == seen) break;
                        seen = prev;
                    }
                }
                break;
            default: atomicAdd(d, p); break;// PLUS: two's complement, same bits for int and uint
        }
    }

    // column-order pass over v: hot values for the value form; bits, rank-ordered values and statistics for the BITS form.
    // kPrepBlocks workgroups, grid-stride; every workgroup writes its own statistics slot (a same-address atomic per wave cost 0.8 ms)
    template<typename T>
    __global__ __launch_bounds__(256) void sell_vprep_kernel(uint n_cols, const T* __restrict__ v, const uint* __restrict__ rank_of, uint n_hot,
                                                             const uint* __restrict__ hot_cols, T* __restrict__ hot_values, uint* __restrict__ vbits, uint* __restrict__ vranked,
                                                             uint* __restrict__ stats, uint n_rows, const T* __restrict__ mask, int op_select,
                                                             unsigned long long* __restrict__ selbits, uint4* __restrict__ next_bits, uint n_bit_quads,
                                                             T* __restrict__ r, T init, uint skip) {
        __shared__ uint s_red[3][256 / kWave];
        // the bitmap is double-buffered: this call fills `vbits` (cleared by the previous call) and clears the other one for the next
        // call — no memset launch per call
        // the first step's mask quad and v quad are requested before anything is stored (at the bench sizes every lane has ONE step of each loop: without
        // this the v load waited behind the selection words' shuffles and store)
        const uint  t0    = (blockIdx.x * 256 + threadIdx.x) * 4;
        const uint  n4    = n_cols & ~3u;
        const bool  pre_m = op_select != SPLAHIP_ALWAYS && t0 + 4 <= n_rows, pre_v = t0 < n4;
        uint4       m_pre = make_uint4(0, 0, 0, 0), x_pre = make_uint4(0, 0, 0, 0);
        if (pre_m) m_pre = *reinterpret_cast<const uint4*>(mask + t0);
        if (pre_v) x_pre = *reinterpret_cast<const uint4*>(v + t0);
        for (uint q = blockIdx.x * 256 + threadIdx.x; q < n_bit_quads; q += gridDim.x * 256) next_bits[q] = make_uint4(0, 0, 0, 0);
        // row selection as one bit per row, and init into every row.  Four consecutive rows per lane (16-byte load and store): a lane's
        // four bits sit at positions 4 l .. 4 l + 3 of its wave's 256-row string; the 16 lanes of a 64-bit word OR their nibbles together
        // with four shuffles and the group's first lane stores the word.  (mask and r are 16-byte aligned: the dispatch checks it.)
        {
            const uint4 init4 = make_uint4(to_bits(init), to_bits(init), to_bits(init), to_bits(init));
            const uint  span  = (n_rows + 255u) & ~255u;// whole 256-row wave steps
            for (uint row = (blockIdx.x * 256 + threadIdx.x) * 4; row < span; row += gridDim.x * 1024) {
                uint nib = 0;
                if (row + 4 <= n_rows) {
                    *reinterpret_cast<uint4*>(r + row) = init4;
                    if (op_select == SPLAHIP_ALWAYS) nib = 0xFu;// (the mask is not read: 4 B per row less)
                    else {
                        const uint4 m = row == t0 ? m_pre : *reinterpret_cast<const uint4*>(mask + row);
                        nib = (eval_sel(op_select, from_bits<T>(m.x)) ? 1u : 0u) | (eval_sel(op_select, from_bits<T>(m.y)) ? 2u : 0u) |
                              (eval_sel(op_select, from_bits<T>(m.z)) ? 4u : 0u) | (eval_sel(op_select, from_bits<T>(m.w)) ? 8u : 0u);
                    }
                } else {
                    for (uint i = 0; i < 4; ++i)
                        if (row + i < n_rows) {
                            r[row + i] = init;
                            nib |= (op_select == SPLAHIP_ALWAYS || eval_sel(op_select, mask[row + i])) ? (1u << i) : 0u;
                        }
                }
                unsigned long long wd = (unsigned long long) nib << (4u * (lane_id() & 15u));
#pragma unroll
                for (uint off = 1; off < 16; off <<= 1) wd |= __shfl_xor(wd, off, kWave);
                if ((lane_id() & 15u) == 0) selbits[row / kWave] = wd;// row = first row of this 16-lane group: a multiple of 64
            }
        }
        // hot values of the value form: a gather by the plan's hot-column list (the rank of a column is only looked up where v is not 0:
        // the pass reads v once and, for a sparse v, almost nothing of rank_of — its cost must not grow with the padded column count of
        // a multi-GPU run)
        for (uint t = blockIdx.x * 256 + threadIdx.x; t < n_hot; t += gridDim.x * 256) hot_values[t] = v[hot_cols[t]];
        uint count = 0, notmin = 0, mx = 0;
        // four columns per lane and step (16-byte loads; hipMalloc'ed operands are 256-byte aligned), tail by single columns.  The rank
        // lookups of a lane's non-zero entries are requested together and the bit / value scatters are fire-and-forget: one round trip
        // per step, not one per non-zero entry
        auto step = [&](uint j, const uint (&b)[4], uint nb) {
            uint rank[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint i = 0; i < 4; ++i)
                if (i < nb && b[i] != skip) rank[i] = rank_of[j + i];
#pragma unroll
            for (uint i = 0; i < 4; ++i)
                if (i < nb && b[i] != skip) {
                    atomicOr(&vbits[(rank[i] + 1u) >> 5], 1u << ((rank[i] + 1u) & 31u));// (bit rank + 1: SellCfg::bits_code)
                    vranked[rank[i]] = b[i];
                    count += 1;
                    notmin = max(notmin, ~b[i]);
                    mx     = max(mx, b[i]);
                }
        };
        for (uint j = t0; j < n4; j += gridDim.x * 1024) {
            const uint4 x    = j == t0 ? x_pre : *reinterpret_cast<const uint4*>(v + j);
            const uint  b[4] = {x.x, x.y, x.z, x.w};
            step(j, b, 4);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && n4 < n_cols) {
            uint b[4] = {0, 0, 0, 0};
            for (uint i = 0; n4 + i < n_cols; ++i) b[i] = to_bits(v[n4 + i]);
            step(n4, b, n_cols - n4);
        }
        count  = wave_sum(count);
        notmin = wave_max(notmin);
        mx     = wave_max(mx);
        if (lane_id() == 0) {
            s_red[0][wave_id()] = count;
            s_red[1][wave_id()] = notmin;
            s_red[2][wave_id()] = mx;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint w = 1; w < 256 / kWave; ++w) {
                count += s_red[0][w];
                notmin = max(notmin, s_red[1][w]);
                mx     = max(mx, s_red[2][w]);
            }
            stats[4 * blockIdx.x + SellCfg::kStatCount]  = count;
            stats[4 * blockIdx.x + SellCfg::kStatNotMin] = notmin;
            stats[4 * blockIdx.x + SellCfg::kStatMax]    = mx;
        }
    }

    template<typename T>
    __global__ __launch_bounds__(256) void sell_hot_pack_kernel(const uint* __restrict__ hot_cols, uint n_hot, const T* __restrict__ v, T* __restrict__ hot_values) {
        const uint s = blockIdx.x * 256 + threadIdx.x;
        if (s < n_hot) hot_values[s] = v[hot_cols[s]];
    }

    __device__ __forceinline__ uint sell_stream_load(const uint* p) { return __builtin_nontemporal_load(p); }

    // BITS = false: Aj2 holds hot-slot / raw column words, s_hot the hot VALUES, cold lanes gather v.
    // BITS = true : Aj2 holds ranks, s_hot the hot BITS (vbits), cold ranks read the global bitmap, set bits take const_bits or
    //               gather vranked; `gate` (the statistics) != nullptr makes the launch a no-op unless its form is the chosen one.
    // The slice walk of one workgroup.  s_hot / s_next live in the calling kernel (the two forms share one LDS allocation).
    // VXM (value form only): the pull form of vxm over the transposed matrix with EVERY index of v stored — products are mult(v, x),
    // a row folds from its first product (no init) and touched[row] tells whether the output entry exists (reference
    // src/cpu/cpu_vxm.hpp:101-107); r is only written for touched rows.
    template<typename T, class Mult, class Add, bool BITS, bool VXM = false, uint KU = SellCfg::kU>
    __device__ __forceinline__ void sell_mxv_body(T* s_hot, uint* s_next_p, uint n_rows, const uint* __restrict__ Ap, uint n_slices,
                                                  const uint* __restrict__ vrow, const uint* __restrict__ vlen,
                                                  const uint* __restrict__ vslot, const uint* __restrict__ off,
                                                  const uint* __restrict__ Aj2, const uint* __restrict__ Ax2,
                                                  const T* __restrict__ hot_values, uint n_hot,
                                                  const T* __restrict__ mask, const T* __restrict__ v,
                                                  T* __restrict__ r, T* __restrict__ partial,
                                                  Mult mult, Add add_raw, int op_select, T init,
                                                  bool is_const, uint cbits,
                                                  const uint* __restrict__ vbits, const uint* __restrict__ vranked,
                                                  const uint* __restrict__ selbits, const uint* __restrict__ sorder,
                                                  uint* __restrict__ touched = nullptr, T ident = T(), uint xbits = 0) {
        using C = SellCfg;
        const Reassoc<Add> add(add_raw);// lane partials of row-exclusive slices start at products (ops.hpp: fp32 MIN / MAX drop NaN products)
        const bool xu = Ax2 == nullptr;// every matrix value is the word `xbits`: the value stream is not read
#ifdef SPLAHIP_SELL_TIMING
        unsigned long long t_entry = 0;
        SELL_NOW(t_entry)
#endif
        static_assert(!(BITS && VXM), "the pull form of vxm uses the value form");
        uint& s_next = *s_next_p;// next of this workgroup's slices nobody has taken yet (ordinal)

        const bool always = op_select == SPLAHIP_ALWAYS;
        if (threadIdx.x == 0) s_next = C::kWaves;// wave w starts with ordinal w
        const T cval = from_bits<T>(cbits);

        // 0. hot values -> LDS, 16 bytes per lane (the table is allocated in whole quads)
        {
            const uint4* src = reinterpret_cast<const uint4*>(hot_values);
            uint4*       dst = reinterpret_cast<uint4*>(s_hot);
            const uint   nq  = (n_hot + 3u) / 4u;
            for (uint q = threadIdx.x; q < nq; q += C::kTpb) dst[q] = src[q];
        }

        // 1. rows that no slice lane will write: empty rows and unselected rows.  Four consecutive rows per lane and step: the
        //    row pointers and mask values arrive as 16-byte loads; the stores stay per row (a selected non-empty row is written
        //    by its slice lane, possibly from another workgroup, and must not be touched here).
#if SPLAHIP_SELL_EXP != 3 /* diagnostic 3: no row pass */
        if constexpr (!BITS) {// (bit form: sell_vprep already wrote init to every row)
            const uint per = ((n_rows + gridDim.x - 1) / gridDim.x + 3u) & ~3u;
            const uint rb  = blockIdx.x * per;
            const uint re  = min(n_rows, rb + per);
            for (uint row = rb + 4 * threadIdx.x; row < re; row += 4 * C::kTpb) {
                if (row + 4 <= re) {
                    const uint4 a    = *reinterpret_cast<const uint4*>(Ap + row);
                    const uint  a4   = Ap[row + 4];
                    const uint  b[5] = {a.x, a.y, a.z, a.w, a4};
                    uint4       m    = make_uint4(0, 0, 0, 0);
                    if (!always) m = *reinterpret_cast<const uint4*>(mask + row);
                    const uint mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                    for (uint i = 0; i < 4; ++i)
                        if (b[i + 1] == b[i] || (!always && !eval_sel(op_select, from_bits<T>(mm[i])))) {
                            if constexpr (VXM) touched[row + i] = 0;
                            else
                                r[row + i] = init;
                        }
                } else {
                    for (uint i = row; i < re; ++i)
                        if (Ap[i + 1] == Ap[i] || (!always && !eval_sel(op_select, mask[i]))) {
                            if constexpr (VXM) touched[i] = 0;
                            else
                                r[i] = init;
                        }
                }
            }
        }
#endif
        __syncthreads();

        // 2. slices.  Workgroup b owns slices b, b + grid, b + 2 grid, ... (sorted by width, so every workgroup holds the same mix);
        //    its 16 waves take them widest first from an LDS counter.  A static deal cannot balance: which slices are skipped
        //    depends on the mask (a BFS step skips the widest ones, the visited hubs), and a lane walks its chunk sequentially, so
        //    the wave that starts a 256-wide slice last sets the launch time.
        //    The selection of a slice's rows (a bit of sell_vprep's bitmap in the BITS form, else the mask word) is fetched one slice
        //    ahead and the header two ahead, so a lane knows whether its row is selected BEFORE it asks for entries: the stream
        //    loads are predicated per lane, and a 128-byte line whose 32 lanes are all unselected (the chunks of a visited hub)
        //    is never requested.
        const uint lane = lane_id();
        const uint w    = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
        // position q of the widest-first order -> slice id (the plan's permutation: row-exclusive slices sit ahead of the sorted chunks in
        // memory but are taken by width like everything else)
        auto phys = [&](uint q) -> uint { return q < n_slices ? (sorder ? sorder[q] : q) : n_slices; };
        uint sA   = phys(blockIdx.x + w * gridDim.x);
        if (sA >= n_slices) return;

        auto take_slice = [&]() -> uint {
            uint ord = 0;
            if (lane == 0) ord = atomicAdd(&s_next, 1u);
            return phys(blockIdx.x + __builtin_amdgcn_readfirstlane(ord) * gridDim.x);
        };
        // the word that decides a lane's selection: nothing for ALWAYS, the row's bit word (BITS) or its mask value
        auto sel_word = [&](uint rowinfo, uint len) -> uint {
            if (always || len == 0) return 0u;
            const uint row = rowinfo & ~C::kPartial;
            if constexpr (BITS) return selbits[row >> 5];
            else
                return to_bits(mask[row]);
        };
        auto is_selected = [&](uint word, uint rowinfo, uint len) -> bool {
            if (len == 0) return false;
            if (always) return true;
            if constexpr (BITS) return ((word >> (rowinfo & 31u)) & 1u) != 0;
            else
                return eval_sel(op_select, from_bits<T>(word));
        };

#ifdef SPLAHIP_SELL_TIMING
        unsigned long long t_a = 0, t_b = 0, t_c = 0, t_d = 0, acc0 = 0, acc1 = 0, acc2 = 0, nsl = 0, t_begin = 0, n_live = 0, n_rounds = 0, n_width = 0, n_lanes = 0;
        SELL_NOW(t_begin)
#endif
        uint rowA = vrow[sA * kWave + lane], lenA = vlen[sA * kWave + lane], baseA = off[sA];
        uint sB   = take_slice();
        bool moreB = sB < n_slices;
        uint rowB = 0, lenB = 0, baseB = 0;
        if (moreB) {
            rowB  = vrow[sB * kWave + lane];
            lenB  = vlen[sB * kWave + lane];
            baseB = off[sB];
        }
        uint wordA = sel_word(rowA, lenA);
        // the slice after B: its ID is resolved a whole slice ahead of its header (ordinal from the LDS counter, then the plan's permutation —
        // a dependent global load that would otherwise stall every pass of the loop for a full memory latency)
        uint sC = moreB ? take_slice() : n_slices;
        while (true) {
            SELL_NOW(t_a)
            // ---- slice A
            const uint rowinfo = rowA;
            const uint base    = __builtin_amdgcn_readfirstlane(baseA);
            const uint s_cur   = sA;
            const uint row     = rowinfo & ~C::kPartial;
            const bool chunk   = (rowinfo & C::kPartial) != 0;
            const bool selected = is_selected(wordA, rowinfo, lenA);
            const uint mylen    = selected ? lenA : 0u;
            const uint* pj = Aj2 + base + lane;
            const uint* px = Ax2 + base + lane;
            uint        jn[KU], xn[KU];
#pragma unroll
            for (uint u = 0; u < KU; ++u) {// first group: only lanes that will fold the entry ask for it
                jn[u] = BITS ? 0u : C::kHotFlag;// a lane without an entry reads the never-set bit (code 0) / hot slot 0 and folds nothing
                xn[u] = xu ? xbits : 0u;
                if (u < mylen) {
                    jn[u] = sell_stream_load(pj + u * kWave);
                    if (!xu) xn[u] = sell_stream_load(px + u * kWave);
                }
            }
            // (requested AFTER A's first group: the loads below sit in uniform branches, so the wait for anything older than them is a
            // wait for them too — ahead of A's loads they cost every slice a second memory round trip)
            // C's header, D's id and B's selection word travel while A is folded
            uint       rowC = 0, lenC = 0, baseC = 0;
            const bool moreC = moreB && sC < n_slices;
            if (moreC) {
                rowC  = vrow[sC * kWave + lane];
                lenC  = vlen[sC * kWave + lane];
                baseC = off[sC];
            }
            uint sD = n_slices;
            if (moreC) sD = take_slice();
            uint wordB = 0;
            if (moreB) wordB = sel_word(rowB, lenB);

            SELL_NOW(t_b)
            SELL_STAMP(t_c)
            const unsigned long long live = __ballot(selected);
            if (live != 0ull) {
                // chunks are sorted by length: the first selected lane holds the longest selected chunk
                const uint width = __builtin_amdgcn_readlane(mylen, __ffsll((long long) live) - 1);
#ifdef SPLAHIP_SELL_TIMING
                n_live += 1, n_rounds += (width + KU - 1) / KU, n_width += width, n_lanes += __popcll(live);
#endif
                T          acc   = init;
                bool       has   = !chunk && !VXM;// a row's only chunk folds from init (mxv); chunks of a long row and vxm rows fold from their first product
                if constexpr (BITS) {// a chunk may hold no product at all: its partial is then op_add's identity (0 for every BITS semiring)
                    if (chunk) acc = ident;// (0 for the zero family; init for min-plus: min(init, x) is folded with init again anyway)
                    has = true;
                }
                for (uint e = 0; e < width; e += KU) {
                    uint jc[KU], xc[KU];
#pragma unroll
                    for (uint u = 0; u < KU; ++u) {
                        jc[u] = jn[u];
                        xc[u] = xn[u];
                    }
                    pj += KU * kWave;
                    px += KU * kWave;
                    // prefetch the next group before this group's gathers are issued (uniform conditions)
#pragma unroll
                    for (uint u = 0; u < KU; ++u) {
                        if (e + KU + u < mylen) {
                            jn[u] = sell_stream_load(pj + u * kWave);
                            if (!xu) xn[u] = sell_stream_load(px + u * kWave);
                        }
                    }
                    T    vv[KU];
                    bool on[KU];
                    if constexpr (BITS) {// the group's LDS words are requested together, then tested (one LDS round trip per group, not per entry)
                        uint a[KU];
#pragma unroll
                        for (uint u = 0; u < KU; ++u) a[u] = to_bits(s_hot[jc[u] >> 7]);
#pragma unroll
                        for (uint u = 0; u < KU; ++u) {
                            on[u] = (e + u < mylen) & (((a[u] >> (jc[u] & 31u)) & 1u) != 0);
                            vv[u] = cval;
                        }
                    }
#pragma unroll
                    for (uint u = 0; u < KU && !BITS; ++u) {
                        const bool take = e + u < mylen;
                        on[u]           = take;
                        const bool hot  = (jc[u] & C::kHotFlag) != 0;
#if SPLAHIP_SELL_EXP == 1 /* diagnostic: no gather at all */
                        vv[u] = from_bits<T>(jc[u] | (hot ? 1u : 0u) | (take ? 2u : 0u));
#elif SPLAHIP_SELL_EXP == 2 /* diagnostic: every gather from LDS */
                        vv[u] = s_hot[(jc[u] & 0x7FFFu)];
#else
                        // the LDS read is unconditional (slot 0 for cold lanes) and lands in its own register, so the cold lanes'
                        // global load does not have to wait for it
                        const T a = s_hot[hot ? (jc[u] & ~C::kHotFlag) : 0u];
                        T       b = T();
#if SPLAHIP_SELL_EXP == 6 /* diagnostic: every XCD's cold gathers inside one eighth of v */
                        if (take && !hot) b = v[(jc[u] & 0x7FFFFu) + (blockIdx.x & 7u) * 0x80000u];
#else
                        if (take && !hot) b = v[jc[u]];
#endif
                        vv[u] = hot ? a : b;
#endif
                    }
                    if constexpr (BITS) {
                        if (!is_const) {// set bits fetch their value from the rank-ordered copy of v (all of the group's gathers in flight together)
#pragma unroll
                            for (uint u = 0; u < KU; ++u)
                                if (on[u]) vv[u] = from_bits<T>(vranked[C::code_rank(jc[u])]);
                        }
                    }
#pragma unroll
                    for (uint u = 0; u < KU; ++u) {
                        const bool take = on[u];
                        const T    p    = VXM ? mult(vv[u], from_bits<T>(xc[u])) : mult(from_bits<T>(xc[u]), vv[u]);
                        const T    both = add(acc, p);
                        acc             = take ? (has ? both : p) : acc;
                        has             = has || take;
                    }
                }
                if constexpr (BITS) {
                    if (chunk) {// a row-exclusive slice: the 64 lane results are one row's — fold them in the wave, one atomic into r[row]
                        bool pv = selected;
                        T    t  = acc;
                        wave_reduce_ordered(t, pv, add);
                        if (lane == 0 && pv && to_bits(t) != to_bits(ident)) atomic_fold(r + row, t, functor_op(add));// (folding the identity changes nothing)
                    } else if (selected && to_bits(acc) != to_bits(init)) {// the prep pass wrote init to every row: a row without a product
                        r[row] = acc;                                        // (most rows of a BFS level) costs no scattered 4-byte store
                    }
                } else if (selected) {
                    if (chunk) partial[vslot[s_cur * kWave + lane]] = acc;
                    else {
                        r[row] = acc;
                        if constexpr (VXM) touched[row] = 1;
                    }
                }
            }
#ifdef SPLAHIP_SELL_TIMING
            SELL_NOW(t_d)
            acc0 += t_b - t_a, acc1 += t_c - t_b, acc2 += t_d - t_c, nsl += 1;
            if (!moreB && lane == 0) {
                const uint gw = blockIdx.x * C::kWaves + w;
                if (gw < 4096) {
                    g_sell_t[0][gw] = acc0, g_sell_t[1][gw] = acc1, g_sell_t[2][gw] = acc2, g_sell_t[3][gw] = nsl;
                    g_sell_t[4][gw] = t_d - t_begin;
                    g_sell_t[6][gw] = n_live, g_sell_t[7][gw] = n_rounds, g_sell_t[8][gw] = n_width, g_sell_t[9][gw] = n_lanes;
                    g_sell_t[10][gw] = t_begin - t_entry;
                }
            }
#endif
            if (!moreB) break;
            sA = sB, rowA = rowB, lenA = lenB, baseA = baseB, wordA = wordB;
            sB = sC, moreB = moreC, rowB = rowC, lenB = lenC, baseB = baseC;
            sC = sD;
        }
    }

    // n products of the same value p folded into acc: what the frontier form below needs instead of one add per entry.
    // PLUS / BXOR (integer types only — the dispatch keeps other types away) count; every other bit-form op_add is idempotent.
    template<typename T, class Add>
    __device__ __forceinline__ T fold_repeated(T acc, T p, uint n, const Add& add) {
        if (n == 0) return acc;
        const int op = functor_op(add);
        if constexpr (std::is_integral<T>::value) {
            if (op == SPLAHIP_PLUS) return from_bits<T>(to_bits(acc) + n * to_bits(p));
            if (op == SPLAHIP_BXOR) return (n & 1u) ? from_bits<T>(to_bits(acc) ^ to_bits(p)) : acc;
        }
        return add(acc, p);
    }

    // FRONTIER form of the bit form: every non-skipped value of v is the same word AND every stored matrix value is the same word (a BFS
    // step over an adjacency matrix) — all products are one value pc, a row's result is fold_repeated(init, pc, hits), and the slice walk
    // only COUNTS set bits.  What that buys, measured with per-wave s_memtime stamps on the general walk (R-MAT 22 BFS step: 16 slices
    // per wave, 9.6 of them with a selected lane, mean selected width 13.5 entries): the walk is bound by instruction issue, not by bytes
    // — 16 predicated loads (six instructions each) and 16 folds of ~9 VALU per group whatever the slice's width.  Here:
    //   * loads are bounds-checked BUFFER loads over the slice (descriptor: width x 256 bytes; voffset: the lane's column, or beyond the
    //     buffer for an unselected lane): a lane without an entry at a position gets code 0 — the never-set bit — without a request, a
    //     branch or a select, the per-load offset is an immediate / a scalar, and the load count is static (exact vmcnt waits);
    //   * a slice of at most 16 positions is walked in quads under UNIFORM branches (a width-3 slice issues 4 loads and 4 folds, not 16);
    //   * a wider slice keeps a sliding window of 32 loads per lane in flight, consumed in place eight at a time (the slot's old code is
    //     taken apart before the slot is re-requested, so the slot stays in one register across the back edge);
    //   * an entry is four instructions: code >> 5 (LDS byte address), ds_read, shift by the code's low five bits, add.
    template<typename T, class Mult, class Add>
    __device__ __forceinline__ void sell_frontier_body(uint* s_bits, uint* s_next_p, uint n_slices, const uint* __restrict__ vrow, const uint* __restrict__ vlen,
                                                       const uint* __restrict__ off, const uint* __restrict__ Aj2r, const uint* __restrict__ vbits,
                                                       const uint* __restrict__ selbits, const uint* __restrict__ sorder, T* __restrict__ r, Mult mult, Add add,
                                                       int op_select, T init, T ident, uint cbits, uint xbits) {
        using C = SellCfg;
        uint&      s_next = *s_next_p;
        const bool always = op_select == SPLAHIP_ALWAYS;
#ifdef SPLAHIP_SELL_TIMING
        unsigned long long t_entry = 0;
        SELL_NOW(t_entry)
#endif
        if (threadIdx.x == 0) s_next = C::kWaves;
        __syncthreads();
        // The bitmap's LDS-resident words are REQUESTED first (16 bytes per lane, ten quads into registers), then the wave resolves its first
        // slices — id, header, selection word: a chain of four dependent round trips — and only then are the quads stored to LDS: the two
        // used to run one after the other in every workgroup's prologue.
        static_assert(C::kBitWords % 4 == 0, "the bitmap is copied in quads");
        constexpr uint kFillPer = (C::kBitWords / 4 + C::kTpb - 1) / C::kTpb;
        uint4          fill[kFillPer];
        {
            const uint4* src = reinterpret_cast<const uint4*>(vbits);
#pragma unroll
            for (uint k = 0; k < kFillPer; ++k) fill[k] = src[min(threadIdx.x + k * C::kTpb, C::kBitWords / 4 - 1u)];
        }
        const T    pc   = mult(from_bits<T>(xbits), from_bits<T>(cbits));
        const uint lane = lane_id();
        const uint w    = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
        auto phys = [&](uint q) -> uint { return q < n_slices ? (sorder ? sorder[q] : q) : n_slices; };
        auto take_slice = [&]() -> uint {
            uint ord = 0;
            if (lane == 0) ord = atomicAdd(&s_next, 1u);
            return phys(blockIdx.x + __builtin_amdgcn_readfirstlane(ord) * gridDim.x);
        };
        auto sel_word = [&](uint rowinfo, uint len) -> uint { return (always || len == 0) ? 0u : selbits[(rowinfo & ~C::kPartial) >> 5]; };
        const unsigned char* lds = reinterpret_cast<const unsigned char*>(s_bits);
        auto                 hit = [&](uint code, uint word) -> uint { return (word >> (code & 31u)) & 1u; };
        auto                 lds_word = [&](uint code) -> uint { return *reinterpret_cast<const uint*>(lds + (code >> 5)); };

        uint       sA    = phys(blockIdx.x + w * gridDim.x);
        const bool haveA = sA < n_slices;
        uint       rowA = 0, lenA = 0, baseA = 0, sB = n_slices, rowB = 0, lenB = 0, baseB = 0, wordA = 0, sC = n_slices;
        bool       moreB = false;
        if (haveA) {
            rowA = vrow[sA * kWave + lane], lenA = vlen[sA * kWave + lane], baseA = off[sA];
            sB    = take_slice();
            moreB = sB < n_slices;
            if (moreB) {
                rowB  = vrow[sB * kWave + lane];
                lenB  = vlen[sB * kWave + lane];
                baseB = off[sB];
            }
            wordA = sel_word(rowA, lenA);
            sC    = moreB ? take_slice() : n_slices;// (resolved a slice ahead of its header: see sell_mxv_body)
        }
        {
            uint4* dst = reinterpret_cast<uint4*>(s_bits);
#pragma unroll
            for (uint k = 0; k < kFillPer; ++k) {
                const uint q = threadIdx.x + k * C::kTpb;
                if (q < C::kBitWords / 4) dst[q] = fill[k];
            }
        }
        __syncthreads();
        if (!haveA) return;
#ifdef SPLAHIP_SELL_TIMING
        unsigned long long t_a = 0, t_b = 0, t_c = 0, t_d = 0, acc0 = 0, acc1 = 0, acc2 = 0, nsl = 0, t_begin = 0, n_live = 0, n_rounds = 0, n_width = 0, n_lanes = 0;
        SELL_NOW(t_begin)
#endif
        while (true) {
            SELL_NOW(t_a)
            // ---- slice A: who is selected, how far the longest selected lane goes
            const uint row      = rowA & ~C::kPartial;
            const bool selected = lenA != 0 && (always || ((wordA >> (rowA & 31u)) & 1u) != 0);
            const uint mylen    = selected ? lenA : 0u;
            const unsigned long long live = __ballot(selected);
            const uint base  = __builtin_amdgcn_readfirstlane(baseA);
            const uint width = live ? __builtin_amdgcn_readlane(mylen, __ffsll((long long) live) - 1) : 0u;// lanes are sorted by length
            // C's header, D's id and B's selection word travel with A's entries
            uint       rowC = 0, lenC = 0, baseC = 0;
            const bool moreC = moreB && sC < n_slices;
            if (moreC) {
                rowC  = vrow[sC * kWave + lane];
                lenC  = vlen[sC * kWave + lane];
                baseC = off[sC];
            }
            uint sD = n_slices;
            if (moreC) sD = take_slice();
            uint wordB = 0;
            if (moreB) wordB = sel_word(rowB, lenB);

            SELL_NOW(t_b)
            if (live != 0ull) {
#ifdef SPLAHIP_SELL_TIMING
                n_live += 1, n_rounds += (width + 3) / 4, n_width += width, n_lanes += __popcll(live);
#endif
                const uint voff = selected ? lane * 4u : 0x80000000u;
                // position t of this lane = t_first (uniform) + t_imm.  The uniform part goes into the DESCRIPTOR — base moved to position t_first, the
                // record count cut to what is left of the slice (scalar arithmetic) — not into the load's scalar offset, which the range check of a raw
                // buffer load need not cover: a request past the slice's width then returns 0 by the checked vector offset + immediate alone
                auto request = [&](uint t_first, uint t_imm) -> uint {
                    const uint                   left = width > t_first ? (width - t_first) * (kWave * 4u) : 0u;
                    const __amdgpu_buffer_rsrc_t rs   = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint*>(Aj2r + base + (size_t) t_first * kWave), 0, left, 0x00020000);
                    return __builtin_amdgcn_raw_buffer_load_b32(rs, voff + t_imm * (kWave * 4u), 0, 2 /* nt */);
                };
                uint hits = 0;
                if (width <= 16u) {
                    uint code[16], word[16];
#pragma unroll
                    for (uint g = 0; g < 4; ++g)
                        if (4 * g < width) {
#pragma unroll
                            for (uint i = 0; i < 4; ++i) code[4 * g + i] = request(0, 4 * g + i);
                        }
#pragma unroll
                    for (uint g = 0; g < 4; ++g)
                        if (4 * g < width) {
#pragma unroll
                            for (uint i = 0; i < 4; ++i) word[4 * g + i] = lds_word(code[4 * g + i]);
                        }
#pragma unroll
                    for (uint g = 0; g < 4; ++g)
                        if (4 * g < width) {
#pragma unroll
                            for (uint i = 0; i < 4; ++i) hits += hit(code[4 * g + i], word[4 * g + i]);
                        }
                } else {
                    constexpr uint D = 32, NB = 8;// entry t lives in window slot t % D
                    uint           q[D];
#pragma unroll
                    for (uint u = 0; u < D; ++u) q[u] = request(u & ~(NB - 1u), u & (NB - 1u));
                    uint e = 0;
                    for (; e + D < width; e += D) {
#pragma unroll
                        for (uint b = 0; b < D; b += NB) {
                            uint addr[NB], sh[NB], word[NB];
#pragma unroll
                            for (uint i = 0; i < NB; ++i) {
                                addr[i] = q[b + i] >> 5;
                                sh[i]   = q[b + i] & 31u;
                            }
#pragma unroll
                            for (uint i = 0; i < NB; ++i) asm volatile("" : "+v"(addr[i]), "+v"(sh[i])::"memory");// (pins the lines above ahead of the requests)
#pragma unroll
                            for (uint i = 0; i < NB; ++i) q[b + i] = request(e + D + b, i);
#pragma unroll
                            for (uint i = 0; i < NB; ++i) word[i] = *reinterpret_cast<const uint*>(lds + addr[i]);
#pragma unroll
                            for (uint i = 0; i < NB; ++i) hits += (word[i] >> sh[i]) & 1u;
                        }
                    }
#pragma unroll
                    for (uint b = 0; b < D; b += NB) {
                        uint word[NB];
#pragma unroll
                        for (uint i = 0; i < NB; ++i) word[i] = lds_word(q[b + i]);
#pragma unroll
                        for (uint i = 0; i < NB; ++i) hits += hit(q[b + i], word[i]);
                    }
                }
                SELL_NOW(t_c)
                if ((__builtin_amdgcn_readfirstlane(rowA) & C::kPartial) != 0) {// a row-exclusive slice: 64 lanes of ONE row
                    const uint total = wave_sum(hits);
                    const T    t     = fold_repeated(ident, pc, total, add);
                    if (lane == 0 && total != 0 && to_bits(t) != to_bits(ident)) atomic_fold(r + row, t, functor_op(add));
                } else if (selected && hits != 0) {
                    const T t = fold_repeated(init, pc, hits, add);
                    if (to_bits(t) != to_bits(init)) r[row] = t;// (the prep pass wrote init to every row)
                }
            }
#ifdef SPLAHIP_SELL_TIMING
            SELL_NOW(t_d)
            if (live == 0ull) t_c = t_b;
            acc0 += t_b - t_a, acc1 += t_c - t_b, acc2 += t_d - t_c, nsl += 1;
            if (!moreB && lane == 0) {
                const uint gw = blockIdx.x * C::kWaves + w;
                if (gw < 4096) {
                    g_sell_t[0][gw] = acc0, g_sell_t[1][gw] = acc1, g_sell_t[2][gw] = acc2, g_sell_t[3][gw] = nsl;
                    g_sell_t[4][gw] = t_d - t_begin;
                    g_sell_t[6][gw] = n_live, g_sell_t[7][gw] = n_rounds, g_sell_t[8][gw] = n_width, g_sell_t[9][gw] = n_lanes;
                    g_sell_t[10][gw] = t_begin - t_entry;
                }
            }
#endif
            if (!moreB) break;
            sA = sB, rowA = rowB, lenA = lenB, baseA = baseB, wordA = wordB;
            sB = sC, moreB = moreC, rowB = rowC, lenB = lenC, baseB = baseC;
            sC = sD;
        }
    }

    // value form only (semirings / types the bit form cannot serve, or SPLAHIP_SELL_BITS=0)
    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(SellCfg::kTpb) void sell_mxv_kernel(uint n_rows, const uint* __restrict__ Ap, uint n_slices, const uint* __restrict__ vrow,
                                                                     const uint* __restrict__ vlen, const uint* __restrict__ vslot, const uint* __restrict__ off,
                                                                     const uint* __restrict__ Aj2, const uint* __restrict__ Ax2, const T* __restrict__ hot_values,
                                                                     uint n_hot, const T* __restrict__ mask, const T* __restrict__ v, T* __restrict__ r,
                                                                     T* __restrict__ partial, Mult mult, Add add, int op_select, T init,
                                                                     const uint* __restrict__ sorder) {
        __shared__ T    s_hot[SellCfg::kHot];
        __shared__ uint s_next;
        sell_mxv_body<T, Mult, Add, false>(s_hot, &s_next, n_rows, Ap, n_slices, vrow, vlen, vslot, off, Aj2, Ax2, hot_values, n_hot, mask, v, r, partial, mult, add,
                                           op_select, init, false, 0u, nullptr, nullptr, nullptr, sorder);
    }

    // the pull form of vxm (identity: every index of v stored) over the sliced-ELL copy of the TRANSPOSED matrix
    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(SellCfg::kTpb) void sell_vxm_kernel(uint n_rows, const uint* __restrict__ Ap, uint n_slices, const uint* __restrict__ vrow,
                                                                     const uint* __restrict__ vlen, const uint* __restrict__ vslot, const uint* __restrict__ off,
                                                                     const uint* __restrict__ Aj2, const uint* __restrict__ Ax2, const T* __restrict__ hot_values,
                                                                     uint n_hot, const T* __restrict__ mask, const T* __restrict__ v, T* __restrict__ r,
                                                                     uint* __restrict__ touched, T* __restrict__ partial, Mult mult, Add add, int op_select,
                                                                     const uint* __restrict__ sorder) {
        __shared__ T    s_hot[SellCfg::kHot];
        __shared__ uint s_next;
        sell_mxv_body<T, Mult, Add, false, true>(s_hot, &s_next, n_rows, Ap, n_slices, vrow, vlen, vslot, off, Aj2, Ax2, hot_values, n_hot, mask, v, r, partial,
                                                 mult, add, op_select, T(0), false, 0u, nullptr, nullptr, nullptr, sorder, touched);
    }
    template<typename T, class Add, bool VXM = false>
    __device__ __forceinline__ void sell_combine_block(uint block, uint n_lrows, const uint* __restrict__ lr_row, const uint* __restrict__ lr_first,
                                                       const T* __restrict__ partial, const T* __restrict__ mask, T* __restrict__ r, Add add, int op_select,
                                                       T init, uint* __restrict__ touched = nullptr);// (defined below, next to its mxv user)
    template<typename T, class Add>
    __global__ __launch_bounds__(256) void sell_vxm_combine_kernel(uint n_lrows, const uint* __restrict__ lr_row, const uint* __restrict__ lr_first,
                                                                   const T* __restrict__ partial, const T* __restrict__ mask, T* __restrict__ r,
                                                                   uint* __restrict__ touched, Add add, int op_select) {
        sell_combine_block<T, Add, true>(blockIdx.x, n_lrows, lr_row, lr_first, partial, mask, r, add, op_select, T(0), touched);
    }

    // both forms in one launch: wave 0 of every workgroup reads sell_vprep's statistics (12-48 KB; all 4096 waves doing so cost
    // 0.5 M L2 requests), the workgroup then walks its slices in the form they choose — no host round trip and no second launch.
    // Workgroup 0 leaves the decision behind the statistics for sell_cold.  force: 0 = decide, 1 = bit form.
    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(SellCfg::kTpb) void sell_mxv_dual_kernel(uint n_rows, const uint* __restrict__ Ap, uint n_slices, const uint* __restrict__ vrow,
                                                                          const uint* __restrict__ vlen, const uint* __restrict__ vslot,
                                                                          const uint* __restrict__ off, const uint* __restrict__ Aj2, const uint* __restrict__ Aj2r,
                                                                          const uint* __restrict__ Ax2, const T* __restrict__ hot_values, uint n_hot,
                                                                          const T* __restrict__ mask, const T* __restrict__ v, T* __restrict__ r,
                                                                          T* __restrict__ partial, Mult mult, Add add, int op_select, T init, uint* __restrict__ stats,
                                                                          uint n_cols, int force, const uint* __restrict__ vbits, const uint* __restrict__ vranked,
                                                                          const uint* __restrict__ selbits, const uint* __restrict__ sorder, T ident, uint xbits) {
        using C = SellCfg;
        static_assert(C::kBitWords <= C::kHot, "the two forms share one LDS table");
        __shared__ T    s_hot[C::kHot];
        __shared__ uint s_next;
        __shared__ uint s_dec[3];
        if (threadIdx.x < kWave) {
            bool       is_const;
            uint       cbits;
            const bool bits_form = sell_bits_form(stats, n_cols, force, is_const, cbits);
            if (threadIdx.x == 0) {
                s_dec[0] = bits_form ? 1u : 0u;
                s_dec[1] = is_const ? 1u : 0u;
                s_dec[2] = cbits;
                if (blockIdx.x == 0) {
                    stats[C::kDecision]     = s_dec[0];
                    stats[C::kDecision + 1] = s_dec[1];
                    stats[C::kDecision + 2] = s_dec[2];
                }
            }
        }
        __syncthreads();
        if (s_dec[0] != 0u && s_dec[1] != 0u && Ax2 == nullptr)// all products are one value: the walk only counts set bits
            sell_frontier_body<T, Mult, Add>(reinterpret_cast<uint*>(s_hot), &s_next, n_slices, vrow, vlen, off, Aj2r, vbits, selbits, sorder, r, mult, add, op_select, init,
                                             ident, s_dec[2], xbits);
        else if (s_dec[0] != 0u && Ax2 == nullptr)// structure-only: no value stream, so a group is twice as many entries for the same loads in flight
            sell_mxv_body<T, Mult, Add, true, false, 2 * C::kU>(s_hot, &s_next, n_rows, Ap, n_slices, vrow, vlen, vslot, off, Aj2r, Ax2, reinterpret_cast<const T*>(vbits),
                                                                C::kBitWords, mask, v, r, partial, mult, add, op_select, init, s_dec[1] != 0u, s_dec[2], vbits, vranked,
                                                                selbits, sorder, nullptr, ident, xbits);
        else if (s_dec[0] != 0u)
            sell_mxv_body<T, Mult, Add, true>(s_hot, &s_next, n_rows, Ap, n_slices, vrow, vlen, vslot, off, Aj2r, Ax2, reinterpret_cast<const T*>(vbits), C::kBitWords,
                                              mask, v, r, partial, mult, add, op_select, init, s_dec[1] != 0u, s_dec[2], vbits, vranked, selbits, sorder, nullptr, ident, xbits);
        else
            sell_mxv_body<T, Mult, Add, false>(s_hot, &s_next, n_rows, Ap, n_slices, vrow, vlen, vslot, off, Aj2, Ax2, hot_values, n_hot, mask, v, r, partial, mult,
                                               add, op_select, init, false, 0u, nullptr, nullptr, nullptr, sorder, nullptr, T(), xbits);
    }

    // BITS form, entries of columns beyond the LDS-resident ranks (ascending rank: the bitmap reads are sequential): a set bit on a
    // selected row folds its product into r[row] with the integer atomic of op_add.  Runs after the main kernel and sell_combine.
    template<typename T, class Mult>
    __device__ __forceinline__ void sell_cold_block(uint block, uint n_cold, const uint* __restrict__ cold_rank, const unsigned long long* __restrict__ cold_entry,
                                                    const uint* __restrict__ stats, const uint* __restrict__ vbits, const uint* __restrict__ vranked,
                                                    const T* __restrict__ mask, T* __restrict__ r, Mult mult, int op_add, int op_select) {
        const bool is_const = stats[SellCfg::kDecision + 1] != 0u;
        const uint cbits    = stats[SellCfg::kDecision + 2];
        const uint k        = block * 256 + threadIdx.x;
        if (k >= n_cold) return;
        const uint rank = cold_rank[k];
        if (((vbits[(rank + 1u) >> 5] >> ((rank + 1u) & 31u)) & 1u) == 0) return;
        const unsigned long long e = cold_entry[k];
        const uint row = (uint) (e >> 32);
        if (op_select != SPLAHIP_ALWAYS && !eval_sel(op_select, mask[row])) return;
        const T    x = from_bits<T>(is_const ? cbits : vranked[rank]);
        const T    p = mult(from_bits<T>((uint) e), x);
        // MIN / MAX / BOR only ever move r[row] one way: a product that cannot move the value read now cannot move the final one either,
        // and most cold products of a shortest-path step do not beat what the LDS-resident columns already gave the row — a plain
        // load instead of a read-modify-write in the L2
        if (op_add == SPLAHIP_MIN || op_add == SPLAHIP_MAX || op_add == SPLAHIP_BOR) {
            const T cur = from_bits<T>(__hip_atomic_load(reinterpret_cast<const uint*>(r + row), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (op_add == SPLAHIP_MIN && !(p < cur)) return;
            if (op_add == SPLAHIP_MAX && !(cur < p)) return;
            if (op_add == SPLAHIP_BOR && (to_bits(cur) | to_bits(p)) == to_bits(cur)) return;
        }
        atomic_fold(r + row, p, op_add);
    }

    // rows longer than kC: r[row] = init (+) partial[first] (+) partial[first + 1] ... in chunk order.  One wave per row: lane l
    // folds a contiguous run of the row's partials, then an in-order wave reduction joins the runs (associative op_add).
    template<typename T, class Add, bool VXM>
    __device__ __forceinline__ void sell_combine_block(uint block, uint n_lrows, const uint* __restrict__ lr_row, const uint* __restrict__ lr_first,
                                                       const T* __restrict__ partial, const T* __restrict__ mask, T* __restrict__ r, Add add_raw, int op_select,
                                                       T init, uint* __restrict__ touched) {
        const Reassoc<Add> add(add_raw);
        const uint i = block * (256 / kWave) + threadIdx.x / kWave;
        if (i >= n_lrows) return;
        const uint row = lr_row[i];
        if (op_select != SPLAHIP_ALWAYS && !eval_sel(op_select, mask[row])) return;// the main kernel's row pass wrote init (vxm: touched = 0)
        const uint k0 = lr_first[i], k1 = lr_first[i + 1];
        const uint c  = (k1 - k0 + kWave - 1) / kWave;
        const uint ks = k0 + lane_id() * c, ke = min(k1, ks + c);
        T          acc   = init;
        bool       valid = false;
        for (uint k = ks; k < ke; ++k) {
            acc   = valid ? add(acc, partial[k]) : partial[k];
            valid = true;
        }
        wave_reduce_ordered(acc, valid, add);
        if (lane_id() == 0) {
            if constexpr (VXM) {// every lane of a long row holds an entry, and every entry is a product: the row is touched
                r[row]       = acc;
                touched[row] = 1;
            } else {
                r[row] = valid ? add(init, acc) : init;
            }
        }
    }

    // value-form calls of semirings the bit form cannot serve: the long rows' partials
    template<typename T, class Add>
    __global__ __launch_bounds__(256) void sell_combine_kernel(uint n_lrows, const uint* __restrict__ lr_row, const uint* __restrict__ lr_first,
                                                               const T* __restrict__ partial, const T* __restrict__ mask, T* __restrict__ r, Add add,
                                                               int op_select, T init) {
        sell_combine_block<T, Add>(blockIdx.x, n_lrows, lr_row, lr_first, partial, mask, r, add, op_select, init);
    }
    // the launch behind sell_mxv_dual: the cold list when the workgroups chose the bit form (blocks [0, cold_blocks)), else the long rows'
    // partials (the blocks behind them) — one launch whichever form ran
    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(256) void sell_post_kernel(uint cold_blocks, uint n_cold, const uint* __restrict__ cold_rank,
                                                            const unsigned long long* __restrict__ cold_entry, const uint* __restrict__ stats,
                                                            const uint* __restrict__ vbits, const uint* __restrict__ vranked, uint n_lrows,
                                                            const uint* __restrict__ lr_row, const uint* __restrict__ lr_first, const T* __restrict__ partial,
                                                            const T* __restrict__ mask, T* __restrict__ r, Mult mult, Add add, int op_select, T init) {
        const bool bits_form = stats[SellCfg::kDecision] != 0u;
        if (blockIdx.x < cold_blocks) {
            if (bits_form) sell_cold_block<T, Mult>(blockIdx.x, n_cold, cold_rank, cold_entry, stats, vbits, vranked, mask, r, mult, functor_op(add), op_select);
        } else if (!bits_form) {
            sell_combine_block<T, Add>(blockIdx.x - cold_blocks, n_lrows, lr_row, lr_first, partial, mask, r, add, op_select, init);
        }
    }

}// namespace splahip
