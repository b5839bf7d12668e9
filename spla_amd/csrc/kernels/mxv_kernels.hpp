// mxv_kernels.hpp — device code of the masked SpMV (see mxv.hip for the contract, the plan and the dispatch).
//
// Work decomposition ("wave tiles"): the CSR nnz stream is cut into work items of at most ~1 Ki entries, each
// processed by ONE 64-lane wavefront with no workgroup barrier (the four waves of a workgroup share nothing, so
// waves of one CU sit in different phases and their HBM streams, gathers and LDS folds overlap):
//   * run item     — consecutive rows whose first entries fall into one kWin-entry window of the nnz stream
//                    (at most kRowsCap rows, every row <= kLongT entries): the slice of Aj/Ax is read once with
//                    16-byte loads, products go to a wave-private LDS stage, each row is folded out of LDS;
//   * segment item — a kSeg-entry piece of a row longer than kLongT: streamed, multiplied, reduced across the
//                    wave with an in-order segmented shuffle reduction; rows made of several segments are
//                    finished by the tiny `mxv_combine` kernel from per-item partials (deterministic, no atomics).
#pragma once

#include "block_utils.hpp"
#include "ops.hpp"

namespace splahip {

#ifdef SPLAHIP_PHASE_TIMING
    // DIAGNOSTIC BUILD ONLY (tools/exp/exp_phase.py): where a run item's wave spends its life.  Summed s_memtime deltas
    // between fully drained points: [0] header, [1] entry stream + Ap, [2] gathers, [3] LDS stage + fold + stores, [4] items
    static __device__ unsigned int g_phase[4][1 << 18];// per item, no atomics (contention would dominate)
#define SPLAHIP_STAMP(t)                                         \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    t = __builtin_amdgcn_s_memtime();
#else
#define SPLAHIP_STAMP(t)
#endif

    struct MxvCfg {
        static constexpr uint kWin      = 512; // nnz window that defines a run item
        static constexpr uint kLongT    = 508; // rows longer than this are cut into segment items
        static constexpr uint kSeg      = 1020;// entries per segment item (+3 alignment slack <= 256 quads)
        static constexpr uint kStage    = 1024;// LDS product stage per wave, entries (run item <= kWin-1+kLongT+3)
        static constexpr uint kRowsCap  = 128; // max rows per run item
        static constexpr uint kShortMax = 32;  // rows up to this length are folded by one lane
        static constexpr uint kTpb      = 256; // 4 waves per workgroup, one item per wave
        static constexpr uint kWaves    = kTpb / kWave;
        static constexpr uint kSegFlag  = 0x80000000u;
    };

    // ------------------------------------------------------------------------------------------------------------
    // plan construction: per row, how many items start at it
    // ------------------------------------------------------------------------------------------------------------
    template<class C>
    __device__ __forceinline__ uint row_item_count(const uint* __restrict__ Ap, uint r) {
        const uint s   = Ap[r];
        const uint len = Ap[r + 1] - s;
        if (len > C::kLongT) return (len + C::kSeg - 1) / C::kSeg;
        if (r == 0 || (r % C::kRowsCap) == 0) return 1;
        const uint ps = Ap[r - 1];
        if (s - ps > C::kLongT) return 1;// previous row is long: it ends a run
        return (s / C::kWin) != (ps / C::kWin) ? 1 : 0;
    }

    // pass 1: items per chunk of 1024 rows
    template<class C>
    __global__ __launch_bounds__(256) void plan_count_kernel(const uint* __restrict__ Ap, uint n_rows,
                                                             uint* __restrict__ chunk_items, uint* __restrict__ n_long) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            items = 0, longs = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            const uint r = first + u;
            if (r < n_rows) {
                items += row_item_count<C>(Ap, r);
                longs += (Ap[r + 1] - Ap[r]) > C::kLongT ? 1u : 0u;
            }
        }
        uint total;
        (void) block_exclusive_sum<256>(items, s_tmp, total);
        if (threadIdx.x == 0) chunk_items[blockIdx.x] = total;
        longs = wave_sum(longs);
        if (lane_id() == 0 && longs) atomicAdd(n_long, longs);
    }

    // pass 3: write the items in row order: item_row[i] = first row (| kSegFlag for a segment), item_k0[i] = first entry
    template<class C>
    __global__ __launch_bounds__(256) void plan_write_kernel(const uint* __restrict__ Ap, uint n_rows, uint nnz,
                                                             const uint* __restrict__ chunk_offset, uint n_chunks,
                                                             uint* __restrict__ item_row, uint* __restrict__ item_k0) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            cnt[4];
        uint            items = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            const uint r = first + u;
            cnt[u]       = r < n_rows ? row_item_count<C>(Ap, r) : 0;
            items += cnt[u];
        }
        uint total;
        uint pos = chunk_offset[blockIdx.x] + block_exclusive_sum<256>(items, s_tmp, total);
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            const uint r = first + u;
            if (cnt[u] == 0) continue;
            const uint s   = Ap[r];
            const uint len = Ap[r + 1] - s;
            if (len > C::kLongT) {
                for (uint g = 0; g < cnt[u]; ++g) {
                    item_row[pos] = r | C::kSegFlag;
                    item_k0[pos]  = s + g * C::kSeg;
                    ++pos;
                }
            } else {
                item_row[pos] = r;
                item_k0[pos]  = s;
                ++pos;
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const uint n_items = chunk_offset[n_chunks];
            item_row[n_items]  = n_rows;
            item_k0[n_items]   = nnz;
        }
    }

    // ------------------------------------------------------------------------------------------------------------
    // main kernel
    //
    // The kernels are issue-bound as much as memory-bound (profiles/r01_*): the per-entry code below is kept
    // branch-free.  A work item's slice [k0, k1) is read as whole aligned quads; the up to 3 entries before k0 / after
    // k1 that ride along belong to neighbouring items: they are real entries (valid column ids), their products are
    // computed and staged but never folded, so run items need no per-entry predicate at all.
    // ------------------------------------------------------------------------------------------------------------

    // last i in [0, nr) with s_ap[i] <= k, for s_ap[0] <= k < s_ap[nr]
    __device__ __forceinline__ uint find_row(const uint* s_ap, uint nr, uint k) {
        uint lo = 0, hi = nr;
        while (hi - lo > 1) {
            const uint mid = lo + ((hi - lo) >> 1);
            if (s_ap[mid] <= k) lo = mid;
            else
                hi = mid;
        }
        return lo;
    }

    __device__ __forceinline__ void wave_lds_fence() {
        // LDS operations of one wave execute in issue order, so a wave's own ds_write -> ds_read needs no hardware wait;
        // this is only a COMPILER barrier.  (An __builtin_amdgcn_fence here made hipcc emit s_waitcnt vmcnt(0), which
        // drained the prefetched global loads of the persistent kernel at every use.)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }

    using u64 = unsigned long long;

    // per-row selected flags of a run item (<= 128 rows) as two ballot masks, uniform across the wave
    struct SelMask {
        u64 m0, m1;
        __device__ __forceinline__ bool row(uint i) const { return ((i < 64 ? m0 >> i : m1 >> (i - 64)) & 1ull) != 0; }
        __device__ __forceinline__ bool any_in(uint a, uint b) const {// any selected row in [a, b], a <= b < 128
            u64 r0 = 0, r1 = 0;
            if (a < 64) r0 = (~0ull << a) & (~0ull >> (63u - min(b, 63u)));
            if (b >= 64) r1 = (~0ull << (a > 64 ? a - 64 : 0u)) & (~0ull >> (63u - (b - 64)));
            return ((m0 & r0) | (m1 & r1)) != 0;
        }
    };

    // One aligned quad of (column, value) pairs.  Only the last quad of the whole matrix can reach past the arrays.
    __device__ __forceinline__ void load_quad(const uint* __restrict__ Aj, const uint* __restrict__ Ax, uint kq, uint nnz, uint4& jq, uint4& xq) {
        if (kq + 4 <= nnz) {
#if !defined(SPLAHIP_STREAM_PLAIN)
            // non-temporal: the streamed CSR entries are read exactly once and should not evict the gathered vector's lines
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            const u4 a = __builtin_nontemporal_load(reinterpret_cast<const u4*>(Aj + kq));
            const u4 b = __builtin_nontemporal_load(reinterpret_cast<const u4*>(Ax + kq));
            jq = make_uint4(a.x, a.y, a.z, a.w);
            xq = make_uint4(b.x, b.y, b.z, b.w);
#else
            jq = *reinterpret_cast<const uint4*>(Aj + kq);
            xq = *reinterpret_cast<const uint4*>(Ax + kq);
#endif
        } else {
            uint jj[4] = {0, 0, 0, 0}, xx[4] = {0, 0, 0, 0};
            for (uint e = 0; e < 4; ++e)
                if (kq + e < nnz) {
                    jj[e] = Aj[kq + e];
                    xx[e] = Ax[kq + e];
                }
            jq = make_uint4(jj[0], jj[1], jj[2], jj[3]);
            xq = make_uint4(xx[0], xx[1], xx[2], xx[3]);
        }
    }

    // The gather v[j].  SPLAHIP_GATHER_MODE (experiments): 0 plain load, 1 non-temporal, 2 agent-scope (sc1, L1 bypass);
    // measured on MI355X: 1 is 2x slower, 2 is no faster than 0 (profiles/r01_notes.md)
#ifndef SPLAHIP_GATHER_MODE
#define SPLAHIP_GATHER_MODE 0
#endif
    template<typename T>
    __device__ __forceinline__ T gather_load(const T* p) {
#if SPLAHIP_GATHER_MODE == 1
        return __builtin_nontemporal_load(p);
#elif SPLAHIP_GATHER_MODE == 2
        return from_bits<T>(__hip_atomic_load(reinterpret_cast<const uint*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#else
        return *p;
#endif
    }

    // product of one entry.  MXV: mult(Ax[k], v[Aj[k]]).  VXM (pull form over the transposed matrix): mult(v[i], x)
    template<bool VXM, typename T, class Mult>
    __device__ __forceinline__ T entry_product(const Mult& mult, uint xbits, T vj) {
        if constexpr (VXM) return mult(vj, from_bits<T>(xbits));
        else
            return mult(from_bits<T>(xbits), vj);
    }

    template<typename T>
    struct alignas(16) Quad {
        T e[4];
    };

    // The wave-tile SpMV kernel.  VXM = false: masked mxv (every row written, fold starts at init).
    // VXM = true: the pull form of vxm on the transposed matrix: "rows" are output columns, entries whose v index is
    // not stored (`present` bitmap; nullptr = every index stored) are skipped, the fold starts at the first product (no
    // init) and touched[row] tells whether the output entry exists (reference src/cpu/cpu_vxm.hpp:101-107).
    // ORDERED (the reference-order mode, splahip_set_reference_order): every row of a run item is folded by ONE lane strictly left to right from `init`
    // (the reference's own loop, src/cpu/cpu_mxv.hpp:94-99) whatever its length, and segment items are left to spmv_longrow_ordered_kernel below.
    template<class C, typename T, class Mult, class Add, bool VXM, bool ORDERED = false>
    __global__ __launch_bounds__(C::kTpb) void spmv_wavetile_kernel(const uint* __restrict__ item_row,
                                                                    const uint* __restrict__ item_k0,
                                                                    uint n_items,
                                                                    const uint* __restrict__ Ap,
                                                                    const uint* __restrict__ Aj,
                                                                    const uint* __restrict__ Ax,
                                                                    const T* __restrict__ mask,
                                                                    const T* __restrict__ v,
                                                                    const uint* __restrict__ present,
                                                                    T* __restrict__ r,
                                                                    uint* __restrict__ touched,
                                                                    T* __restrict__ partial,
                                                                    uint* __restrict__ partial_ok,
                                                                    Mult mult, Add add_raw, int op_select, T init, uint nnz, const uint* __restrict__ gate = nullptr,
                                                                    const uint* __restrict__ item_list = nullptr, const uint* __restrict__ list_count = nullptr) {
        // gate (mxv.hip: a masked call over a plan that also holds the column-ordered blocks): word 3 of the mask probe's result says which kernels run — 1: these.
        // item_list (mxv_rows.hpp): the launch walks only the listed items (the plan's segment items: the rows too long for the row kernel)
        if (gate != nullptr && gate[3] == 0u) return;
        const Reassoc<Add> add(add_raw);// partials start at products: fp32 MIN / MAX must drop NaN products like the left fold from `init` does
        __shared__ uint    s_ap_all[C::kWaves][C::kRowsCap + 1];
        __shared__ Quad<T> s_prod_all[C::kWaves][C::kStage / 4 + 1];
        __shared__ uint    s_ok_all[C::kWaves][VXM ? C::kStage / 4 + 1 : 1];// VXM: 4 "product defined" bytes per quad

        const uint lane = lane_id();
        const uint w    = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
        uint       item = blockIdx.x * C::kWaves + w;
        if (item_list != nullptr) {
            if (item >= *list_count) return;
            item = item_list[item];
        }
        if (item >= n_items) return;
#ifdef SPLAHIP_PHASE_TIMING
        unsigned long long t_begin = 0, t_header = 0, t_stream = 0, t_gather = 0, t_end = 0;
#endif
        SPLAHIP_STAMP(t_begin)

        const uint e0  = item_row[item];
        const uint e1  = item_row[item + 1];
        const uint k0  = item_k0[item];
        const uint k1  = item_k0[item + 1];
        const uint row = e0 & ~C::kSegFlag;
        const uint a0  = k0 & ~3u;                        // 16-byte aligned origin of the entry slice
        const uint nq  = k1 > a0 ? (k1 - a0 + 3) >> 2 : 0;// quads in the slice, <= 256
        SPLAHIP_STAMP(t_header)

        if (e0 & C::kSegFlag) {
            if constexpr (ORDERED) return;// (the row is folded in order by spmv_longrow_ordered_kernel)
            // ---------------- segment item: a <= kSeg-entry piece of the long row `row`
            const uint rs   = Ap[row];
            const uint len  = Ap[row + 1] - rs;
            const uint nseg = (len + C::kSeg - 1) / C::kSeg;
            if (!eval_sel(op_select, mask[row])) {
                if (nseg == 1 && lane == 0) {// multi-segment rows are finished by the combine kernel
                    if constexpr (VXM) touched[row] = 0;
                    else
                        r[row] = init;
                }
                return;
            }
            uint4 jq[4], xq[4];
#pragma unroll
            for (uint u = 0; u < 4; ++u)
                if (lane + u * kWave < nq) load_quad(Aj, Ax, a0 + 4 * (lane + u * kWave), nnz, jq[u], xq[u]);
            T    part  = init;
            bool valid = false;
#pragma unroll
            for (uint u = 0; u < 4; ++u) {
                if (lane + u * kWave < nq) {
                    const uint kq    = a0 + 4 * (lane + u * kWave);
                    const uint jj[4] = {jq[u].x, jq[u].y, jq[u].z, jq[u].w};
                    const uint xx[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w};
                    T          vv[4];
#pragma unroll
                    for (uint e = 0; e < 4; ++e) vv[e] = gather_load(v + jj[e]);
#pragma unroll
                    for (uint e = 0; e < 4; ++e) {
                        bool take = kq + e >= k0 && kq + e < k1;// only the first / last quad of a segment are partial
                        if constexpr (VXM)
                            if (present != nullptr) take = take && ((present[jj[e] >> 5] >> (jj[e] & 31u)) & 1u) != 0;
                        const T p    = entry_product<VXM>(mult, xx[e], vv[e]);
                        const T both = add(part, p);
                        part         = take ? (valid ? both : p) : part;
                        valid        = valid || take;
                    }
                }
            }
            wave_reduce_ordered(part, valid, add);
            if (lane == 0) {
                if (nseg == 1) {
                    if constexpr (VXM) {
                        touched[row] = valid ? 1u : 0u;
                        if (valid) r[row] = part;
                    } else {
                        r[row] = valid ? add(init, part) : init;
                    }
                } else {
                    partial[item] = part;
                    if constexpr (VXM) partial_ok[item] = valid ? 1u : 0u;
                }
            }
            return;
        }

        // ---------------- run item: rows [row, row + nr), entries [k0, k1)
        const uint nr     = (e1 & ~C::kSegFlag) - row;
        uint*      s_ap   = s_ap_all[w];
        Quad<T>*   s_quad = s_prod_all[w];
        const T*   s_prod = reinterpret_cast<const T*>(s_quad);
        uint*      s_okq  = s_ok_all[w];
        const unsigned char* s_ok = reinterpret_cast<const unsigned char*>(s_okq);

        // 1. row starts -> LDS, selected flags -> two ballot masks.  op_select == ALWAYS needs no mask read at all, and then
        //    the entry stream does not depend on this step: its loads are issued first so both round trips overlap.
        const bool always = op_select == SPLAHIP_ALWAYS;
        uint4      jq[4], xq[4];
        bool       act[4];
        if (always) {
#pragma unroll
            for (uint u = 0; u < 4; ++u) {
                act[u] = lane + u * kWave < nq;
                if (act[u]) load_quad(Aj, Ax, a0 + 4 * (lane + u * kWave), nnz, jq[u], xq[u]);
            }
        }
        bool selA = false, selB = false;
        {
            if (lane <= nr) s_ap[lane] = Ap[row + lane];
            if (lane + kWave <= nr) s_ap[lane + kWave] = Ap[row + lane + kWave];
            if (lane == 0 && 2 * kWave <= nr) s_ap[2 * kWave] = Ap[row + 2 * kWave];
            if (lane < nr) selA = always || eval_sel(op_select, mask[row + lane]);
            if (lane + kWave < nr) selB = always || eval_sel(op_select, mask[row + lane + kWave]);
        }
        SelMask sel;
        sel.m0 = __ballot(selA);
        sel.m1 = __ballot(selB);
        const u64 live0 = nr >= 64 ? ~0ull : ((1ull << nr) - 1ull);
        const u64 live1 = nr <= 64 ? 0ull : (nr >= 128 ? ~0ull : ((1ull << (nr - 64)) - 1ull));
        if ((sel.m0 | sel.m1) == 0ull) {// nothing selected: no matrix entry is touched
            if (lane < nr) {
                if constexpr (VXM) touched[row + lane] = 0;
                else
                    r[row + lane] = init;
            }
            if (lane + kWave < nr) {
                if constexpr (VXM) touched[row + lane + kWave] = 0;
                else
                    r[row + lane + kWave] = init;
            }
            return;
        }
        const bool all_sel = sel.m0 == live0 && sel.m1 == live1;
        wave_lds_fence();

        // 2. stream the slice once (16-byte loads), gather, multiply, stage the products in LDS (16-byte stores)
        {
            if (!always) {
#pragma unroll
                for (uint u = 0; u < 4; ++u) {
                    const uint qi = lane + u * kWave;
                    act[u]        = qi < nq;
                    if (act[u] && !all_sel) {// skip quads that only cover unselected rows
                        const uint ka = max(a0 + 4 * qi, k0), kb = min(a0 + 4 * qi + 3, k1 - 1);
                        act[u]        = sel.any_in(find_row(s_ap, nr, ka), find_row(s_ap, nr, kb));
                    }
                    if (act[u]) load_quad(Aj, Ax, a0 + 4 * qi, nnz, jq[u], xq[u]);
                }
            }
            SPLAHIP_STAMP(t_stream)
#ifdef SPLAHIP_PHASE_TIMING
            T vv_all[4][4];
#pragma unroll
            for (uint u = 0; u < 4; ++u)
                if (act[u]) {
                    const uint jj[4] = {jq[u].x, jq[u].y, jq[u].z, jq[u].w};
#pragma unroll
                    for (uint e = 0; e < 4; ++e) vv_all[u][e] = gather_load(v + jj[e]);
                }
            SPLAHIP_STAMP(t_gather)
#endif
#pragma unroll
            for (uint u = 0; u < 4; ++u) {
                const uint qi = lane + u * kWave;
                if (act[u]) {
                    const uint jj[4] = {jq[u].x, jq[u].y, jq[u].z, jq[u].w};
                    const uint xx[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w};
                    T          vv[4];
#pragma unroll
#ifdef SPLAHIP_PHASE_TIMING
                    for (uint e = 0; e < 4; ++e) vv[e] = vv_all[u][e];
#else
                    for (uint e = 0; e < 4; ++e) vv[e] = gather_load(v + jj[e]);
#endif
                    Quad<T> p;
#pragma unroll
                    for (uint e = 0; e < 4; ++e) p.e[e] = entry_product<VXM>(mult, xx[e], vv[e]);
                    s_quad[qi] = p;
                    if constexpr (VXM) {
                        uint okw = 0x01010101u;
                        if (present != nullptr) {
                            okw = 0;
#pragma unroll
                            for (uint e = 0; e < 4; ++e) okw |= ((present[jj[e] >> 5] >> (jj[e] & 31u)) & 1u) << (8 * e);
                        }
                        s_okq[qi] = okw;
                    }
                } else if (VXM && qi < nq) {
                    s_okq[qi] = 0;
                }
            }
        }
        wave_lds_fence();

        // 3. fold every row out of LDS: a lane per short row (strict left fold in entry order = the reference's order);
        //    rows longer than kShortMax are folded by the whole wave (lane-contiguous odd-length chunks, in-order
        //    segmented shuffle reduction)
#pragma unroll
        for (uint u = 0; u < C::kRowsCap / kWave; ++u) {
            const uint i       = lane + u * kWave;
            bool       is_long = false;
            if (i < nr) {
                T    out = init;
                bool has = !VXM;
                if (sel.row(i)) {
                    const uint s = s_ap[i] - a0, e = s_ap[i + 1] - a0;
                    if constexpr (ORDERED && !VXM) {
                        with_static_op(add_raw, [&](auto op) {// (ordered folds start at init: the op itself, its switch outside the loop)
                            for (uint k = s; k < e; ++k) out = op(out, s_prod[k]);
                        });
                    } else if (ORDERED || e - s <= C::kShortMax) {
                        for (uint k = s; k < e; ++k) {
                            if constexpr (VXM) {
                                if (s_ok[k]) {
                                    out = has ? add(out, s_prod[k]) : s_prod[k];
                                    has = true;
                                }
                            } else {
                                out = add(out, s_prod[k]);
                            }
                        }
                    } else {
                        is_long = true;
                    }
                }
                if (!is_long) {
                    if constexpr (VXM) {
                        touched[row + i] = has ? 1u : 0u;
                        if (has) r[row + i] = out;
                    } else {
                        r[row + i] = out;
                    }
                }
            }
            u64 todo = __ballot(is_long);
            while (todo) {
                const uint l  = __ffsll((long long) todo) - 1;
                todo &= todo - 1;
                const uint li = l + u * kWave;
                const uint s = s_ap[li] - a0, e = s_ap[li + 1] - a0;
                const uint c  = ((e - s + kWave - 1) / kWave) | 1u;
                const uint ks = s + lane * c;
                const uint ke = min(ks + c, e);
                bool       valid = false;
                T          part  = init;
                for (uint k = ks; k < ke; ++k) {
                    if (!VXM || s_ok[k]) {
                        part  = valid ? add(part, s_prod[k]) : s_prod[k];
                        valid = true;
                    }
                }
                wave_reduce_ordered(part, valid, add);
                if (lane == 0) {
                    if constexpr (VXM) {
                        touched[row + li] = valid ? 1u : 0u;
                        if (valid) r[row + li] = part;
                    } else {
                        r[row + li] = valid ? add(init, part) : init;
                    }
                }
            }
        }
#ifdef SPLAHIP_PHASE_TIMING
        SPLAHIP_STAMP(t_end)
        if (lane == 0) {
            const uint slot = item & ((1u << 18) - 1u);
            g_phase[0][slot] = (uint) (t_header - t_begin);
            g_phase[1][slot] = (uint) (t_stream - t_header);
            g_phase[2][slot] = (uint) (t_gather - t_stream);
            g_phase[3][slot] = (uint) (t_end - t_gather);
        }
#endif
    }

    // The reference-order mode's kernel for the rows longer than kLongT (the plan's segment items): one wavefront per ROW — launched per item, the item that
    // is its row's first segment acts — streams the row 64 entries a step (coalesced Aj / Ax, gathers of v, the next step's loads in flight) and adds the
    // step's products ONE AFTER THE OTHER in entry order: lane l's product is read with v_readlane and added by every lane into the same running value, so
    // the result is the reference's sequential fold (src/cpu/cpu_mxv.hpp:94-99; vxm: src/cpu/cpu_vxm.hpp:101-113, the fold starts at the first product)
    // bit for bit.  Slow on purpose: a row of 10^5 entries is a chain of 10^5 dependent additions (~0.4 ms).
    template<class C, typename T, class Mult, class Add, bool VXM>
    __global__ __launch_bounds__(C::kTpb) void spmv_longrow_ordered_kernel(const uint* __restrict__ item_row, const uint* __restrict__ item_k0, uint n_items,
                                                                           const uint* __restrict__ Ap, const uint* __restrict__ Aj, const uint* __restrict__ Ax,
                                                                           const T* __restrict__ mask, const T* __restrict__ v, const uint* __restrict__ present,
                                                                           T* __restrict__ r, uint* __restrict__ touched, Mult mult, Add add, int op_select, T init, uint nnz) {
#pragma clang fp contract(off)
        const uint lane = lane_id();
        const uint item = blockIdx.x * C::kWaves + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
        if (item >= n_items) return;
        const uint e0 = item_row[item];
        if (!(e0 & C::kSegFlag)) return;
        const uint row = e0 & ~C::kSegFlag;
        const uint rs  = Ap[row];
        if (item_k0[item] != rs) return;// not the first segment of its row
        const uint re = Ap[row + 1];
        if (!eval_sel(op_select, mask[row])) {
            if (lane == 0) {
                if constexpr (VXM) touched[row] = 0;
                else
                    r[row] = init;
            }
            return;
        }
        // a step = 256 entries: a lane owns the aligned quad 4 (lane) .. 4 (lane) + 3 of the step (16-byte loads of Aj / Ax, four gathers); the streams of step
        // s + 2 and the gathers of step s + 1 are in flight while step s is added
        const uint a0 = rs & ~3u;
        struct Step {
            uint4 j, x;
            T     p[4];
            uint  ok;// bit e: entry 4 lane + e of the step belongs to the row (and, vxm over a sparse v, its index of v is stored)
        };
        auto load_jx = [&](uint kb, Step& st) {// kb: the step's first (aligned) entry
            const uint kq = kb + 4 * lane;
            st.ok         = 0;
            st.j = make_uint4(0, 0, 0, 0), st.x = make_uint4(0, 0, 0, 0);
            if (kq < re) {
                load_quad(Aj, Ax, kq, nnz, st.j, st.x);
#pragma unroll
                for (uint e = 0; e < 4; ++e) st.ok |= (kq + e >= rs && kq + e < re) ? (1u << e) : 0u;
            }
        };
        auto gather = [&](Step& st) {
            const uint jj[4] = {st.j.x, st.j.y, st.j.z, st.j.w};
            const uint xx[4] = {st.x.x, st.x.y, st.x.z, st.x.w};
#pragma unroll
            for (uint e = 0; e < 4; ++e) {
                const uint j = (st.ok >> e) & 1u ? jj[e] : jj[0];// (an entry of a neighbouring row rides along in the quad: its column is valid too, but need not be read)
                if constexpr (VXM)
                    if (present != nullptr && ((st.ok >> e) & 1u) && ((present[j >> 5] >> (j & 31u)) & 1u) == 0) st.ok &= ~(1u << e);
                st.p[e] = entry_product<VXM>(mult, xx[e], gather_load(v + ((st.ok >> e) & 1u ? j : 0u)));
            }
        };
        T    acc = init;
        bool has = !VXM;
        Step S0, S1, S2;
        load_jx(a0, S0);
        load_jx(a0 + 256, S1);
        gather(S0);
        for (uint kb = a0; kb < re; kb += 256) {
            load_jx(kb + 512, S2);
            gather(S1);
            const u64 all = __ballot(S0.ok == 15u);
            if (all == ~0ull && has) {// (vxm: the fold starts at the FIRST product — its step takes the slow loop below, every later full step this one)
                // a full step: the running value travels DOWN the lanes — in round t lane t takes lane t - 1's value (DPP wave_shr:1; lane 0: the value carried in)
                // and adds its own four products in order; every lane runs every round, lane t's value is final after round t (its left neighbour's was final a
                // round earlier) and recomputing it later gives the same word.  One DPP move and four dependent adds per four entries (the v_readlane form:
                // eight instructions and the SGPR hazards: the R-MAT 23 call 2.9 ms, bound by its longest row's chain).
                T a = acc;
                with_static_op(add, [&](auto op) {// (the op's switch once per step, not once per addition)
#pragma unroll 16
                    for (uint t = 0; t < kWave; ++t) {
                        const T prev = from_bits<T>((uint) __builtin_amdgcn_update_dpp((int) to_bits(acc), (int) to_bits(a), 0x138, 0xF, 0xF, false));
                        a            = op(op(op(op(prev, S0.p[0]), S0.p[1]), S0.p[2]), S0.p[3]);
                    }
                });
                acc = from_bits<T>((uint) __builtin_amdgcn_readlane((int) to_bits(a), (int) kWave - 1));
            } else {
                for (u64 m = __ballot(S0.ok != 0u); m; m &= m - 1) {
                    const uint l  = __ffsll((long long) m) - 1;
                    const uint ok = (uint) __builtin_amdgcn_readlane((int) S0.ok, l);
#pragma unroll
                    for (uint e = 0; e < 4; ++e) {
                        const T pl = from_bits<T>((uint) __builtin_amdgcn_readlane((int) to_bits(S0.p[e]), l));
                        if ((ok >> e) & 1u) {
                            acc = has ? add(acc, pl) : pl;
                            has = true;
                        }
                    }
                }
            }
            S0 = S1;
            S1 = S2;
        }
        if (lane == 0) {
            if constexpr (VXM) {
                touched[row] = has ? 1u : 0u;
                if (has) r[row] = acc;
            } else
                r[row] = acc;
        }
    }

    // finish rows made of several segment items: strict in-order fold of their partials (item i: the first segment of its row does the row)
    template<class C, typename T, class Add, bool VXM>
    __device__ __forceinline__ void spmv_combine_item(uint i, const uint* __restrict__ item_row, const uint* __restrict__ item_k0, uint n_items, const uint* __restrict__ Ap,
                                                      const T* __restrict__ mask, T* __restrict__ r, uint* __restrict__ touched, const T* __restrict__ partial,
                                                      const uint* __restrict__ partial_ok, const Reassoc<Add>& add, int op_select, T init) {
        if (i >= n_items) return;
        const uint e0 = item_row[i];
        if (!(e0 & C::kSegFlag)) return;
        const uint row = e0 & ~C::kSegFlag;
        const uint rs  = Ap[row];
        if (item_k0[i] != rs) return;// not the first segment of its row
        const uint len  = Ap[row + 1] - rs;
        const uint nseg = (len + C::kSeg - 1) / C::kSeg;
        if (nseg == 1) return;// written by the main kernel
        T    out = init;
        bool has = !VXM;
        if (eval_sel(op_select, mask[row])) {
            for (uint g = 0; g < nseg; ++g) {
                if (!VXM || partial_ok[i + g]) {
                    out = has ? add(out, partial[i + g]) : partial[i + g];
                    has = true;
                }
            }
        }
        if constexpr (VXM) {
            touched[row] = has ? 1u : 0u;
            if (has) r[row] = out;
        } else {
            r[row] = out;
        }
    }
    template<class C, typename T, class Add, bool VXM>
    __global__ __launch_bounds__(256) void spmv_combine_kernel(const uint* __restrict__ item_row, const uint* __restrict__ item_k0,
                                                               uint n_items, const uint* __restrict__ Ap,
                                                               const T* __restrict__ mask, T* __restrict__ r, uint* __restrict__ touched,
                                                               const T* __restrict__ partial, const uint* __restrict__ partial_ok,
                                                               Add add_raw, int op_select, T init, const uint* __restrict__ gate = nullptr) {
        if (gate != nullptr && gate[3] == 0u) return;
        const Reassoc<Add> add(add_raw);
        spmv_combine_item<C, T, Add, VXM>(blockIdx.x * 256 + threadIdx.x, item_row, item_k0, n_items, Ap, mask, r, touched, partial, partial_ok, add, op_select, init);
    }

    // ------------------------------------------------------------------------------------------------------------
    // sequential-per-row kernel: exact reference order for ANY op_add, with the reference's early exit
    // ------------------------------------------------------------------------------------------------------------
    template<typename T, class Mult, class Add, bool EARLY>
    __global__ __launch_bounds__(256) void mxv_rowseq_kernel(uint n_rows,
                                                             const uint* __restrict__ Ap,
                                                             const uint* __restrict__ Aj,
                                                             const uint* __restrict__ Ax,
                                                             const T* __restrict__ mask,
                                                             const T* __restrict__ v,
                                                             T* __restrict__ r,
                                                             Mult mult, Add add, int op_select, T init) {
        const uint i  = blockIdx.x * 256 + threadIdx.x;
        const bool in = i < n_rows;
        T          sum = init;
        if constexpr (!EARLY) {
            if (in && eval_sel(op_select, mask[i])) {
                const uint e = Ap[i + 1];
                for (uint k = Ap[i]; k < e; ++k) sum = add(sum, mult(from_bits<T>(Ax[k]), v[Aj[k]]));
            }
            if (in) r[i] = sum;
        } else {
            // early exit (src/cpu/cpu_mxv.hpp:96-98: the row stops behind the first product that moves the sum off init).  A lane walks the first kEarlyLane entries
            // of its own row (most rows stop within a few); what is left of a longer row is searched by the whole WAVE, 64 entries a step (round 6: until then a
            // hub row without an early hit was 10^5 dependent gathers on one lane).  Why the wave may look at 64 entries at once: until the row stops its sum
            // compares equal to init, so with S the sum the step starts from, entry l's candidate add(S, p_l) is what the sequential fold would hold behind entry l
            // — exactly, unless an earlier entry of the step changed the sum's BITS without moving it off init (a zero whose sign flips: MULT, DIV or SECOND as
            // op_add over zero products).  Such a step (rare) is folded one entry after the other down the lanes (DPP wave_shr:1), like the ordered kernels.
            constexpr uint kEarlyLane = 64;
            const bool     sel = in && eval_sel(op_select, mask[i]);
            uint           k   = sel ? Ap[i] : 0u;
            const uint     k1  = sel ? Ap[i + 1] : 0u;
            const uint     ke  = min(k1, k + kEarlyLane);
            bool           done = false;
            for (; k < ke; ++k) {
                sum = add(sum, mult(from_bits<T>(Ax[k]), v[Aj[k]]));
                if (sum != init) {
                    done = true;
                    break;
                }
            }
            u64 todo = __ballot(sel && !done && ke < k1);
            while (todo != 0ull) {
                const int  src = __ffsll((long long) todo) - 1;
                const uint a = __shfl(ke, src, kWave), b = __shfl(k1, src, kWave);
                T          S = from_bits<T>((uint) __shfl((int) to_bits(sum), src, kWave));// the sum the next step starts from (compares equal to init)
                bool       stopped = false;
                for (uint q = a; q < b && !stopped; q += kWave) {
                    const uint kk    = q + lane_id();
                    const bool valid = kk < b;
                    T          p     = init;
                    if (valid) p = mult(from_bits<T>(Ax[kk]), v[Aj[kk]]);
                    const T    c    = valid ? add(S, p) : S;
                    const bool off  = valid && c != init;
                    const bool flip = valid && !off && to_bits(c) != to_bits(S);
                    const u64  m = __ballot(off), f = __ballot(flip);
                    const u64  before = m != 0ull ? ((m & (0ull - m)) - 1ull) : ~0ull;// the lanes in front of the first entry that moves the sum
                    if ((f & before) != 0ull) {
                        // the sum's bits change inside the step before it stops: one entry after the other
                        T x = S;
                        for (uint t = 0; t < kWave; ++t) {
                            const T prev = from_bits<T>((uint) __builtin_amdgcn_update_dpp((int) to_bits(S), (int) to_bits(x), 0x138, 0xF, 0xF, false));
                            x            = valid ? add(prev, p) : prev;
                        }
                        const u64 m2 = __ballot(valid && x != init);
                        if (m2 != 0ull) {
                            S       = from_bits<T>((uint) __shfl((int) to_bits(x), __ffsll((long long) m2) - 1, kWave));
                            stopped = true;
                        } else
                            S = from_bits<T>((uint) __builtin_amdgcn_readlane((int) to_bits(x), (int) kWave - 1));
                    } else if (m != 0ull) {// the first entry that moves the sum: its lane's value is the row's result
                        S       = from_bits<T>((uint) __shfl((int) to_bits(c), __ffsll((long long) m) - 1, kWave));
                        stopped = true;
                    }
                }
                if ((int) lane_id() == src) sum = S;
                todo &= todo - 1ull;
            }
            if (in) r[i] = sum;
        }
    }

}// namespace splahip
