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
    // ------------------------------------------------------------------------------------------------------------

    // last i in [0, nr) with s_ap[i] <= k, for s_ap[0] <= k < s_ap[nr]
    __device__ __forceinline__ uint find_row(const uint* s_ap, uint nr, uint k) {
        uint lo = 0, hi = nr;
        while (hi - lo > 1) {
            const uint mid = (lo + hi) >> 1;
            if (s_ap[mid] <= k) lo = mid;
            else
                hi = mid;
        }
        return lo;
    }

    __device__ __forceinline__ void wave_lds_fence() {
        // LDS operations of one wave are executed in order; this only stops the compiler from reordering across it
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // load up to 4 aligned quads per lane of the entry range [k0, k1) (a0 = k0 & ~3; quad qi covers a0+4qi..+3)
    struct QuadRegs {
        uint4 j[4];
        uint4 x[4];
        bool  act[4];
    };

    __device__ __forceinline__ void load_quad(const uint* __restrict__ Aj, const uint* __restrict__ Ax, uint kq, uint k0, uint k1,
                                              uint nnz, int vec_ok, uint4& jq, uint4& xq) {
        if (vec_ok && kq + 4 <= nnz) {
            jq = *reinterpret_cast<const uint4*>(Aj + kq);
            xq = *reinterpret_cast<const uint4*>(Ax + kq);
        } else {
            uint jj[4] = {0, 0, 0, 0}, xx[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint e = 0; e < 4; ++e) {
                const uint k = kq + e;
                if (k >= k0 && k < k1) {
                    jj[e] = Aj[k];
                    xx[e] = Ax[k];
                }
            }
            jq = make_uint4(jj[0], jj[1], jj[2], jj[3]);
            xq = make_uint4(xx[0], xx[1], xx[2], xx[3]);
        }
    }

    // Operand access of one matrix entry.  MXV: p = mult(Ax[k], v[Aj[k]]), always defined.
    // VXM (pull form over the transposed matrix): p = mult(v[i], x) and only defined when v stores entry i
    // (`present` bitmap; nullptr = every entry stored).
    template<bool VXM, typename T, class Mult>
    __device__ __forceinline__ T entry_product(const Mult& mult, uint xbits, T vj) {
        if constexpr (VXM) return mult(vj, from_bits<T>(xbits));
        else
            return mult(from_bits<T>(xbits), vj);
    }
    __device__ __forceinline__ bool bit_present(const uint* __restrict__ present, uint j) {
        return present == nullptr || ((present[j >> 5] >> (j & 31u)) & 1u) != 0;
    }

    // The wave-tile SpMV kernel.  VXM = false: masked mxv (every row written, fold starts at init).
    // VXM = true: the pull form of vxm on the transposed matrix: "rows" are output columns, entries whose v index is
    // not stored are skipped, the fold starts at the first product (no init) and touched[row] tells whether the
    // output entry exists (reference src/cpu/cpu_vxm.hpp:101-107).
    template<class C, typename T, class Mult, class Add, bool VXM>
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
                                                                    Mult mult, Add add, int op_select, T init,
                                                                    uint nnz, int vec_ok) {
        __shared__ uint          s_ap_all[C::kWaves][C::kRowsCap + 1];
        __shared__ T             s_prod_all[C::kWaves][C::kStage + 4];
        __shared__ unsigned char s_sel_all[C::kWaves][C::kRowsCap];
        __shared__ unsigned char s_ok_all[C::kWaves][VXM ? C::kStage + 4 : 4];

        const uint lane = lane_id();
        const uint w    = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
        const uint item = blockIdx.x * C::kWaves + w;
        if (item >= n_items) return;

        const uint e0  = item_row[item];
        const uint e1  = item_row[item + 1];
        const uint k0  = item_k0[item];
        const uint k1  = item_k0[item + 1];
        const uint row = e0 & ~C::kSegFlag;
        const uint a0  = k0 & ~3u;                        // 16-byte aligned origin of the entry slice
        const uint nq  = k1 > a0 ? (k1 - a0 + 3) >> 2 : 0;// quads in the slice, <= 256

        if (e0 & C::kSegFlag) {
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
            bool  act[4];
#pragma unroll
            for (uint u = 0; u < 4; ++u) {
                const uint qi = lane + u * kWave;
                act[u]        = qi < nq;
                if (act[u]) load_quad(Aj, Ax, a0 + 4 * qi, k0, k1, nnz, vec_ok, jq[u], xq[u]);
            }
            T    part  = init;
            bool valid = false;
#pragma unroll
            for (uint u = 0; u < 4; ++u) {
                if (act[u]) {
                    const uint kq    = a0 + 4 * (lane + u * kWave);
                    const uint jj[4] = {jq[u].x, jq[u].y, jq[u].z, jq[u].w};
                    const uint xx[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w};
                    T          vv[4];
                    bool       ok[4];
#pragma unroll
                    for (uint e = 0; e < 4; ++e) {
                        ok[e] = kq + e >= k0 && kq + e < k1;
                        if (VXM && ok[e]) ok[e] = bit_present(present, jj[e]);
                        vv[e] = ok[e] ? v[jj[e]] : T(0);
                    }
#pragma unroll
                    for (uint e = 0; e < 4; ++e) {
                        if (ok[e]) {
                            const T p = entry_product<VXM>(mult, xx[e], vv[e]);
                            part      = valid ? add(part, p) : p;
                            valid     = true;
                        }
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
        const uint     nr     = (e1 & ~C::kSegFlag) - row;
        uint*          s_ap   = s_ap_all[w];
        T*             s_prod = s_prod_all[w];
        unsigned char* s_sel  = s_sel_all[w];
        unsigned char* s_ok   = s_ok_all[w];

        // 1. row starts and mask -> LDS
        int any = 0, all = 1;
#pragma unroll
        for (uint u = 0; u < (C::kRowsCap + kWave) / kWave; ++u) {
            const uint i = lane + u * kWave;
            if (i <= nr) s_ap[i] = Ap[row + i];
            if (i < nr) {
                const bool s = eval_sel(op_select, mask[row + i]);
                s_sel[i]     = s ? 1 : 0;
                any |= s ? 1 : 0;
                all &= s ? 1 : 0;
            }
        }
        const bool any_sel = __ballot(any) != 0ull;
        const bool all_sel = __ballot(!all) == 0ull;
        if (!any_sel) {// nothing selected: no matrix entry is touched
#pragma unroll
            for (uint u = 0; u < C::kRowsCap / kWave; ++u) {
                const uint i = lane + u * kWave;
                if (i < nr) {
                    if constexpr (VXM) touched[row + i] = 0;
                    else
                        r[row + i] = init;
                }
            }
            return;
        }
        wave_lds_fence();

        // 2. stream the slice once, stage products in LDS
        {
            uint4 jq[4], xq[4];
            bool  act[4];
#pragma unroll
            for (uint u = 0; u < 4; ++u) {
                const uint qi = lane + u * kWave;
                act[u]        = qi < nq;
                if (act[u] && !all_sel) {// skip quads that only cover unselected rows
                    bool need = false;
#pragma unroll
                    for (uint e = 0; e < 4; ++e) {
                        const uint k = a0 + 4 * qi + e;
                        if (k >= k0 && k < k1) need |= s_sel[find_row(s_ap, nr, k)] != 0;
                    }
                    act[u] = need;
                }
                if (act[u]) load_quad(Aj, Ax, a0 + 4 * qi, k0, k1, nnz, vec_ok, jq[u], xq[u]);
            }
#pragma unroll
            for (uint u = 0; u < 4; ++u) {
                const uint qi = lane + u * kWave;
                if (act[u]) {
                    const uint kq    = a0 + 4 * qi;
                    const uint jj[4] = {jq[u].x, jq[u].y, jq[u].z, jq[u].w};
                    const uint xx[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w};
                    T          vv[4];
                    bool       ok[4];
#pragma unroll
                    for (uint e = 0; e < 4; ++e) {
                        ok[e] = kq + e >= k0 && kq + e < k1;
                        if (VXM && ok[e]) ok[e] = bit_present(present, jj[e]);
                        vv[e] = ok[e] ? v[jj[e]] : T(0);
                    }
#pragma unroll
                    for (uint e = 0; e < 4; ++e) {
                        if (ok[e]) s_prod[4 * qi + e] = entry_product<VXM>(mult, xx[e], vv[e]);
                        if constexpr (VXM) s_ok[4 * qi + e] = ok[e] ? 1 : 0;
                    }
                } else if (VXM && qi < nq) {
#pragma unroll
                    for (uint e = 0; e < 4; ++e) s_ok[4 * qi + e] = 0;
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
                if (s_sel[i]) {
                    const uint s = s_ap[i] - a0, e = s_ap[i + 1] - a0;
                    if (e - s <= C::kShortMax) {
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
            unsigned long long todo = __ballot(is_long);
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
    }

    // finish rows made of several segment items: strict in-order fold of their partials
    template<class C, typename T, class Add, bool VXM>
    __global__ __launch_bounds__(256) void spmv_combine_kernel(const uint* __restrict__ item_row, const uint* __restrict__ item_k0,
                                                               uint n_items, const uint* __restrict__ Ap,
                                                               const T* __restrict__ mask, T* __restrict__ r, uint* __restrict__ touched,
                                                               const T* __restrict__ partial, const uint* __restrict__ partial_ok,
                                                               Add add, int op_select, T init) {
        const uint i = blockIdx.x * 256 + threadIdx.x;
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
        const uint i = blockIdx.x * 256 + threadIdx.x;
        if (i >= n_rows) return;
        T sum = init;
        if (eval_sel(op_select, mask[i])) {
            const uint e = Ap[i + 1];
            for (uint k = Ap[i]; k < e; ++k) {
                sum = add(sum, mult(from_bits<T>(Ax[k]), v[Aj[k]]));
                if (EARLY && sum != init) break;
            }
        }
        r[i] = sum;
    }

}// namespace splahip
