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
//     group and the next slice's header, all with static load counts so hipcc can place exact s_waitcnt values.
//   * chunks of rows longer than C write partials; sell_combine folds them in chunk order (associative op_add only).
//   * rows without entries and unselected rows get `init` from a coalesced pass over the workgroup's share of rows.
//
// Semantics: reference Algo_mxv_masked_cpu<T>::execute, src/cpu/cpu_mxv.hpp:56-106.
#pragma once

#include "block_utils.hpp"
#include "ops.hpp"

#ifndef SPLAHIP_SELL_EXP
#define SPLAHIP_SELL_EXP 0
#endif

namespace splahip {

    struct SellCfg {
        static constexpr uint kC       = 256;  // max entries per virtual row (chunk)
        static constexpr uint kHot     = 40000;// LDS-resident values of the gathered vector (156.25 KB of the 160 KB)
        static constexpr uint kTpb     = 1024; // one workgroup per CU
        static constexpr uint kWaves   = kTpb / kWave;
        static constexpr uint kU       = 8;    // entries per lane and group
        static constexpr uint kHotFlag = 0x80000000u;
        static constexpr uint kPartial = 0x80000000u;// in vrow: this chunk belongs to a row longer than kC
    };

    template<typename T>
    __global__ __launch_bounds__(256) void sell_hot_pack_kernel(const uint* __restrict__ hot_cols, uint n_hot, const T* __restrict__ v, T* __restrict__ hot_values) {
        const uint s = blockIdx.x * 256 + threadIdx.x;
        if (s < n_hot) hot_values[s] = v[hot_cols[s]];
    }

    __device__ __forceinline__ uint sell_stream_load(const uint* p) { return __builtin_nontemporal_load(p); }

    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(SellCfg::kTpb) void sell_mxv_kernel(uint n_rows, const uint* __restrict__ Ap, uint n_slices,
                                                                     const uint* __restrict__ vrow, const uint* __restrict__ vlen,
                                                                     const uint* __restrict__ vslot, const uint* __restrict__ off,
                                                                     const uint* __restrict__ Aj2, const uint* __restrict__ Ax2,
                                                                     const T* __restrict__ hot_values, uint n_hot,
                                                                     const T* __restrict__ mask, const T* __restrict__ v,
                                                                     T* __restrict__ r, T* __restrict__ partial,
                                                                     Mult mult, Add add, int op_select, T init) {
        using C = SellCfg;
        __shared__ T s_hot[C::kHot];

        const bool always = op_select == SPLAHIP_ALWAYS;

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
        {
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
                        if (b[i + 1] == b[i] || (!always && !eval_sel(op_select, from_bits<T>(mm[i])))) r[row + i] = init;
                } else {
                    for (uint i = row; i < re; ++i)
                        if (Ap[i + 1] == Ap[i] || (!always && !eval_sel(op_select, mask[i]))) r[i] = init;
                }
            }
        }
#endif
        __syncthreads();

        // 2. slices.  They are sorted by width, so dealing them out boustrophedon (round k ascending, round k + 1 descending over the
        //    launch's waves) gives every wave nearly the same number of entries.
        const uint lane   = lane_id();
        const uint w      = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
        const uint stride = gridDim.x * C::kWaves;
        const uint gw     = blockIdx.x * C::kWaves + w;
        if (gw >= n_slices) return;
        uint round = 0;
        uint s     = gw;

        uint row_n  = vrow[s * kWave + lane];
        uint len_n  = vlen[s * kWave + lane];
        uint base_n = off[s];
        while (true) {
            const uint rowinfo = row_n;
            uint       mylen   = len_n;
            const uint base    = __builtin_amdgcn_readfirstlane(base_n);
            const uint s_cur   = s;
            ++round;
            s               = round * stride + ((round & 1u) ? stride - 1u - gw : gw);
            const bool more = s < n_slices;
            if (more) {// header of this wave's next slice: lands while the current one is processed
                row_n  = vrow[s * kWave + lane];
                len_n  = vlen[s * kWave + lane];
                base_n = off[s];
            }
            const uint row   = rowinfo & ~C::kPartial;
            const bool chunk = (rowinfo & C::kPartial) != 0;
            // width before selection: lane 0 holds the slice's longest chunk
            const uint width0 = __builtin_amdgcn_readfirstlane(mylen);

            // first group + the mask value travel together; rows of the group beyond the slice's width are not fetched
            const uint* pj = Aj2 + base + lane;
            const uint* px = Ax2 + base + lane;
            uint        jn[C::kU], xn[C::kU];
#pragma unroll
            for (uint u = 0; u < C::kU; ++u) {
                jn[u] = C::kHotFlag;
                xn[u] = 0;
                if (u < width0) {
                    jn[u] = sell_stream_load(pj + u * kWave);
                    xn[u] = sell_stream_load(px + u * kWave);
                }
            }
            bool selected = mylen > 0;
            if (!always && selected) selected = eval_sel(op_select, mask[row]);
            if (!selected) mylen = 0;
            const unsigned long long live = __ballot(selected);
            if (live != 0ull) {
                // chunks are sorted by length: the first selected lane holds the longest selected chunk
                const uint width = __builtin_amdgcn_readlane(mylen, __ffsll((long long) live) - 1);
                T          acc   = init;
                bool       has   = !chunk;// a row's only chunk folds from init; chunks of a long row fold from their first product
                for (uint e = 0; e < width; e += C::kU) {
                    uint jc[C::kU], xc[C::kU];
#pragma unroll
                    for (uint u = 0; u < C::kU; ++u) {
                        jc[u] = jn[u];
                        xc[u] = xn[u];
                    }
                    pj += C::kU * kWave;
                    px += C::kU * kWave;
                    // prefetch the next group before this group's gathers are issued (uniform conditions)
#pragma unroll
                    for (uint u = 0; u < C::kU; ++u) {
                        if (e + C::kU + u < width) {
                            jn[u] = sell_stream_load(pj + u * kWave);
                            xn[u] = sell_stream_load(px + u * kWave);
                        }
                    }
                    T vv[C::kU];
#pragma unroll
                    for (uint u = 0; u < C::kU; ++u) {
                        const bool take = e + u < mylen;
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
#pragma unroll
                    for (uint u = 0; u < C::kU; ++u) {
                        const bool take = e + u < mylen;
                        const T    p    = mult(from_bits<T>(xc[u]), vv[u]);
                        const T    both = add(acc, p);
                        acc             = take ? (has ? both : p) : acc;
                        has             = has || take;
                    }
                }
                if (selected) {
                    if (chunk) partial[vslot[s_cur * kWave + lane]] = acc;
                    else
                        r[row] = acc;
                }
            }
            if (!more) break;
        }
    }

    // rows longer than kC: r[row] = init (+) partial[first] (+) partial[first + 1] ... in chunk order.  One wave per row: lane l
    // folds a contiguous run of the row's partials, then an in-order wave reduction joins the runs (associative op_add).
    template<typename T, class Add>
    __global__ __launch_bounds__(256) void sell_combine_kernel(uint n_lrows, const uint* __restrict__ lr_row, const uint* __restrict__ lr_first,
                                                               const T* __restrict__ partial, const T* __restrict__ mask, T* __restrict__ r, Add add,
                                                               int op_select, T init) {
        const uint i = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
        if (i >= n_lrows) return;
        const uint row = lr_row[i];
        if (!eval_sel(op_select, mask[row])) return;// the main kernel's row pass wrote init
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
        if (lane_id() == 0) r[row] = valid ? add(init, acc) : init;
    }

}// namespace splahip
