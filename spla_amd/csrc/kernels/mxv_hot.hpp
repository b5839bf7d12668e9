// mxv_hot.hpp — the persistent, software-pipelined form of the wave-tile masked SpMV with an LDS-resident cache of
// the hottest columns of the gathered vector.
//
// Why (measured on MI355X, profiles/r01_exp_gather.txt): with the CSR stream read once at 16 B/lane the kernel is NOT
// HBM-bound but bound by the divergent 4-byte gathers v[Aj[k]]: every L1-missing gather pulls a 128-B line through the
// CU's vector-memory pipeline (~1.6 clk per gathered lane even on L2 hits), while a random LDS read costs ~0.1 clk per
// lane.  Scale-free graphs concentrate the gathers on few columns (the ~28 K most referenced columns of R-MAT scale 22
// take 55-68 % of all entries), so those columns' values are copied into LDS once per launch and the matrix's column
// ids are re-encoded (bit 31 = "LDS slot") in a private copy of Aj kept with the plan.
//
// Structure: ONE 768-thread workgroup per CU (12 waves, 3 per SIMD) owns 90 KB of LDS for the hot values plus a private
// ~5.6 KB stage + row-pointer ring per wave.  Each wave walks the plan's work items with a stride of the total wave count
// and keeps four items in flight (R: row slice, S: Aj/Ax quads, G: gathers, P: products + fold), each stage's loads being
// issued one iteration before they are consumed.  Results are bit-identical to spmv_wavetile_kernel (same items, same
// fold order; checked by tests/gpu_utils.py::hip_mxv).
//
// STATUS: EXPERIMENTAL, off by default (SPLAHIP_HOT=1).  Measured on MI355X (R-MAT scale 22, fp32): the hot cache halves
// the gather cost (the gap between real and perfectly coalesced column ids falls from ~250 us to ~95-125 us) but the
// persistent skeleton itself is slower than the non-persistent kernel (3-stage, 16 waves: 364 us vs 390 us total; this
// 4-stage version: 450 us): at <= 4 waves per SIMD the loop is issue- and latency-bound, hipcc spills (48 B/lane at 168
// VGPRs) and every scratch reload drains the in-flight prefetches with s_waitcnt vmcnt(0).  Kept for the next round
// (hand-placed waits / a leaner item loop), not used by the product path.
#pragma once

#include "mxv_kernels.hpp"

namespace splahip {

    struct HotCfg {
        static constexpr uint kHot      = 23000;// LDS-resident hot values (90 KB)
        static constexpr uint kTpb      = 768;  // one workgroup per CU, 12 waves (3 per SIMD: 170 VGPRs each)
        static constexpr uint kWaves    = kTpb / kWave;
        static constexpr uint kHotFlag  = 0x80000000u;
    };

    // item descriptor held in SGPR-like uniform registers
    struct ItemDesc {
        uint row, nr, k0, k1;
        bool seg, live;
    };

    template<class C>
    __device__ __forceinline__ ItemDesc load_desc(const uint* __restrict__ item_row, const uint* __restrict__ item_k0, uint item, uint n_items) {
        ItemDesc d;
        d.live = item < n_items;
        if (d.live) {
            const uint e0 = item_row[item], e1 = item_row[item + 1];
            d.k0  = item_k0[item];
            d.k1  = item_k0[item + 1];
            d.row = e0 & ~C::kSegFlag;
            d.seg = (e0 & C::kSegFlag) != 0;
            d.nr  = d.seg ? 1u : (e1 & ~C::kSegFlag) - d.row;
        } else {
            d.row = d.nr = d.k0 = d.k1 = 0;
            d.seg = false;
        }
        return d;
    }

    // per-item state that travels down the software pipeline (uniform across the wave)
    struct ItemState {
        ItemDesc d;
        SelMask  sel;
        bool     all_sel;
        uint     slot;// LDS ring slot of the row-pointer slice
    };

    template<class C, typename T, class Mult, class Add>
    __global__ __launch_bounds__(HotCfg::kTpb) void spmv_hot_kernel(const uint* __restrict__ item_row,
                                                                    const uint* __restrict__ item_k0,
                                                                    uint n_items,
                                                                    const uint* __restrict__ Ap,
                                                                    const uint* __restrict__ Aj_enc,
                                                                    const uint* __restrict__ Ax,
                                                                    const T* __restrict__ mask,
                                                                    const T* __restrict__ v,
                                                                    const T* __restrict__ hot_values, uint n_hot,
                                                                    T* __restrict__ r,
                                                                    T* __restrict__ partial,
                                                                    Mult mult, Add add, int op_select, T init, uint nnz) {
        constexpr uint kSlice = C::kRowsCap + 1;
        constexpr uint kRing  = 3;
        __shared__ T       s_hot[HotCfg::kHot];
        __shared__ uint    s_ap_all[HotCfg::kWaves][kRing][kSlice];
        __shared__ Quad<T> s_prod_all[HotCfg::kWaves][C::kStage / 4 + 1];

        // ---- hot values -> LDS (contiguous, gathered once per launch by hot_gather_kernel)
        for (uint h = threadIdx.x; h < n_hot; h += HotCfg::kTpb) s_hot[h] = hot_values[h];
        __syncthreads();

        const uint lane   = lane_id();
        const uint w      = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
        const uint stride = gridDim.x * HotCfg::kWaves;
        uint       item   = blockIdx.x * HotCfg::kWaves + w;
        Quad<T>*   s_quad = s_prod_all[w];
        const T*   s_prod = reinterpret_cast<const T*>(s_quad);

        // Software pipeline, one item per stage; every stage's loads have a whole iteration to land, so the single
        // "everything has arrived" wait at the top of an iteration never stalls on a load issued in that iteration:
        //   R (i+3): row-pointer slice + mask values   -> registers  (issued at the end of iteration i)
        //   S (i+2): Aj/Ax quads                       -> registers  (issued at the end of iteration i, needs R(i+2) landed in LDS)
        //   G (i+1): gathers v[Aj] / LDS hot slots     -> registers  (issued at the start of iteration i from S(i+1))
        //   P (i)  : products -> LDS stage -> per-row fold -> store
        uint  apA = 0, apB = 0, apC = 0;
        T     mkA = T(0), mkB = T(0);
        uint4 js[4], xs[4];     // S stage: quads of item i+2 (in flight)
        bool  as_[4];
        uint4 xg[4];            // G stage: values of item i+1
        T     gg[4][4];         //          gathered operands of item i+1 (in flight)
        bool  ag[4];
        uint4 xp[4];            // P stage: values of item i
        T     gp[4][4];         //          gathered operands of item i
        bool  ap_[4];
#pragma unroll
        for (uint u = 0; u < 4; ++u) as_[u] = ag[u] = ap_[u] = false;

        auto make_state = [&](uint it, uint slot) {
            ItemState st;
            st.d       = load_desc<C>(item_row, item_k0, it, n_items);
            st.sel.m0  = st.sel.m1 = 0;
            st.all_sel = false;
            st.slot    = slot;
            return st;
        };
        auto issue_rows = [&](const ItemState& st) {// R: Ap slice + mask of an item -> registers
            if (!st.d.live) return;
            if (lane <= st.d.nr) apA = Ap[st.d.row + lane];
            if (lane + kWave <= st.d.nr) apB = Ap[st.d.row + lane + kWave];
            if (lane == 0 && 2 * kWave <= st.d.nr) apC = Ap[st.d.row + 2 * kWave];
            if (lane < st.d.nr) mkA = mask[st.d.row + lane];
            if (lane + kWave < st.d.nr) mkB = mask[st.d.row + lane + kWave];
        };
        auto land_rows = [&](ItemState& st) {// registers -> LDS ring slot + ballot masks
            if (!st.d.live) return;
            uint* s_ap = s_ap_all[w][st.slot];
            if (lane <= st.d.nr) s_ap[lane] = apA;
            if (lane + kWave <= st.d.nr) s_ap[lane + kWave] = apB;
            if (lane == 0 && 2 * kWave <= st.d.nr) s_ap[2 * kWave] = apC;
            st.sel.m0       = __ballot(lane < st.d.nr && eval_sel(op_select, mkA));
            st.sel.m1       = __ballot(lane + kWave < st.d.nr && eval_sel(op_select, mkB));
            const u64 live0 = st.d.nr >= 64 ? ~0ull : ((1ull << st.d.nr) - 1ull);
            const u64 live1 = st.d.nr <= 64 ? 0ull : (st.d.nr >= 128 ? ~0ull : ((1ull << (st.d.nr - 64)) - 1ull));
            st.all_sel      = st.sel.m0 == live0 && st.sel.m1 == live1;
            wave_lds_fence();
        };
        auto issue_stream = [&](const ItemState& st, uint4* J, uint4* X, bool* A) {// S
#pragma unroll
            for (uint u = 0; u < 4; ++u) A[u] = false;
            if (!st.d.live || (st.sel.m0 | st.sel.m1) == 0ull) return;
            const uint  a0   = st.d.k0 & ~3u;
            const uint  nq   = st.d.k1 > a0 ? (st.d.k1 - a0 + 3) >> 2 : 0;
            const uint* s_ap = s_ap_all[w][st.slot];
#pragma unroll
            for (uint u = 0; u < 4; ++u) {
                const uint qi = lane + u * kWave;
                A[u]          = qi < nq;
                if (A[u] && !st.all_sel && !st.d.seg) {// skip quads that only cover unselected rows
                    const uint ka = max(a0 + 4 * qi, st.d.k0), kb = min(a0 + 4 * qi + 3, st.d.k1 - 1);
                    A[u]          = st.sel.any_in(find_row(s_ap, st.d.nr, ka), find_row(s_ap, st.d.nr, kb));
                }
                if (A[u]) load_quad(Aj_enc, Ax, a0 + 4 * qi, nnz, J[u], X[u]);
            }
        };
        auto issue_gathers = [&](const uint4* J, const bool* A, T (*G)[4]) {// G: branch-free; a hot column reads its LDS slot and
#pragma unroll                                                          //    sends its global load to v[0] (one shared line)
            for (uint u = 0; u < 4; ++u) {
                if (A[u]) {
                    const uint jj[4] = {J[u].x, J[u].y, J[u].z, J[u].w};
#pragma unroll
                    for (uint e = 0; e < 4; ++e) {
                        const bool hot = (jj[e] & HotCfg::kHotFlag) != 0;
                        const uint idx = jj[e] & ~HotCfg::kHotFlag;
                        const T    a   = s_hot[hot ? idx : 0u];
                        const T    b   = gather_load(v + (hot ? 0u : idx));
                        G[u][e]        = hot ? a : b;
                    }
                }
            }
        };

        // ---- prologue: fill the pipeline
        ItemState P  = make_state(item, 0);
        ItemState G1 = make_state(item + stride, 1);
        ItemState S2 = make_state(item + 2 * stride, 2);
        issue_rows(P);
        land_rows(P);
        issue_stream(P, js, xs, as_);
        issue_rows(G1);
        land_rows(G1);
        issue_gathers(js, as_, gp);// P's gathers
#pragma unroll
        for (uint u = 0; u < 4; ++u) xp[u] = xs[u], ap_[u] = as_[u];
        issue_stream(G1, js, xs, as_);
        issue_rows(S2);
        land_rows(S2);
        issue_gathers(js, as_, gg);// G1's gathers
#pragma unroll
        for (uint u = 0; u < 4; ++u) xg[u] = xs[u], ag[u] = as_[u];
        issue_stream(S2, js, xs, as_);
        ItemState R3 = make_state(item + 3 * stride, 0);// slot 0 is still P's: landed only after P is done
        issue_rows(R3);

        while (P.d.live) {
            // ---- P: item i — products, fold, store (its operands were gathered during the previous iteration)
            const uint  a0   = P.d.k0 & ~3u;
            const uint* s_ap = s_ap_all[w][P.slot];
            if (P.d.seg) {
                const uint len  = s_ap[1] - s_ap[0];
                const uint nseg = (len + C::kSeg - 1) / C::kSeg;
                if (P.sel.m0 == 0ull) {
                    if (nseg == 1 && lane == 0) r[P.d.row] = init;
                } else {
                    T    part  = init;
                    bool valid = false;
#pragma unroll
                    for (uint u = 0; u < 4; ++u) {
                        if (ap_[u]) {
                            const uint kq    = a0 + 4 * (lane + u * kWave);
                            const uint xx[4] = {xp[u].x, xp[u].y, xp[u].z, xp[u].w};
#pragma unroll
                            for (uint e = 0; e < 4; ++e) {
                                const bool take = kq + e >= P.d.k0 && kq + e < P.d.k1;
                                const T    p    = mult(from_bits<T>(xx[e]), gp[u][e]);
                                const T    both = add(part, p);
                                part            = take ? (valid ? both : p) : part;
                                valid           = valid || take;
                            }
                        }
                    }
                    wave_reduce_ordered(part, valid, add);
                    if (lane == 0) {
                        if (nseg == 1) r[P.d.row] = valid ? add(init, part) : init;
                        else
                            partial[item] = part;
                    }
                }
            } else if ((P.sel.m0 | P.sel.m1) == 0ull) {
                if (lane < P.d.nr) r[P.d.row + lane] = init;
                if (lane + kWave < P.d.nr) r[P.d.row + lane + kWave] = init;
            } else {
#pragma unroll
                for (uint u = 0; u < 4; ++u) {
                    if (ap_[u]) {
                        const uint xx[4] = {xp[u].x, xp[u].y, xp[u].z, xp[u].w};
                        Quad<T>    p;
#pragma unroll
                        for (uint e = 0; e < 4; ++e) p.e[e] = mult(from_bits<T>(xx[e]), gp[u][e]);
                        s_quad[lane + u * kWave] = p;
                    }
                }
                wave_lds_fence();
#pragma unroll
                for (uint u = 0; u < C::kRowsCap / kWave; ++u) {
                    const uint i       = lane + u * kWave;
                    bool       is_long = false;
                    if (i < P.d.nr) {
                        T out = init;
                        if (P.sel.row(i)) {
                            const uint s = s_ap[i] - a0, e = s_ap[i + 1] - a0;
                            if (e - s <= C::kShortMax) {
                                for (uint k = s; k < e; ++k) out = add(out, s_prod[k]);
                            } else {
                                is_long = true;
                            }
                        }
                        if (!is_long) r[P.d.row + i] = out;
                    }
                    u64 todo = __ballot(is_long);
                    while (todo) {
                        const uint l = __ffsll((long long) todo) - 1;
                        todo &= todo - 1;
                        const uint li = l + u * kWave;
                        const uint s = s_ap[li] - a0, e = s_ap[li + 1] - a0;
                        const uint c  = ((e - s + kWave - 1) / kWave) | 1u;
                        const uint ks = s + lane * c;
                        const uint ke = min(ks + c, e);
                        bool       valid = false;
                        T          part  = init;
                        for (uint k = ks; k < ke; ++k) {
                            part  = valid ? add(part, s_prod[k]) : s_prod[k];
                            valid = true;
                        }
                        wave_reduce_ordered(part, valid, add);
                        if (lane == 0) r[P.d.row + li] = valid ? add(init, part) : init;
                    }
                }
                wave_lds_fence();// the stage is rewritten by the next item
            }

            // ---- advance the pipeline: everything issued during the previous iteration has had this whole P stage to land
            const uint freed = P.slot;
            item += stride;
            P = G1;
#pragma unroll
            for (uint u = 0; u < 4; ++u) {
                xp[u]  = xg[u];
                ap_[u] = ag[u];
#pragma unroll
                for (uint e = 0; e < 4; ++e) gp[u][e] = gg[u][e];
            }
            G1 = S2;
            issue_gathers(js, as_, gg);// G(i+1) from the landed quads of S
#pragma unroll
            for (uint u = 0; u < 4; ++u) xg[u] = xs[u], ag[u] = as_[u];
            S2      = R3;
            S2.slot = freed;
            land_rows(S2);                 // R -> LDS slot just freed by the finished item
            issue_stream(S2, js, xs, as_); // S(i+2)
            R3 = make_state(item + 3 * stride, 0);
            issue_rows(R3);                // R(i+3)
        }
    }

    // hot_values[s] = v[hot_cols[s]]: the once-per-launch gather of the hot columns
    template<typename T>
    __global__ __launch_bounds__(256) void hot_gather_kernel(const uint* __restrict__ hot_cols, uint n_hot, const T* __restrict__ v, T* __restrict__ hot_values) {
        const uint s = blockIdx.x * 256 + threadIdx.x;
        if (s < n_hot) hot_values[s] = v[hot_cols[s]];
    }

    // ---- hot-set construction (plan time) ------------------------------------------------------------------------------
    __global__ __launch_bounds__(256) void col_count_kernel(const uint* __restrict__ Aj, uint nnz, uint* __restrict__ count) {
        const uint k = blockIdx.x * 256 + threadIdx.x;
        if (k < nnz) atomicAdd(&count[Aj[k]], 1u);
    }
    __global__ __launch_bounds__(256) void col_key_kernel(const uint* __restrict__ count, uint n_cols, uint* __restrict__ key, uint* __restrict__ col) {
        const uint j = blockIdx.x * 256 + threadIdx.x;
        if (j < n_cols) {
            key[j] = ~count[j];// ascending sort of ~count = descending count; the stable sort keeps ascending column order among ties
            col[j] = j;
        }
    }
    __global__ __launch_bounds__(256) void slot_fill_kernel(const uint* __restrict__ sorted_col, const uint* __restrict__ sorted_key, uint n_hot,
                                                            uint* __restrict__ slot_of, uint* __restrict__ hot_cols) {
        const uint s = blockIdx.x * 256 + threadIdx.x;
        if (s < n_hot) {
            hot_cols[s]            = sorted_col[s];
            slot_of[sorted_col[s]] = (~sorted_key[s]) >= 2u ? s : 0xFFFFFFFFu;// a column referenced once gains nothing from the cache
        }
    }
    __global__ __launch_bounds__(256) void col_encode_kernel(const uint* __restrict__ Aj, uint nnz, const uint* __restrict__ slot_of, uint* __restrict__ Aj_enc) {
        const uint k = blockIdx.x * 256 + threadIdx.x;
        if (k < nnz) {
            const uint j = Aj[k];
            const uint s = slot_of[j];
            Aj_enc[k]    = s != 0xFFFFFFFFu ? (s | HotCfg::kHotFlag) : j;
        }
    }

}// namespace splahip
