// mxm_spa.hip — SpGEMM by row classes: Gustavson's row-by-row product with the accumulator of one result row kept in LDS, one column block at a
// time (the reference keeps a dense accumulator row of width n_cols per row of A, src/cpu/cpu_mxm.hpp:93-113 — this is that, cut to what 160 KB of
// LDS hold), for every op_add that has an LDS atomic (mxv_cob.hpp: cob_acc_kind — PLUS, MIN / MAX / BOR / BAND / BXOR on int and uint, LOR / LAND).
//
// Semantics: reference Algo_mxm_cpu<T>::execute, cpu_mxm.hpp:85-114: R[i,j] = fold over k in A[i,:]'s order of add(., mult(a_ik, b_kj)) from init;
// an entry exists where at least one product exists and its value != init; columns ascending.  For an associative + commutative op_add the fold
// order is free: int / uint results are bit-identical, fp32 PLUS is summed in a double slice and rounded once (the 1e-5 bar).  Everything else
// (MINUS, FIRST, SECOND, fp32 MULT, fp32 MIN / MAX ...) keeps the expand / sort / compress form of mxm.hip, which folds in the reference's order.
//
// Row classes (by the row's product count P_i = sum over k of nnz(B[k,:])):
//   light  P_i <= kLight : compacted into a CSR of their own and handed to the ESC form (a global sort of few products is cheap; a workgroup that
//                          zeroes and scans an LDS block per row is not)
//   heavy  the rest      : one workgroup per row, rows taken largest first from a device counter.  Two passes over the same traversal:
//                          SYMBOLIC (a bitmap of the touched columns -> the row's entry count -> Rp), then NUMERIC (the LDS accumulator block,
//                          ds_add / ds_min / ds_or ..., entries emitted in ascending column order straight into the result arrays).
//   dense  (fp32 MULT / PLUS only) where the leading h x h corners of A and B are dense tiles, their product runs on the matrix cores
//                          (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains) and is pre-loaded into the accumulators of the rows it belongs to;
//                          the sparse traversal then skips those products (see "dense corner" below).
//
// The traversal of one (row i, column block [c0, c1)): for chunks of 1024 entries (i, k) of A's row — a thread per k finds the part of B[k,:]
// inside the block (B's rows are column-sorted: a cursor per k kept in scratch memory advances from block to block, one binary search for the
// block's end), a workgroup scan turns the part lengths into product offsets, and the products are dealt evenly over the threads (a binary
// search of the offsets in LDS finds a product's k) — a hub's 100 K-entry row and a 3-entry row keep every lane busy alike.
#include "block_utils.hpp"
#include "common.hpp"
#include "mxv_cob.hpp"
#include "ops.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <unordered_map>
#include <vector>

extern "C" int splahip_mxm(int type, uint32_t a_rows, uint32_t b_cols, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, const uint32_t* d_Bp,
                           const uint32_t* d_Bj, const void* d_Bx, uint32_t* d_Rp, uint32_t** d_Rj, void** d_Rx, uint32_t* h_nnz, uint64_t* h_products, int op_mult,
                           int op_add, uint32_t init_bits, uint64_t block_products, void* stream);
extern "C" int splahip_free(void* d_ptr);

namespace splahip {

    thread_local bool tl_mxm_force_esc = false;// set around the light rows' call into the ESC form
    thread_local uint32_t tl_last_dense[6] = {0, 0, 0, 0, 0, 0};// the dense window of the last mxm of this thread: i0, k0, j0, rows, inner, columns (splahip_mxm_last_dense)
    thread_local uint32_t tl_last_dense_h = 0, tl_last_heavy = 0, tl_last_light_nnz = 0, tl_last_ranked = 0;// what the last mxm of this thread did (splahip_mxm_last_info)
    thread_local float    tl_last_gemm_ms = 0.0f;// the two MFMA GEMMs (values, product counts) of the last call's dense window, timed with events when SPLAHIP_MXM_DENSE_TIME=1

    namespace {

        using u64 = unsigned long long;
        constexpr unsigned kTpb   = 1024;
#ifndef SPLAHIP_MXM_LONG
#define SPLAHIP_MXM_LONG 128
#endif
        constexpr unsigned kLongPart = SPLAHIP_MXM_LONG;// products of one (A entry, column block) part from which a wavefront walks the part on its own
#ifndef SPLAHIP_MXM_WALK_DEPTH
#define SPLAHIP_MXM_WALK_DEPTH 2// (whole call at R-MAT 18: depth 1 / 2 38.3 - 38.5 ms, 3 39.3, 4 39.4 - 39.9, 6 41.9, 8 43.4 - 44.0: a step holds ~4.4 products per thread, deeper turns search and load for nothing)
#endif
#ifndef SPLAHIP_MXM_RANKED_DEPTH
#define SPLAHIP_MXM_RANKED_DEPTH 2// (whole call: 2 -> 38.2 - 38.5 ms, 4 -> 38.5 - 38.8, 8 -> 39.0 - 39.4)
#endif
        constexpr unsigned kRankedDepth = SPLAHIP_MXM_RANKED_DEPTH;// the same in the ranked form's two walks (whole rows: tens of products per thread)
        constexpr unsigned kWalkDepth = SPLAHIP_MXM_WALK_DEPTH;// products per thread in flight in the flat walk of a block step
#ifndef SPLAHIP_MXM_LIGHT_DEFAULT
#define SPLAHIP_MXM_LIGHT_DEFAULT 2048// (a light row has to fit one batch)
#endif
        u64 light_threshold() {// rows with at most this many products go to the ESC form (read per call: splahip_env_refresh() takes effect)
            const char* e = knob_env("SPLAHIP_MXM_LIGHT");
            return e && *e ? (u64) std::strtoull(e, nullptr, 10) : (u64) SPLAHIP_MXM_LIGHT_DEFAULT;
        }

        struct SpaArgs {
            const uint32_t *Ap, *Aj, *Ax, *Bp, *Bj, *Bx;
            // B's entries as (column, value word) PAIRS, written once per call (pairs_kernel): the numeric walks read 8 bytes per product from ONE line instead of a
            // column word and a value word from two — a short part (the ~4 entries of a row of B inside one accumulator block) was two or three line requests
            const uint2* Bjx;
            // work items, largest first: a heavy row, or — for a row of millions of products — one range of its column blocks (the blocks of a row are
            // independent: a hub's row is walked by many workgroups at once)
            const uint32_t* item_row;
            const uint32_t* item_c0;
            const uint32_t* item_c1;
            const uint32_t* item_out;// NUMERIC in: first result entry of the item
            uint32_t        n_heavy; // items
            uint32_t        b_cols;
            uint32_t        block;  // columns per LDS block
            // boundaries of every row of B at the numeric blocks' edges, found once per call: bblk[k * (nb + 1) + j] = first entry of B[k,:] whose column is
            // >= j * bunit (j = nb: the row's end).  With it a (row, block) step needs no search and no cursor.  nullptr: search + cursors
            const uint32_t* bblk;
            uint32_t        bunit;  // columns per numeric block
            uint32_t        nb;     // numeric blocks
            uint32_t*       cursor; // gridDim.x x max_deg words (only read / written when the row needs more than one block)
            uint32_t        max_deg;
            uint32_t*       counts; // SYMBOLIC out: entries of item q
            uint32_t*       Rj;
            uint32_t*       Rx;
            uint32_t*       dropped;// NUMERIC out: entries whose value equals init
            // dense window (fp32 MULT / PLUS): rows [d_i0, d_i0 + d_hi) x inner indices [d_k0, d_k0 + d_hk) of A and [d_k0 ..) x columns [d_j0, d_j0 + d_hj) of B are
            // dense tiles anywhere in the matrices (sides multiples of 32); their product is in dense_c (row-major d_hi x d_hj) with dense_n = number of
            // products per cell; d_hi = 0: no dense window
            uint32_t     d_i0, d_k0, d_j0, d_hi, d_hk, d_hj;
            const float* dense_c;
            const float* dense_n;
        };

        __device__ __forceinline__ uint32_t lower_bound_dev(const uint32_t* __restrict__ a, uint32_t lo, uint32_t hi, uint32_t x) {// first index in [lo, hi) with a[index] >= x
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (a[mid] < x) lo = mid + 1;
                else
                    hi = mid;
            }
            return lo;
        }

#ifdef SPLAHIP_SPA_TIMING
        // DIAGNOSTIC BUILD ONLY (tools/exp/exp_spa_timing.py): s_memtime ticks of workgroup b (thread 0) in the phases of the blocked numeric form:
        // [0] part lengths + scan, [1] product walk, [2] emit, [3] steps
        static __device__ unsigned long long g_spa_t[4][1024];
#define SPA_T(t) t = __builtin_amdgcn_s_memtime();
#else
#define SPA_T(t)
#endif
        // LDS: [accumulator block: `block` slice words (NUMERIC)] [touched bitmap: block / 32 words] [offsets 1025] [part starts 1024] [long-part lengths 1024] [a values 1024] [scan scratch]
        template<typename T, class Mult, int ACC, bool NUMERIC>
        __global__ __launch_bounds__(kTpb) void mxm_spa_kernel(SpaArgs a, Mult mult, DynBin add, T init) {
            using word = typename CobSlice<T, ACC>::word;
            extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
            word*     s_acc  = reinterpret_cast<word*>(s_raw);
            uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_raw + (NUMERIC ? (size_t) a.block * sizeof(word) : 0));
            uint32_t* s_pref = s_bits + a.block / 32;
            uint32_t* s_seg  = s_pref + kTpb + 1;
            uint32_t* s_len  = s_seg + kTpb;// long parts: their length (0 for a short one)
            uint32_t* s_av   = s_len + kTpb;
            uint32_t* s_tmp  = s_av + kTpb;// kTpb / 64 + 2 words
            uint32_t* s_lidx = s_tmp + kTpb / 64 + 4;// the long parts of a walk, in no particular order (kTpb words)
            __shared__ uint32_t s_nlong[2];          // how many: one counter per parity of the walk's number (zeroed for the next walk while this one runs)
            const uint32_t      tid = threadIdx.x;
            uint32_t*           cur = a.cursor + (size_t) blockIdx.x * a.max_deg;
            uint32_t            n_dropped = 0;
#ifdef SPLAHIP_SPA_TIMING
            unsigned long long tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0, acc_a = 0, acc_b = 0, acc_c = 0, n_steps = 0;
#endif
#ifdef SPLAHIP_SPA_WATCHDOG
            const long long wd_t0 = wall_clock64();
#define SPA_WD(code, va, vb)                                                                                                           \
    if (wall_clock64() - wd_t0 > 200000000ll) {                                                                                       \
        if ((threadIdx.x & 63) == 0) printf("[spa watchdog] block %u wave %u loop %d: %u %u\n", blockIdx.x, threadIdx.x / 64, code, (unsigned) (va), (unsigned) (vb)); \
        __builtin_trap();                                                                                                            \
    }
#else
#define SPA_WD(code, va, vb)
#endif
            // the accumulator block and the touched bitmap start clean and are left clean by every emit pass (it resets what it reads): a block in
            // which the row holds no product costs a scan of its part lengths and nothing else
            uint32_t walk_no = 0;// (uniform) walks made so far: which of the two long-part counters this walk uses
            {
                if (tid < 2) s_nlong[tid] = 0;
                for (uint32_t w = tid; w < a.block / 32; w += kTpb) s_bits[w] = 0;
                if constexpr (NUMERIC) {
                    const word id = [] {
                        if constexpr (sizeof(word) == 8) return -0.0;
                        else
                            return cob_identity<T, ACC>();
                    }();
                    for (uint32_t j = tid; j < a.block; j += kTpb) s_acc[j] = id;
                }
                __syncthreads();
            }
            // rows are sorted by descending product count: dealing them round-robin gives every workgroup a similar share, largest first
            for (uint32_t q = blockIdx.x; q < a.n_heavy; q += gridDim.x) {
                SPA_WD(1, q, a.n_heavy)
                const uint32_t i = a.item_row[q], cb = a.item_c0[q], ce = a.item_c1[q], a0 = a.Ap[i], deg = a.Ap[i + 1] - a0;
                uint32_t       run = 0;// entries of the item emitted so far (the same in every thread)
                // a row of at most kTpb entries (one part per thread) keeps its inner index, value and part boundaries in registers across the column
                // blocks: the end of a part in one block is its start in the next, and the boundary behind that is requested a block ahead — a block
                // step then starts without the Aj -> boundary-table chain (two dependent round trips of the ~5 a step had; with one workgroup per CU
                // nothing else hides them)
                const bool single = a.bblk != nullptr && deg <= kTpb;
                uint32_t   h_k = 0, h_b1 = 0, h_b = 0, h_e = 0, h_av = 0;
                const uint32_t* h_t = nullptr;
                auto            end_of = [&](uint32_t c_end) -> uint32_t { return c_end >= a.b_cols ? h_b1 : h_t[c_end / a.bunit]; };
                if (single && tid < deg) {
                    h_k  = a.Aj[a0 + tid];
                    h_b1 = a.Bp[h_k + 1];
                    h_t  = a.bblk + (size_t) h_k * (a.nb + 1);
                    h_b  = h_t[cb / a.bunit];
                    h_e  = end_of(min(ce, cb + a.block));
                    if constexpr (NUMERIC) h_av = a.Ax[a0 + tid];
                }
                for (uint32_t c0 = cb; c0 < ce; c0 += a.block) {
                    SPA_WD(2, c0, a.block)
                    const uint32_t c1 = min(ce, c0 + a.block), width = c1 - c0, words = (width + 31) / 32;
                    const bool     last = c1 == ce, to_end = c1 >= a.b_cols;
                    // this row meets the dense window inside this block?  Then the window's cells (i, j) are pre-loaded from what the matrix cores computed, and the
                    // walk below leaves out the products (i, k, j) with k and j inside the window: by cutting the front of B[k, :]'s run when the window starts
                    // where the block starts (a window at the matrices' origin: the common case of a hub corner), product by product otherwise
                    const bool     dense_rows = a.d_hi && i - a.d_i0 < a.d_hi && c0 < a.d_j0 + a.d_hj && c1 > a.d_j0;
                    const bool     dense_cut  = dense_rows && a.d_j0 == c0;
                    const uint32_t d_j1       = a.d_j0 + a.d_hj;
                    uint32_t       block_products = dense_rows ? 1u : 0u;// (uniform) anything to emit from this block?
                    if (dense_rows) {
                        for (uint32_t j = max(c0, a.d_j0) + tid; j < min(c1, d_j1); j += kTpb) {
                            const size_t cell = (size_t) (i - a.d_i0) * a.d_hj + (j - a.d_j0);
                            if (a.dense_n[cell] > 0.5f) {
                                (void) __hip_atomic_fetch_or(s_bits + ((j - c0) >> 5), 1u << ((j - c0) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if constexpr (NUMERIC && sizeof(word) == 8) s_acc[j - c0] = (double) a.dense_c[cell];
                            }
                        }
                        __syncthreads();
                    }
                    SPA_T(tp0)
#ifdef SPLAHIP_SPA_TIMING
                    unsigned long long walk = 0;
#endif
                    for (uint32_t kc = 0; kc < deg; kc += kTpb) {
                        SPA_WD(3, kc, deg)
                        const uint32_t nk  = min(kTpb, deg - kc);
                        uint32_t       len = 0;
                        bool           in_win = false;// this part's inner index lies in the dense window and the window's columns are NOT cut off the run: product by product
                        if (single) {
                            if (tid < nk) {
                                uint32_t b = h_b;
                                const uint32_t e = h_e;
                                h_b              = e;
                                if (!last) h_e = end_of(min(ce, c1 + a.block));// the boundary behind the NEXT block travels during this one
                                in_win = dense_rows && h_k - a.d_k0 < a.d_hk;
                                if (in_win && dense_cut) b = lower_bound_dev(a.Bj, b, min(e, h_b1), d_j1), in_win = false;// products (i, k, j) of the window live in the dense tile
                                len        = e - b;
                                s_seg[tid] = b;
                                if constexpr (NUMERIC) s_av[tid] = h_av;
                            }
                        } else if (tid < nk) {
                            const uint32_t k  = a.Aj[a0 + kc + tid];
                            const uint32_t b1 = a.Bp[k + 1];
                            uint32_t       b, e;
                            if (a.bblk) {// (c0 and c1 are multiples of the numeric block width, or the last column)
                                const uint32_t* t = a.bblk + (size_t) k * (a.nb + 1);
                                b                 = t[c0 / a.bunit];
                                e                 = to_end ? b1 : t[c1 / a.bunit];
                            } else {
                                b = c0 != cb ? cur[kc + tid] : (cb == 0 ? a.Bp[k] : lower_bound_dev(a.Bj, a.Bp[k], b1, cb));
                                e = to_end ? b1 : lower_bound_dev(a.Bj, b, b1, c1);
                            }
                            in_win = dense_rows && k - a.d_k0 < a.d_hk;
                            if (in_win && dense_cut) b = lower_bound_dev(a.Bj, b, min(e, b1), d_j1), in_win = false;// products (i, k, j) of the window live in the dense tile: skip them
                            len              = e - b;
                            s_seg[tid]       = b;
                            if (!last && !a.bblk) cur[kc + tid] = e;
                            if constexpr (NUMERIC) s_av[tid] = a.Ax[a0 + kc + tid];
                        }
                        // two classes of parts (the run of B[k, :] inside this block).  SHORT ones go through the flat walk below: a thread per
                        // product, which finds its part by a 10-step search of the prefix sums.  LONG ones (a hub row of B: most of the products of
                        // a heavy row) are walked by a wavefront each, lanes striding the run — no search, and the column / value words of a step
                        // are one coalesced run of B's row
                        const bool is_long = len >= kLongPart;
                        s_len[tid]         = (is_long ? len : 0u) | (in_win ? 0x80000000u : 0u);// bit 31: leave out the products whose column lies in the dense window
                        // the long parts are LISTED (an LDS counter, any order): a wave then visits the long parts only.  Every wave reading all nk part lengths in
                        // turn to find the few long ones — 64 dependent LDS reads per wave and walk — was half of the walk phase (round 5)
                        uint32_t* const n_long_p = &s_nlong[walk_no & 1u];
                        if (is_long) s_lidx[__hip_atomic_fetch_add(n_long_p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = tid;
                        uint32_t       total;
                        const uint32_t ex = block_exclusive_sum<kTpb>(is_long ? 0u : len, s_tmp, total);
                        s_pref[tid]       = ex;// threads beyond nk hold `total`: a sentinel for the search
                        __syncthreads();
                        const uint32_t n_long   = *n_long_p;
                        const bool     any_long = n_long != 0;
                        if (tid == 0) s_nlong[(walk_no + 1u) & 1u] = 0;// (the next walk's counter: its appends come behind this walk's closing barrier)
                        ++walk_no;
                        block_products |= total | (any_long ? 1u : 0u);
                        SPA_T(tp1)
                        // kWalkDepth products per thread in flight: their searches and column / value loads first (indices clamped, so the loads
                        // stand outside every branch and the compiler waits once), then their LDS atomics.  One product per turn was one global round trip per
                        // turn: a block step of ~4.5 K products walked in five dependent round trips (round 5: 28.9 ms of a 57 ms call at R-MAT 18)
                        for (uint32_t p0 = tid; p0 < total; p0 += kWalkDepth * kTpb) {
                            SPA_WD(4, p0, total)
                            uint32_t lo4[kWalkDepth], col4[kWalkDepth], x4[kWalkDepth];
#pragma unroll
                            for (int u = 0; u < (int) kWalkDepth; ++u) {
                                const uint32_t p = min(p0 + (uint32_t) u * kTpb, total - 1u);
                                uint32_t       lo = 0, hi = nk;// largest t with s_pref[t] <= p
                                while (hi - lo > 1) {
                                    const uint32_t mid = (lo + hi) >> 1;
                                    if (s_pref[mid] <= p) lo = mid;
                                    else
                                        hi = mid;
                                }
                                const uint32_t idx = s_seg[lo] + (p - s_pref[lo]);
                                lo4[u]  = lo;
                                if constexpr (NUMERIC) {
                                    const uint2 cx = a.Bjx[idx];
                                    col4[u] = cx.x, x4[u] = cx.y;
                                } else
                                    col4[u] = a.Bj[idx], x4[u] = 0;
                            }
#pragma unroll
                            for (int u = 0; u < (int) kWalkDepth; ++u) {
                                if (p0 + (uint32_t) u * kTpb >= total) break;
                                const uint32_t lo = lo4[u], col = col4[u] - c0;
                                if (dense_rows && (s_len[lo] >> 31) && col + c0 - a.d_j0 < a.d_hj) continue;// (in the dense tile)
                                (void) __hip_atomic_fetch_or(s_bits + (col >> 5), 1u << (col & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if constexpr (NUMERIC) {
                                    const T prod = mult(from_bits<T>(s_av[lo]), from_bits<T>(x4[u]));
                                    if constexpr (sizeof(word) == 8) (void) __hip_atomic_fetch_add(s_acc + col, (double) prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    else
                                        cob_accumulate<T, ACC>(s_acc + col, prod);
                                }
                            }
                        }
                        if (any_long) {
                            for (uint32_t lj = tid / 64; lj < n_long; lj += kTpb / 64) {
                                const uint32_t t = s_lidx[lj];
                                const uint32_t n = s_len[t] & 0x7FFFFFFFu;
                                const bool     win = dense_rows && (s_len[t] >> 31);
                                const uint32_t b0 = s_seg[t];
                                T              av = T();
                                if constexpr (NUMERIC) av = from_bits<T>(s_av[t]);
                                for (uint32_t p = tid & 63u; p < n; p += 256u) {// four steps of 64 in flight per lane (clamped loads, as above)
                                    uint32_t col4[4], x4[4];
#pragma unroll
                                    for (int u = 0; u < 4; ++u) {
                                        const uint32_t pp = min(p + 64u * (uint32_t) u, n - 1u);
                                        if constexpr (NUMERIC) {
                                            const uint2 cx = a.Bjx[b0 + pp];
                                            col4[u] = cx.x, x4[u] = cx.y;
                                        } else
                                            col4[u] = a.Bj[b0 + pp], x4[u] = 0;
                                    }
#pragma unroll
                                    for (int u = 0; u < 4; ++u) {
                                        if (p + 64u * (uint32_t) u >= n) break;
                                        const uint32_t col = col4[u] - c0;
                                        if (win && col + c0 - a.d_j0 < a.d_hj) continue;// (in the dense tile)
                                        (void) __hip_atomic_fetch_or(s_bits + (col >> 5), 1u << (col & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        if constexpr (NUMERIC) {
                                            const T prod = mult(av, from_bits<T>(x4[u]));
                                            if constexpr (sizeof(word) == 8) (void) __hip_atomic_fetch_add(s_acc + col, (double) prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                            else
                                                cob_accumulate<T, ACC>(s_acc + col, prod);
                                        }
                                    }
                                }
                            }
                        }
                        __syncthreads();
#ifdef SPLAHIP_SPA_TIMING
                        SPA_T(tp2)
                        walk += tp2 - tp1;
#endif
                    }
                    SPA_T(tp2)
                    // the block's entries, ascending: a popcount scan over the bitmap words (which are cleared on the way, like the accumulators)
                    if (block_products)
                    for (uint32_t w0 = 0; w0 < words; w0 += kTpb) {
                        SPA_WD(5, w0, words)
                        const uint32_t w    = w0 + tid;
                        const uint32_t bits = w < words ? s_bits[w] : 0u;
                        if (bits) s_bits[w] = 0u;
                        uint32_t       total;
                        const uint32_t ex = block_exclusive_sum<kTpb>(__popc(bits), s_tmp, total);
                        if constexpr (NUMERIC) {
                            // a thread per OUTPUT entry, not per bitmap word: entry o of the block belongs to the word whose prefix count covers it (a
                            // 10-step search of the word offsets, kept in LDS) and is that word's k-th set bit — so consecutive threads write consecutive
                            // entries of Rj / Rx.  A thread per word wrote its 3 - 4 entries one by one at a stride: the result stores (1.28 G entries,
                            // 10 GB per R-MAT 18 call) were ~38 % of a block step
                            s_pref[tid] = ex;  // (the part arrays are free between the walks)
                            s_seg[tid]  = bits;
                            __syncthreads();
                            const uint32_t base = a.item_out[q] + run, n_words = min(kTpb, words - w0);
                            for (uint32_t o = tid; o < total; o += kTpb) {
                                uint32_t lo = 0, hi = n_words;// largest word index with s_pref[index] <= o
                                while (hi - lo > 1) {
                                    const uint32_t mid = (lo + hi) >> 1;
                                    if (s_pref[mid] <= o) lo = mid;
                                    else
                                        hi = mid;
                                }
                                uint32_t b = s_seg[lo];
                                for (uint32_t k = o - s_pref[lo]; k != 0; --k) b &= b - 1;// drop the k lowest set bits
                                const uint32_t j = (w0 + lo) * 32 + (uint32_t) __ffs((int) b) - 1;
                                uint32_t       accw;
                                if constexpr (sizeof(word) == 8) {
                                    accw     = to_bits((float) s_acc[j]);
                                    s_acc[j] = -0.0;
                                } else {
                                    accw     = s_acc[j];
                                    s_acc[j] = cob_identity<T, ACC>();
                                }
                                const T val      = cob_finish<T, ACC>(accw, init, add);
                                a.Rj[base + o]   = c0 + j;
                                a.Rx[base + o]   = to_bits(val);
                                if (!(val != init)) ++n_dropped;// the reference drops r == init (cpu_mxm.hpp:110): compacted afterwards when any
                            }
                            __syncthreads();// (s_pref / s_seg are the next walk's again)
                        }
                        run += total;
                    }
                    __syncthreads();
#ifdef SPLAHIP_SPA_TIMING
                    SPA_T(tp3)
                    acc_a += (tp2 - tp0) - walk, acc_b += walk, acc_c += tp3 - tp2, n_steps += 1;
#endif
                }
                if constexpr (!NUMERIC) {
                    if (tid == 0) a.counts[q] = run;
                }
            }
            if constexpr (NUMERIC) {
                if (n_dropped) atomicAdd(a.dropped, n_dropped);
#ifdef SPLAHIP_SPA_TIMING
                if (tid == 0 && blockIdx.x < 1024) g_spa_t[0][blockIdx.x] = acc_a, g_spa_t[1][blockIdx.x] = acc_b, g_spa_t[2][blockIdx.x] = acc_c, g_spa_t[3][blockIdx.x] = n_steps;
#endif
            }
        }

        // NUMERIC, "ranked" form — for an item whose entry count (known from the symbolic pass) fits one accumulator block: no column blocks at all.  Per
        // super-block of `wr` columns: pass A sets the touched bitmap (as the symbolic pass did), a popcount scan over its words gives every touched column its
        // RANK among the item's entries, pass B walks the products again and adds each one into acc[rank(column)], and the entries leave in rank order =
        // ascending column order.  A row of a few thousand entries spread over 260 K columns costs two walks over its products instead of 43 block steps.
        // LDS: [acc: cap slice words] [bitmap: wr / 32] [word ranks: wr / 32] [offsets 1025] [part starts 1024] [a values 1024] [scan scratch]
        template<typename T, class Mult, int ACC>
        __global__ __launch_bounds__(kTpb) void mxm_spa_ranked_kernel(SpaArgs a, uint32_t cap, uint32_t wr, Mult mult, DynBin add, T init) {
            using word = typename CobSlice<T, ACC>::word;
            extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
            word*          s_acc  = reinterpret_cast<word*>(s_raw);
            uint32_t*      s_bits = reinterpret_cast<uint32_t*>(s_raw + (size_t) cap * sizeof(word));
            uint32_t*      s_rank = s_bits + wr / 32;
            uint32_t*      s_pref = s_rank + wr / 32;
            uint32_t*      s_seg  = s_pref + kTpb + 1;
            uint32_t*      s_av   = s_seg + kTpb;
            uint32_t*      s_tmp  = s_av + kTpb;
            const uint32_t tid    = threadIdx.x;
            uint32_t       n_dropped = 0;
            const word     id = [] {
                if constexpr (sizeof(word) == 8) return -0.0;
                else
                    return cob_identity<T, ACC>();
            }();
            for (uint32_t q = blockIdx.x; q < a.n_heavy; q += gridDim.x) {
                const uint32_t i = a.item_row[q], cb = a.item_c0[q], ce = a.item_c1[q], a0 = a.Ap[i], deg = a.Ap[i + 1] - a0;
                uint32_t       run = 0;
                for (uint32_t s0 = cb; s0 < ce; s0 += wr) {
                    const uint32_t s1 = min(ce, s0 + wr), words = (s1 - s0 + 31) / 32;
                    const bool     to_end = s1 >= a.b_cols;
                    // one walk over the products of (row i, columns [s0, s1)): fn(value word of a_ik, column - s0, value word of b_kj — read, with the column, from the
                    // pairs when `pairs` says so); four products per thread in flight, like the blocked form's walk
                    auto walk = [&](auto pairs, auto fn) {
                        constexpr bool kPairs = decltype(pairs)::value;
                        for (uint32_t kc = 0; kc < deg; kc += kTpb) {
                            const uint32_t nk  = min(kTpb, deg - kc);
                            uint32_t       len = 0;
                            if (tid < nk) {
                                const uint32_t  k = a.Aj[a0 + kc + tid];
                                const uint32_t* t = a.bblk + (size_t) k * (a.nb + 1);
                                const uint32_t  b = t[s0 / a.bunit], e = to_end ? a.Bp[k + 1] : t[s1 / a.bunit];
                                len               = e - b;
                                s_seg[tid]        = b;
                                s_av[tid]         = a.Ax[a0 + kc + tid];
                            }
                            uint32_t       total;
                            const uint32_t ex = block_exclusive_sum<kTpb>(len, s_tmp, total);
                            s_pref[tid]       = ex;
                            __syncthreads();
                            for (uint32_t p0 = tid; p0 < total; p0 += kRankedDepth * kTpb) {
                                uint32_t lo4[kRankedDepth], col4[kRankedDepth], x4[kRankedDepth];
#pragma unroll
                                for (int u = 0; u < (int) kRankedDepth; ++u) {
                                    const uint32_t p = min(p0 + (uint32_t) u * kTpb, total - 1u);
                                    uint32_t       lo = 0, hi = nk;
                                    while (hi - lo > 1) {
                                        const uint32_t mid = (lo + hi) >> 1;
                                        if (s_pref[mid] <= p) lo = mid;
                                        else
                                            hi = mid;
                                    }
                                    const uint32_t idx = s_seg[lo] + (p - s_pref[lo]);
                                    lo4[u]             = lo;
                                    if constexpr (kPairs) {
                                        const uint2 cx = a.Bjx[idx];
                                        col4[u] = cx.x, x4[u] = cx.y;
                                    } else
                                        col4[u] = a.Bj[idx], x4[u] = 0;
                                }
#pragma unroll
                                for (int u = 0; u < (int) kRankedDepth; ++u) {
                                    if (p0 + (uint32_t) u * kTpb >= total) break;
                                    fn(s_av[lo4[u]], col4[u] - s0, x4[u]);
                                }
                            }
                            __syncthreads();
                        }
                    };
                    for (uint32_t w = tid; w < words; w += kTpb) s_bits[w] = 0;
                    __syncthreads();
                    walk(std::false_type{}, [&](uint32_t, uint32_t col, uint32_t) { (void) __hip_atomic_fetch_or(s_bits + (col >> 5), 1u << (col & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); });
                    // ranks: exclusive popcount prefix over the bitmap words
                    uint32_t cnt = 0;
                    for (uint32_t w0 = 0; w0 < words; w0 += kTpb) {
                        const uint32_t w = w0 + tid;
                        uint32_t       total;
                        const uint32_t ex = block_exclusive_sum<kTpb>(w < words ? __popc(s_bits[w]) : 0, s_tmp, total);
                        if (w < words) s_rank[w] = cnt + ex;
                        cnt += total;
                    }
                    for (uint32_t j = tid; j < cnt; j += kTpb) s_acc[j] = id;
                    __syncthreads();
                    walk(std::true_type{}, [&](uint32_t av, uint32_t col, uint32_t xw) {
                        const uint32_t pos  = s_rank[col >> 5] + __popc(s_bits[col >> 5] & ((1u << (col & 31)) - 1u));
                        const T        prod = mult(from_bits<T>(av), from_bits<T>(xw));
                        if constexpr (sizeof(word) == 8) (void) __hip_atomic_fetch_add(s_acc + pos, (double) prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else
                            cob_accumulate<T, ACC>(s_acc + pos, prod);
                    });
                    // entries in rank order = ascending columns
                    const uint32_t out = a.item_out[q] + run;
                    for (uint32_t w = tid; w < words; w += kTpb) {
                        uint32_t p = s_rank[w];
                        for (uint32_t b = s_bits[w]; b; b &= b - 1, ++p) {
                            uint32_t accw;
                            if constexpr (sizeof(word) == 8) accw = to_bits((float) s_acc[p]);
                            else
                                accw = s_acc[p];
                            const T val   = cob_finish<T, ACC>(accw, init, add);
                            a.Rj[out + p] = s0 + w * 32 + (uint32_t) __ffs((int) b) - 1;
                            a.Rx[out + p] = to_bits(val);
                            if (!(val != init)) ++n_dropped;
                        }
                    }
                    run += cnt;
                    __syncthreads();
                }
            }
            if (n_dropped) atomicAdd(a.dropped, n_dropped);
        }

        // (column, value word) pairs of B's first n entries (n read on the device: hint[1], left by max_col_tail_kernel)
        __global__ __launch_bounds__(256) void pairs_kernel(const uint32_t* __restrict__ hint, const uint32_t* __restrict__ Bj, const uint32_t* __restrict__ Bx, uint2* __restrict__ out) {
            const uint32_t n = hint[1];
            for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) out[k] = make_uint2(Bj[k], Bx[k]);
        }
        // hint[0] = the largest inner index A refers to (max_col_kernel); hint[1] = entries of the rows of B up to it = what a call can read of B
        __global__ void max_col_tail_kernel(uint32_t n_rows, const uint32_t* __restrict__ Ap, uint32_t* __restrict__ hint, const uint32_t* __restrict__ Bp) {
            hint[1] = Ap[n_rows] ? Bp[hint[0] + 1] : 0u;// (A without entries refers to no row of B)
        }

        // bblk[k][j]: a thread per (row of B, block edge)
        __global__ __launch_bounds__(256) void bblk_kernel(uint32_t b_rows, uint32_t nb, uint32_t bunit, const uint32_t* __restrict__ Bp, const uint32_t* __restrict__ Bj,
                                                           uint32_t* __restrict__ bblk) {
            const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x;
            if (t >= (size_t) b_rows * (nb + 1)) return;
            const uint32_t k = (uint32_t) (t / (nb + 1)), j = (uint32_t) (t % (nb + 1));
            bblk[t]          = j == nb ? Bp[k + 1] : lower_bound_dev(Bj, Bp[k], Bp[k + 1], j * bunit);
        }

        // light rows -> a CSR of their own (heavy rows empty): a thread per row copies its few entries
        __global__ __launch_bounds__(256) void light_len_kernel(uint32_t n_rows, const uint32_t* __restrict__ Ap, const u64* __restrict__ P, uint32_t* __restrict__ len, u64 kLight) {
            const uint32_t i = blockIdx.x * 256 + threadIdx.x;
            if (i <= n_rows) len[i] = (i < n_rows && P[i] <= kLight) ? Ap[i + 1] - Ap[i] : 0u;
        }
        __global__ __launch_bounds__(256) void light_copy_kernel(uint32_t n_rows, const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ Aj, const uint32_t* __restrict__ Ax,
                                                                 const u64* __restrict__ P, const uint32_t* __restrict__ Lp, uint32_t* __restrict__ Lj, uint32_t* __restrict__ Lx, u64 kLight) {
            const uint32_t i = blockIdx.x * 256 + threadIdx.x;
            if (i >= n_rows || P[i] > kLight) return;
            uint32_t o = Lp[i];
            for (uint32_t e = Ap[i]; e < Ap[i + 1]; ++e, ++o) Lj[o] = Aj[e], Lx[o] = Ax[e];
        }
        // per-row entry counts of the result: light rows from the ESC result, heavy rows from the symbolic pass
        __global__ __launch_bounds__(256) void count_light_kernel(uint32_t n_rows, const uint32_t* __restrict__ Lrp, uint32_t* __restrict__ cnt) {
            const uint32_t i = blockIdx.x * 256 + threadIdx.x;
            if (i <= n_rows) cnt[i] = i < n_rows ? Lrp[i + 1] - Lrp[i] : 0u;
        }
        __global__ __launch_bounds__(256) void count_heavy_kernel(uint32_t n_heavy, const uint32_t* __restrict__ rows, const uint32_t* __restrict__ counts, uint32_t* __restrict__ cnt) {
            const uint32_t q = blockIdx.x * 256 + threadIdx.x;
            if (q < n_heavy) cnt[rows[q]] = counts[q];
        }
        // ---- light rows in LDS (expand - sort - compress inside one workgroup) ------------------------------------------------------------------------
        // A BATCH = consecutive light rows holding at most kBatchCap products and kBatchRows rows.  One workgroup of 256 threads per batch:
        //   expand   every product (i, k, j) of the batch's rows gets the 64-bit key  row-in-batch << 44 | column j << 12 | product number  and its value word —
        //            the products of a row are numbered in the order of A[i, :]'s entries (and, per entry, of B[k, :]'s), so ascending keys list every result
        //            cell's products in the reference's fold order (src/cpu/cpu_mxm.hpp:98-104);
        //   sort     a bitonic sort of the keys in LDS (the keys are distinct: no stability question), the value words stay where they are;
        //   compress the first key of every (row, column) run folds its run in order from init, drops a result equal to init (cpu_mxm.hpp:110) and the kept
        //            entries leave compacted, rows in order, columns ascending, into the batch's slice of a staging array; per row: where its entries start and
        //            how many there are.
        // No global sort, no expanded products in HBM (the ESC form wrote 12 bytes per product and radix-sorted them in five passes: 15 ms of a 61 ms R-MAT 18
        // call for 3 % of its products), exact for every op (fp32 PLUS included).  The rows' entries are copied to their place once the row pointers are known.
#ifndef SPLAHIP_MXM_BATCH_CAP
#define SPLAHIP_MXM_BATCH_CAP 2048// (round 5, whole call at R-MAT 18: 4096 products / 512 threads 40.75 ms, 2048 / 512 39.8, 2048 / 256 40.0, 1024 / 256 40.85: four workgroups per CU hide the batch's load chains)
#endif
#ifndef SPLAHIP_MXM_BATCH_TPB
#define SPLAHIP_MXM_BATCH_TPB 512
#endif
        constexpr uint32_t kBatchCap = SPLAHIP_MXM_BATCH_CAP, kBatchRows = 512, kBatchTpb = SPLAHIP_MXM_BATCH_TPB;
        // the batch's sort key gives the product number 12 bits and the batch row 9 bits (below): a build with a larger batch would corrupt the column field silently
        static_assert(kBatchCap <= 4096 && kBatchRows <= 512 && kBatchCap % 2 == 0, "light_batch_kernel: the key layout holds 12 bits of product number and 9 bits of batch row");
        template<typename T, class Mult, class Add>
        __global__ __launch_bounds__(kBatchTpb) void light_batch_kernel(const uint32_t* __restrict__ batch_first, const uint32_t* __restrict__ light_rows, const uint32_t* __restrict__ light_eoff,
                                                                  const u64* __restrict__ batch_poff,
                                                                  const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ Aj, const T* __restrict__ Ax,
                                                                  const uint32_t* __restrict__ Bp, const uint32_t* __restrict__ Bj, const T* __restrict__ Bx, uint32_t* __restrict__ Sj,
                                                                  T* __restrict__ Sx, uint32_t* __restrict__ lstart_lo, uint32_t* __restrict__ lstart_hi, uint32_t* __restrict__ lcnt,
                                                                  Mult mult, Add add, T init) {
            __shared__ u64      s_key[kBatchCap];
            __shared__ uint32_t s_val[kBatchCap];
            __shared__ uint32_t s_pref[kBatchTpb + 1], s_seg[kBatchTpb], s_av[kBatchTpb], s_rowl[kBatchTpb], s_tmp[kBatchTpb / 64 + 2];
            __shared__ uint32_t s_rcnt[kBatchRows], s_rfirst[kBatchRows], s_eoff[kBatchRows + 1];
            const uint32_t tid = threadIdx.x, b = blockIdx.x;
            const uint32_t r0 = batch_first[b], nr = batch_first[b + 1] - r0;
            const u64      out0 = batch_poff[b];
            for (uint32_t r = tid; r < nr; r += kBatchTpb) s_rcnt[r] = 0u, s_rfirst[r] = 0xFFFFFFFFu;
            for (uint32_t r = tid; r <= nr; r += kBatchTpb) s_eoff[r] = light_eoff[r0 + r] - light_eoff[r0];// entries of the batch's rows before row r
            __syncthreads();
            // expand: the batch's A entries, 256 at a time in row order (a thread finds its entry's row by a search over the rows' entry offsets)
            uint32_t n_prod = 0;// (uniform)
            {
                const uint32_t total_e = s_eoff[nr];
                for (uint32_t e0 = 0; e0 < total_e; e0 += kBatchTpb) {
                    const uint32_t e = e0 + tid;
                    uint32_t       len = 0;
                    if (e < total_e) {
                        uint32_t lo = 0, hi = nr;// largest row index with s_eoff[index] <= e
                        while (hi - lo > 1) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (s_eoff[mid] <= e) lo = mid;
                            else
                                hi = mid;
                        }
                        const uint32_t ae = Ap[light_rows[r0 + lo]] + (e - s_eoff[lo]), k = Aj[ae];
                        len               = Bp[k + 1] - Bp[k];
                        s_seg[tid]        = Bp[k];
                        s_av[tid]         = to_bits(Ax[ae]);
                        s_rowl[tid]       = lo;
                    }
                    uint32_t       total;
                    const uint32_t ex = block_exclusive_sum<kBatchTpb>(len, s_tmp, total);
                    s_pref[tid]       = ex;
                    if (tid == 0) s_pref[kBatchTpb] = total;
                    __syncthreads();
                    const uint32_t nk = min(kBatchTpb, total_e - e0);
                    for (uint32_t p = tid; p < total; p += kBatchTpb) {
                        uint32_t lo = 0, hi = nk;// largest entry index with s_pref[index] <= p
                        while (hi - lo > 1) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (s_pref[mid] <= p) lo = mid;
                            else
                                hi = mid;
                        }
                        const uint32_t idx = s_seg[lo] + (p - s_pref[lo]), at = n_prod + p;
                        s_key[at]          = ((u64) s_rowl[lo] << 44) | ((u64) Bj[idx] << 12) | at;
                        s_val[at]          = to_bits(mult(from_bits<T>(s_av[lo]), Bx[idx]));
                    }
                    n_prod += total;
                    __syncthreads();
                }
            }
            if (n_prod == 0) {
                for (uint32_t r = tid; r < nr; r += kBatchTpb) lcnt[light_rows[r0 + r]] = 0u;
                return;
            }
            // sort: bitonic over the next power of two (the padding keys are the largest)
            uint32_t n2 = 1;
            while (n2 < n_prod) n2 <<= 1;
            for (uint32_t p = n_prod + tid; p < n2; p += kBatchTpb) s_key[p] = ~0ull;
            __syncthreads();
            for (uint32_t k = 2; k <= n2; k <<= 1)
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t t = tid; t < n2 / 2; t += kBatchTpb) {
                        const uint32_t i = ((t / j) * 2 * j) + (t % j), l = i + j;// the pair (i, i + j); ascending where (i & k) == 0
                        const u64      x = s_key[i], y = s_key[l];
                        const bool     up = (i & k) == 0;
                        if ((x > y) == up) s_key[i] = y, s_key[l] = x;
                    }
                    __syncthreads();
                }
            // compress: heads fold their runs; kept results are compacted in order
            uint32_t kept_before = 0;// (uniform)
            for (uint32_t p0 = 0; p0 < n_prod; p0 += kBatchTpb) {
                const uint32_t p = p0 + tid;
                bool           keep = false;
                T              r    = init;
                uint32_t       rl = 0, col = 0;
                if (p < n_prod) {
                    const u64 key = s_key[p];
                    if (p == 0 || (s_key[p - 1] >> 12) != (key >> 12)) {// first product of a result cell
                        rl  = (uint32_t) (key >> 44);
                        col = (uint32_t) (key >> 12);
                        for (uint32_t q = p; q < n_prod && (s_key[q] >> 12) == (key >> 12); ++q) r = add(r, from_bits<T>(s_val[(uint32_t) s_key[q] & 0xFFFu]));
                        keep = r != init;
                    }
                }
                uint32_t       total;
                const uint32_t ex = block_exclusive_sum<kBatchTpb>(keep ? 1u : 0u, s_tmp, total);
                if (keep) {
                    const uint32_t o = kept_before + ex;
                    Sj[out0 + o] = col, Sx[out0 + o] = r;
                    (void) __hip_atomic_fetch_add(s_rcnt + rl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    (void) __hip_atomic_fetch_min(s_rfirst + rl, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                kept_before += total;
            }
            __syncthreads();
            for (uint32_t r = tid; r < nr; r += kBatchTpb) {
                const uint32_t row = light_rows[r0 + r];
                const u64      st  = out0 + (s_rcnt[r] ? s_rfirst[r] : 0u);
                lcnt[row] = s_rcnt[r], lstart_lo[row] = (uint32_t) st, lstart_hi[row] = (uint32_t) (st >> 32);
            }
        }
        // the light rows' entries from the staging array (or the ESC form's compact result) to their place in the result
        __global__ __launch_bounds__(256) void place_light_kernel(uint32_t n_rows, const uint32_t* __restrict__ lstart_lo, const uint32_t* __restrict__ lstart_hi,
                                                                  const uint32_t* __restrict__ lcnt, const uint32_t* __restrict__ Sj, const uint32_t* __restrict__ Sx,
                                                                  const uint32_t* __restrict__ Rp, uint32_t* __restrict__ Rj, uint32_t* __restrict__ Rx) {
            const uint32_t i = blockIdx.x * (256 / 64) + threadIdx.x / 64;// a wavefront per row: coalesced copies
            if (i >= n_rows) return;
            const uint32_t n = lcnt[i], o = Rp[i];
            if (n == 0) return;
            const u64 b = ((u64) lstart_hi[i] << 32) | lstart_lo[i];
            for (uint32_t k = threadIdx.x & 63u; k < n; k += 64) Rj[o + k] = Sj[b + k], Rx[o + k] = Sx[b + k];
        }
        // (the ESC form's result as per-row starts and counts)
        __global__ __launch_bounds__(256) void light_from_rowptr_kernel(uint32_t n_rows, const uint32_t* __restrict__ Lrp, uint32_t* __restrict__ lstart_lo, uint32_t* __restrict__ lstart_hi,
                                                                        uint32_t* __restrict__ lcnt) {
            const uint32_t i = blockIdx.x * 256 + threadIdx.x;
            if (i < n_rows) lstart_lo[i] = Lrp[i], lstart_hi[i] = 0u, lcnt[i] = Lrp[i + 1] - Lrp[i];
        }
        // compaction of the entries equal to init (rare: a numeric cancellation)
        template<typename T>
        __global__ __launch_bounds__(256) void keep_flag_kernel(uint32_t nnz, const uint32_t* __restrict__ Rx, T init, uint32_t* __restrict__ keep) {
            const uint32_t k = blockIdx.x * 256 + threadIdx.x;
            if (k <= nnz) keep[k] = (k < nnz && from_bits<T>(Rx[k]) != init) ? 1u : 0u;
        }
        __global__ __launch_bounds__(256) void keep_scatter_kernel(uint32_t nnz, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ Rj,
                                                                   const uint32_t* __restrict__ Rx, uint32_t* __restrict__ Oj, uint32_t* __restrict__ Ox) {
            const uint32_t k = blockIdx.x * 256 + threadIdx.x;
            if (k < nnz && keep[k]) Oj[rank[k]] = Rj[k], Ox[rank[k]] = Rx[k];
        }
        __global__ __launch_bounds__(256) void keep_rowptr_kernel(uint32_t n_rows, const uint32_t* __restrict__ rank, uint32_t* __restrict__ Rp) {
            const uint32_t i = blockIdx.x * 256 + threadIdx.x;
            if (i <= n_rows) Rp[i] = rank[Rp[i]];
        }

        // ---- dense tiles on the matrix cores ---------------------------------------------------------------------------------------------------
        // Where are they?  Per 32-row panel of M: entries per 32-column tile (a histogram in LDS: n_cols / 32 counters); the tiles holding >= thr entries (half
        // full: 512) are listed as (panel, tile, count).  A panel with fewer than thr entries holds none and costs two loads.
        __global__ __launch_bounds__(256) void tile_fill_kernel(uint32_t n_rows, uint32_t n_tiles, const uint32_t* __restrict__ Mp, const uint32_t* __restrict__ Mj, uint32_t thr,
                                                                uint32_t cap, uint32_t* __restrict__ n_out, uint32_t* __restrict__ out) {
            extern __shared__ uint32_t s_cnt[];
            const uint32_t n_panels = (n_rows + 31) / 32;
            for (uint32_t p = blockIdx.x; p < n_panels; p += gridDim.x) {
                const uint32_t e0 = Mp[p * 32], e1 = Mp[min(n_rows, p * 32 + 32)];
                if (e1 - e0 < thr) continue;// (uniform)
                for (uint32_t t = threadIdx.x; t < n_tiles; t += 256) s_cnt[t] = 0;
                __syncthreads();
                for (uint32_t e = e0 + threadIdx.x; e < e1; e += 256) (void) __hip_atomic_fetch_add(s_cnt + (Mj[e] >> 5), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __syncthreads();
                for (uint32_t t = threadIdx.x; t < n_tiles; t += 256)
                    if (s_cnt[t] >= thr) {
                        const uint32_t at = atomicAdd(n_out, 1u);
                        if (at < cap) out[3 * at] = p, out[3 * at + 1] = t, out[3 * at + 2] = s_cnt[t];
                    }
                __syncthreads();
            }
        }
        // densify M[r0 : r0 + hr, c0 : c0 + hc] (row-major hr x hc, zeros where no entry) and its 0 / 1 structure; `bad` is raised by a value that is not finite
        // (0 x inf in a dense tile would invent a NaN the sparse product does not have); count = entries of the window
        __global__ __launch_bounds__(256) void densify_kernel(uint32_t r0, uint32_t c0, uint32_t hc, const uint32_t* __restrict__ Mp, const uint32_t* __restrict__ Mj,
                                                              const float* __restrict__ Mx, float* __restrict__ D, float* __restrict__ S, uint32_t* __restrict__ bad,
                                                              uint32_t* __restrict__ count) {
            const uint32_t i = blockIdx.x;// one workgroup per row of the window
            uint32_t       c = 0;
            for (uint32_t e = Mp[r0 + i] + threadIdx.x; e < Mp[r0 + i + 1]; e += 256) {
                const uint32_t j = Mj[e];
                if (j >= c0 + hc) break;// columns ascend; (threads further right see larger columns: they stop too)
                if (j < c0) continue;
                const float x = Mx[e];
                D[(size_t) i * hc + (j - c0)] = x;
                S[(size_t) i * hc + (j - c0)] = 1.0f;
                ++c;
                if (!(fabsf(x) <= 3.0e38f)) atomicOr(bad, 1u);
            }
            if (c) atomicAdd(count, c);
        }
        // C[M x N] = A[M x K] x B[K x N] (row-major, sides multiples of 32), fp32 on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).  A workgroup of four
        // wavefronts owns a 64 x 64 tile of C, a wavefront its 32 x 32 quarter; K runs in steps of 32 through LDS: the 64 x 32 panel of A and the 32 x 64
        // panel of B are staged with coalesced 16-byte loads (the panel of the NEXT step is requested before the matrix cores work on this one), then every
        // wavefront issues 16 MFMAs per step from LDS.  Operand layout of the instruction (cdna_hip_programming.md 3): lane l supplies A[row = l % 32][k = l / 32]
        // and B[k = l / 32][col = l % 32]; accumulator register r of lane l holds C[row = 8 * (r / 4) + (l / 32) * 4 + r % 4][col = l % 32].  (A's panel is kept
        // with a row stride of 33 words: the 32 lanes of a half-wave read one column of it.)
        __global__ __launch_bounds__(256) void dense_gemm_mfma_kernel(uint32_t M, uint32_t N, uint32_t K, const float* __restrict__ A, const float* __restrict__ B,
                                                                      float* __restrict__ C) {
            typedef float f16v __attribute__((ext_vector_type(16)));
            __shared__ float s_a[2][64][33];
            __shared__ float s_b[2][32][64];
            const uint32_t   tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
            const uint32_t   ti = blockIdx.y * 64, tj = blockIdx.x * 64;     // the workgroup's tile of C
            const uint32_t   wi = (wv >> 1) * 32, wj = (wv & 1) * 32;        // the wavefront's quarter
            const uint32_t   m = lane & 31, kh = lane >> 5;
            // staging: thread t loads A[ti + t / 4][k0 + 8 (t % 4) .. + 8) (two float4) and B[k0 + t / 8][tj + 8 (t % 8) .. + 8) (two float4)
            const uint32_t ar = tid >> 2, ac = (tid & 3) * 8, br = tid >> 3, bc = (tid & 7) * 8;
            float4         ra[2], rb[2];
            auto fetch = [&](uint32_t k0) {
                const bool a_in = ti + ar < M, b_in = tj + bc < N;// (sides are multiples of 32: a half tile is all in or all out per 8 words... of 32)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    ra[h] = a_in ? *reinterpret_cast<const float4*>(A + (size_t) (ti + ar) * K + k0 + ac + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                    rb[h] = b_in ? *reinterpret_cast<const float4*>(B + (size_t) (k0 + br) * N + tj + bc + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            };
            auto stage = [&](int buf) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    s_a[buf][ar][ac + 4 * h + 0] = ra[h].x, s_a[buf][ar][ac + 4 * h + 1] = ra[h].y, s_a[buf][ar][ac + 4 * h + 2] = ra[h].z, s_a[buf][ar][ac + 4 * h + 3] = ra[h].w;
                    *reinterpret_cast<float4*>(&s_b[buf][br][bc + 4 * h]) = rb[h];
                }
            };
            f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            fetch(0);
            stage(0);
            __syncthreads();
            for (uint32_t k0 = 0, it = 0; k0 < K; k0 += 32, ++it) {
                const int  buf  = (int) (it & 1u);
                const bool more = k0 + 32 < K;
                if (more) fetch(k0 + 32);// in flight under the MFMAs below
#pragma unroll
                for (int kk = 0; kk < 32; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_a[buf][wi + m][kk + kh], s_b[buf][kk + kh][wj + m], acc, 0, 0, 0);
                if (more) stage(buf ^ 1);
                __syncthreads();
            }
            if (ti + wi < M && tj + wj < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t row = 8 * (r / 4) + kh * 4 + (r % 4);
                    C[(size_t) (ti + wi + row) * N + tj + wj + m] = acc[r];
                }
            }
        }

        uint32_t env_u32(const char* name, uint32_t dflt) {
            const char* e = knob_env(name);
            return e && *e ? (uint32_t) std::strtoul(e, nullptr, 10) : dflt;
        }

        // scratch memory of a call: from the calling thread's scratch arena (common.hpp; mxm_spa opens the ScratchScope), or — with the arena switched off —
        // stream-ordered from the device's pool.  (hipMalloc / hipFree took a driver round trip each and the frees synchronised the device; hipFreeAsync still costs
        // the host 25 - 170 us per buffer: ~25 buffers per call)
        struct Scratch {
            hipStream_t        s = nullptr;
            std::vector<void*> pooled;
            explicit Scratch(hipStream_t stream) : s(stream) {}
            Scratch(const Scratch&)            = delete;
            Scratch& operator=(const Scratch&) = delete;
            ~Scratch() {
                for (void* p : pooled)
                    if (p) (void) hipFreeAsync(p, s);
            }
            template<typename U>
            hipError_t alloc(U** out, size_t count) {
                void*        p     = nullptr;
                const size_t bytes = sizeof(U) * (count ? count : 1);
                hipError_t   e;
                if (scratch_active()) e = scratch_take(&p, bytes);
                else {
                    e = device_malloc_async(&p, bytes, s);
                    if (e == hipSuccess) pooled.push_back(p);
                }
                *out = static_cast<U*>(p);
                return e;
            }
        };

        // PINNED host memory for the few MB a call moves between host and device (row products and counts down, batch / item lists and row pointers up): a bump
        // allocator per host thread, reset at the start of a call and kept for the life of the process (like runtime.hip's mailbox).  A copy from or to pageable
        // memory is staged by the runtime and blocks the host: ~0.15 ms per small copy, 0.7 ms per MB (round-5 trace: 14 ms of a 54 ms call with the GPU idle)
        // (one per host thread, listed in a process-wide registry like the device-side caches of runtime.hip: splahip_mxm_trim frees the chunks of every thread that is
        // not inside a call and deletes the arenas of threads that have exited — the thread_local holder's destructor only marks its arena an orphan)
        struct PinnedArena {
            struct Chunk {
                unsigned char* p;
                size_t         cap, used;
            };
            std::mutex         m;
            std::vector<Chunk> chunks;
            int                in_call = 0;
            bool               orphan  = false;
            void reset() {
                std::lock_guard<std::mutex> g(m);
                for (Chunk& c : chunks) c.used = 0;
            }
            void* take(size_t bytes) {
                bytes = (bytes + 255) & ~size_t(255);
                std::lock_guard<std::mutex> g(m);
                for (Chunk& c : chunks)
                    if (c.cap - c.used >= bytes) {
                        void* r = c.p + c.used;
                        c.used += bytes;
                        return r;
                    }
                const size_t cap = std::max(bytes, size_t(8) << 20);
                void*        p   = nullptr;
                if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
                    (void) hipGetLastError();
                    return nullptr;
                }
                chunks.push_back({static_cast<unsigned char*>(p), cap, bytes});
                return p;
            }
        };
        struct PinnedRegistry {
            std::mutex                m;
            std::vector<PinnedArena*> all;
        };
        PinnedRegistry& pinned_registry() {
            static PinnedRegistry* r = new PinnedRegistry;
            return *r;
        }
        struct PinnedHolder {
            PinnedArena* a;
            PinnedHolder() : a(new PinnedArena) {
                auto& r = pinned_registry();
                std::lock_guard<std::mutex> g(r.m);
                r.all.push_back(a);
            }
            ~PinnedHolder() {
                std::lock_guard<std::mutex> g(a->m);
                a->orphan = true;
            }
        };
        PinnedArena& arena() {
            static thread_local PinnedHolder h;
            return *h.a;
        }
        struct PinnedCall {// the calling thread's arena is in use for the length of a call (a trim from another thread leaves it alone)
            PinnedCall() {
                PinnedArena& a = arena();
                std::lock_guard<std::mutex> g(a.m);
                ++a.in_call;
            }
            ~PinnedCall() {
                PinnedArena& a = arena();
                std::lock_guard<std::mutex> g(a.m);
                --a.in_call;
            }
        };
        // host -> device through the arena (the source may go out of scope at once); device -> host INTO the arena (valid after the next synchronisation of s,
        // until the thread's next mxm call)
        hipError_t h2d(void* dst, const void* src, size_t bytes, hipStream_t s) {
            if (bytes == 0) return hipSuccess;
            void* st = arena().take(bytes);
            if (!st) {
                const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
                return e != hipSuccess ? e : hipStreamSynchronize(s);
            }
            std::memcpy(st, src, bytes);
            return hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, s);
        }
        template<typename U>
        hipError_t d2h(U** host, const U* src, size_t count, hipStream_t s) {
            U* st = static_cast<U*>(arena().take(sizeof(U) * std::max<size_t>(1, count)));
            *host = st;
            if (!st) return hipErrorOutOfMemory;
            return count ? hipMemcpyAsync(st, src, sizeof(U) * count, hipMemcpyDeviceToHost, s) : hipSuccess;
        }
        // SPLAHIP_MXM_HOSTTRACE=1: HOST time between the marks of a call, no synchronisation (what the host does while the GPU waits; diagnostic)
        void host_mark(const char* what) {
            if (knob_env("SPLAHIP_MXM_HOSTTRACE") == nullptr) return;// (read per call: a cached hash lookup)
            static thread_local auto t0 = std::chrono::steady_clock::now();
            const auto               t1 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[mxm-host] %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        }

        hipError_t scan_u32(uint32_t* in, uint32_t* out, size_t n, Scratch& tmp, hipStream_t s) {
            size_t     bytes = 0;
            hipError_t e     = rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
            void*      t     = nullptr;
            if (e == hipSuccess) e = tmp.alloc(reinterpret_cast<unsigned char**>(&t), bytes ? bytes : 4);
            if (e == hipSuccess) e = rocprim::exclusive_scan(t, bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
            return e;
        }

        // SPLAHIP_MXM_TRACE=1: the phases of the row-class form on stderr, each closed by a device synchronisation (diagnostic)
        void spa_trace(const char* what, hipStream_t s) {
            static const bool on = knob_env("SPLAHIP_MXM_TRACE") != nullptr;
            if (!on) return;
            static thread_local auto t0 = std::chrono::steady_clock::now();
            const hipError_t         e  = hipStreamSynchronize(s);
            const auto               t1 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[mxm-spa] %-22s %9.2f ms  (%s)\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count(), hipGetErrorString(e));
            t0 = t1;
        }

        template<typename T, class Mult, int ACC>
        int spa_run(uint32_t a_rows, uint32_t b_cols, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const uint32_t* Bp, const uint32_t* Bj, const void* Bx, uint32_t* Rp,
                    uint32_t** Rj, void** Rx, uint32_t* h_nnz, const u64* hP, const u64* dP, const uint32_t* hAp, const uint32_t* h_order, int op_mult, int op_add,
                    uint32_t init_bits, uint32_t b_rows_hint, const uint32_t* d_hint, uint32_t b_nnz_hint, hipStream_t s) {
            using word      = typename CobSlice<T, ACC>::word;
            constexpr int t = TypeId<T>::value;
            const u64     kLight = light_threshold();
            Scratch       tmp(s);
            hipError_t    e = hipSuccess;
            spa_trace("start", s);
            host_mark("call start");
            // B as (column, value) pairs for the numeric walks (8 bytes per entry of scratch, one pass)
            uint2* d_bjx = nullptr;
            e            = tmp.alloc(&d_bjx, b_nnz_hint);
            if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm (row classes): %s", hipGetErrorString(e));
            if (b_nnz_hint) hipLaunchKernelGGL(pairs_kernel, dim3(std::min<uint32_t>(4096u, (b_nnz_hint + 255) / 256)), dim3(256), 0, s, d_hint, Bj, static_cast<const uint32_t*>(Bx), d_bjx);
            // heavy rows, largest first: the front of h_order (rows by descending product count, sorted on the device)
            std::vector<uint32_t> heavy;
            uint32_t              max_deg = 1;
            for (uint32_t k = 0; k < a_rows && hP[h_order[k]] > kLight; ++k) {
                const uint32_t i = h_order[k];
                heavy.push_back(i);
                max_deg = std::max(max_deg, hAp[i + 1] - hAp[i]);
            }
            const bool any_light = heavy.size() < a_rows && hP[h_order[heavy.size()]] > 0;
            // (the work items are listed further down, while the light rows' kernel runs)

            // ---- light rows: expand - sort - compress inside a workgroup's LDS (light_batch_kernel); SPLAHIP_MXM_LIGHT_LDS=0: the global ESC form -------
            uint32_t *Lrj = nullptr, *Lrx = nullptr, l_nnz = 0;// (the ESC form's result arrays, when it ran)
            uint32_t *d_lcnt = nullptr, *d_lstart_lo = nullptr, *d_lstart_hi = nullptr, *Sj = nullptr, *Sx = nullptr, *stage_j = nullptr, *stage_x = nullptr;
            struct FreeStage {
                uint32_t *&j, *&x;
                ~FreeStage() {
                    if (j) (void) handoff_release(j);
                    if (x) (void) handoff_release(x);
                }
            } free_stage{stage_j, stage_x};
            if (any_light) {
                e = tmp.alloc(&d_lcnt, (size_t) a_rows + 1);
                if (e == hipSuccess) e = tmp.alloc(&d_lstart_lo, (size_t) a_rows);
                if (e == hipSuccess) e = tmp.alloc(&d_lstart_hi, (size_t) a_rows);
                if (e == hipSuccess) e = hipMemsetAsync(d_lcnt, 0, 4 * ((size_t) a_rows + 1), s);
                if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm (row classes): %s", hipGetErrorString(e));
            }
            if (any_light && kLight <= kBatchCap && env_u32("SPLAHIP_MXM_LIGHT_LDS", 1) != 0) {
                // batches of consecutive light rows: at most kBatchCap products and kBatchRows rows each
                std::vector<uint32_t> lrows, eoff(1, 0u), bfirst(1, 0u);
                std::vector<u64>      poff(1, 0ull);
                lrows.reserve(a_rows), eoff.reserve((size_t) a_rows + 1);
                u64                   in_batch = 0;
                for (uint32_t i = 0; i < a_rows; ++i) {
                    if (hP[i] == 0 || hP[i] > kLight) continue;
                    if (in_batch + hP[i] > kBatchCap || lrows.size() - bfirst.back() >= kBatchRows) {
                        bfirst.push_back((uint32_t) lrows.size());
                        poff.push_back(poff.back() + in_batch);
                        in_batch = 0;
                    }
                    lrows.push_back(i);
                    eoff.push_back(eoff.back() + (hAp[i + 1] - hAp[i]));
                    in_batch += hP[i];
                }
                bfirst.push_back((uint32_t) lrows.size());
                poff.push_back(poff.back() + in_batch);
                const uint32_t n_batches = (uint32_t) bfirst.size() - 1;
                host_mark("light batches listed");
                uint32_t *d_lrows = nullptr, *d_eoff = nullptr, *d_bfirst = nullptr;
                u64*      d_poff = nullptr;
                e = tmp.alloc(&d_lrows, lrows.size());
                if (e == hipSuccess) e = tmp.alloc(&d_eoff, eoff.size());
                if (e == hipSuccess) e = tmp.alloc(&d_bfirst, bfirst.size());
                if (e == hipSuccess) e = tmp.alloc(&d_poff, poff.size());
                // (the staging arrays come from the hand-off cache: two hipMalloc / hipFree pairs of half a GB per call cost more than the kernel)
                if (e == hipSuccess) e = handoff_alloc(reinterpret_cast<void**>(&Sj), sizeof(uint32_t) * (size_t) std::max<u64>(1, poff.back()));
                if (e == hipSuccess) stage_j = Sj, e = handoff_alloc(reinterpret_cast<void**>(&Sx), sizeof(uint32_t) * (size_t) std::max<u64>(1, poff.back()));
                if (e == hipSuccess) stage_x = Sx;
                if (e == hipSuccess) e = h2d(d_lrows, lrows.data(), 4 * lrows.size(), s);
                if (e == hipSuccess) e = h2d(d_eoff, eoff.data(), 4 * eoff.size(), s);
                if (e == hipSuccess) e = h2d(d_bfirst, bfirst.data(), 4 * bfirst.size(), s);
                if (e == hipSuccess) e = h2d(d_poff, poff.data(), 8 * poff.size(), s);
                if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm (light rows): %s", hipGetErrorString(e));
                hipLaunchKernelGGL((light_batch_kernel<T, Mult, DynBin>), dim3(n_batches), dim3(kBatchTpb), 0, s, d_bfirst, d_lrows, d_eoff, d_poff, Ap, Aj, static_cast<const T*>(Ax), Bp, Bj,
                                   static_cast<const T*>(Bx), Sj, reinterpret_cast<T*>(Sx), d_lstart_lo, d_lstart_hi, d_lcnt, Mult(op_mult), DynBin(op_add), from_bits<T>(init_bits));
                e = hipGetLastError();// (no synchronisation: the lists went up through the pinned arena, and the host has the heavy rows to sort meanwhile)
                if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm (light rows): %s", hipGetErrorString(e));
            } else if (any_light) {
                uint32_t *len = nullptr, *Lp = nullptr, *Lj = nullptr, *Lx = nullptr, *Lrp = nullptr;
                e = tmp.alloc(&len, (size_t) a_rows + 1);
                if (e == hipSuccess) e = tmp.alloc(&Lp, (size_t) a_rows + 1);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(light_len_kernel, dim3((a_rows + 256) / 256), dim3(256), 0, s, a_rows, Ap, dP, len, kLight);
                    e = scan_u32(len, Lp, (size_t) a_rows + 1, tmp, s);
                }
                uint32_t l_entries = 0;
                if (e == hipSuccess) e = hipMemcpyAsync(&l_entries, Lp + a_rows, 4, hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = hipStreamSynchronize(s);
                if (e == hipSuccess) e = tmp.alloc(&Lj, l_entries);
                if (e == hipSuccess) e = tmp.alloc(&Lx, l_entries);
                if (e == hipSuccess) e = tmp.alloc(&Lrp, (size_t) a_rows + 1);
                if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm (row classes): %s", hipGetErrorString(e));
                hipLaunchKernelGGL(light_copy_kernel, dim3((a_rows + 255) / 256), dim3(256), 0, s, a_rows, Ap, Aj, static_cast<const uint32_t*>(Ax), dP, Lp, Lj, Lx, kLight);
                tl_mxm_force_esc = true;
                ScratchPause pause;// (the ESC form allocates per row block: it keeps to the stream-ordered pool)
                void*     lrx    = nullptr;
                uint32_t  esc_nnz = 0;
                const int st     = splahip_mxm(t, a_rows, b_cols, Lp, Lj, Lx, Bp, Bj, Bx, Lrp, &Lrj, &lrx, &esc_nnz, nullptr, op_mult, op_add, init_bits, 0, s);
                tl_mxm_force_esc = false;
                if (st != SPLAHIP_OK) return st;
                Lrx = static_cast<uint32_t*>(lrx);
                Sj = Lrj, Sx = Lrx;
                hipLaunchKernelGGL(light_from_rowptr_kernel, dim3((a_rows + 255) / 256), dim3(256), 0, s, a_rows, Lrp, d_lstart_lo, d_lstart_hi, d_lcnt);
            }
            host_mark("light rows enqueued");
            spa_trace("light rows", s);
            struct FreeLight {
                uint32_t *&j, *&x;
                ~FreeLight() {
                    if (j) (void) splahip_free(j);
                    if (x) (void) splahip_free(x);
                }
            } free_light{Lrj, Lrx};

            // ---- heavy rows: the work items (listed on the host while the light rows' kernel runs; nothing here depends on the dense window) ----------------
            int dev = 0, cus = 256;
            (void) hipGetDevice(&dev);
            (void) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            const uint32_t extra = sizeof(uint32_t) * (5 * kTpb + 1 + kTpb / 64 + 4);
            // numeric: accumulator block + bitmap; symbolic: bitmap only (a much wider block)
            // columns per accumulator block of the blocked numeric form.  SPLAHIP_MXM_BLOCK_KB: its LDS footprint, default 128 KB = one workgroup
            // per CU.  Every (row, block) step pays a scan of the row's part lengths, an emit scan and six barriers whatever it holds, and a
            // hub row meets every block: R-MAT 18 per call 107 / 92 / 84 / 77 ms with 48 (two workgroups per CU, 43 blocks) / 64 / 96 / 128 KB (16 blocks)
            const uint32_t blk_kb  = std::min(128u, std::max(16u, env_u32("SPLAHIP_MXM_BLOCK_KB", 128)));
            const uint32_t blk_num = blk_kb * 1024u / (uint32_t) sizeof(word) / 2048u * 2048u;
            const uint32_t blk_sym = std::max(blk_num, std::min<uint32_t>((1u << 20) - 32768u, b_cols + blk_num - 1) / blk_num * blk_num);// a multiple of blk_num; bitmap + chunk arrays < 160 KB
            const uint32_t lds_num = blk_num * (uint32_t) sizeof(word) + blk_num / 8 + extra, lds_sym = blk_sym / 8 + extra;
            // items: a heavy row; a row of more than kSplit products is cut into ranges of whole numeric blocks (its column blocks are independent), so that
            // the few hub rows that carry most of the products are walked by many workgroups at once
            const u64      kSplit = env_u32("SPLAHIP_MXM_SPLIT", 1u << 19);
            const uint32_t nb_num = (b_cols + blk_num - 1) / blk_num;
            struct Item {
                uint32_t row, c0, c1;
                u64      est;
            };
            std::vector<Item> items;
            for (uint32_t i : heavy) {
                const uint32_t pieces = (uint32_t) std::min<u64>(nb_num, std::max<u64>(1, (hP[i] + kSplit - 1) / kSplit));
                for (uint32_t pc = 0; pc < pieces; ++pc) {
                    const uint32_t b0 = (uint32_t) ((u64) nb_num * pc / pieces), b1 = (uint32_t) ((u64) nb_num * (pc + 1) / pieces);
                    if (b1 > b0) items.push_back({i, b0 * blk_num, std::min(b_cols, b1 * blk_num), hP[i] / pieces});
                }
            }
            std::stable_sort(items.begin(), items.end(), [](const Item& x, const Item& y) { return x.est > y.est; });
            const uint32_t        n_items = (uint32_t) items.size();
            host_mark("items listed");
            std::vector<uint32_t> h_item(3 * (size_t) n_items);
            for (uint32_t q = 0; q < n_items; ++q) h_item[q] = items[q].row, h_item[n_items + q] = items[q].c0, h_item[2 * (size_t) n_items + q] = items[q].c1;

            // ---- dense window (fp32 MULT / PLUS): a block A[i0 : i0 + hi, k0 : k0 + hk] x B[k0 : k0 + hk, j0 : j0 + hj] of tiles that are at least half full and
            // finite — ANYWHERE in the matrices (sides multiples of 32, at most SPLAHIP_MXM_DENSE_H = 4096: six window-sized fp32 scratch arrays, 384 MB at most) — goes through the matrix cores ---------------------
            uint32_t d_i0 = 0, d_k0 = 0, d_j0 = 0, d_hi = 0, d_hk = 0, d_hj = 0;
            float *  dA = nullptr, *dB = nullptr, *dC = nullptr, *dN = nullptr, *sA = nullptr, *sB = nullptr;
            if constexpr (std::is_same<T, float>::value && ACC == COB_ADD) {
                const uint32_t b_rows = b_rows_hint, h_max = std::max(32u, env_u32("SPLAHIP_MXM_DENSE_H", 4096u)) & ~31u;
                if (op_mult == SPLAHIP_MULT && env_u32("SPLAHIP_MXM_DENSE", 1) != 0 && a_rows >= 32 && b_cols >= 32 && b_rows >= 32) {
                    // 1. the half-full 32 x 32 tiles of A (row panels x inner tiles) and of B (inner panels x column tiles)
                    constexpr uint32_t kCap = 1u << 16;// listed tiles per matrix (more: the densest region is still among them in practice; the window is verified below anyway)
                    struct Tile {
                        uint32_t p, t, c;
                    };
                    auto list_tiles = [&](uint32_t n_rows, uint32_t n_cols, const uint32_t* Mp, const uint32_t* Mj, std::vector<Tile>& out) -> hipError_t {
                        const uint32_t n_tiles = (n_cols + 31) / 32;
                        if ((size_t) n_tiles * 4 > 150u * 1024u) return hipSuccess;// (the histogram lives in LDS: matrices of up to 1.2 M columns)
                        uint32_t *cnt = nullptr, *lst = nullptr;
                        hipError_t e2 = tmp.alloc(&cnt, 1);
                        if (e2 == hipSuccess) e2 = tmp.alloc(&lst, 3 * (size_t) kCap);
                        if (e2 == hipSuccess) e2 = hipMemsetAsync(cnt, 0, 4, s);
                        if (e2 != hipSuccess) return e2;
                        if (n_tiles * 4 > 64 * 1024) (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        hipLaunchKernelGGL(tile_fill_kernel, dim3(std::min(4096u, (n_rows + 31) / 32)), dim3(256), n_tiles * 4, s, n_rows, n_tiles, Mp, Mj, 512u, kCap, cnt, lst);
                        uint32_t n = 0;
                        e2         = hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, s);
                        if (e2 == hipSuccess) e2 = hipStreamSynchronize(s);
                        n = std::min(n, kCap);
                        std::vector<uint32_t> h(3 * (size_t) n);
                        if (e2 == hipSuccess && n) e2 = hipMemcpy(h.data(), lst, 12 * (size_t) n, hipMemcpyDeviceToHost);
                        for (uint32_t q = 0; e2 == hipSuccess && q < n; ++q) out.push_back({h[3 * q], h[3 * q + 1], h[3 * q + 2]});
                        return e2;
                    };
                    std::vector<Tile> ta, tb;
                    e = list_tiles(a_rows, b_rows, Ap, Aj, ta);
                    if (e == hipSuccess && !ta.empty()) e = list_tiles(b_rows, b_cols, Bp, Bj, tb);
                    if (e == hipSuccess && !ta.empty() && !tb.empty()) {
                        // 2. the seed: the fullest tile (ip, kp) of A whose inner panel kp holds a half-full tile (kp, jp) of B (the fullest of them); then the runs of
                        // half-full tiles through the seed along rows, inner indices (both matrices) and columns give the window's sides
                        auto key = [](uint32_t p, uint32_t t) { return ((u64) p << 32) | t; };
                        std::unordered_map<u64, uint32_t> sa, sb;
                        std::unordered_map<uint32_t, Tile> best_b;// per inner panel: B's fullest tile
                        for (const Tile& t : ta) sa[key(t.p, t.t)] = t.c;
                        for (const Tile& t : tb) {
                            sb[key(t.p, t.t)] = t.c;
                            auto it = best_b.find(t.p);
                            if (it == best_b.end() || it->second.c < t.c) best_b[t.p] = t;
                        }
                        std::sort(ta.begin(), ta.end(), [](const Tile& x, const Tile& y) { return x.c != y.c ? x.c > y.c : (x.p != y.p ? x.p < y.p : x.t < y.t); });
                        const Tile* seed = nullptr;
                        for (const Tile& t : ta)
                            if (best_b.count(t.t)) {
                                seed = &t;
                                break;
                            }
                        if (seed) {
                            const uint32_t ip = seed->p, kp = seed->t, jp = best_b[kp].t, span = h_max / 32;
                            auto in_a = [&](uint32_t p, uint32_t t) { return sa.count(key(p, t)) != 0; };
                            auto in_b = [&](uint32_t p, uint32_t t) { return sb.count(key(p, t)) != 0; };
                            uint32_t i_lo = ip, i_hi = ip + 1, k_lo = kp, k_hi = kp + 1, j_lo = jp, j_hi = jp + 1;
                            while (i_lo > 0 && i_hi - i_lo < span && in_a(i_lo - 1, kp)) --i_lo;
                            while (i_hi - i_lo < span && in_a(i_hi, kp)) ++i_hi;
                            while (k_lo > 0 && k_hi - k_lo < span && in_a(ip, k_lo - 1) && in_b(k_lo - 1, jp)) --k_lo;
                            while (k_hi - k_lo < span && in_a(ip, k_hi) && in_b(k_hi, jp)) ++k_hi;
                            while (j_lo > 0 && j_hi - j_lo < span && in_b(kp, j_lo - 1)) --j_lo;
                            while (j_hi - j_lo < span && in_b(kp, j_hi)) ++j_hi;
                            // 3. the window is VERIFIED (both blocks half full over their whole area, every value finite) and shrunk towards the seed until it is:
                            // the runs above only looked along the seed's row, column and inner panel
                            for (int attempt = 0; attempt < 12 && d_hi == 0 && e == hipSuccess; ++attempt) {
                                const uint32_t i0 = i_lo * 32, k0 = k_lo * 32, j0 = j_lo * 32;
                                const uint32_t hi = std::min(i_hi * 32, a_rows & ~31u) - i0, hk = std::min({k_hi * 32, b_rows & ~31u}) - k0, hj = std::min(j_hi * 32, b_cols & ~31u) - j0;
                                if ((int32_t) hi < 32 || (int32_t) hk < 32 || (int32_t) hj < 32) break;
                                uint32_t* stat = nullptr;// bad, count A, count B
                                e              = tmp.alloc(&stat, 4);
                                if (e == hipSuccess) e = tmp.alloc(&dA, (size_t) hi * hk);
                                if (e == hipSuccess) e = tmp.alloc(&sA, (size_t) hi * hk);
                                if (e == hipSuccess) e = tmp.alloc(&dB, (size_t) hk * hj);
                                if (e == hipSuccess) e = tmp.alloc(&sB, (size_t) hk * hj);
                                if (e == hipSuccess) e = tmp.alloc(&dC, (size_t) hi * hj);
                                if (e == hipSuccess) e = tmp.alloc(&dN, (size_t) hi * hj);
                                if (e != hipSuccess) break;
                                (void) hipMemsetAsync(stat, 0, 16, s);
                                for (float* q : {dA, sA}) (void) hipMemsetAsync(q, 0, sizeof(float) * (size_t) hi * hk, s);
                                for (float* q : {dB, sB}) (void) hipMemsetAsync(q, 0, sizeof(float) * (size_t) hk * hj, s);
                                hipLaunchKernelGGL(densify_kernel, dim3(hi), dim3(256), 0, s, i0, k0, hk, Ap, Aj, static_cast<const float*>(Ax), dA, sA, stat, stat + 1);
                                hipLaunchKernelGGL(densify_kernel, dim3(hk), dim3(256), 0, s, k0, j0, hj, Bp, Bj, static_cast<const float*>(Bx), dB, sB, stat, stat + 2);
                                uint32_t hs[3] = {0, 0, 0};
                                e              = hipMemcpyAsync(hs, stat, 12, hipMemcpyDeviceToHost, s);
                                if (e == hipSuccess) e = hipStreamSynchronize(s);
                                if (e != hipSuccess) break;
                                if (!hs[0] && hs[1] >= 0.5 * hi * hk && hs[2] >= 0.5 * hk * hj) {
                                    const dim3 grid((hj + 63) / 64, (hi + 63) / 64);
                                    const bool timed = env_u32("SPLAHIP_MXM_DENSE_TIME", 0) != 0;// (measurement runs: bench.py --workload mxm-dense)
                                    hipEvent_t ev0 = nullptr, ev1 = nullptr;
                                    if (timed && hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess) (void) hipEventRecord(ev0, s);
                                    hipLaunchKernelGGL(dense_gemm_mfma_kernel, grid, dim3(256), 0, s, hi, hj, hk, dA, dB, dC);
                                    hipLaunchKernelGGL(dense_gemm_mfma_kernel, grid, dim3(256), 0, s, hi, hj, hk, sA, sB, dN);
                                    tl_last_gemm_ms = 0.0f;
                                    if (ev0 && ev1) {
                                        (void) hipEventRecord(ev1, s);
                                        if (hipEventSynchronize(ev1) == hipSuccess) (void) hipEventElapsedTime(&tl_last_gemm_ms, ev0, ev1);
                                    }
                                    if (ev0) (void) hipEventDestroy(ev0);
                                    if (ev1) (void) hipEventDestroy(ev1);
                                    d_i0 = i0, d_k0 = k0, d_j0 = j0, d_hi = hi, d_hk = hk, d_hj = hj;
                                } else {// halve the longest side, keeping the seed's tile inside
                                    uint32_t &lo = (i_hi - i_lo >= k_hi - k_lo && i_hi - i_lo >= j_hi - j_lo) ? i_lo : (k_hi - k_lo >= j_hi - j_lo ? k_lo : j_lo);
                                    uint32_t &up = (&lo == &i_lo) ? i_hi : (&lo == &k_lo ? k_hi : j_hi);
                                    const uint32_t at = (&lo == &i_lo) ? ip : (&lo == &k_lo ? kp : jp);
                                    if (up - lo <= 1) break;// the seed's tiles themselves are not finite: no window
                                    const uint32_t half = (up - lo + 1) / 2;
                                    if (at - lo < half) up = lo + half;
                                    else
                                        lo = up - half;
                                }
                            }
                        }
                    }
                    if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm (dense window): %s", hipGetErrorString(e));
                }
            }
            const uint32_t dense_h = d_hi;// (non-zero: a dense window exists)

            // ---- heavy rows: work items, symbolic pass, row pointers + item offsets, numeric pass -----------------------------------------------
            const uint32_t grid_sym = std::max(1u, std::min(n_items, (uint32_t) cus * std::max(1u, std::min(2u, (160u * 1024u - 256u) / lds_sym))));
            const uint32_t grid_num = std::max(1u, std::min(n_items, (uint32_t) cus * std::max(1u, std::min(2u, (160u * 1024u - 256u) / lds_num))));
            const uint32_t grid     = std::max(grid_sym, grid_num);
            uint32_t *d_item = nullptr, *d_out = nullptr, *d_counts = nullptr, *d_cursor = nullptr, *d_dropped = nullptr;
            e = tmp.alloc(&d_item, 3 * (size_t) n_items);
            if (e == hipSuccess) e = tmp.alloc(&d_out, n_items);
            if (e == hipSuccess) e = tmp.alloc(&d_counts, n_items);
            if (e == hipSuccess) e = tmp.alloc(&d_cursor, (size_t) grid * max_deg);
            if (e == hipSuccess) e = tmp.alloc(&d_dropped, 1);
            if (e == hipSuccess && n_items) e = h2d(d_item, h_item.data(), 4 * h_item.size(), s);
            if (e == hipSuccess) e = hipMemsetAsync(d_dropped, 0, 4, s);
            if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm (row classes): %s", hipGetErrorString(e));
            // block boundaries of B's rows, once per call (up to 1 GiB; beyond that the kernels search and keep cursors)
            uint32_t* d_bblk = nullptr;
            if (nb_num > 1 && (u64) b_rows_hint * (nb_num + 1) * 4ull <= (1ull << 30)) {
                e = tmp.alloc(&d_bblk, (size_t) b_rows_hint * (nb_num + 1));
                if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm (row classes): %s", hipGetErrorString(e));
                const size_t cells = (size_t) b_rows_hint * (nb_num + 1);
                hipLaunchKernelGGL(bblk_kernel, dim3((unsigned) ((cells + 255) / 256)), dim3(256), 0, s, b_rows_hint, nb_num, blk_num, Bp, Bj, d_bblk);
            }
            SpaArgs a{Ap, Aj, static_cast<const uint32_t*>(Ax), Bp, Bj, static_cast<const uint32_t*>(Bx), d_bjx, d_item, d_item + n_items, d_item + 2 * (size_t) n_items, d_out, n_items, b_cols,
                      blk_sym, d_bblk, blk_num, nb_num, d_cursor, max_deg, d_counts, nullptr, nullptr, d_dropped, d_i0, d_k0, d_j0, d_hi, d_hk, d_hj, dC, dN};
            const T init = from_bits<T>(init_bits);
            if (n_items) {
                // (a per-device setting and cheap: set on every call rather than remembered in a process-wide flag)
                SPLAHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mxm_spa_kernel<T, Mult, ACC, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                SPLAHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mxm_spa_kernel<T, Mult, ACC, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                hipLaunchKernelGGL((mxm_spa_kernel<T, Mult, ACC, false>), dim3(grid_sym), dim3(kTpb), lds_sym, s, a, Mult(op_mult), DynBin(op_add), init);
                SPLAHIP_TRY(hipGetLastError());
                spa_trace("symbolic pass", s);
            }
            // row pointers and item offsets on the host (a few hundred thousand counters): entries per row = the light rows' ESC result or the sum of the
            // row's items; an item's entries start at Rp[row] + the counts of the row's items to its left
            host_mark("symbolic pass enqueued");
            std::vector<uint32_t> h_Rp((size_t) a_rows + 1), h_out(n_items), h_cnt_own;
            uint32_t *            h_counts = nullptr, *h_cnt = nullptr;
            SPLAHIP_TRY(d2h(&h_counts, d_counts, n_items, s));
            if (d_lcnt) SPLAHIP_TRY(d2h(&h_cnt, d_lcnt, a_rows, s));
            else {
                h_cnt_own.assign((size_t) a_rows + 1, 0u);
                h_cnt = h_cnt_own.data();
            }
            SPLAHIP_TRY(hipStreamSynchronize(s));
            host_mark("counts on the host");
            if (d_lcnt) {
                u64 ln = 0;
                for (uint32_t i = 0; i < a_rows; ++i) ln += h_cnt[i];
                l_nnz = (uint32_t) std::min<u64>(ln, 0xFFFFFFFFull);
            }
            u64 total = l_nnz;
            for (uint32_t q = 0; q < n_items; ++q) total += h_counts[q];
            if (total > 0xFFFFFFF0ull) return fail(SPLAHIP_NOT_IMPLEMENTED, "the product holds %llu entries: not representable with 32-bit row pointers", total);
            // (the items of a row in ascending column order: a stable counting sort by column block, then rows take their items in that order — no comparison sort
            // of ~10^5 items per call)
            {
                std::vector<uint32_t> by_c0(n_items), first((size_t) nb_num + 2, 0u);
                for (uint32_t q = 0; q < n_items; ++q) ++first[items[q].c0 / blk_num + 1];
                for (size_t b = 1; b < first.size(); ++b) first[b] += first[b - 1];
                for (uint32_t q = 0; q < n_items; ++q) by_c0[first[items[q].c0 / blk_num]++] = q;
                for (uint32_t q : by_c0) {
                    h_out[q] = h_cnt[items[q].row];// offset inside the row so far
                    h_cnt[items[q].row] += h_counts[q];
                }
            }
            h_Rp[0] = 0;
            for (uint32_t i = 0; i < a_rows; ++i) h_Rp[i + 1] = h_Rp[i] + h_cnt[i];
            for (uint32_t q = 0; q < n_items; ++q) h_out[q] += h_Rp[items[q].row];
            SPLAHIP_TRY(h2d(Rp, h_Rp.data(), 4 * ((size_t) a_rows + 1), s));
            if (n_items) SPLAHIP_TRY(h2d(d_out, h_out.data(), 4 * (size_t) n_items, s));
            host_mark("row pointers");
            spa_trace("row pointers", s);
            uint32_t *oj = nullptr, *ox = nullptr;
            e = handoff_alloc(reinterpret_cast<void**>(&oj), sizeof(uint32_t) * (size_t) total);
            if (e == hipSuccess) e = handoff_alloc(reinterpret_cast<void**>(&ox), sizeof(uint32_t) * (size_t) total);
            if (e != hipSuccess) {
                if (oj) (void) handoff_release(oj);
                return fail(SPLAHIP_ERROR, "mxm result allocation: %s", hipGetErrorString(e));
            }
            struct ResultGuard {// the result arrays go back to the hand-off cache on every early return below; disarmed when they are handed to the caller
                uint32_t **j, **x;
                bool       armed = true;
                ~ResultGuard() {
                    if (armed) {
                        if (*j) (void) handoff_release(*j);
                        if (*x) (void) handoff_release(*x);
                    }
                }
            } result_guard{&oj, &ox};
            spa_trace("result allocation", s);
            if (d_lcnt && l_nnz) hipLaunchKernelGGL(place_light_kernel, dim3((a_rows + 3) / 4), dim3(256), 0, s, a_rows, d_lstart_lo, d_lstart_hi, d_lcnt, Sj, Sx, Rp, oj, ox);
            if (n_items) {
                // items whose entries fit one accumulator block take the ranked form (two walks, no column blocks); hub rows, rows that meet the dense corner and
                // everything when B's boundary table was too large keep the blocked form
                const uint32_t wr  = std::max(1u, 245760u / blk_num) * blk_num;
                const uint32_t cap = ((160u * 1024u - 256u - 2u * (wr / 8) - (uint32_t) sizeof(uint32_t) * (3 * kTpb + 1 + kTpb / 64 + 4)) / (uint32_t) sizeof(word)) & ~63u;// what is left of the LDS
                const bool            ranked_on = d_bblk && env_u32("SPLAHIP_MXM_RANKED", 1) != 0;
                auto                  is_ranked = [&](uint32_t q) { return ranked_on && h_counts[q] <= cap && !(d_hi && items[q].row - d_i0 < d_hi); };
                uint32_t              n_rk = 0;
                for (uint32_t q = 0; q < n_items; ++q) n_rk += is_ranked(q) ? 1u : 0u;
                const uint32_t        n_bl = n_items - n_rk;
                std::vector<uint32_t> rk(4 * (size_t) n_rk), bl(4 * (size_t) n_bl);// [rows | c0 | c1 | out], the items' order kept
                for (uint32_t q = 0, ir = 0, ib = 0; q < n_items; ++q) {
                    const bool     ranked = is_ranked(q);
                    uint32_t*      dst    = ranked ? rk.data() : bl.data();
                    const size_t   n = ranked ? n_rk : n_bl, at = ranked ? ir++ : ib++;
                    dst[at] = items[q].row, dst[n + at] = items[q].c0, dst[2 * n + at] = items[q].c1, dst[3 * n + at] = h_out[q];
                }
                if (knob_env("SPLAHIP_MXM_TRACE")) {// class sizes (diagnostic)
                    u64 re = 0, be = 0, rp = 0, bp = 0, bsteps = 0;
                    for (uint32_t q = 0; q < n_items; ++q) {
                        const bool ranked = is_ranked(q);
                        (ranked ? re : be) += h_counts[q];
                        (ranked ? rp : bp) += items[q].est;
                        if (!ranked) bsteps += (items[q].c1 - items[q].c0 + blk_num - 1) / blk_num;
                    }
                    std::fprintf(stderr, "[mxm-spa] items: ranked %zu (entries %llu, products ~%llu), blocked %zu (entries %llu, products ~%llu, block steps %llu), cap %u\n",
                                 (size_t) n_rk, (unsigned long long) re, (unsigned long long) rp, (size_t) n_bl, (unsigned long long) be, (unsigned long long) bp,
                                 (unsigned long long) bsteps, cap);
                }
                uint32_t *d_rk = nullptr, *d_bl = nullptr;
                e = tmp.alloc(&d_rk, rk.size());
                if (e == hipSuccess) e = tmp.alloc(&d_bl, bl.size());
                if (e == hipSuccess) e = h2d(d_rk, rk.data(), 4 * rk.size(), s);
                if (e == hipSuccess) e = h2d(d_bl, bl.data(), 4 * bl.size(), s);
                if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm (row classes): %s", hipGetErrorString(e));
                host_mark("item classes");
                a.Rj = oj, a.Rx = ox;
                if (const uint32_t n = n_bl) {
                    a.block = blk_num, a.item_row = d_bl, a.item_c0 = d_bl + n, a.item_c1 = d_bl + 2 * (size_t) n, a.item_out = d_bl + 3 * (size_t) n, a.n_heavy = n;
                    hipLaunchKernelGGL((mxm_spa_kernel<T, Mult, ACC, true>), dim3(std::min(n, grid_num)), dim3(kTpb), lds_num, s, a, Mult(op_mult), DynBin(op_add), init);
                }
                if (const uint32_t n = n_rk) {
                    SPLAHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mxm_spa_ranked_kernel<T, Mult, ACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                    const uint32_t lds_r = cap * (uint32_t) sizeof(word) + 2 * (wr / 8) + sizeof(uint32_t) * (3 * kTpb + 1 + kTpb / 64 + 4);
                    a.item_row = d_rk, a.item_c0 = d_rk + n, a.item_c1 = d_rk + 2 * (size_t) n, a.item_out = d_rk + 3 * (size_t) n, a.n_heavy = n;
                    hipLaunchKernelGGL((mxm_spa_ranked_kernel<T, Mult, ACC>), dim3(std::min(n, (uint32_t) cus)), dim3(kTpb), lds_r, s, a, cap, wr, Mult(op_mult), DynBin(op_add), init);
                }
                tl_last_ranked = n_rk;
            }
            host_mark("numeric pass enqueued");
            SPLAHIP_TRY(hipStreamSynchronize(s));
            host_mark("numeric pass done");
            spa_trace("numeric pass", s);
            uint32_t dropped = 0;
            e                = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(&dropped, d_dropped, 4, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e == hipSuccess && dropped) {// entries that folded to init: squeeze them out (flags, scan, scatter, new row pointers)
                uint32_t *keep = nullptr, *rank = nullptr, *nj = nullptr, *nx = nullptr;
                const uint32_t n = (uint32_t) total;
                e                = tmp.alloc(&keep, (size_t) n + 1);
                if (e == hipSuccess) e = tmp.alloc(&rank, (size_t) n + 1);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(keep_flag_kernel<T>, dim3((n + 256) / 256), dim3(256), 0, s, n, ox, init, keep);
                    e = scan_u32(keep, rank, (size_t) n + 1, tmp, s);
                }
                if (e == hipSuccess) e = handoff_alloc(reinterpret_cast<void**>(&nj), sizeof(uint32_t) * (size_t) (n - dropped));
                if (e == hipSuccess) e = handoff_alloc(reinterpret_cast<void**>(&nx), sizeof(uint32_t) * (size_t) (n - dropped));
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(keep_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, keep, rank, oj, ox, nj, nx);
                    hipLaunchKernelGGL(keep_rowptr_kernel, dim3((a_rows + 256) / 256), dim3(256), 0, s, a_rows, rank, Rp);
                    e = hipStreamSynchronize(s);
                }
                if (e == hipSuccess) {
                    (void) handoff_release(oj);
                    (void) handoff_release(ox);
                    oj = nj, ox = nx, total -= dropped;
                }
            }
            if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm (row classes): %s", hipGetErrorString(e));// (result_guard releases oj / ox)
            spa_trace("compaction / done", s);
            result_guard.armed = false;
            *Rj = oj, *Rx = ox, *h_nnz = (uint32_t) total;
            tl_last_dense_h = dense_h, tl_last_heavy = n_items, tl_last_light_nnz = l_nnz;
            tl_last_dense[0] = d_i0, tl_last_dense[1] = d_k0, tl_last_dense[2] = d_j0, tl_last_dense[3] = d_hi, tl_last_dense[4] = d_hk, tl_last_dense[5] = d_hj;
            return SPLAHIP_OK;
        }

        template<typename T, int ACC>
        int spa_by_mult(uint32_t a_rows, uint32_t b_cols, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const uint32_t* Bp, const uint32_t* Bj, const void* Bx, uint32_t* Rp,
                        uint32_t** Rj, void** Rx, uint32_t* h_nnz, const u64* hP, const u64* dP, const uint32_t* hAp, const uint32_t* h_order, int op_mult, int op_add,
                        uint32_t init_bits, uint32_t b_rows_hint, const uint32_t* d_hint, uint32_t b_nnz_hint, hipStream_t s) {
            if (op_mult == SPLAHIP_MULT) return spa_run<T, StaBin<SPLAHIP_MULT>, ACC>(a_rows, b_cols, Ap, Aj, Ax, Bp, Bj, Bx, Rp, Rj, Rx, h_nnz, hP, dP, hAp, h_order, op_mult, op_add, init_bits, b_rows_hint, d_hint, b_nnz_hint, s);
            return spa_run<T, DynBin, ACC>(a_rows, b_cols, Ap, Aj, Ax, Bp, Bj, Bx, Rp, Rj, Rx, h_nnz, hP, dP, hAp, h_order, op_mult, op_add, init_bits, b_rows_hint, d_hint, b_nnz_hint, s);
        }

        __global__ __launch_bounds__(256) void max_col_kernel(uint32_t n_rows, const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ Aj, uint32_t* __restrict__ out) {
            const uint32_t nnz = Ap[n_rows];
            uint32_t       m   = 0;
            for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < nnz; k += gridDim.x * 256) m = max(m, Aj[k]);
            m = wave_max(m);
            if (lane_id() == 0) atomicMax(out, m);
        }
        __global__ __launch_bounds__(256) void rowprod_kernel(uint32_t n_rows, const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ Aj, const uint32_t* __restrict__ Bp,
                                                              u64* __restrict__ P, uint32_t* __restrict__ ids) {
            // a wavefront per row, lanes over its entries (a thread per row walked a hub's 10^5 entries alone: 3.3 ms of an R-MAT 18 call)
            const uint32_t i = blockIdx.x * (256 / 64) + threadIdx.x / 64;
            if (i >= n_rows) return;
            u64 sum = 0;
            for (uint32_t e = Ap[i] + (threadIdx.x & 63u); e < Ap[i + 1]; e += 64) sum += Bp[Aj[e] + 1] - Bp[Aj[e]];
            for (unsigned off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
            if ((threadIdx.x & 63u) == 0) P[i] = sum, ids[i] = i;
        }

    }// namespace

    // -1: the row-class form does not apply (op_add without an LDS atomic, no heavy row, switched off): the caller runs the ESC form
    int mxm_spa(int type, uint32_t a_rows, uint32_t b_cols, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const uint32_t* Bp, const uint32_t* Bj, const void* Bx, uint32_t* Rp,
                uint32_t** Rj, void** Rx, uint32_t* h_nnz, unsigned long long* h_products, int op_mult, int op_add, uint32_t init_bits, hipStream_t s) {
        if (!tl_mxm_force_esc) {
            tl_last_dense_h = tl_last_heavy = tl_last_light_nnz = 0;
            for (uint32_t& w : tl_last_dense) w = 0;
        }
        if (tl_mxm_force_esc || env_u32("SPLAHIP_MXM_SPA", 1) == 0) return -1;
        const int acc = cob_acc_kind(op_add, type);
        if (acc == COB_NONE || b_cols == 0) return -1;
        arena().reset();
        PinnedCall   pinned_call;
        ScratchScope scratch_scope(s);
        Scratch      tmp(s);
        u64*      dP    = nullptr;
        uint32_t* d_max = nullptr;
        if (tmp.alloc(&dP, a_rows) != hipSuccess || tmp.alloc(&d_max, 2) != hipSuccess) return -1;
        SPLAHIP_TRY(hipMemsetAsync(d_max, 0, 8, s));
        // rows by descending product count, ties by row number (a stable device sort: the same order std::stable_sort gave, which cost the host 4 ms per R-MAT 18 call)
        u64*      dPs   = nullptr;
        uint32_t *d_ids = nullptr, *d_order = nullptr;
        if (tmp.alloc(&dPs, a_rows) != hipSuccess || tmp.alloc(&d_ids, a_rows) != hipSuccess || tmp.alloc(&d_order, a_rows) != hipSuccess) return -1;
        hipLaunchKernelGGL(rowprod_kernel, dim3((a_rows + 3) / 4), dim3(256), 0, s, a_rows, Ap, Aj, Bp, dP, d_ids);
        {
            size_t bytes = 0;
            SPLAHIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, bytes, dP, dPs, d_ids, d_order, a_rows, 0, 64, s));
            unsigned char* t = nullptr;
            if (tmp.alloc(&t, bytes ? bytes : 4) != hipSuccess) return -1;
            SPLAHIP_TRY(rocprim::radix_sort_pairs_desc(t, bytes, dP, dPs, d_ids, d_order, a_rows, 0, 64, s));
        }
        // the number of rows of B is not an argument of splahip_mxm: the largest inner index A refers to bounds what the dense corner may touch
        hipLaunchKernelGGL(max_col_kernel, dim3(1024), dim3(256), 0, s, a_rows, Ap, Aj, d_max);
        hipLaunchKernelGGL(max_col_tail_kernel, dim3(1), dim3(1), 0, s, a_rows, Ap, d_max, Bp);
        u64*      hP    = nullptr;
        uint32_t *hAp = nullptr, *h_maxp = nullptr, *h_order = nullptr;
        SPLAHIP_TRY(d2h(&hP, dP, a_rows, s));
        SPLAHIP_TRY(d2h(&h_order, d_order, a_rows, s));
        SPLAHIP_TRY(d2h(&hAp, Ap, (size_t) a_rows + 1, s));
        SPLAHIP_TRY(d2h(&h_maxp, d_max, 2, s));
        SPLAHIP_TRY(hipStreamSynchronize(s));
        u64       total  = 0;
        bool      heavy  = false;
        const u64 kLight = light_threshold();
        for (uint32_t i = 0; i < a_rows; ++i) total += hP[i], heavy = heavy || hP[i] > kLight;
        if (!heavy) return -1;
        if (h_products) *h_products = total;
        const uint32_t h_max = *h_maxp;
        const uint32_t b_rows_hint = h_max + 1;
#define SPLAHIP_SPA(TT, AA) return spa_by_mult<TT, AA>(a_rows, b_cols, Ap, Aj, Ax, Bp, Bj, Bx, Rp, Rj, Rx, h_nnz, hP, dP, hAp, h_order, op_mult, op_add, init_bits, b_rows_hint, d_max, h_maxp[1], s)
        switch (type) {
            case SPLAHIP_FLOAT:
                if (acc == COB_ADD) SPLAHIP_SPA(float, COB_ADD);
                if (acc == COB_LOR) SPLAHIP_SPA(float, COB_LOR);
                if (acc == COB_LAND) SPLAHIP_SPA(float, COB_LAND);
                break;
            case SPLAHIP_INT:
                if (acc == COB_ADD) SPLAHIP_SPA(int32_t, COB_ADD);
                if (acc == COB_MIN) SPLAHIP_SPA(int32_t, COB_MIN);
                if (acc == COB_MAX) SPLAHIP_SPA(int32_t, COB_MAX);
                if (acc == COB_OR) SPLAHIP_SPA(int32_t, COB_OR);
                if (acc == COB_AND) SPLAHIP_SPA(int32_t, COB_AND);
                if (acc == COB_XOR) SPLAHIP_SPA(int32_t, COB_XOR);
                if (acc == COB_LOR) SPLAHIP_SPA(int32_t, COB_LOR);
                if (acc == COB_LAND) SPLAHIP_SPA(int32_t, COB_LAND);
                break;
            case SPLAHIP_UINT:
                if (acc == COB_ADD) SPLAHIP_SPA(uint32_t, COB_ADD);
                if (acc == COB_MIN) SPLAHIP_SPA(uint32_t, COB_MIN);
                if (acc == COB_MAX) SPLAHIP_SPA(uint32_t, COB_MAX);
                if (acc == COB_OR) SPLAHIP_SPA(uint32_t, COB_OR);
                if (acc == COB_AND) SPLAHIP_SPA(uint32_t, COB_AND);
                if (acc == COB_XOR) SPLAHIP_SPA(uint32_t, COB_XOR);
                if (acc == COB_LOR) SPLAHIP_SPA(uint32_t, COB_LOR);
                if (acc == COB_LAND) SPLAHIP_SPA(uint32_t, COB_LAND);
                break;
        }
#undef SPLAHIP_SPA
        return -1;
    }

}// namespace splahip

// which row classes the last splahip_mxm of this thread used: heavy rows (LDS accumulator form), entries produced by the light rows' ESC call,
// side of the dense corner that ran on the matrix cores (0 = none)
namespace splahip {
    // the pinned staging chunks of every host thread that is not inside an mxm call go back to the driver; arenas of threads that have exited are deleted
    void mxm_pinned_trim() {
        auto& r = pinned_registry();
        std::lock_guard<std::mutex> rg(r.m);
        for (size_t k = 0; k < r.all.size();) {
            PinnedArena* a    = r.all[k];
            bool         drop = false;
            {
                std::lock_guard<std::mutex> g(a->m);
                if (a->in_call == 0) {
                    for (auto& c : a->chunks) (void) hipHostFree(c.p);
                    a->chunks.clear();
                    drop = a->orphan;
                }
            }
            if (drop) {
                delete a;
                r.all.erase(r.all.begin() + (long) k);
            } else
                ++k;
        }
    }
}// namespace splahip

extern "C" int splahip_mxm_last_dense_ms(float* gemm_ms) {
    if (gemm_ms) *gemm_ms = splahip::tl_last_dense_h ? splahip::tl_last_gemm_ms : 0.0f;
    return SPLAHIP_OK;
}

extern "C" int splahip_mxm_last_info(uint32_t* n_heavy_rows, uint32_t* light_nnz, uint32_t* dense_h) {
    if (n_heavy_rows) *n_heavy_rows = splahip::tl_last_heavy;
    if (light_nnz) *light_nnz = splahip::tl_last_light_nnz;
    if (dense_h) *dense_h = splahip::tl_last_dense_h;
    return SPLAHIP_OK;
}

// where the last mxm of the calling thread found its dense window (rows / inner indices / columns of A x B that went through the matrix cores): six words
// i0, k0, j0, rows, inner, columns; rows == 0: none
extern "C" int splahip_mxm_last_dense(uint32_t* window6) {
    SPLAHIP_REQUIRE(window6 != nullptr, "null argument");
    for (int k = 0; k < 6; ++k) window6[k] = splahip::tl_last_dense[k];
    return SPLAHIP_OK;
}

#ifdef SPLAHIP_SPA_TIMING
// DIAGNOSTIC BUILD ONLY: per-workgroup phase ticks of the last blocked numeric launch (4 x 1024 64-bit words)
extern "C" int splahip_debug_spa_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(splahip::g_spa_t), sizeof(unsigned long long) * 4 * 1024) == hipSuccess ? SPLAHIP_OK : SPLAHIP_ERROR;
}
#endif
