// mxv_rows.hpp — masked SpMV over the SELECTED rows only (round 4; reworked in round 6): what a mask that selects few entries runs instead of streaming the whole matrix.
//
// Semantics: reference Algo_mxv_masked_cpu<T>::execute, src/cpu/cpu_mxv.hpp:88-103 — an unselected row costs the reference nothing (:89-92) and its own perf
// test selects every 10th row (tests/test_mxv.cpp:91-130).  The column-ordered blocks (mxv_cob.hpp) stream every entry of a range whatever the mask, and the
// wave-tile CSR kernel still visits every work item.  Here, by row length:
//   rows_compact_kernel  one streaming pass over mask and Ap: r = init everywhere; the selected rows of 1 .. 32 entries are appended to a list (an LDS buffer per
//                        workgroup, one global atomic per flush), the selected MEDIUM rows (33 .. 508 entries) to a second list (from the end of the same array)
//   rows_short_kernel    the listed short rows, 64 per wave: coalesced Aj / Ax quads, gathers of v, products staged in LDS, a lane folds its row strictly left to
//                        right from init (the reference's order: bit-exact, fp32 sums included)
//   rows_task_kernel     a wave per task: a selected medium row, or a segment item (<= 1020 entries) of a selected long row — the whole task requested at once
//   long rows            the plan's rows of more than 508 entries are listed once per plan (rows_long_list_kernel); per call rows_long_select_kernel appends the
//                        segment items of the SELECTED ones; rows of several segments are finished by the wave-tile plan's combine kernel
// Round 6 (profiles/r06_notes.md): until then a medium row was folded by ONE lane of a 64-row batch — a batch of 64 neighbouring medium rows was a pole of ~300 us —
// and the list pass searched the plan (17 dependent loads) once per selected long row: a random 10 % of R-MAT 23's rows took 416 us against 300 on the blocks.
// A version that cut the work by selected ENTRIES (tiles of a virtual stream of selected 16-byte quads, rows re-derived from mask / Ap per tile) was built, is
// correct and was slower on every mask (177 us for the tiles of that call, 134 against 86 us on light rows): a 256-thread tile is a chain of barriers, the waves of
// this form overlap their phases freely.
// op_add must be associative and commutative (a task's lanes take its entries strided) — the same calls the wave-tile kernel serves.  Every kernel takes the
// mask probe's words as its gate and returns at once unless word 3 says the row kernels serve the call (cob.hip: cob_mask_probe_kernel).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_utils.hpp"
#include "mask_probe.hpp"
#include "mxv_kernels.hpp"
#include "ops.hpp"

namespace splahip {

    // The lists of a call live in kRowsRegions areas of the row space (area g = the rows of the compaction workgroups [g, g + 1) * blocks / kRowsRegions): an area's short
    // rows from its start upwards, its medium rows from its end downwards, its two counts one 8-byte word on a 128-byte line of its own (ctl[32 g], ctl[32 g + 1]) — the
    // compaction workgroups all flush at about the same time, same-address atomics are served one by one (~6 ns), and one counter for the whole matrix was 1 K of them in
    // a row.  The consumers turn a list index into (area, offset) with the eight running sums (rows_region_sums).
    constexpr uint32_t kRowsRegions = 8, kRowsCntStride = 32, kRowsCtlSet = kRowsRegions * kRowsCntStride;// (words of one set: [3] the segment items listed, [4..7] of the first set the gate)
    struct RowsSums {
        uint32_t pre[kRowsRegions + 1];
        __device__ __forceinline__ uint32_t region_of(uint32_t i) const {
            uint32_t g = 0;
#pragma unroll
            for (uint32_t k = 1; k < kRowsRegions; ++k) g += i >= pre[k] ? 1u : 0u;
            return g;
        }
        // (the running sum in front of region g: a select chain, the index is not a compile-time constant)
        __device__ __forceinline__ uint32_t before(uint32_t g) const {
            uint32_t b = 0;
#pragma unroll
            for (uint32_t k = 1; k < kRowsRegions; ++k) b = g >= k ? pre[k] : b;
            return b;
        }
    };
    __device__ __forceinline__ RowsSums rows_region_sums(const uint32_t* __restrict__ ctl, uint32_t which) {// which: 0 short rows, 1 medium rows
        RowsSums s;
        uint32_t c[kRowsRegions];
#pragma unroll
        for (uint32_t g = 0; g < kRowsRegions; ++g) c[g] = ctl[g * kRowsCntStride + which];
        s.pre[0] = 0;
#pragma unroll
        for (uint32_t g = 0; g < kRowsRegions; ++g) s.pre[g + 1] = s.pre[g] + c[g];
        return s;
    }
    constexpr uint32_t kRowsShortMax = 508;// = MxvCfg's longest row of a run item: longer rows are segment items of the wave-tile plan
    constexpr uint32_t kRowsLaneMax  = 32; // rows up to this length are folded by one lane of a 64-row batch (rows_short_kernel); up to kRowsShortMax: a wave each

    // One streaming pass over mask and Ap, 8 rows per thread and step (16-byte loads): r = init for every row (the row kernels overwrite the selected ones
    // behind this launch); list[0 .. count[0]) = the selected rows with 1 .. kRowsLaneMax entries, in no particular order (clusters of neighbouring rows: an LDS
    // buffer per workgroup, one global atomic per flush); list[n_rows - 1 - i], i < count[1] = the selected MEDIUM rows (count[0 .. 1] are one 8-byte word for the last flush) (kRowsLaneMax + 1 .. kRowsShortMax entries:
    // a wavefront each, rows_task_kernel), one atomic per wave.  Longer rows are the plan's business (rows_long_select_kernel).
    template<typename T, bool SMALL = false>
    __device__ __forceinline__ void rows_compact_body(const uint32_t vb, const uint32_t vg, uint32_t n_rows, const uint32_t* __restrict__ Ap, const T* __restrict__ mask, int op_select, T init,
                                                               T* __restrict__ r, uint32_t* __restrict__ list, uint32_t* __restrict__ ctl, uint32_t area,
                                                               const uint32_t* __restrict__ gate) {
        if (gate[3] == 0u) return;
        // this workgroup's area of the lists (vg is a multiple of kRowsRegions; area = its workgroups' rows): the area's count pair, first and last list slot
        const uint32_t  region     = vb / (vg / kRowsRegions);
        uint32_t* const count      = ctl + region * kRowsCntStride;
        const uint32_t  area_first = region * area, area_last = min(n_rows, (region + 1u) * area) - 1u;
        constexpr uint32_t kBuf = SMALL ? 2048 : 4096;// (SMALL: 26 KB of LDS per workgroup instead of 37 — six workgroups per CU instead of four)
        __shared__ uint32_t s_buf[kBuf + 2048];
        constexpr uint32_t  kBufM = SMALL ? 512 : 1024;
        __shared__ uint32_t s_med[kBufM + 2048];// the medium rows, buffered the same way (a global atomic per wave and step was ~5 K same-address atomics of ~15 ns each)
        __shared__ uint32_t s_n, s_base, s_nm, s_basem;
        if (threadIdx.x == 0) s_n = 0, s_nm = 0;
        __syncthreads();
        const uint32_t per = ((n_rows + vg - 1) / vg + 2047u) & ~2047u;// a contiguous share of the rows per workgroup, whole steps of 2048 rows
        const uint32_t b0 = vb * per, b1 = min(n_rows, b0 + per);
        const uint4    iv = make_uint4(to_bits(init), to_bits(init), to_bits(init), to_bits(init));
        // the NEXT step's words are requested before this step's are looked at (a step is loads -> scans -> LDS -> barriers: without this the workgroup's
        // memory requests and its barriers take turns); a thread whose 8 rows are not all inside the share and the matrix loads them one by one when their step comes
        auto whole = [&](uint32_t base) { return base < b1 && base + threadIdx.x * 8 + 8 <= b1 && base + threadIdx.x * 8 + 8 < n_rows; };
        uint4 na0 = make_uint4(0, 0, 0, 0), na1 = na0, nm0 = na0, nm1 = na0;
        uint32_t na8 = 0;
        auto request = [&](uint32_t base) {
            if (whole(base)) {// (Ap and mask are 16-byte aligned arrays)
                const uint32_t row0 = base + threadIdx.x * 8;
                na0 = *reinterpret_cast<const uint4*>(Ap + row0), na1 = *reinterpret_cast<const uint4*>(Ap + row0 + 4);
                nm0 = *reinterpret_cast<const uint4*>(mask + row0), nm1 = *reinterpret_cast<const uint4*>(mask + row0 + 4);
                na8 = Ap[row0 + 8];
            }
        };
        request(b0);
        for (uint32_t base = b0; base < b1; base += 2048) {
            const uint32_t row0 = base + threadIdx.x * 8;
            uint32_t       ap[9];
            uint32_t       mk[8];
            if (whole(base)) {
                const uint4 a0 = na0, a1 = na1, m0 = nm0, m1 = nm1;
                ap[0] = a0.x, ap[1] = a0.y, ap[2] = a0.z, ap[3] = a0.w, ap[4] = a1.x, ap[5] = a1.y, ap[6] = a1.z, ap[7] = a1.w, ap[8] = na8;
                mk[0] = m0.x, mk[1] = m0.y, mk[2] = m0.z, mk[3] = m0.w, mk[4] = m1.x, mk[5] = m1.y, mk[6] = m1.z, mk[7] = m1.w;
                request(base + 2048);
                *reinterpret_cast<uint4*>(r + row0)     = iv;
                *reinterpret_cast<uint4*>(r + row0 + 4) = iv;
            } else {
                request(base + 2048);
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j) {
                    const bool in = row0 + j < b1;
                    ap[j]         = in ? Ap[row0 + j] : 0u;
                    mk[j]         = in ? to_bits(mask[row0 + j]) : 0u;
                    if (in) r[row0 + j] = init;
                }
                ap[8] = row0 + 8 <= b1 ? Ap[row0 + 8] : (row0 < b1 ? Ap[b1] : 0u);
                for (uint32_t j = 0; j < 8; ++j)
                    if (row0 + j >= b1) ap[j] = ap[8];// (rows behind the share: length 0)
            }
            uint32_t take = 0, cnt = 0, med = 0, n_med = 0;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t d = ap[j + 1] - ap[j];
                if (row0 + j < b1 && d != 0 && d <= kRowsShortMax && eval_sel<T>(op_select, from_bits<T>(mk[j]))) {
                    if (d <= kRowsLaneMax) take |= 1u << j, ++cnt;
                    else
                        med |= 1u << j, ++n_med;
                }
            }
            {// the medium rows: positions in the workgroup's second buffer (one LDS atomic per wave)
                const uint32_t inc_m = wave_inclusive_sum(n_med);
                const uint32_t tot_m = (uint32_t) __builtin_amdgcn_readlane((int) inc_m, (int) kWave - 1);
                if (tot_m) {
                    uint32_t base_m = 0;
                    if (lane_id() == 0) base_m = atomicAdd(&s_nm, tot_m);
                    base_m      = (uint32_t) __builtin_amdgcn_readfirstlane((int) base_m);
                    uint32_t at = base_m + inc_m - n_med;
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j)
                        if (med & (1u << j)) s_med[at++] = row0 + j;
                }
            }
            // positions: the wave's counts scanned, one LDS atomic per wave
            uint32_t inc = cnt;
#pragma unroll
            for (uint32_t off = 1; off < kWave; off <<= 1) {
                const uint32_t y = __shfl_up(inc, off, kWave);
                if (lane_id() >= off) inc += y;
            }
            uint32_t wbase = 0;
            if (lane_id() == kWave - 1 && inc) wbase = atomicAdd(&s_n, inc);
            wbase        = __shfl(wbase, kWave - 1, kWave);
            uint32_t pos = wbase + inc - cnt;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j)
                if (take & (1u << j)) s_buf[pos++] = row0 + j;
            __syncthreads();
            if (base + 2048 >= b1) {// the share's last step: both buffers go out behind ONE global atomic on the pair (short rows | medium rows) — the workgroups
                                    // all get here at about the same time, and same-address atomics are served one by one (~6 ns each)
                const uint32_t n = s_n, nm = s_nm;
                if (threadIdx.x == 0 && (n | nm)) {
                    const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(count), (unsigned long long) n | ((unsigned long long) nm << 32));
                    s_base = (uint32_t) old, s_basem = (uint32_t) (old >> 32);
                }
                __syncthreads();
                for (uint32_t i = threadIdx.x; i < n; i += 256) list[area_first + s_base + i] = s_buf[i];
                for (uint32_t i = threadIdx.x; i < nm; i += 256) list[area_last - (s_basem + i)] = s_med[i];
            } else {
                if (s_nm >= kBufM) {// flush the medium rows: stored from the END of the list downwards (uniform: s_nm is read behind the barrier)
                    const uint32_t n = s_nm;
                    if (threadIdx.x == 0) s_basem = atomicAdd(count + 1, n);
                    __syncthreads();
                    for (uint32_t i = threadIdx.x; i < n; i += 256) list[area_last - (s_basem + i)] = s_med[i];
                    __syncthreads();
                    if (threadIdx.x == 0) s_nm = 0;
                    __syncthreads();
                }
                if (s_n >= kBuf) {// flush (uniform: s_n is read behind the barrier)
                    const uint32_t n = s_n;
                    if (threadIdx.x == 0) s_base = atomicAdd(count, n);
                    __syncthreads();
                    for (uint32_t i = threadIdx.x; i < n; i += 256) list[area_first + s_base + i] = s_buf[i];
                    __syncthreads();
                    if (threadIdx.x == 0) s_n = 0;
                    __syncthreads();
                }
            }
        }
    }

    // The listed rows, a wave at a time: 64 listed rows' bounds arrive in one round trip (lane l = row l); the wave then takes as many of them as fit its LDS stage
    // (kRowsStage entries, rows padded to 16-byte quads; a listed row holds at most kRowsLaneMax entries: two rounds at most), requests ALL their Aj / Ax quads at once (a lane's quads belong to whichever rows the scan of the quad
    // counts says), gathers v for all of them, stages the products, and lane l folds row l's products out of LDS — strictly left to right from init, the
    // reference's own order (cpu_mxv.hpp:94-99): bit-exact for every op_add, fp32 sums included.  Three round trips per ~1000 entries instead of two per row.
    constexpr uint32_t kRowsStage = 1024;// entries of a wave's product stage (256 quads: 4 per lane)
    template<typename T, class Mult, class Add>
    __device__ __forceinline__ void rows_short_body(const uint32_t vb, const uint32_t vg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, uint32_t area,
                                                             const uint32_t* __restrict__ Ap,
                                                             const uint32_t* __restrict__ Aj, const uint32_t* __restrict__ Ax, const T* __restrict__ v, T* __restrict__ r, Mult mult,
                                                             Add add, T init, uint32_t nnz, const uint32_t* __restrict__ gate) {
        if (gate[3] == 0u) return;
        __shared__ T        s_prod_all[256 / kWave][kRowsStage];
        __shared__ uint32_t s_q_all[256 / kWave][kWave + 1];// exclusive scan of the rows' quad counts
        __shared__ uint32_t s_a_all[256 / kWave][kWave];    // first (16-byte aligned) entry of the rows' quads
        const RowsSums sums = rows_region_sums(count, 0);
        const uint32_t n    = sums.pre[kRowsRegions];
        const uint32_t lane = lane_id(), wv = threadIdx.x / kWave;
        T* const        s_prod = s_prod_all[wv];
        uint32_t* const s_q    = s_q_all[wv];
        uint32_t* const s_a    = s_a_all[wv];
        const uint32_t  wave = vb * (256 / kWave) + wv, n_waves = vg * (256 / kWave);
        for (uint32_t i0 = wave * kWave; i0 < n; i0 += n_waves * kWave) {
            uint32_t my_row = 0, k0 = 0, k1 = 0;
            if (i0 + lane < n) {
                const uint32_t g = sums.region_of(i0 + lane);
                my_row = list[g * area + (i0 + lane - sums.before(g))];
                k0     = Ap[my_row];
                k1     = Ap[my_row + 1];
            }
            const uint32_t a0 = k0 & ~3u;                            // 16-byte aligned origin of the row's quads
            const uint32_t nq = k1 > k0 ? (k1 - a0 + 3) >> 2 : 0;   // quads of the row (<= 9)
            uint32_t       done = 0;                                 // rows of this batch already folded
            const uint32_t here = min(kWave, n - i0);
            while (done < here) {
                // rows [done, upto) fit the stage: scan the quad counts of the rows not yet done
                uint32_t inc = lane >= done && lane < here ? nq : 0;
#pragma unroll
                for (uint32_t off = 1; off < kWave; off <<= 1) {
                    const uint32_t y = __shfl_up(inc, off, kWave);
                    if (lane >= off) inc += y;
                }
                const unsigned long long fits = __ballot(lane >= done && lane < here && inc <= kRowsStage / 4);
                const uint32_t upto = done + (uint32_t) __popcll(fits);// (a single row always fits: nq <= 128; the fitting rows are a prefix of the rest)
                const uint32_t ex   = inc - (lane >= done && lane < here ? nq : 0);
                s_q[lane]           = ex;
                s_a[lane]           = a0;
                const uint32_t total = __shfl(inc, (int) (upto - 1), kWave);// quads of the rows [done, upto)
                if (lane == 0) s_q[kWave] = total;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // the quads, 4 per lane: which row does virtual quad vq belong to (largest row in [done, upto) with s_q[row] <= vq)
                uint4    jq[4], xq[4];
                uint32_t qrow[4], qk[4];
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) {
                    const uint32_t vq = lane + u * kWave;
                    qrow[u]           = 0xFFFFFFFFu;
                    if (vq < total) {
                        uint32_t lo = done, hi = upto;// s_q[lo] <= vq < s_q[hi]
                        while (hi - lo > 1) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (s_q[mid] <= vq) lo = mid;
                            else
                                hi = mid;
                        }
                        qrow[u] = lo;
                        qk[u]   = s_a[lo] + 4 * (vq - s_q[lo]);// (through LDS, not a shuffle: the lanes diverge here)
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u)
                    if (qrow[u] != 0xFFFFFFFFu) {
                        if (qk[u] + 4 <= nnz) {
                            jq[u] = *reinterpret_cast<const uint4*>(Aj + qk[u]);
                            xq[u] = *reinterpret_cast<const uint4*>(Ax + qk[u]);
                        } else {// the last quad of the arrays may be cut short: its words one by one (what lies behind them is never folded)
                            jq[u] = make_uint4(Aj[qk[u]], qk[u] + 1 < nnz ? Aj[qk[u] + 1] : 0u, qk[u] + 2 < nnz ? Aj[qk[u] + 2] : 0u, 0u);
                            xq[u] = make_uint4(Ax[qk[u]], qk[u] + 1 < nnz ? Ax[qk[u] + 1] : 0u, qk[u] + 2 < nnz ? Ax[qk[u] + 2] : 0u, 0u);
                        }
                    }
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u)
                    if (qrow[u] != 0xFFFFFFFFu) {
                        const uint32_t jj[4] = {jq[u].x, jq[u].y, jq[u].z, jq[u].w};
                        const uint32_t xx[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w};
                        T              vv[4];
#pragma unroll
                        for (uint32_t e = 0; e < 4; ++e) vv[e] = v[jj[e]];
                        const uint32_t vq = lane + u * kWave;
#pragma unroll
                        for (uint32_t e = 0; e < 4; ++e) s_prod[4 * vq + e] = mult(from_bits<T>(xx[e]), vv[e]);
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane >= done && lane < upto) {
                    T              sum = init;
                    const uint32_t s0  = 4 * ex + (k0 - a0), s1 = s0 + (k1 - k0);
                    for (uint32_t k = s0; k < s1; ++k) sum = add(sum, s_prod[k]);
                    r[my_row] = sum;
                }
                done = upto;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }

    // The plan's long rows, listed once per plan: long_row[i] = a row of more than kRowsShortMax entries, long_first[i] = its first segment item (in no particular order).
    __global__ __launch_bounds__(256) void rows_long_list_kernel(const uint32_t* __restrict__ item_row, const uint32_t* __restrict__ item_k0, uint32_t n_items,
                                                                 const uint32_t* __restrict__ Ap, uint32_t seg_flag, uint32_t* __restrict__ count, uint32_t* __restrict__ long_row,
                                                                 uint32_t* __restrict__ long_first) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i >= n_items) return;
        const uint32_t e0 = item_row[i];
        if (!(e0 & seg_flag)) return;
        const uint32_t row = e0 & ~seg_flag;
        if (item_k0[i] != Ap[row]) return;// not the first segment of its row
        const uint32_t at = atomicAdd(count, 1u);
        long_row[at] = row, long_first[at] = i;
    }

    // per call: the segment items of the SELECTED long rows -> long_items[0 .. *n_listed)
    template<typename T>
    __device__ __forceinline__ void rows_long_select_body(const uint32_t vb, const uint32_t vg, uint32_t n_long, const uint32_t* __restrict__ long_row, const uint32_t* __restrict__ long_first,
                                                                   const uint32_t* __restrict__ Ap, const T* __restrict__ mask, int op_select, uint32_t seg_len,
                                                                   uint32_t* __restrict__ long_items, uint32_t* __restrict__ n_listed, const uint32_t* __restrict__ gate) {
        if (gate[3] == 0u) return;
        const uint32_t i   = vb * 256 + threadIdx.x;
        const uint32_t row = i < n_long ? long_row[i] : 0u;
        const bool     sel = i < n_long && eval_sel<T>(op_select, mask[row]);
        const uint32_t nseg = sel ? (Ap[row + 1] - Ap[row] + seg_len - 1) / seg_len : 0u, first = sel ? long_first[i] : 0u;
        // one global atomic per WAVE (the counts scanned across its lanes): same-address atomics are served one by one, ~6 ns each
        const uint32_t inc = wave_inclusive_sum(nseg), total = (uint32_t) __builtin_amdgcn_readlane((int) inc, (int) kWave - 1);
        if (total == 0u) return;
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(n_listed, total);
        const uint32_t at = (uint32_t) __builtin_amdgcn_readfirstlane((int) base) + inc - nseg;
        for (uint32_t g = 0; g < nseg; ++g) long_items[at + g] = (first + g) | (nseg == 1u ? 0x80000000u : 0u);// (bit 31: the row's only segment — rows_task_body finishes the row)
    }

    // The wave tasks of a call, a wave per task in turn (a fixed grid: the lists' lengths are only known on the device): first the listed segment items of the
    // selected long rows, then the selected medium rows (from the end of `list` downwards).  The whole task — at most 1020 entries, 256 aligned quads — is requested
    // at once (4 quads per lane, then their 16 gathers); a lane folds its quads in entry order, the lanes are joined in order (op_add is associative and commutative
    // on this path); a segment's partial goes to the combine kernel, a medium row or a row of one segment is finished here (r = add(init, fold)).
    template<typename T, class Mult, class Add>
    __device__ __forceinline__ void rows_task_body(const uint32_t vb, const uint32_t vg, const uint32_t* __restrict__ long_items, const uint32_t* __restrict__ ctl, const uint32_t* __restrict__ list, uint32_t n_rows, uint32_t area,
                                                            const uint32_t* __restrict__ item_row,
                                                            const uint32_t* __restrict__ item_k0, const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ Aj,
                                                            const uint32_t* __restrict__ Ax, const T* __restrict__ v, T* __restrict__ r, T* __restrict__ partial, Mult mult,
                                                            Add add_raw, T init, uint32_t seg_flag, uint32_t seg_len, uint32_t nnz, const uint32_t* __restrict__ gate) {
        if (gate[3] == 0u) return;
        const Reassoc<Add> add(add_raw);
        const RowsSums     sums    = rows_region_sums(ctl, 1);
        const uint32_t     n_items = ctl[3], n = n_items + sums.pre[kRowsRegions];
        const uint32_t     lane = lane_id(), stride = vg * (256 / kWave);
        uint32_t           w = vb * (256 / kWave) + threadIdx.x / kWave;
        if (w >= n) return;
        // A task's header is two dependent round trips (its list word, then its bounds) in front of the two of its entries (quads, gathers): the NEXT task's header
        // is requested under this task's entries — its list word before the quads, its bounds before the gathers (loads return in order: they are there when the
        // quads / gathers are).  A segment item that is its row's only segment carries bit 31 (rows_long_select_body): no look at Ap to tell.
        auto list_word = [&](uint32_t t) {
            if (t < n_items) return long_items[t];
            const uint32_t m = t - n_items, g = sums.region_of(m);// the m-th medium row: area g, counted from the area's end
            return list[min(n_rows, (g + 1u) * area) - 1u - (m - sums.before(g))];
        };
        struct Head {
            uint32_t row, k0, k1, item;
            bool     whole;// the task is the whole row: finished here
        };
        auto bounds = [&](uint32_t t, uint32_t word) {
            Head h;
            if (t >= n_items) h.row = word, h.k0 = Ap[word], h.k1 = Ap[word + 1], h.item = 0u, h.whole = true;
            else {
                h.item = word & 0x7FFFFFFFu, h.whole = (word >> 31) != 0u;
                h.row = item_row[h.item] & ~seg_flag, h.k0 = item_k0[h.item], h.k1 = item_k0[h.item + 1];
            }
            return h;
        };
        Head cur = bounds(w, list_word(w));
        for (;;) {
            const uint32_t wn   = w + stride;
            const bool     more = wn < n;
            const uint32_t next_word = more ? list_word(wn) : 0u;
            const uint32_t k0 = cur.k0, k1 = cur.k1;
            const uint32_t a0 = k0 & ~3u, nq = k1 > a0 ? (k1 - a0 + 3) >> 2 : 0;// <= 256
            uint4          jq[4], xq[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
                if (lane + u * kWave < nq) load_quad(Aj, Ax, a0 + 4 * (lane + u * kWave), nnz, jq[u], xq[u]);
            Head nxt = cur;
            if (more) nxt = bounds(wn, next_word);
            T    part  = init;
            bool valid = false;
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
                if (lane + u * kWave < nq) {
                    const uint32_t kq    = a0 + 4 * (lane + u * kWave);
                    const uint32_t jj[4] = {jq[u].x, jq[u].y, jq[u].z, jq[u].w};
                    const uint32_t xx[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w};
                    T              vv[4];
#pragma unroll
                    for (uint32_t e = 0; e < 4; ++e) vv[e] = v[jj[e]];
#pragma unroll
                    for (uint32_t e = 0; e < 4; ++e) {
                        const bool take = kq + e >= k0 && kq + e < k1;// only the first / last quad of a segment are partial
                        const T    p    = mult(from_bits<T>(xx[e]), vv[e]);
                        const T    both = add(part, p);
                        part            = take ? (valid ? both : p) : part;
                        valid           = valid || take;
                    }
                }
            // (a lane holds the quads lane, lane + 64, ...: joined in lane order the fold is a re-association, as everywhere on this path)
            wave_reduce_ordered(part, valid, add);
            if (lane == 0) {
                if (cur.whole) r[cur.row] = valid ? add(init, part) : init;// a medium row, or a row of one segment: finished here (the combine kernel skips it)
                else
                    partial[cur.item] = part;
            }
            if (!more) break;
            w = wn, cur = nxt;
        }
    }

    // ---- the launches of a call (mxv.hip: launch_rows).  The two passes that read the mask and do not depend on one another share a launch — the first workgroups
    // list the segment items of the selected long rows, the others compact the selected short / medium rows (+ r = init): 41.5 us against 36 + 11 (R-MAT 23, a random
    // 10 % of the rows).  Measured and NOT done: the short rows' batches and the wave tasks in one launch (even / odd workgroups): 143.6 us against 42 + 83 — both are
    // bound by the same line requests, and each kind of workgroup then carries the other's registers and LDS ("50 %, no hub row", where there are no tasks: 234 against
    // 170 us) — tools/gpu_r6_rows3.sh.
    template<typename T, bool SMALL = false>
    __global__ __launch_bounds__(256) void rows_select_kernel(MaskProbeArgs probe, uint32_t select_blocks, uint32_t n_rows, const uint32_t* __restrict__ Ap, const T* __restrict__ mask,
                                                              int op_select, T init, T* __restrict__ r, uint32_t* __restrict__ list, uint32_t* __restrict__ ctl,
                                                              uint32_t* __restrict__ ctl_next, uint32_t area, uint32_t n_long, const uint32_t* __restrict__ long_row,
                                                              const uint32_t* __restrict__ long_first, uint32_t seg_len, uint32_t* __restrict__ long_items,
                                                              const uint32_t* __restrict__ gate) {
        // the list lengths of the NEXT call over the plan start at zero (two sets of counters, used in turn: nobody else touches the other set during this launch)
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x < kRowsRegions) {
            ctl_next[threadIdx.x * kRowsCntStride] = 0u, ctl_next[threadIdx.x * kRowsCntStride + 1u] = 0u;
            if (threadIdx.x == 0) ctl_next[3] = 0u;
        }
        if (blockIdx.x < probe.blocks) {// this call's mask probe (mask_probe.hpp): a hint for the next call, no part of this one
            mask_probe_body<T>(blockIdx.x, probe.blocks, probe, Ap, mask, op_select);
            return;
        }
        const uint32_t b = blockIdx.x - probe.blocks, g = gridDim.x - probe.blocks;
        if (b < select_blocks) rows_long_select_body<T>(b, select_blocks, n_long, long_row, long_first, Ap, mask, op_select, seg_len, long_items, ctl + 3, gate);
        else
            rows_compact_body<T, SMALL>(b - select_blocks, g - select_blocks, n_rows, Ap, mask, op_select, init, r, list, ctl, area, gate);
    }
    // the short rows' batches with the combine pass (mxv_kernels.hpp: the rows of several segments, from their partials) as the launch's first workgroups: launched BEHIND
    // the wave tasks, whose partials it folds — the combine pass as a launch of its own was 5 us at the end of every call with a selected long row
    template<class C, typename T, class Mult, class Add>
    __global__ __launch_bounds__(256) void rows_short_combine_kernel(uint32_t combine_blocks, const uint32_t* __restrict__ item_row, const uint32_t* __restrict__ item_k0, uint32_t n_items,
                                                                     const T* __restrict__ mask, const T* __restrict__ partial, int op_select, const uint32_t* __restrict__ list,
                                                                     const uint32_t* __restrict__ count, uint32_t area, const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ Aj,
                                                                     const uint32_t* __restrict__ Ax, const T* __restrict__ v, T* __restrict__ r, Mult mult, Add add, T init, uint32_t nnz,
                                                                     const uint32_t* __restrict__ gate) {
        if (blockIdx.x < combine_blocks) {
            if (gate[3] == 0u) return;
            const Reassoc<Add> radd(add);
            spmv_combine_item<C, T, Add, false>(blockIdx.x * 256 + threadIdx.x, item_row, item_k0, n_items, Ap, mask, r, nullptr, partial, nullptr, radd, op_select, init);
            return;
        }
        rows_short_body<T, Mult, Add>(blockIdx.x - combine_blocks, gridDim.x - combine_blocks, list, count, area, Ap, Aj, Ax, v, r, mult, add, init, nnz, gate);
    }
    // (the passes as launches of their own; compaction and selection apart: SPLAHIP_ROWS_FUSED=0, A/B runs)
    template<typename T>
    __global__ __launch_bounds__(256) void rows_compact_kernel(uint32_t n_rows, const uint32_t* __restrict__ Ap, const T* __restrict__ mask, int op_select, T init, T* __restrict__ r,
                                                               uint32_t* __restrict__ list, uint32_t* __restrict__ count, uint32_t* __restrict__ ctl_next, uint32_t area,
                                                               const uint32_t* __restrict__ gate) {
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x < kRowsRegions) {
            ctl_next[threadIdx.x * kRowsCntStride] = 0u, ctl_next[threadIdx.x * kRowsCntStride + 1u] = 0u;
            if (threadIdx.x == 0) ctl_next[3] = 0u;
        }
        rows_compact_body<T>(blockIdx.x, gridDim.x, n_rows, Ap, mask, op_select, init, r, list, count, area, gate);
    }
    template<typename T>
    __global__ __launch_bounds__(256) void rows_long_select_kernel(uint32_t n_long, const uint32_t* __restrict__ long_row, const uint32_t* __restrict__ long_first,
                                                                   const uint32_t* __restrict__ Ap, const T* __restrict__ mask, int op_select, uint32_t seg_len,
                                                                   uint32_t* __restrict__ long_items, uint32_t* __restrict__ n_listed, const uint32_t* __restrict__ gate) {
        rows_long_select_body<T>(blockIdx.x, gridDim.x, n_long, long_row, long_first, Ap, mask, op_select, seg_len, long_items, n_listed, gate);
    }
    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(256) void rows_short_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, uint32_t area, const uint32_t* __restrict__ Ap,
                                                             const uint32_t* __restrict__ Aj, const uint32_t* __restrict__ Ax, const T* __restrict__ v, T* __restrict__ r, Mult mult,
                                                             Add add, T init, uint32_t nnz, const uint32_t* __restrict__ gate) {
        rows_short_body<T, Mult, Add>(blockIdx.x, gridDim.x, list, count, area, Ap, Aj, Ax, v, r, mult, add, init, nnz, gate);
    }
    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(256) void rows_task_kernel(const uint32_t* __restrict__ long_items, const uint32_t* __restrict__ ctl, const uint32_t* __restrict__ list, uint32_t n_rows,
                                                            uint32_t area, const uint32_t* __restrict__ item_row, const uint32_t* __restrict__ item_k0, const uint32_t* __restrict__ Ap,
                                                            const uint32_t* __restrict__ Aj, const uint32_t* __restrict__ Ax, const T* __restrict__ v, T* __restrict__ r,
                                                            T* __restrict__ partial, Mult mult, Add add, T init, uint32_t seg_flag, uint32_t seg_len, uint32_t nnz,
                                                            const uint32_t* __restrict__ gate) {
        rows_task_body<T, Mult, Add>(blockIdx.x, gridDim.x, long_items, ctl, list, n_rows, area, item_row, item_k0, Ap, Aj, Ax, v, r, partial, mult, add, init, seg_flag, seg_len, nnz, gate);
    }

}// namespace splahip
