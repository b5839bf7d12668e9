// mask_probe.hpp — the mask probe of a masked call over a plan that holds the column-ordered blocks (cob.hip: cob_mask_probe_prepare / cob_mask_probe_launch).
// The body is shared by the probe's own kernel (a call served by the blocks) and by the first launch of the selected-rows side (mxv_rows.hpp: rows_select_kernel),
// whose first workgroups run it beside the compaction pass: as a launch of its own it was 12 us in front of a 170 us call.
#pragma once

#include "block_utils.hpp"
#include "ops.hpp"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace splahip {

    struct MaskProbeArgs {
        uint32_t           blocks = 0;// workgroups the probe wants (0: no probe for this call)
        uint32_t           n_rows = 0, stride = 1, rows_x10 = 0, nnz = 0, seq = 0, heavy_slots = 0, heavy_deg = 0xFFFFFFFFu, long_x10 = 0;
        uint32_t*          probe      = nullptr;// device accumulators: [0..1] selected entries, [2] ticket, [4..5] cost
        volatile uint32_t* host_words = nullptr;// pinned: [0..1] selected entries, [2] verdict, [3] sequence number
        const uint32_t*    hub_row    = nullptr;
    };

    // The mask probe of a masked call (op_select other than ALWAYS): how many ENTRIES does the selection cover?  A 1-in-`stride` sample of the rows (2048-row
    // groups, every row of a group: coalesced loads of mask and Ap), scaled — a few MB, ~4 us at R-MAT 23; visiting every hub row instead cost 65 us of scattered
    // 4-byte gathers (profiles/r04_notes.md 3).  The last workgroup to finish compares what the selected-rows kernels would cost — every selected row is
    // ~20 line requests of its own: ~rows_x10 / 10 blocks-entries per selected entry, plus their passes over mask / Ap / r — with the blocks' stream of the whole
    // matrix, and leaves the estimate and the verdict (1: the row kernels) in pinned HOST memory.  Both sides compute the reference's result for any mask, so the
    // verdict is only a hint: the host reads the latest one that has arrived before it enqueues its NEXT masked call over the plan (no synchronisation, no gated
    // launches: enqueueing both sides behind a device-side decision cost a dense masked call 4 - 5 us per launch that returned at once, 45 us in all).
    template<typename T>
    __device__ __forceinline__ void mask_probe_body(const uint32_t vb, const uint32_t vg, const MaskProbeArgs a, const uint32_t* __restrict__ Ap, const T* __restrict__ mask,
                                                    int op_select) {
        __shared__ unsigned long long s_sum[2][256 / kWave];
        unsigned long long sel_entries = 0, sel_cost = 0;// entries of the selected rows; what they cost the row kernels, in tenths of a blocks-entry
        const uint32_t n_groups = (a.n_rows + a.stride * 2048u - 1) / (a.stride * 2048u);
        for (uint32_t gi = vb; gi < n_groups; gi += vg) {
            // (which of the stride's groups is sampled varies pseudo-randomly: a generator's labels put its hubs at regular strides)
            const uint32_t row0 = (gi * a.stride + ((gi * 2654435761u) >> 9) % a.stride) * 2048u + threadIdx.x * 8u;
            if (row0 < a.n_rows) {// (every load issued before the first use; a row behind the matrix's end re-reads the last one)
                T        mk[8];
                uint32_t ap[9];
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j) mk[j] = mask[min(row0 + j, a.n_rows - 1)];
#pragma unroll
                for (uint32_t j = 0; j < 9; ++j) ap[j] = Ap[min(row0 + j, a.n_rows)];
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j)
                    if (row0 + j < a.n_rows && ap[j + 1] - ap[j] < a.heavy_deg && eval_sel<T>(op_select, mk[j])) {
                        sel_entries += (unsigned long long) (ap[j + 1] - ap[j]) * a.stride;
                        sel_cost += (unsigned long long) (ap[j + 1] - ap[j]) * a.stride * a.rows_x10;
                    }
            }
        }
        // the heaviest rows are no matter for a sample (a generator's first rows hold a quarter of a scale-free matrix: one sampled group of them would say anything):
        // the first two hub ranges' rows — the a.heavy_slots rows of degree >= a.heavy_deg — are visited one by one
        for (uint32_t s0 = vb * 256 + threadIdx.x; s0 < a.heavy_slots; s0 += vg * 256) {
            const uint32_t row = a.hub_row[s0];
            if (row != 0xFFFFFFFFu) {
                const uint32_t d = Ap[row + 1] - Ap[row];
                if (d >= a.heavy_deg && eval_sel<T>(op_select, mask[row])) sel_entries += d, sel_cost += (unsigned long long) d * a.long_x10;
            }
        }
        for (unsigned off = kWave / 2; off > 0; off >>= 1) sel_entries += __shfl_xor(sel_entries, off, kWave), sel_cost += __shfl_xor(sel_cost, off, kWave);
        if (lane_id() == 0) s_sum[0][threadIdx.x / kWave] = sel_entries, s_sum[1][threadIdx.x / kWave] = sel_cost;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long tot = 0, cost = 0;
            for (uint32_t w = 0; w < 256 / kWave; ++w) tot += s_sum[0][w], cost += s_sum[1][w];
            if (tot) {
                atomicAdd(reinterpret_cast<unsigned long long*>(a.probe), tot);
                atomicAdd(reinterpret_cast<unsigned long long*>(a.probe + 4), cost);
            }
            __threadfence();
            if (atomicAdd(&a.probe[2], 1u) == vg - 1) {// the last workgroup: every other one's sum is behind its fence
                __threadfence();
                const unsigned long long sel_nnz = __hip_atomic_load(reinterpret_cast<unsigned long long*>(a.probe), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // what the HOST reads before its next masked call over this plan (pinned memory, no synchronisation: a hint that lags by a call)
                a.host_words[0] = (uint32_t) sel_nnz, a.host_words[1] = (uint32_t) (sel_nnz >> 32);
                const unsigned long long sel_cost_all = __hip_atomic_load(reinterpret_cast<unsigned long long*>(a.probe + 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.host_words[2] = sel_cost_all / 10ull + a.nnz / 6ull < a.nnz ? 1u : 0u;
                __threadfence_system();
                a.host_words[3] = a.seq;
                // the accumulators return to zero for the next probe (nobody else touches them in this launch)
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.probe), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&a.probe[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.probe + 4), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

}// namespace splahip
