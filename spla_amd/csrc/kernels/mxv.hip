// mxv.hip — masked sparse-matrix x dense-vector over an arbitrary (op_mult, op_add) semiring on gfx950.
//
// Semantics: reference Algo_mxv_masked_cpu<T>::execute, src/cpu/cpu_mxv.hpp:56-106 (the parity oracle):
//     r[i] = select(mask[i]) ? ((init (+) p_0) (+) p_1 ...) : init,   p_k = mult(Ax[k], v[Aj[k]]), k ascending in row i
// Entry point replaced: Algo_mxv_masked_cl<T>::execute, src/opencl/cl_mxv.hpp:65-74 (not ported: the reference
// launches one workgroup-row per CSR row; here the nnz stream is cut into ~1 Ki-entry wave tiles, mxv_kernels.hpp).
//
// Data layout in HBM: plain CSR exactly as the reference's CpuCsr/CLCsr (Ap[n_rows+1], Aj[nnz], Ax[nnz], 4-byte
// words, reference src/cpu/cpu_formats.hpp:163-175) plus the plan: item_row[n_items+1], item_k0[n_items+1]
// (first row / first entry of every work item) and a partial[n_items] scratch for multi-segment rows.
//
// Kernel `mxv_wavetile` (associative + commutative op_add, no early exit) — HBM-bound; algorithmic bytes per call:
//     8*nnz_sel + 4*(n_rows+1) + 4*n_rows (mask) + 4*n_rows (r) + 4*n_cols (v once).
// Kernel `mxv_rowseq` (any op_add, optional early exit): one lane per row, strict left fold straight from HBM with
// the reference's early-exit rule.  Exact for non-associative / non-commutative op_add.
#include "block_utils.hpp"
#include "common.hpp"
#include "mxv_first_hit.hpp"
#include "mxv_kernels.hpp"
#include "mxv_rows.hpp"
#include "ops.hpp"
#include "plan.hpp"

#include <cstdlib>
#include <cstring>

using namespace splahip;

using Cfg = MxvCfg;

namespace splahip {
    // the semirings: 0 annihilates op_mult and is op_add's identity (sell_skip_family == 1) and op_add is idempotent (PLUS and BXOR count, they do not qualify)
    bool first_hit_family(int type, int op_mult, int op_add) {
        return type != SPLAHIP_FLOAT && sell_skip_family(type, op_mult, op_add) == 1 && (op_add == SPLAHIP_BOR || op_add == SPLAHIP_MAX);
    }
    // every stored value of the matrix is one word: said by the owner of the value array (splahip_csr_plan_set_uniform_values), or known from the attach pass
    // of a layout copy (the plan alone does not own the caller's value array)
    bool uniform_values(const splahip_plan_s* plan, uint32_t n_cols, uint32_t* xbits) {
        if (plan->values_uniform == 1) return *xbits = plan->values_bits, true;
        if (plan->sell && plan->sell->n_cols == n_cols && plan->sell->x_uniform) return *xbits = plan->sell->x_bits, true;
        if (plan->cob && plan->cob->n_cols == n_cols && plan->cob->x_uniform) return *xbits = plan->cob->x_bits, true;
        return false;
    }
}// namespace splahip

namespace {

    enum class Variant { WaveTileStatic,
                         WaveTileDynamic,
                         RowSeq,
                         RowSeqEarly };

    // semirings with an ahead-of-time instantiated wave-tile kernel; the others go through the runtime-switched functor
    bool static_semiring(int type, int op_mult, int op_add) {
        if (op_mult == SPLAHIP_MULT && op_add == SPLAHIP_PLUS) return true;
        if (op_mult == SPLAHIP_PLUS && op_add == SPLAHIP_MIN) return true;
        if (op_mult == SPLAHIP_LAND && op_add == SPLAHIP_LOR) return true;
        if (op_mult == SPLAHIP_BAND && op_add == SPLAHIP_BOR) return type != SPLAHIP_FLOAT;
        return false;
    }

    Variant pick_variant(int type, int op_mult, int op_add, int early_exit) {
        if (early_exit) return Variant::RowSeqEarly;
        {// A/B runs only: SPLAHIP_FORCE_ROWSEQ=1 sends everything to the one-lane-per-row kernel
            const char* e = knob_env("SPLAHIP_FORCE_ROWSEQ");
            if (e && e[0] == '1') return Variant::RowSeq;
        }
        if (!fold_is_reorderable(op_add, type) || !binop_is_commutative(op_add)) return Variant::RowSeq;
        return static_semiring(type, op_mult, op_add) ? Variant::WaveTileStatic : Variant::WaveTileDynamic;
    }

    bool sell_disabled() {
        const char* e = knob_env("SPLAHIP_SELL");
        return e && e[0] == '0';
    }

    thread_local const uint32_t* tl_wavetile_gate = nullptr;// masked calls over a plan with blocks: the mask probe's words (the kernels run when word 3 is 1)

    template<typename T, class Mult, class Add>
    int launch_wavetile(splahip_plan plan, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const void* mask,
                        const void* v, void* r, int op_mult, int op_add, int op_select, uint32_t init_bits, hipStream_t st) {
        const uint grid = (plan->n_items + Cfg::kWaves - 1) / Cfg::kWaves;
        hipLaunchKernelGGL((spmv_wavetile_kernel<Cfg, T, Mult, Add, false>), dim3(grid), dim3(Cfg::kTpb), 0, st,
                           plan->d_item_row, plan->d_item_k0, plan->n_items, Ap, Aj, static_cast<const uint32_t*>(Ax),
                           static_cast<const T*>(mask), static_cast<const T*>(v), static_cast<const uint32_t*>(nullptr),
                           static_cast<T*>(r), static_cast<uint32_t*>(nullptr), reinterpret_cast<T*>(plan->d_partial),
                           static_cast<uint32_t*>(nullptr), Mult(op_mult), Add(op_add), op_select,
                           from_bits<T>(init_bits), plan->nnz, tl_wavetile_gate);
        SPLAHIP_TRY(hipGetLastError());
        if (plan->n_long > 0) {
            hipLaunchKernelGGL((spmv_combine_kernel<Cfg, T, Add, false>), dim3((plan->n_items + 255) / 256), dim3(256), 0, st,
                               plan->d_item_row, plan->d_item_k0, plan->n_items, Ap, static_cast<const T*>(mask),
                               static_cast<T*>(r), static_cast<uint32_t*>(nullptr), reinterpret_cast<const T*>(plan->d_partial),
                               static_cast<const uint32_t*>(nullptr), Add(op_add), op_select, from_bits<T>(init_bits), tl_wavetile_gate);
            SPLAHIP_TRY(hipGetLastError());
        }
        return SPLAHIP_OK;
    }

    // the ORDERED kernels (mxv_kernels.hpp: ORDERED, spmv_longrow_ordered_kernel): run items with one lane per row, long rows with one wavefront per row — every
    // selected row a strict left fold from init in entry order, the reference's own loop.  They serve the reference-order mode (an fp32 PLUS; AddF = the static
    // functor: the chain of a long row is one dependent addition per entry) and, since round 6, every op_add that must not be re-ordered (MINUS, DIV, FIRST, SECOND,
    // MINUS_POW2, fp32 MULT ...: AddF = the runtime-switched functor) — until then a lane per row straight from memory, a hub row of 10^5 entries one lane's.
    template<typename T, class AddF>
    int launch_ordered(splahip_plan plan, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const void* mask, const void* v, void* r, int op_mult, int op_add,
                       int op_select, uint32_t init_bits, hipStream_t st) {
        const uint grid = (plan->n_items + Cfg::kWaves - 1) / Cfg::kWaves;
        hipLaunchKernelGGL((spmv_wavetile_kernel<Cfg, T, DynBin, AddF, false, true>), dim3(grid), dim3(Cfg::kTpb), 0, st, plan->d_item_row, plan->d_item_k0, plan->n_items, Ap, Aj,
                           static_cast<const uint32_t*>(Ax), static_cast<const T*>(mask), static_cast<const T*>(v), static_cast<const uint32_t*>(nullptr), static_cast<T*>(r),
                           static_cast<uint32_t*>(nullptr), reinterpret_cast<T*>(plan->d_partial), static_cast<uint32_t*>(nullptr), DynBin(op_mult), AddF(op_add), op_select,
                           from_bits<T>(init_bits), plan->nnz, static_cast<const uint32_t*>(nullptr));
        SPLAHIP_TRY(hipGetLastError());
        if (plan->n_long > 0) {
            hipLaunchKernelGGL((spmv_longrow_ordered_kernel<Cfg, T, DynBin, AddF, false>), dim3(grid), dim3(Cfg::kTpb), 0, st, plan->d_item_row, plan->d_item_k0, plan->n_items, Ap, Aj,
                               static_cast<const uint32_t*>(Ax), static_cast<const T*>(mask), static_cast<const T*>(v), static_cast<const uint32_t*>(nullptr), static_cast<T*>(r),
                               static_cast<uint32_t*>(nullptr), DynBin(op_mult), AddF(op_add), op_select, from_bits<T>(init_bits), plan->nnz);
            SPLAHIP_TRY(hipGetLastError());
        }
        return SPLAHIP_OK;
    }

    template<typename T, bool EARLY>
    int launch_rowseq(uint32_t n_rows, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const void* mask,
                      const void* v, void* r, int op_mult, int op_add, int op_select, uint32_t init_bits, hipStream_t st) {
        hipLaunchKernelGGL((mxv_rowseq_kernel<T, DynBin, DynBin, EARLY>), dim3((n_rows + 255) / 256), dim3(256), 0, st,
                           n_rows, Ap, Aj, static_cast<const uint32_t*>(Ax), static_cast<const T*>(mask),
                           static_cast<const T*>(v), static_cast<T*>(r), DynBin(op_mult), DynBin(op_add), op_select,
                           from_bits<T>(init_bits));
        SPLAHIP_TRY(hipGetLastError());
        return SPLAHIP_OK;
    }

    // ---- BFS-like calls over a structure-only matrix: a row stops at its first hit (mxv_first_hit.hpp) -------------------------------------------------
    // the semirings: 0 annihilates op_mult and is op_add's identity (sell_skip_family == 1) and op_add is idempotent (PLUS and BXOR count, they do not qualify)
    uint32_t knob_u32(const char* name, uint32_t dflt) {
        const char* e = knob_env(name);
        return e && *e ? (uint32_t) std::strtoul(e, nullptr, 10) : dflt;
    }
    int first_hit_prepare(splahip_plan plan, uint32_t n_cols) {
        if (plan->d_fh_ctl && plan->fh_cols == n_cols) return SPLAHIP_OK;
        if (plan->d_fh_ctl) return SPLAHIP_ERROR;// (a plan serves one matrix: one column count)
        const size_t bit_words = ((size_t) n_cols + 2047u) / 2048u * 64u;
        uint32_t *   bits = nullptr, *ctl = nullptr, *host = nullptr, *host_dev = nullptr;
        hipError_t   e    = device_malloc(reinterpret_cast<void**>(&bits), sizeof(uint32_t) * bit_words);
        if (e == hipSuccess) e = device_malloc(reinterpret_cast<void**>(&ctl), sizeof(uint32_t) * (8 + 8 * (size_t) kVstatBlocks));
        if (e == hipSuccess) e = hipMemset(ctl, 0, sizeof(uint32_t) * (8 + 8 * (size_t) kVstatBlocks));
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&host), 8 * sizeof(uint32_t), hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&host_dev), host, 0);
        if (e != hipSuccess) {// (nothing half-built stays with the plan: the other forms serve its calls)
            if (bits) (void) hipFree(bits);
            if (ctl) (void) hipFree(ctl);
            if (host) (void) hipHostFree(host);
            (void) hipGetLastError();
            return SPLAHIP_ERROR;
        }
        for (int k = 0; k < 8; ++k) host[k] = 0;
        plan->d_fh_bits = bits, plan->d_fh_ctl = ctl, plan->h_fh = host, plan->d_h_fh = host_dev;
        plan->fh_cols   = n_cols;
        return SPLAHIP_OK;
    }
    // Enqueue the pass over v (bitmap, uniformity, a sampled estimate of the walk's cost; its verdict also lands in pinned host memory) and say whether THIS call
    // runs on the first-hit kernel: when the latest finished pass — of an earlier call — saw a frontier (one non-zero word) AND a walk that mostly ends within a
    // row's first entries: the sampled rows that would be walked past their fourth entry hold at most SPLAHIP_FIRST_HIT_TAIL_PCT % (15) of the sampled selected
    // entries.  (R-MAT 22: the level of a traversal that reaches the most vertices 62 us against 92 on the sliced-ELL copy; a random 2 % of the columns as
    // frontier 211 against 62 — profiles/r04_first_hit_sweep.txt: most rows never hit, and scanning whole CSR rows a lane each is what the layout copies do
    // better.)  The kernel is correct for any vector and mask (it sees this call's own verdict), so a stale hint only costs time; a traversal's levels differ,
    // so the hint pays for repeated or slowly changing steps.  A plan whose calls keep not qualifying is probed every 16th call.
    template<typename T>
    int first_hit_probe(splahip_plan plan, uint32_t n_cols, const uint32_t* Ap, const uint32_t* Aj, const void* mask, int op_select, const void* v, bool early_exit,
                        hipStream_t st, bool* take) {
        *take = false;
        if (knob_u32("SPLAHIP_FIRST_HIT", 1) == 0 || n_cols == 0) return SPLAHIP_OK;
        int rc = first_hit_prepare(plan, n_cols);
        if (rc != SPLAHIP_OK) return SPLAHIP_OK;// (no workspace: the other forms serve the call)
        const uint32_t pct = knob_u32("SPLAHIP_FIRST_HIT_TAIL_PCT", 15);
        uint32_t h[8];// (the pass writes two 16-byte halves, the call's sequence number in each: a torn read shows as two different numbers)
        for (int k = 0; k < 8; ++k) h[k] = plan->h_fh[k];
        *take = h[3] != 0u && h[3] == h[7] && h[0] != 0u && h[2] != 0u && h[4] != 0u && (uint64_t) h[5] * 100u <= (uint64_t) h[4] * pct;
        // an early_exit call needs no hint: what it would run otherwise is the one-lane-per-row CSR kernel, which walks the same way but reads the value words, gathers
        // words of v instead of bits, and leaves a long row without a hit to ONE lane (this kernel hands it to the wave); a vector that is not a frontier is folded with
        // the reference's early-exit rule by the same launch
        if (early_exit) *take = true;
        if (!*take) {
            plan->fh_misses += 1;
            if (plan->fh_misses > 4 && (plan->fh_misses & 15u) != 0u) return SPLAHIP_OK;
        } else
            plan->fh_misses = 0;
        plan->fh_seq = plan->fh_seq + 1 ? plan->fh_seq + 1 : 1;
        hipLaunchKernelGGL((vstat_kernel<T>), dim3(kVstatBlocks), dim3(kVstatTpb), 0, st, n_cols, static_cast<const T*>(v), plan->d_fh_bits, plan->d_fh_ctl + 8, plan->d_fh_ctl + 4,
                           plan->d_fh_ctl, plan->d_h_fh, plan->fh_seq, plan->n_rows, Ap, Aj, static_cast<const T*>(mask), op_select);
        SPLAHIP_TRY(hipGetLastError());
        return SPLAHIP_OK;
    }
    template<typename T>
    int launch_first_hit(splahip_plan plan, uint32_t n_rows, const uint32_t* Ap, const uint32_t* Aj, const void* mask, const void* v, void* r, int op_mult, int op_add,
                         int op_select, uint32_t init_bits, uint32_t xbits, int early_exit, hipStream_t st) {
        const dim3 grid((n_rows + 255) / 256);
        if constexpr (!std::is_same<T, float>::value) {
            if (op_mult == SPLAHIP_BAND && op_add == SPLAHIP_BOR) {
                hipLaunchKernelGGL((mxv_first_hit_kernel<T, StaBin<SPLAHIP_BAND>, StaBin<SPLAHIP_BOR>, false>), grid, dim3(256), 0, st, n_rows, Ap, Aj, static_cast<const T*>(mask),
                                   static_cast<const T*>(v), static_cast<T*>(r), StaBin<SPLAHIP_BAND>(), StaBin<SPLAHIP_BOR>(), op_select, from_bits<T>(init_bits), xbits,
                                   plan->d_fh_ctl, plan->d_fh_bits, static_cast<uint32_t*>(nullptr), early_exit);
                SPLAHIP_TRY(hipGetLastError());
                return SPLAHIP_OK;
            }
        }
        hipLaunchKernelGGL((mxv_first_hit_kernel<T, DynBin, DynBin, false>), grid, dim3(256), 0, st, n_rows, Ap, Aj, static_cast<const T*>(mask), static_cast<const T*>(v),
                           static_cast<T*>(r), DynBin(op_mult), DynBin(op_add), op_select, from_bits<T>(init_bits), xbits, plan->d_fh_ctl, plan->d_fh_bits,
                           static_cast<uint32_t*>(nullptr), early_exit);
        SPLAHIP_TRY(hipGetLastError());
        return SPLAHIP_OK;
    }

    // workspaces of the selected-rows path, allocated on a plan's first masked call that may take it; the plan's long rows are listed here, once
    int rows_prepare(splahip_plan plan, const uint32_t* Ap, hipStream_t st) {
        if (plan->rows_ready) return SPLAHIP_OK;
        // (nothing half-built stays with the plan when a step fails: the next call starts over)
        uint32_t *list = nullptr, *ctl = nullptr, *items = nullptr, *lrow = nullptr, *lfirst = nullptr;
        // the call's selected short rows from the front, its selected medium rows from the end
        hipError_t e = device_malloc(reinterpret_cast<void**>(&list), sizeof(uint32_t) * ((size_t) plan->n_rows + 1));
        if (e == hipSuccess) e = device_malloc(reinterpret_cast<void**>(&ctl), sizeof(uint32_t) * 2 * kRowsCtlSet);// two sets of list lengths (the calls use them in turn), kRowsRegions count pairs each on lines of their own (mxv_rows.hpp)// [0] short rows, [1] medium rows (one 8-byte pair), [3] segment items listed by the running call (zeroed by its mask probe), [2] long rows of the plan, [4..7] the row kernels' gate: open
        const uint32_t open_gate[4] = {0, 0, 0, 1};
        if (e == hipSuccess) e = hipMemcpy(ctl + 4, open_gate, sizeof(open_gate), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(ctl, 0, 4 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(ctl + 8, 0, (2 * kRowsCtlSet - 8) * sizeof(uint32_t));
        if (e == hipSuccess) e = device_malloc(reinterpret_cast<void**>(&items), sizeof(uint32_t) * ((size_t) plan->n_items + 1));
        if (e == hipSuccess && plan->n_long) e = device_malloc(reinterpret_cast<void**>(&lrow), sizeof(uint32_t) * (size_t) plan->n_long);
        if (e == hipSuccess && plan->n_long) e = device_malloc(reinterpret_cast<void**>(&lfirst), sizeof(uint32_t) * (size_t) plan->n_long);
        if (e == hipSuccess && plan->n_long && plan->n_items) {
            hipLaunchKernelGGL(rows_long_list_kernel, dim3((plan->n_items + 255) / 256), dim3(256), 0, st, plan->d_item_row, plan->d_item_k0, plan->n_items, Ap, Cfg::kSegFlag, ctl + 2,
                               lrow, lfirst);
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            for (void* p : {(void*) list, (void*) ctl, (void*) items, (void*) lrow, (void*) lfirst})
                if (p) (void) hipFree(p);
            return fail(SPLAHIP_ERROR, "selected-rows workspace: %s", hipGetErrorString(e));
        }
        plan->d_rows_list = list, plan->d_rows_ctl = ctl, plan->d_long_items = items, plan->d_long_rows = lrow, plan->d_long_first = lfirst;
        plan->rows_ready = true;
        return SPLAHIP_OK;
    }

    // the selected-rows path (mxv_rows.hpp), every launch gated on the mask probe's decision
    template<typename T, class Mult, class Add>
    int launch_rows(splahip_plan plan, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const void* mask, const void* v, void* r, int op_mult, int op_add,
                    int op_select, uint32_t init_bits, const uint32_t* gate, const MaskProbeArgs& probe, hipStream_t st) {
        const T        init = from_bits<T>(init_bits);
        const uint32_t n    = plan->n_rows;
        // the list lengths: two sets of counters used in turn — a call's first launch returns the OTHER set to zero for the call after it
        if (plan->rows_dirty) {// (a call whose launches failed may have left either set behind: both start over; [2] the plan's long-row count and [4..7] the gate stay)
            SPLAHIP_TRY(hipMemsetAsync(plan->d_rows_ctl, 0, 2 * sizeof(uint32_t), st));
            SPLAHIP_TRY(hipMemsetAsync(plan->d_rows_ctl + 3, 0, sizeof(uint32_t), st));
            SPLAHIP_TRY(hipMemsetAsync(plan->d_rows_ctl + 8, 0, (2 * kRowsCtlSet - 8) * sizeof(uint32_t), st));
            plan->rows_dirty = false;
        }
        uint32_t* const ctl      = plan->d_rows_ctl + (plan->rows_parity ? kRowsCtlSet : 0);
        uint32_t* const ctl_next = plan->d_rows_ctl + (plan->rows_parity ? 0 : kRowsCtlSet);
        plan->rows_parity ^= 1u;
        // (ctl[0] short rows, [1] medium rows, [3] segment items listed: returned to zero by the first launch of the PREVIOUS rows-side call over the plan)
        // (workgroups: every flush of a workgroup's list is a same-address atomic — 256 / 512 / 1024 / 2048 / 4096 workgroups at R-MAT 23, 10 % of the light rows: 103 / 87 /
        // 86 / 99 / 113 us per call)
        const uint32_t compact_blocks = std::max(kRowsRegions, knob_u32("SPLAHIP_ROWS_COMPACT_BLOCKS", 1024) / kRowsRegions * kRowsRegions),// whole workgroups per list area
                       // (short rows / wave tasks: a workgroup per 512 rows of the matrix, 1024 ... 8192 — R-MAT 23, 4096 / 8192 of each: random 10 % 161.4 / 159.2 us, "50 %, no
                       // hub row" 155.7 / 152.9 us, three repeats each: tools/gpu_r6_rows8.sh)
                       by_rows      = std::min(8192u, std::max(1024u, n / 512u)),// (R-MAT 21: 2048 / 4096 / 8192 of each: random 10 % 63.0 / 60.6 / 60.6 us; R-MAT 20: level)
                       short_blocks = knob_u32("SPLAHIP_ROWS_SHORT_BLOCKS", by_rows), task_blocks = knob_u32("SPLAHIP_ROWS_LONG_BLOCKS", by_rows);
        // rows per list area = the rows of its compaction workgroups (rows_compact_body: a workgroup's share is whole steps of 2048 rows)
        const uint32_t per_block = ((n + compact_blocks - 1) / compact_blocks + 2047u) & ~2047u, area = per_block * (compact_blocks / kRowsRegions);
        const bool     has_long = plan->n_items && plan->n_long > 0;// the selected long rows' segment items (a plan without long rows lists none)
        if (knob_u32("SPLAHIP_ROWS_FUSED", 1)) {// the call's mask probe and the two mask passes share a launch (mxv_rows.hpp)
            const uint32_t select_blocks = has_long ? (plan->n_long + 255) / 256 : 0u;
            const dim3     grid(probe.blocks + select_blocks + compact_blocks);
            if (knob_u32("SPLAHIP_ROWS_SMALL_LDS", 1))
                hipLaunchKernelGGL((rows_select_kernel<T, true>), grid, dim3(256), 0, st, probe, select_blocks, n, Ap, static_cast<const T*>(mask), op_select, init, static_cast<T*>(r),
                                   plan->d_rows_list, ctl, ctl_next, area, plan->n_long, plan->d_long_rows, plan->d_long_first, Cfg::kSeg, plan->d_long_items, gate);
            else
                hipLaunchKernelGGL((rows_select_kernel<T, false>), grid, dim3(256), 0, st, probe, select_blocks, n, Ap, static_cast<const T*>(mask), op_select, init, static_cast<T*>(r),
                                   plan->d_rows_list, ctl, ctl_next, area, plan->n_long, plan->d_long_rows, plan->d_long_first, Cfg::kSeg, plan->d_long_items, gate);
        } else {
            cob_mask_probe_launch(probe, TypeId<T>::value, Ap, mask, op_select, st);
            hipLaunchKernelGGL((rows_compact_kernel<T>), dim3(compact_blocks), dim3(256), 0, st, n, Ap, static_cast<const T*>(mask), op_select, init, static_cast<T*>(r),
                               plan->d_rows_list, ctl, ctl_next, area, gate);
            if (has_long)
                hipLaunchKernelGGL((rows_long_select_kernel<T>), dim3((plan->n_long + 255) / 256), dim3(256), 0, st, plan->n_long, plan->d_long_rows, plan->d_long_first, Ap,
                                   static_cast<const T*>(mask), op_select, Cfg::kSeg, plan->d_long_items, ctl + 3, gate);
        }
        const bool combine = plan->n_items && plan->n_long > 0;// (rows of several segments: their partials are folded behind the tasks)
        const bool ride    = combine && knob_u32("SPLAHIP_ROWS_COMBINE_FUSED", 1) != 0;
        if (!ride)
            hipLaunchKernelGGL((rows_short_kernel<T, Mult, Add>), dim3(short_blocks), dim3(256), 0, st, plan->d_rows_list, ctl, area, Ap, Aj, static_cast<const uint32_t*>(Ax),
                               static_cast<const T*>(v), static_cast<T*>(r), Mult(op_mult), Add(op_add), init, plan->nnz, gate);
        hipLaunchKernelGGL((rows_task_kernel<T, Mult, Add>), dim3(task_blocks), dim3(256), 0, st, plan->d_long_items, ctl, plan->d_rows_list, n, area, plan->d_item_row,
                           plan->d_item_k0, Ap, Aj, static_cast<const uint32_t*>(Ax), static_cast<const T*>(v), static_cast<T*>(r), reinterpret_cast<T*>(plan->d_partial),
                           Mult(op_mult), Add(op_add), init, Cfg::kSegFlag, Cfg::kSeg, plan->nnz, gate);
        if (ride) {// the short rows behind the tasks, the combine pass as the same launch's first workgroups (mxv_rows.hpp)
            const uint32_t combine_blocks = (plan->n_items + 255) / 256;
            hipLaunchKernelGGL((rows_short_combine_kernel<Cfg, T, Mult, Add>), dim3(combine_blocks + short_blocks), dim3(256), 0, st, combine_blocks, plan->d_item_row, plan->d_item_k0,
                               plan->n_items, static_cast<const T*>(mask), reinterpret_cast<const T*>(plan->d_partial), op_select, plan->d_rows_list, ctl, area, Ap, Aj,
                               static_cast<const uint32_t*>(Ax), static_cast<const T*>(v), static_cast<T*>(r), Mult(op_mult), Add(op_add), init, plan->nnz, gate);
        }
        {
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) {
                plan->rows_dirty = true;
                return fail(SPLAHIP_ERROR, "selected-rows kernels: %s", hipGetErrorString(le));
            }
        }
        if (combine && !ride) {
            hipLaunchKernelGGL((spmv_combine_kernel<Cfg, T, Add, false>), dim3((plan->n_items + 255) / 256), dim3(256), 0, st, plan->d_item_row, plan->d_item_k0, plan->n_items, Ap,
                               static_cast<const T*>(mask), static_cast<T*>(r), static_cast<uint32_t*>(nullptr), reinterpret_cast<const T*>(plan->d_partial),
                               static_cast<const uint32_t*>(nullptr), Add(op_add), op_select, init, gate);
            SPLAHIP_TRY(hipGetLastError());
        }
        return SPLAHIP_OK;
    }
    template<typename T>
    int rows_typed(splahip_plan plan, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                   uint32_t init_bits, const uint32_t* gate, const MaskProbeArgs& probe, hipStream_t st) {
        if (op_mult == SPLAHIP_MULT && op_add == SPLAHIP_PLUS)
            return launch_rows<T, StaBin<SPLAHIP_MULT>, StaBin<SPLAHIP_PLUS>>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, gate, probe, st);
        if (op_mult == SPLAHIP_PLUS && op_add == SPLAHIP_MIN)
            return launch_rows<T, StaBin<SPLAHIP_PLUS>, StaBin<SPLAHIP_MIN>>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, gate, probe, st);
        return launch_rows<T, DynBin, DynBin>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, gate, probe, st);
    }

#define SPLAHIP_WT(MUL, ADD) launch_wavetile<T, StaBin<MUL>, StaBin<ADD>>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st)

    template<typename T>
    int dispatch_typed(splahip_plan plan, uint32_t n_rows, const uint32_t* Ap, const uint32_t* Aj, const void* Ax,
                       const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                       uint32_t init_bits, int early_exit, hipStream_t st) {
        constexpr int type = TypeId<T>::value;
        Variant variant = pick_variant(type, op_mult, op_add, early_exit);
        // reference-order mode: an fp32 sum is folded exactly as the reference folds it (every other fold of the wave-tile kernels is exact as it is)
        if (reference_order_calls() && type == SPLAHIP_FLOAT && op_add == SPLAHIP_PLUS && !early_exit) {
            if (plan->n_items != 0 && ((reinterpret_cast<uintptr_t>(Aj) | reinterpret_cast<uintptr_t>(Ax)) & 15u) == 0)
                return launch_ordered<T, StaBin<SPLAHIP_PLUS>>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
            variant = Variant::RowSeq;// (an unaligned slice of a larger allocation: one lane per row straight from memory — the same fold)
        }
        // the wave-tile kernels read Aj / Ax with 16-byte loads: an unaligned slice of a larger allocation takes the row kernel
        if (((reinterpret_cast<uintptr_t>(Aj) | reinterpret_cast<uintptr_t>(Ax)) & 15u) != 0 &&
            (variant == Variant::WaveTileStatic || variant == Variant::WaveTileDynamic))
            variant = Variant::RowSeq;
        switch (variant) {
            case Variant::RowSeqEarly:
                return launch_rowseq<T, true>(n_rows, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
            case Variant::RowSeq:
                // an op_add that must not be re-ordered: the ordered kernels (the same fold, a hub row no longer one lane's); a plan without items (an empty
                // matrix) or an unaligned slice of a larger allocation keeps the lane-per-row kernel
                if (plan->n_items != 0 && ((reinterpret_cast<uintptr_t>(Aj) | reinterpret_cast<uintptr_t>(Ax)) & 15u) == 0 && !knob_env("SPLAHIP_FORCE_ROWSEQ"))
                    return launch_ordered<T, DynBin>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
                return launch_rowseq<T, false>(n_rows, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
            case Variant::WaveTileDynamic:
                return launch_wavetile<T, DynBin, DynBin>(plan, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init_bits, st);
            case Variant::WaveTileStatic:
                if (op_mult == SPLAHIP_MULT && op_add == SPLAHIP_PLUS) return SPLAHIP_WT(SPLAHIP_MULT, SPLAHIP_PLUS);
                if (op_mult == SPLAHIP_PLUS && op_add == SPLAHIP_MIN) return SPLAHIP_WT(SPLAHIP_PLUS, SPLAHIP_MIN);
                if (op_mult == SPLAHIP_LAND && op_add == SPLAHIP_LOR) return SPLAHIP_WT(SPLAHIP_LAND, SPLAHIP_LOR);
                if constexpr (type != SPLAHIP_FLOAT) {
                    if (op_mult == SPLAHIP_BAND && op_add == SPLAHIP_BOR) return SPLAHIP_WT(SPLAHIP_BAND, SPLAHIP_BOR);
                }
                break;
        }
        return fail(SPLAHIP_NOT_IMPLEMENTED, "no kernel variant for this semiring");
    }
#undef SPLAHIP_WT

    void plan_release(splahip_plan plan) {
        if (!plan) return;
        if (plan->h_fh) (void) hipHostFree(const_cast<uint32_t*>(plan->h_fh));
        if (plan->d_fh_bits) (void) hipFree(plan->d_fh_bits);
        if (plan->d_fh_ctl) (void) hipFree(plan->d_fh_ctl);
        if (plan->d_rows_list) (void) hipFree(plan->d_rows_list);
        if (plan->d_rows_ctl) (void) hipFree(plan->d_rows_ctl);
        if (plan->d_long_items) (void) hipFree(plan->d_long_items);
        if (plan->d_long_rows) (void) hipFree(plan->d_long_rows);
        if (plan->d_long_first) (void) hipFree(plan->d_long_first);
        if (plan->d_item_row) (void) hipFree(plan->d_item_row);
        if (plan->d_item_k0) (void) hipFree(plan->d_item_k0);
        if (plan->d_partial) (void) hipFree(plan->d_partial);
        if (plan->d_partial_ok) (void) hipFree(plan->d_partial_ok);
        if (plan->d_ws) (void) hipFree(plan->d_ws);
        sell_release(plan->sell);
        cob_release(plan->cob);
        delete plan;
    }

}// namespace

extern "C" {

int splahip_csr_plan_create(splahip_plan* out, uint32_t n_rows, uint32_t nnz, const uint32_t* d_Ap, void* stream) {
    SPLAHIP_REQUIRE(out != nullptr, "plan is null");
    SPLAHIP_REQUIRE(d_Ap != nullptr || n_rows == 0, "d_Ap is null");
    SPLAHIP_REQUIRE(n_rows < Cfg::kSegFlag, "n_rows must be < 2^31");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);

    auto* plan   = new splahip_plan_s();
    plan->n_rows = n_rows;
    plan->nnz    = nnz;
    if (n_rows == 0) {
        *out = plan;
        return SPLAHIP_OK;
    }

    const uint32_t n_chunks = (n_rows + 1023u) / 1024u;
    uint32_t*      d_chunk  = nullptr;// n_chunks + 1 offsets, then 1 word n_long
    hipError_t     e        = hipMalloc(reinterpret_cast<void**>(&d_chunk), sizeof(uint32_t) * (size_t) (n_chunks + 2));
    if (e != hipSuccess) {
        delete plan;
        return fail(SPLAHIP_ERROR, "hipMalloc failed for plan scratch: %s", hipGetErrorString(e));
    }
    uint32_t* d_nlong = d_chunk + n_chunks + 1;
    e                 = hipMemsetAsync(d_nlong, 0, sizeof(uint32_t), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(plan_count_kernel<Cfg>, dim3(n_chunks), dim3(256), 0, s, d_Ap, n_rows, d_chunk, d_nlong);
        hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, d_chunk, n_chunks);
        e = hipGetLastError();
    }
    uint32_t h_tail[2] = {0, 0};// {n_items, n_long}
    if (e == hipSuccess) e = hipMemcpyAsync(h_tail, d_chunk + n_chunks, sizeof(h_tail), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) {
        plan->n_items      = h_tail[0];
        plan->n_long       = h_tail[1];
        const size_t bytes = sizeof(uint32_t) * ((size_t) plan->n_items + 1);
        e                  = hipMalloc(reinterpret_cast<void**>(&plan->d_item_row), bytes);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&plan->d_item_k0), bytes);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&plan->d_partial), bytes);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&plan->d_partial_ok), bytes);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(plan_write_kernel<Cfg>, dim3(n_chunks), dim3(256), 0, s, d_Ap, n_rows, nnz, d_chunk, n_chunks,
                           plan->d_item_row, plan->d_item_k0);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void) hipFree(d_chunk);
    if (e != hipSuccess) {
        plan_release(plan);
        return fail(SPLAHIP_ERROR, "plan construction failed: %s", hipGetErrorString(e));
    }
    *out = plan;
    return SPLAHIP_OK;
}

int splahip_csr_plan_destroy(splahip_plan plan) {
    plan_release(plan);
    return SPLAHIP_OK;
}

namespace {
    // smallest and largest word of an array
    __global__ __launch_bounds__(256) void words_range_kernel(size_t n, const uint32_t* __restrict__ x, uint32_t* __restrict__ out) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0;
        for (size_t k = (size_t) blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t) gridDim.x * 256) {
            const uint32_t a = x[k];
            lo = min(lo, a), hi = max(hi, a);
        }
        lo = ~wave_max(~lo);
        hi = wave_max(hi);
        __shared__ uint32_t s_lo[256 / kWave], s_hi[256 / kWave];// (one pair of same-address atomics per workgroup, not per wave)
        if (lane_id() == 0) s_lo[threadIdx.x / kWave] = lo, s_hi[threadIdx.x / kWave] = hi;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint32_t k = 1; k < 256 / kWave; ++k) lo = min(lo, s_lo[k]), hi = max(hi, s_hi[k]);
            atomicMax(&out[0], ~lo);
            atomicMax(&out[1], hi);
        }
    }
}// namespace

int splahip_words_range(const void* d_words, size_t n, uint32_t* lo, uint32_t* hi, void* stream) {
    SPLAHIP_REQUIRE(lo && hi, "null output");
    *lo = 0xFFFFFFFFu, *hi = 0;
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_words != nullptr, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    DevMem      out;
    SPLAHIP_TRY(out.alloc(s, 8));
    SPLAHIP_TRY(hipMemsetAsync(out.p, 0, 8, s));
    hipLaunchKernelGGL(words_range_kernel, dim3(1024), dim3(256), 0, s, n, static_cast<const uint32_t*>(d_words), out.as<uint32_t>());
    uint32_t h[2] = {0, 0};
    SPLAHIP_TRY(hipMemcpyAsync(h, out.p, 8, hipMemcpyDeviceToHost, s));
    SPLAHIP_TRY(hipStreamSynchronize(s));
    *lo = ~h[0], *hi = h[1];
    return SPLAHIP_OK;
}

int splahip_csr_plan_set_uniform_values(splahip_plan plan, int is_uniform, uint32_t value_bits) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    plan->values_uniform = is_uniform ? 1 : 0, plan->values_bits = value_bits;
    return SPLAHIP_OK;
}

int splahip_csr_plan_hint_products(splahip_plan plan, uint64_t products) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    plan->products_hint = products;
    return SPLAHIP_OK;
}

int splahip_csr_plan_first_hit(splahip_plan plan, int* taken, int* uniform, uint32_t* nonzeros, uint32_t* tail_pct) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    if (taken) *taken = plan->fh_taken;
    if (uniform) *uniform = plan->h_fh && plan->h_fh[3] != 0u ? (int) plan->h_fh[0] : -1;
    if (nonzeros) *nonzeros = plan->h_fh ? plan->h_fh[2] : 0u;
    if (tail_pct) *tail_pct = plan->h_fh && plan->h_fh[4] ? (uint32_t) ((uint64_t) plan->h_fh[5] * 100u / plan->h_fh[4]) : 100u;
    return SPLAHIP_OK;
}

int splahip_csr_plan_info(splahip_plan plan, uint32_t* n_items, uint32_t* n_long) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    if (n_items) *n_items = plan->n_items;
    if (n_long) *n_long = plan->n_long;
    return SPLAHIP_OK;
}

const char* splahip_mxv_variant_of(splahip_plan plan, int type, int op_mult, int op_add, int early_exit) {
    uint32_t xb = 0;
    if (plan && first_hit_family(type, op_mult, op_add) && plan->fh_taken && plan->nnz != 0 &&
        (plan->values_uniform == 1 || (plan->sell && uniform_values(plan, plan->sell->n_cols, &xb)) || (plan->cob && uniform_values(plan, plan->cob->n_cols, &xb))))
        return "mxv_first_hit<structure-only matrix, frontier vector: a row stops at its first hit>";// (what the plan's last such call ran on)
    if (!early_exit && plan && plan->cob && cob_serves_mxv(plan, type, op_mult, op_add, plan->cob->n_cols, nullptr) &&
        !(plan->sell && !sell_disabled() && sell_skip_family(type, op_mult, op_add) == 1))
        return type == SPLAHIP_FLOAT && op_add == SPLAHIP_PLUS ? "mxv_cob<column-ordered blocks, fp64 LDS accumulation>"
               : type == SPLAHIP_FLOAT && (op_add == SPLAHIP_MIN || op_add == SPLAHIP_MAX) ? "mxv_cob<column-ordered blocks, order-preserving integer keys in LDS>"
                                                                                            : "mxv_cob<column-ordered blocks, LDS accumulation>";
    const bool ref_order = reference_order_calls() && type == SPLAHIP_FLOAT && op_add == SPLAHIP_PLUS;
    if (plan && plan->sell && !early_exit && !ref_order && fold_is_reorderable(op_add, type) && (plan->sell->n_xslices == 0 || binop_is_commutative(op_add)) && !sell_disabled()) {
        if (sell_skip_family(type, op_mult, op_add) != 0)// the device picks the bit form or the value form from the input vector's statistics
            return static_semiring(type, op_mult, op_add) ? "mxv_sell<static semiring; sparse-vector bit form | value form>" : "mxv_sell<runtime ops; sparse-vector bit form | value form>";
        return static_semiring(type, op_mult, op_add) ? "mxv_sell<static semiring>" : "mxv_sell<runtime ops>";
    }
    return splahip_mxv_variant(type, op_mult, op_add, early_exit);
}

const char* splahip_mxv_variant(int type, int op_mult, int op_add, int early_exit) {
    if (reference_order_calls() && type == SPLAHIP_FLOAT && op_add == SPLAHIP_PLUS && !early_exit) return "mxv_ordered<reference fold order: a lane per row, a wavefront per long row>";
    switch (pick_variant(type, op_mult, op_add, early_exit)) {
        case Variant::WaveTileStatic: return "mxv_wavetile<static semiring>";
        case Variant::WaveTileDynamic: return "mxv_wavetile<runtime ops>";
        case Variant::RowSeq: return "mxv_ordered<an op_add that must not be re-ordered: a lane per row, a wavefront per long row>";
        case Variant::RowSeqEarly: return "mxv_rowseq<early_exit>";
    }
    return "?";
}

int splahip_mxv_masked(splahip_plan plan, int type, uint32_t n_rows, uint32_t n_cols,
                       const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax,
                       const void* d_mask, const void* d_v, void* d_r,
                       int op_mult, int op_add, int op_select, uint32_t init_bits, int early_exit, void* stream) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    SPLAHIP_REQUIRE(plan->n_rows == n_rows, "plan was built for a different matrix (n_rows)");
    SPLAHIP_REQUIRE(type == SPLAHIP_INT || type == SPLAHIP_UINT || type == SPLAHIP_FLOAT, "unknown element type");
    SPLAHIP_REQUIRE(binop_valid_for(op_mult, type), "op_multiply not defined for this type");
    SPLAHIP_REQUIRE(binop_valid_for(op_add, type), "op_add not defined for this type");
    SPLAHIP_REQUIRE(op_select >= 0 && op_select < SPLAHIP_SELOP_COUNT, "unknown op_select");
    op_add = canonical_add(type, op_add, init_bits);// fp32 MIN / MAX from a NaN init: the reference's fold keeps init (ops.hpp) -> FIRST, the row kernel
    if (n_rows == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_Ap && d_mask && d_r, "null device pointer");
    SPLAHIP_REQUIRE(plan->nnz == 0 || (d_Aj && d_Ax && d_v), "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    // a BFS-like step over a structure-only matrix (one value word, a frontier vector, an idempotent fold): a row is settled by its first hit, whatever
    // early_exit says (mxv_first_hit.hpp; reference src/algorithm.cpp:97-99 is this call with early_exit = true)
    plan->fh_taken = 0;
    uint32_t xbits = 0;
    if (first_hit_family(type, op_mult, op_add) && uniform_values(plan, n_cols, &xbits) && d_r != d_mask && d_r != d_v && plan->nnz != 0) {
        bool take = false;
        int  rc   = type == SPLAHIP_INT ? first_hit_probe<int32_t>(plan, n_cols, d_Ap, d_Aj, d_mask, op_select, d_v, early_exit != 0, s, &take)
                                        : first_hit_probe<uint32_t>(plan, n_cols, d_Ap, d_Aj, d_mask, op_select, d_v, early_exit != 0, s, &take);
        if (rc != SPLAHIP_OK) return rc;
        if (take) {
            plan->fh_taken = 1;
            return type == SPLAHIP_INT ? launch_first_hit<int32_t>(plan, n_rows, d_Ap, d_Aj, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, xbits, early_exit, s)
                                       : launch_first_hit<uint32_t>(plan, n_rows, d_Ap, d_Aj, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, xbits, early_exit, s);
        }
    }
    // the column-ordered-blocks copy (splahip_csr_plan_attach_cob): associative + commutative op_add with an LDS atomic, no early exit
    // When the plan also holds a sliced-ELL copy, the semirings of its sparse-vector (bit) form whose skippable value is 0 (BAND / BOR, LAND / LOR,
    // ... on int / uint: a BFS step, where a few per cent of v are non-zero and half the rows are masked out) stay with it: it reads only the
    // selected rows' entries and one bit per column; every other call is faster over the column-ordered blocks (SPLAHIP_PREFER=cob|sell overrides).
    if (!early_exit && cob_serves_mxv(plan, type, op_mult, op_add, n_cols, stream) && d_r != d_mask && d_r != d_v &&
        ((reinterpret_cast<uintptr_t>(d_r) | reinterpret_cast<uintptr_t>(d_mask)) & 15u) == 0) {// (its row pass moves r and mask 16 bytes at a time)
        const char* pref     = knob_env("SPLAHIP_PREFER");
        const bool  sell_ok  = plan->sell && plan->sell->n_cols == n_cols && !sell_disabled() && (plan->sell->n_xslices == 0 || binop_is_commutative(op_add));
        const bool  use_sell = sell_ok && (pref ? pref[0] == 's' : sell_skip_family(type, op_mult, op_add) == 1);
        if (!use_sell) {
            // A mask that selects few ENTRIES (the reference skips unselected rows, src/cpu/cpu_mxv.hpp:89-92; its own perf test selects every 10th row,
            // tests/test_mxv.cpp:91-130): the blocks stream every entry of a range that holds a selected row, the CSR wave-tile kernels only the selected rows'.
            // Every masked call enqueues a small probe of its mask whose verdict lands in pinned host memory; the host enqueues whichever side the LATEST FINISHED
            // probe favours (no round trip; both sides are correct for any mask, so a stale verdict only costs time).  op_select = ALWAYS needs none of it.
            const Variant wv = pick_variant(type, op_mult, op_add, 0);
            const bool    wt = (wv == Variant::WaveTileStatic || wv == Variant::WaveTileDynamic) && plan->n_items != 0 &&
                            ((reinterpret_cast<uintptr_t>(d_Aj) | reinterpret_cast<uintptr_t>(d_Ax) | reinterpret_cast<uintptr_t>(d_Ap)) & 15u) == 0;
            if (wt && op_select != SPLAHIP_ALWAYS && rows_prepare(plan, d_Ap, s) != SPLAHIP_OK) return SPLAHIP_ERROR;
            MaskProbeArgs probe;// this call's probe (a hint for the NEXT call): the first workgroups of the row kernels' first launch, or a launch of its own in front of the blocks
            if (wt && cob_mask_probe_prepare(plan, op_select, &probe) != 0) {
                switch (type) {
                    case SPLAHIP_INT: return rows_typed<int32_t>(plan, d_Ap, d_Aj, d_Ax, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, plan->d_rows_ctl + 4, probe, s);
                    case SPLAHIP_UINT: return rows_typed<uint32_t>(plan, d_Ap, d_Aj, d_Ax, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, plan->d_rows_ctl + 4, probe, s);
                    default: return rows_typed<float>(plan, d_Ap, d_Aj, d_Ax, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, plan->d_rows_ctl + 4, probe, s);
                }
            }
            return cob_mxv_probed(plan, type, d_Ap, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, probe, stream);
        }
    }
    // the sliced-ELL copy (splahip_csr_plan_attach_sell) serves every associative op_add without early exit; its lanes fold in
    // entry order, so op_add need not be commutative.  SPLAHIP_SELL=0 keeps the CSR kernels (A/B runs, tests).
    // (rows with row-exclusive slices are folded lane-strided: then op_add must be commutative too — FIRST / SECOND keep the row kernel)
    const bool ref_order = reference_order_calls() && type == SPLAHIP_FLOAT && op_add == SPLAHIP_PLUS;// (the CSR kernels' ordered forms serve it: dispatch_typed)
    if (plan->sell && !early_exit && !ref_order && fold_is_reorderable(op_add, type) && (plan->sell->n_xslices == 0 || binop_is_commutative(op_add)) &&
        plan->sell->n_cols == n_cols && !sell_disabled() && d_r != d_mask && d_r != d_v &&// (its prep pass writes r before the kernel reads mask / v)
        ((reinterpret_cast<uintptr_t>(d_Ap) | reinterpret_cast<uintptr_t>(d_mask) | reinterpret_cast<uintptr_t>(d_v) | reinterpret_cast<uintptr_t>(d_r)) & 15u) == 0)// 16-byte accesses of Ap / mask / v / r
        return sell_mxv(plan, type, d_Ap, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, stream);
    switch (type) {
        case SPLAHIP_INT:
            return dispatch_typed<int32_t>(plan, n_rows, d_Ap, d_Aj, d_Ax, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, early_exit, s);
        case SPLAHIP_UINT:
            return dispatch_typed<uint32_t>(plan, n_rows, d_Ap, d_Aj, d_Ax, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, early_exit, s);
        default:
            return dispatch_typed<float>(plan, n_rows, d_Ap, d_Aj, d_Ax, d_mask, d_v, d_r, op_mult, op_add, op_select, init_bits, early_exit, s);
    }
}
}

#ifdef SPLAHIP_PHASE_TIMING
// DIAGNOSTIC BUILD ONLY: copy out (reset != 0: and clear) the per-item phase ticks of mxv_kernels.hpp, 4 x 2^18 words
extern "C" int splahip_debug_phase(unsigned int* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(splahip::g_phase), sizeof(unsigned int) * 4 * (1 << 18)) != hipSuccess) return SPLAHIP_ERROR;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(splahip::g_phase)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned int) * 4 * (1 << 18)) != hipSuccess) return SPLAHIP_ERROR;
    }
    return SPLAHIP_OK;
}
#endif
