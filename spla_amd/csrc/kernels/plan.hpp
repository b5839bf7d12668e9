// plan.hpp — the opaque splahip_plan: work decomposition of one device CSR matrix (built by mxv.hip, used by mxv / vxm).
#pragma once

#include <stdint.h>

// sliced-ELL copy of the matrix + hot-column table (sell.hip / mxv_sell.hpp); owned by the plan
struct splahip_sell_s {
    uint32_t  n_cols     = 0;
    size_t    bytes      = 0;// device memory the copy holds
    uint32_t  n_vrows    = 0;// virtual rows (chunks), padded to a multiple of 64
    uint32_t  n_slices   = 0;
    uint32_t  n_slots    = 0;// entries of Aj2 / Ax2 incl. in-slice padding (the allocation adds a read-ahead tail)
    uint32_t  n_hot      = 0;
    uint32_t  n_lrows    = 0;// rows longer than one chunk
    uint32_t  n_partials = 0;
    uint32_t  n_xslices  = 0;// leading row-exclusive slices (rows longer than kXL): selection is uniform per slice; the fold order inside
                             // such a row is lane-strided, so op_add must also be commutative when n_xslices > 0
    uint32_t* d_sorder   = nullptr;// n_slices: slice ids by descending width (the order the workgroups take them in)
    uint32_t* d_vrow     = nullptr;// n_vrows: row id | bit 31 "chunk of a long row"
    uint32_t* d_vlen     = nullptr;// n_vrows: entries of the chunk (0 = padding lane)
    uint32_t* d_vslot    = nullptr;// n_vrows: partial slot of a long row's chunk
    uint32_t* d_off      = nullptr;// n_slices + 1: first entry of every slice
    uint32_t* d_Aj2      = nullptr;
    uint32_t* d_Ax2      = nullptr;
    uint32_t* d_hot_cols = nullptr;// n_hot column ids in LDS-slot order
    uint32_t* d_hot_vals = nullptr;// n_hot words, packed per launch
    uint32_t* d_lr_row   = nullptr;// n_lrows
    uint32_t* d_lr_first = nullptr;// n_lrows + 1
    uint32_t* d_partial  = nullptr;// n_partials
    // "sparse input vector" form (int / uint semirings whose mult has 0 as annihilator and whose add has 0 as identity):
    uint32_t* d_rank_of  = nullptr;// n_cols: position of a column in the descending reference-count order (slot s < n_hot == rank s)
    uint32_t* d_Aj2r     = nullptr;// the entries' columns as ranks, same layout as d_Aj2
    uint32_t* d_vbits    = nullptr;// per launch: bit `rank` = v[column] != 0; followed by sell_vprep's per-workgroup statistics
    uint32_t* d_vranked  = nullptr;// per launch: v in rank order (written for the non-zero entries only)
    uint32_t  n_bit_words = 0;     // words of ONE bitmap; d_vbits = bitmap 0 | bitmap 1 | statistics (double-buffered: a call fills one and
                                   // clears the other for the next call)
    mutable uint32_t bit_buffer = 0;// which bitmap the next call fills
    // value statistics of the copy, taken on the first min-plus call of a type (sell.hip: ax_stats): -1 = not taken yet
    // every stored matrix value has the same bits (a graph's adjacency matrix of ones): the bit-form kernel then streams the rank words
    // only, 4 B per entry instead of 8 (taken once at attach time; sell.hip)
    bool     x_uniform = false;
    uint32_t x_bits    = 0;
    mutable int      ax_type    = -1;
    mutable uint32_t ax_bad     = 0;// some value is negative (INT / FLOAT) or NaN
    mutable uint32_t ax_max     = 0;// largest value's bits (INT / UINT; meaningful when !ax_bad)
    uint32_t* d_selbits   = nullptr;// per launch: bit `row` = op_select(mask[row]) (the slices' rows are length-sorted: a bit is 32x denser than
                                    // a mask word, and the popular lengths keep ascending row order, so a slice's 64 tests share a few lines)
    // entries whose column rank is beyond the LDS-resident bits: out of d_Aj2r (coded as a never-set bit), listed here by ascending rank
    uint32_t  n_cold       = 0;
    uint32_t* d_cold_rank  = nullptr;// n_cold
    uint64_t* d_cold_entry = nullptr;// n_cold: row << 32 | value bits
};

// column-ordered blocks copy of the matrix (cob.hip / mxv_cob.hpp); owned by the plan
struct splahip_cob_sched_s {// one work schedule: the pieces (consecutive chunks of one range), one workgroup each, largest first
    uint32_t  n_tiles   = 0;// pieces = workgroups of the kernel
    uint32_t  n_multi   = 0;// ranges cut into several pieces (each has a shared accumulator + a ticket)
    uint32_t* d_tiles   = nullptr;// n_tiles x CobTile
    uint32_t* d_ranges  = nullptr;// n_ranges x CobRange (n_pieces differs between the schedules)
    uint32_t  lds_bytes = 0;// dynamic LDS of a workgroup: the larger of a hub slice and a natural slice + the spare words
    // the run-time schedule (mxv_cob_dyn.hpp): a persistent grid of `slots` workgroups shares every range's chunks through the range's cursor
    uint32_t  slots     = 0;      // workgroups of the persistent launch (one per CU)
    uint32_t  n_jobs    = 0;      // listed jobs: range ids in start order, a range once per worker the cost model gives it
    uint32_t* d_jobs    = nullptr;
    uint32_t* d_dyn_ctl = nullptr;// [0] next listed job (== slots between calls), [1] workgroups that have left (0 between calls)
};
struct splahip_cob_s {
    uint32_t  n_rows   = 0;
    uint32_t  n_cols   = 0;
    uint32_t  R        = 0;// slots per range (the range's result slice lives in LDS)
    uint32_t  Rn       = 0;// slots of a natural range (>= R; > R: an fp32 sum keeps the natural ranges in float words)
    uint32_t  acc_stride = 0;// words between two shared accumulators: max(R, Rn)
    uint32_t  hub_finish  = 0;// who finishes the hub ranges' rows: 0 the range's last piece, 1 a kernel of its own behind the main launch (SPLAHIP_COB_HUB_FINISH)
    uint32_t  ticket_words = 0;// words of d_ticket: arrival counters, then the finishers' counters
    bool      half_groups = false;// two double slices fit a CU's LDS: the fp32 PLUS forms run 8-wave workgroups, two per CU
    uint32_t  col_bits = 0;// a chunk never straddles a window of 2^col_bits columns (so its columns span < 2^16: the decoded word is slot << 16 | column - hdr)
    uint32_t  n_chunks = 0;// 256-slot chunks
    uint32_t  n_ranges = 0;// hub ranges first, then the natural ranges in row order
    uint32_t  n_hub_ranges = 0;
    uint32_t  n_hub_rows   = 0;
    size_t    bytes    = 0;// device memory the copy holds
    // [0]: 4-byte slice words (int / uint, LOR / LAND, fp32 MIN / MAX keys), [1]: 8-byte slice words (fp32 PLUS); they differ in the workgroups a CU
    // can hold.  n_tiles == 0: this slice width does not fit the LDS at this R
    splahip_cob_sched_s sched[2];
    uint32_t* d_hdr    = nullptr;// n_chunks: column of the chunk's first slot
    uint16_t* d_sl     = nullptr;// n_chunks x 256: slot in range (padding / bridging slots: R + lane)
    uint8_t*  d_dl     = nullptr;// n_chunks x 256: column delta to the slot before it in sorted order (0 for a chunk's first slot)
    uint32_t* d_ax     = nullptr;// n_chunks x 256 value words
    uint16_t* d_slot16 = nullptr;// n_rows (padded to 8): slot of the row in its natural range, 0xFFFF no entry, 0xFFFE hub row
    uint32_t* d_hub_row = nullptr;// n_hub_ranges x R: row of a hub slot, 0xFFFFFFFF padding
    uint32_t* d_acc    = nullptr;// max(sched[].n_multi) x R words: shared accumulators, at the identity of `acc_identity` between calls
    uint32_t* d_ticket = nullptr;// max(sched[].n_multi) arrival counters, 0 between calls
    bool      dyn      = false;  // calls run the run-time schedule (mxv_cob_dyn.hpp); then d_acc holds one accumulator per RANGE
    uint32_t* d_spans  = nullptr;// n_ranges x (first chunk, chunks)
    uint32_t* d_dyn_words = nullptr;// n_ranges cursors, n_ranges state words (all 0 between calls), n_ranges finish flags (the number of the call that set them)
    uint32_t* d_fin_tasks = nullptr;// the finishing kernel's row tasks: n_fin_tasks x (row0, row1, -, range): <= 2048 consecutive rows of one natural range
    uint32_t  n_fin_tasks = 0;
    uint16_t* d_dyn_weight = nullptr;// per range: modelled cost of one chunk, 1/16 of a plain stream chunk's
    uint32_t* d_dyn_ranges = nullptr;// n_ranges x CobRange as the row pass wants them
    mutable uint32_t dyn_seq = 0;    // call counter: a range whose workers merged carries it in its finish flag
    uint32_t  dyn_min_hub = 0, dyn_min_nat = 0;// unclaimed rounds per worker below which a workgroup without a listed job no longer joins a hub / natural range
    mutable uint32_t acc_identity = 0;
    mutable bool     acc_armed    = false;
    uint32_t* d_vbits  = nullptr;// per call: bit c = v[c] != 0 (sparse-input-vector form, mxv_cob.hpp MODE 1 / 2)
    // hot-column relabelling (round 4; cob.hip): the K most referenced columns of a matrix whose labels carry no locality take the ids 0 .. K-1 in descending
    // reference-count order, every other column c the id n_hot_pad + c; a call gathers from d_vperm = [v[hot_cols[0 .. K)] | padding | v] (one pass per call)
    uint32_t  n_vcols   = 0;      // columns of the vector the kernels gather from: n_cols, or n_hot_pad + n_cols
    uint32_t  n_hot     = 0;      // K (0: no relabelling)
    uint32_t  n_hot_pad = 0;      // K rounded up to whole 64-word blocks
    uint32_t* d_hot_cols = nullptr;// K column ids in descending reference-count order
    uint32_t* d_vperm   = nullptr;// n_hot_pad + n_cols words, written per call
    // masked calls (round 4): the mask probe — d_probe: [0..1] entries of the selected rows (64 bit; a sample of the rows, scaled), [2] workgroups done, [4..5] what the
    // selection costs the row kernels (zero between probes); h_probe (pinned host memory the probe's last workgroup writes): [0..1] the estimate, [2] the verdict (1: the selected-rows kernels of mxv_rows.hpp
    // would serve such a mask faster than the blocks), [3] the probe's sequence number (0: none has finished yet)
    uint32_t*          d_probe   = nullptr;
    volatile uint32_t* h_probe   = nullptr;
    uint32_t*          d_h_probe = nullptr;// h_probe as the device sees it
    mutable uint32_t   probe_seq = 0;
    mutable int        last_rows = 0;// which side served the last masked call (1: the selected-rows kernels)
    uint32_t           heavy_slots = 0, heavy_deg = 0xFFFFFFFFu;// the probe's exact stratum: the first heavy_slots hub slots hold the rows of degree >= heavy_deg
    bool      x_uniform = false; // every stored value of the matrix is the same word (found at attach time) ...
    uint32_t  x_bits    = 0;     // ... this one: the sparse-vector form then does not read the value stream
    // value statistics of the copy's value words, per type (found by the first call of a type that needs them): any negative / NaN value,
    // the largest value (int / uint), any -0.0f (float)
    mutable int      ax_type = -1;
    mutable uint32_t ax_bad = 1, ax_max = 0, ax_negzero = 1;
};

struct splahip_plan_s {
    uint32_t  n_rows  = 0;
    uint32_t  nnz     = 0;
    uint32_t  n_items = 0;
    uint32_t  n_long  = 0;         // rows cut into segment items
    uint32_t* d_item_row   = nullptr;// n_items + 1
    uint32_t* d_item_k0    = nullptr;// n_items + 1
    uint32_t* d_partial    = nullptr;// n_items words of scratch (per-segment partial results)
    uint32_t* d_partial_ok = nullptr;// n_items words of scratch (vxm: does the segment hold a product)
    // call workspace, grown on demand and kept with the plan so steady-state calls do not allocate (vxm pull form)
    void*     d_ws         = nullptr;
    size_t    ws_bytes     = 0;
    // masked calls that select few entries (mxv_rows.hpp): the list of selected short rows of the running call, and the plan's segment items
    uint32_t* d_rows_list  = nullptr;// per block of 2048 rows: quads of its selected short rows -> offsets in the call's virtual quad stream (+ the total), built on the first such call
    bool      rows_dirty   = false;  // a selected-rows call failed to launch: both sets of list lengths are cleared before the next one
    uint32_t  rows_parity  = 0;      // which of the two sets of list lengths in d_rows_ctl ([0..3] / [8..11]) the next selected-rows call uses
    uint32_t* d_rows_ctl   = nullptr;// [0] short rows, [1] medium rows, [3] segment items listed by the running call, [2] the plan's long rows (listed once), [4..7] the open gate
    uint32_t* d_long_items = nullptr;// the segment items of the running call's selected long rows
    uint32_t* d_long_rows  = nullptr;// the plan's long rows (more than 508 entries), in no particular order
    uint32_t* d_long_first = nullptr;// ... and the first segment item of each
    uint32_t  n_long_items = 0;
    bool      rows_ready   = false;
    // BFS-like calls over a structure-only matrix (mxv_first_hit.hpp): per-call bitmap and statistics of the input vector, the verdict in pinned host memory
    uint32_t*          d_fh_bits  = nullptr;// one bit per column (+ padding to whole 2048-column steps)
    uint32_t*          d_fh_ctl   = nullptr;// [0..3] vflag (uniform, the word, non-zero count), [4] ticket, [8 ..] per-workgroup statistics
    volatile uint32_t* h_fh       = nullptr;// pinned: [0] uniform, [1] the word, [2] non-zero count, [3] sequence number of the call that wrote them
    uint32_t*          d_h_fh     = nullptr;// its device address
    uint32_t           fh_cols    = 0;
    uint32_t           fh_seq     = 0;
    uint32_t           fh_misses  = 0;      // consecutive calls whose hint said "not a frontier": the probe backs off
    int                fh_taken   = 0;      // did the last BFS-like call run on the first-hit kernel
    // what the owner of the value array says about it (splahip_csr_plan_set_uniform_values): -1 nothing, 0 several words, 1 every stored value is values_bits
    int                values_uniform = -1;
    uint32_t           values_bits    = 0;
    uint64_t           products_hint  = 0;  // products of the NEXT pull call over this (transposed) plan, if its caller counted them (splahip_csr_plan_hint_products; one shot)
    splahip_sell_s* sell   = nullptr;// splahip_csr_plan_attach_sell
    splahip_cob_s*  cob    = nullptr;// splahip_csr_plan_attach_cob
};

namespace splahip {
    // BFS-like semirings: 0 annihilates op_mult and is op_add's identity (sell_skip_family == 1) and op_add is idempotent (mxv_first_hit.hpp)
    bool first_hit_family(int type, int op_mult, int op_add);
    // every stored value of the plan's matrix is one word: said by the owner of the arrays, or known from the attach pass of a layout copy
    bool uniform_values(const splahip_plan_s* plan, uint32_t n_cols, uint32_t* xbits);
    void sell_release(splahip_sell_s* sell);
    void cob_release(splahip_cob_s* cob);
    // does the plan hold a column-ordered-blocks copy that can serve this op_add on this type (an LDS atomic exists, SPLAHIP_COB != 0)
    bool cob_serves(const splahip_plan_s* plan, int type, int op_add, uint32_t n_cols);
    // the masked SpMV's question: the above, plus fp32 MIN / MAX over op_mult = PLUS when the matrix holds no -0.0f (first asked: one value pass on `stream`)
    bool cob_serves_mxv(const splahip_plan_s* plan, int type, int op_mult, int op_add, uint32_t n_cols, void* stream);
    // masked SpMV over the column-ordered-blocks copy (associative + commutative op_add with an LDS atomic, no early exit); SPLAHIP status
    int cob_mxv(const splahip_plan_s* plan, int type, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                uint32_t init_bits, void* stream);
    // masked calls (op_select other than ALWAYS): enqueue a probe of the mask (how many entries does the selection cover) whose verdict lands in pinned host memory,
    // and return the verdict of the latest probe that has FINISHED — 1: the selected-rows kernels (mxv_rows.hpp) serve such a mask faster than the blocks, which
    // stream the whole matrix whatever the mask.  Both compute the reference's result for any mask: the verdict is a hint that lags by a call.
    struct MaskProbeArgs;// (mask_probe.hpp)
    int  cob_mask_probe_prepare(const splahip_plan_s* plan, int op_select, MaskProbeArgs* out);
    void cob_mask_probe_launch(const MaskProbeArgs& a, int type, const uint32_t* Ap, const void* mask, int op_select, void* stream);
    int  cob_mxv_probed(const splahip_plan_s* plan, int type, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                        uint32_t init_bits, const MaskProbeArgs& probe, void* stream);
    // the same over a second column part of the same rows: r[row] = add(r[row], this part's fold) where the row is selected and holds an entry here
    int cob_mxv_accumulate(const splahip_plan_s* plan, int type, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add,
                           int op_select, void* stream);
    // the pull form of vxm with every index of v stored, over the copy attached to the TRANSPOSED matrix's plan (out values + touched flags)
    bool cob_vxm_serves(const splahip_plan_s* plan_T, int type, int op_add, uint32_t n_in);
    int  cob_vxm(const splahip_plan_s* plan_T, int type, const uint32_t* Tp, const void* mask, const void* v, void* out, uint32_t* touched, int op_mult, int op_add,
                 int op_select, void* stream);
    // may the sparse-vector (BITS) form serve this semiring: 0 no, 1 zero family (0 annihilates op_mult and is op_add's identity), 2 min-plus
    // family (an entry whose v value equals init cannot lower the minimum when no matrix value is negative: checked per call)
    int sell_skip_family(int type, int op_mult, int op_add);
    // the pull form of vxm with every index of v stored, over the sliced-ELL copy attached to the TRANSPOSED matrix's plan: out values and
    // touched flags per output column (associative + commutative op_add); returns a SPLAHIP status
    int sell_vxm(const splahip_plan_s* plan_T, int type, const uint32_t* Tp, const void* mask, const void* v, void* out, uint32_t* touched, int op_mult,
                 int op_add, int op_select, void* stream);
    // the sliced-ELL masked SpMV (associative op_add, no early exit); returns a SPLAHIP status
    int sell_mxv(const splahip_plan_s* plan, int type, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                 uint32_t init_bits, void* stream);
}// namespace splahip
