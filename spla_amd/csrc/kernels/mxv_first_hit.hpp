// mxv_first_hit.hpp — the BFS-like masked SpMV over plain CSR: a row stops at its first hit.
//
// Reference semantics: src/cpu/cpu_mxv.hpp:88-103 (r[i] = init folded with every product of a selected row; early_exit stops a row once sum != init).
// A call WITHOUT early exit may stop a row early too whenever the rest of the row cannot change the fold.  That is the case for the traversal step of
// reference src/algorithm.cpp:97-99 and everything shaped like it:
//   * every stored value of the matrix is the same word x (known from a layout copy's attach pass: structure-only matrices),
//   * every non-zero entry of the input vector is the same word c (a frontier: found per call by vstat_kernel below),
//   * 0 annihilates op_mult and is op_add's identity (sell_skip_family == 1: BAND / MULT / LAND products, BOR or uint MAX folds), and op_add is idempotent.
// Then every product is 0 or p = mult(x, c), and a row's fold is add(init, p) if the row holds a hit — an entry whose column is non-zero in v — and init
// otherwise: the first hit settles the row.  The kernel reads column ids only (no value stream), and one BIT per column of a bitmap of v (n_cols / 8 bytes:
// L2-resident) instead of a word of v.  Rows of the largest BFS levels stop after a few entries: the R-MAT 22 step reads a fraction of the selected entries.
// When the vector of a call turns out NOT to be uniform (vflag[0] == 0) the same launch folds every row completely, sequentially, in the reference's
// order: correct for any vector, slow — the host only takes this path when the previous call's vector was uniform and dense enough (mxv.hip).
#pragma once

#include "common.hpp"
#include "ops.hpp"

namespace splahip {

    // One pass over v: the bitmap of its non-zero entries (bits[c >> 5] bit c & 31), and whether all of them hold the same word.
    // part: kVstatBlocks x 8 words of per-workgroup statistics (count, min, max of the non-zero words; sampled selected entries, sampled tail entries); ticket: one word, 0 between calls.
    // The last workgroup to finish reduces them: vflag[0] = 1 if uniform (or empty), vflag[1] = the word, vflag[2] = non-zero count; and sums the workgroups' samples of
    // what the first-hit walk would cost on this call (below); the verdicts and the call's sequence number go to pinned host memory (host_words[0..7]: the NEXT call's hint).
#ifndef SPLAHIP_VSTAT_BLOCKS
#define SPLAHIP_VSTAT_BLOCKS 128// (the closing ticket is a same-address atomic per workgroup: 32 / 64 / 128 / 256 / 512 workgroups -> 26.8 / 17.3 / 14.5 / 16.5 - 17.3 / 24.0 us at 4.2 M columns)
#endif
    constexpr uint32_t kVstatTpb    = 512;
    constexpr uint32_t kVstatBlocks = SPLAHIP_VSTAT_BLOCKS;// one workgroup per CU: the closing ticket is a same-address atomic per workgroup (~10 ns each, serialised)
    template<typename T>
    __global__ __launch_bounds__(kVstatTpb) void vstat_kernel(uint32_t n_cols, const T* __restrict__ v, uint32_t* __restrict__ bits, uint32_t* __restrict__ part,
                                                        uint32_t* __restrict__ ticket, uint32_t* __restrict__ vflag, volatile uint32_t* __restrict__ host_words, uint32_t seq,
                                                        uint32_t n_rows, const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ Aj, const T* __restrict__ mask,
                                                        int op_select) {
        __shared__ uint32_t s_red[3][kVstatTpb / kWave];
        __shared__ uint32_t s_last;
        uint32_t            count = 0, mn = 0xFFFFFFFFu, mx = 0;
        const uint32_t*     w     = reinterpret_cast<const uint32_t*>(v);
        // What would the first-hit walk cost on THIS call?  A sample of kVstatBlocks x 16 = 2048 rows, evenly spaced, 16 per workgroup (its first wave: four lanes
        // per row, one of the row's first four entries each; the loads run under the wave's share of the stream below): a selected row that holds more than four
        // entries and no hit among its first four is a TAIL row — the walk goes on into it.  The host takes the first-hit kernel while the tail rows' entries stay
        // a small share of the sampled selected entries (mxv.hip).
        // (three dependent rounds — mask / Ap, the column ids, the words of v — threaded through the wave's first step of the stream: the rounds wait for their own
        // loads only, the 32 stream loads issued in between stay in flight)
        uint32_t       s_sel = 0, s_tail = 0;
#ifdef SPLAHIP_VSTAT_NOSAMPLE
        const bool     sampler = false;// (diagnostic build: what the sample costs)
#else
        const bool     sampler = threadIdx.x < kWave;
#endif
        const uint32_t s_i     = threadIdx.x & 3u;
        uint32_t       s_k0 = 0, s_k1 = 0, s_c = 0, s_v = 0;
        bool           s_has = false;
        int            s_stage = 0;
        if (sampler) {
            const uint32_t step = max(1u, n_rows / (kVstatBlocks * 16u));
            const uint64_t row  = (uint64_t) (blockIdx.x * 16u + threadIdx.x / 4u) * step;
            const bool     sel  = row < n_rows && eval_sel(op_select, mask[row]);
            s_k0 = sel ? Ap[row] : 0u, s_k1 = sel ? Ap[row + 1] : 0u;
        }
        auto sample_columns = [&]() {
            s_has = s_k0 + s_i < s_k1;
            s_c   = s_has ? Aj[s_k0 + s_i] : 0u;
        };
        auto sample_values = [&]() { s_v = s_has ? w[s_c] : 0u; };
        // a wave takes 2048 consecutive columns per step: 32 coalesced loads in flight per lane, one ballot each = the 64 bits of the columns it covers; lane j keeps
        // ballot j and the wave stores its 32 words of 64 bits in one coalesced store (bits is 8-byte aligned: a hipMalloc'ed array)
        const uint32_t       lane    = lane_id();
        const uint32_t       wave    = (blockIdx.x * kVstatTpb + threadIdx.x) / kWave, n_waves = gridDim.x * (kVstatTpb / kWave);
        unsigned long long*  bits64  = reinterpret_cast<unsigned long long*>(bits);
        for (uint64_t base = (uint64_t) wave * 2048u; base < n_cols; base += (uint64_t) n_waves * 2048u) {
            uint32_t e[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const uint64_t c = base + (uint32_t) j * kWave + lane;
                e[j]             = c < n_cols ? w[c] : 0u;
            }
            if (sampler && s_stage == 0) sample_columns(), s_stage = 1;
            unsigned long long mine = 0ull;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const unsigned long long b = __ballot(e[j] != 0u);
                if ((int) lane == j) mine = b;
                if (e[j] != 0u) count += 1, mn = min(mn, e[j]), mx = max(mx, e[j]);
            }
            if (sampler && s_stage == 1) sample_values(), s_stage = 2;
            if (lane < 32u && base + (uint64_t) lane * kWave < n_cols) bits64[(base >> 6) + lane] = mine;
        }
        if (sampler) {
            if (s_stage == 0) sample_columns();
            if (s_stage <= 1) sample_values();
            uint32_t any = s_has && s_v != 0u ? 1u : 0u;
            any |= (uint32_t) __shfl_xor((int) any, 1, kWave);
            any |= (uint32_t) __shfl_xor((int) any, 2, kWave);
            if (s_i == 0u && s_k1 > s_k0) {
                s_sel = s_k1 - s_k0;
                if (!any && s_k1 - s_k0 > 4u) s_tail = s_k1 - s_k0;
            }
            s_sel  = wave_sum(s_sel);
            s_tail = wave_sum(s_tail);
        }
        count = wave_sum(count);
        mn    = ~wave_max(~mn);
        mx    = wave_max(mx);
        if (lane_id() == 0) s_red[0][threadIdx.x / kWave] = count, s_red[1][threadIdx.x / kWave] = mn, s_red[2][threadIdx.x / kWave] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint32_t k = 1; k < kVstatTpb / kWave; ++k) count += s_red[0][k], mn = min(mn, s_red[1][k]), mx = max(mx, s_red[2][k]);
            part[8 * blockIdx.x + 0] = count, part[8 * blockIdx.x + 1] = mn, part[8 * blockIdx.x + 2] = mx, part[8 * blockIdx.x + 3] = s_sel, part[8 * blockIdx.x + 4] = s_tail;
            __threadfence();
            s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (s_last == 0u) return;
        // the last workgroup: the other workgroups' slots — one 16-byte load per thread (nobody on this CU has read these lines in this launch, and the ticket's
        // acquire orders the loads behind the other workgroups' releases; agent-scope atomic loads would be issued and waited for one word at a time)
        count = 0, mn = 0xFFFFFFFFu, mx = 0;
        uint32_t sample_sel = 0, sample_tail = 0;
        for (uint32_t k = threadIdx.x; k < gridDim.x; k += kVstatTpb) {
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            const u4v p = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(part) + 2 * k);
            const u4v q = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(part) + 2 * k + 1);
            count += p.x, mn = min(mn, p.y), mx = max(mx, p.z), sample_sel += p.w, sample_tail += q.x;
        }
        count = wave_sum(count);
        mn    = ~wave_max(~mn);
        mx    = wave_max(mx);
        __syncthreads();
        if (lane_id() == 0) s_red[0][threadIdx.x / kWave] = count, s_red[1][threadIdx.x / kWave] = mn, s_red[2][threadIdx.x / kWave] = mx;
        __syncthreads();
        if (threadIdx.x == 0)
            for (uint32_t k = 1; k < kVstatTpb / kWave; ++k) mn = min(mn, s_red[1][k]), mx = max(mx, s_red[2][k]);
        uint32_t sel_entries = wave_sum(sample_sel), tail_entries = wave_sum(sample_tail);
        __syncthreads();
        if (lane_id() == 0) s_red[1][threadIdx.x / kWave] = sel_entries, s_red[2][threadIdx.x / kWave] = tail_entries;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint32_t k = 1; k < kVstatTpb / kWave; ++k) count += s_red[0][k];
            // (min / max were folded into thread 0's registers before the slots were reused)
            const uint32_t uniform = (count == 0u || mn == mx) ? 1u : 0u;
            for (uint32_t k = 1; k < kVstatTpb / kWave; ++k) sel_entries += s_red[1][k], tail_entries += s_red[2][k];
            vflag[0] = uniform, vflag[1] = count ? mx : 0u, vflag[2] = count;
            *ticket  = 0u;
            // two 16-byte stores to pinned host memory, the call's sequence number in both: the host trusts the words when the two numbers agree (no system-scope
            // fence and second round trip at the end of the launch)
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            u4v*      hw = reinterpret_cast<u4v*>(const_cast<uint32_t*>(host_words));
            const u4v a  = {uniform, count ? mx : 0u, count, seq};
            const u4v b  = {sel_entries, tail_entries, 0u, seq};
            __builtin_nontemporal_store(a, hw);
            __builtin_nontemporal_store(b, hw + 1);
        }
    }

    // One wave per 64 consecutive rows, a lane per row (see the walk below); a hub row without a hit would otherwise keep one lane busy for its 10^5 entries.
#ifndef SPLAHIP_FH_LANE
#define SPLAHIP_FH_LANE 64
#endif
#ifndef SPLAHIP_FH_UNROLL
#define SPLAHIP_FH_UNROLL 2// (measured on the R-MAT 22 traversal step, kernel + pass over v: 1 -> 56 - 63 us, 2 -> 53.5, 3 -> 56.7, 4 -> 63, 8 -> 92: what is read past the hit is wasted)
#endif
    constexpr uint32_t kFirstHitLane = SPLAHIP_FH_LANE;// entries of its own row a lane walks (SPLAHIP_FH_UNROLL at a time) before the wave takes the row over
    // VXM = false: the masked SpMV (bits = vstat_kernel's bitmap of the non-zero entries of v; vflag[0] uniform, [1] the word, [2] the non-zero count); early_exit: the
    //   call's flag — it only matters when v turns out not to be a frontier (the complete fold then stops a row like the reference does).
    // VXM = true: the pull form of vxm over the TRANSPOSED matrix (reference src/cpu/cpu_vxm.hpp:92-125): "rows" are output indices, bits = the bitmap of the
    //   STORED indices of the sparse input vector (an entry whose index is not stored gives no product), v = its dense image, products are mult(v, x), the fold
    //   starts at the first product (no init) and touched[row] says whether the output entry exists; vflag[0] / vflag[1] = max over the stored value words of
    //   ~word / of word (vx_range_kernel): one stored word c  <=>  ~vflag[0] == vflag[1].  Every product is then mult(c, x): the first stored neighbour settles
    //   the output entry.
    template<typename T, class Mult, class Add, bool VXM>
    __global__ __launch_bounds__(256) void mxv_first_hit_kernel(uint32_t n_rows, const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ Aj, const T* __restrict__ mask,
                                                                const T* __restrict__ v, T* __restrict__ r, Mult mult, Add add, int op_select, T init, uint32_t xbits,
                                                                const uint32_t* __restrict__ vflag, const uint32_t* __restrict__ bits, uint32_t* __restrict__ touched, int early_exit) {
        const uint32_t row     = blockIdx.x * 256 + threadIdx.x;
        const bool     in      = row < n_rows;
        const bool     sel     = in && eval_sel(op_select, mask[row]);
        const uint32_t k0      = sel ? Ap[row] : 0u;
        const uint32_t k1      = sel ? Ap[row + 1] : 0u;
        const bool     uniform = VXM ? ~vflag[0] == vflag[1] : vflag[0] != 0u;// (the same word for every lane of the launch)
        const T        x       = from_bits<T>(xbits);
        T              sum     = init;
        bool           found_any = false;
        if (uniform) {
            const T    p     = VXM ? mult(from_bits<T>(vflag[1]), x) : mult(x, from_bits<T>(vflag[1]));
            const T    hit   = VXM ? p : add(init, p);
            const bool moves = VXM ? true : (to_bits(hit) != to_bits(init) && vflag[2] != 0u);// (a product that leaves init alone: nothing to look for)
            // every lane walks the first kFirstHitLane entries of its own row, SPLAHIP_FH_UNROLL at a time (the rows of a large BFS level are settled within a few
            // entries, the long ones too); what is left of a long row without a hit so far is walked by the whole wave afterwards, 64 entries per step
            const uint32_t ke = min(k1, k0 + kFirstHitLane);
            bool           found_own = false;
            if (moves) {
                constexpr uint32_t KU = SPLAHIP_FH_UNROLL;
                for (uint32_t k = k0; k < ke; k += KU) {
                    uint32_t c[KU], w[KU], any = 0;
#pragma unroll
                    for (uint32_t u = 0; u < KU; ++u) c[u] = Aj[min(k + u, ke - 1)];
#pragma unroll
                    for (uint32_t u = 0; u < KU; ++u) w[u] = bits[c[u] >> 5];
#pragma unroll
                    for (uint32_t u = 0; u < KU; ++u) any |= w[u] >> (c[u] & 31u);
                    if ((any & 1u) != 0u) {
                        found_own = true;
                        break;
                    }
                }
            }
            if (found_own) sum = hit, found_any = true;
            unsigned long long todo = __ballot(moves && !found_own && ke < k1);
            while (todo != 0ull) {
                const int      src = __ffsll((long long) todo) - 1;
                const uint32_t a = __shfl(ke, src, kWave), b = __shfl(k1, src, kWave);
                bool           found = false;
                for (uint32_t k = a + lane_id(); !found && k - lane_id() < b; k += kWave) {
                    bool mine = false;
                    if (k < b) {
                        const uint32_t c = Aj[k];
                        mine             = ((bits[c >> 5] >> (c & 31u)) & 1u) != 0u;
                    }
                    found = __ballot(mine) != 0ull;
                }
                if ((int) lane_id() == src && found) sum = hit, found_any = true;
                todo &= todo - 1ull;
            }
        } else if (sel) {// any vector: the complete fold, in entry order
            if constexpr (VXM) {
                for (uint32_t k = k0; k < k1; ++k) {
                    const uint32_t c = Aj[k];
                    if (((bits[c >> 5] >> (c & 31u)) & 1u) == 0u) continue;
                    const T p = mult(v[c], x);
                    sum       = found_any ? add(sum, p) : p;
                    found_any = true;
                }
            } else {// (early_exit: the reference's rule, cpu_mxv.hpp:96-98 — the row stops behind the product that moves the sum off init)
                for (uint32_t k = k0; k < k1; ++k) {
                    sum = add(sum, mult(x, v[Aj[k]]));
                    if (early_exit && to_bits(sum) != to_bits(init)) break;
                }
            }
        }
        if constexpr (VXM) {
            if (in) {
                touched[row] = found_any ? 1u : 0u;
                if (found_any) r[row] = sum;
            }
        } else {
            if (in) r[row] = sum;
        }
    }

    // vflag[0] = max(~word), vflag[1] = max(word) over the stored values of a sparse vector (both 0 before the launch).  One pair of atomics per WORKGROUP, at most
    // 64 workgroups: same-address atomics serialise (one pair per wave of a 1024-workgroup grid cost 24 us for 10^5 values)
    template<int = 0>// (a template: the header is included by two translation units)
    __global__ __launch_bounds__(256) void vx_range_kernel(uint32_t nvals, const uint32_t* __restrict__ vx, uint32_t* __restrict__ vflag) {
        __shared__ uint32_t s_lo[256 / kWave], s_hi[256 / kWave];
        uint32_t            lo = 0xFFFFFFFFu, hi = 0;
        for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < nvals; k += gridDim.x * 256) {
            const uint32_t a = vx[k];
            lo = min(lo, a), hi = max(hi, a);
        }
        lo = ~wave_max(~lo);
        hi = wave_max(hi);
        if (lane_id() == 0) s_lo[threadIdx.x / kWave] = lo, s_hi[threadIdx.x / kWave] = hi;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint32_t k = 1; k < 256 / kWave; ++k) lo = min(lo, s_lo[k]), hi = max(hi, s_hi[k]);
            atomicMax(&vflag[0], ~lo);
            atomicMax(&vflag[1], hi);
        }
    }

}// namespace splahip
