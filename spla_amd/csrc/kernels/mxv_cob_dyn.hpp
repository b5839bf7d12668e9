// mxv_cob_dyn.hpp — the column-ordered blocks (mxv_cob.hpp) walked by a PERSISTENT grid that shares every range's chunks at run time.
//
// Semantics: the same as cob_mxv_kernel (reference Algo_mxv_masked_cpu<T>::execute, src/cpu/cpu_mxv.hpp:88-103; the pull form of vxm,
// src/cpu/cpu_vxm.hpp:92-125): the LDS slice of a range, the products added with the LDS atomic of op_add, the row pass of mxv_cob.hpp.
//
// Why (profiles/r03_notes.md 4 / 9, r04_notes.md 2 / 5): the static launch cuts a range into PIECES of modelled cost, one workgroup each.  A piece's real
// speed is 80 - 700 ns per chunk depending on how cold its lines of v are and on what the rest of the chip is doing, so the pieces pack to 79 - 83 % of the
// CU-time (R-MAT 23: 58 ms-CU of work in a 274 us span) and the launch ends with ~40 us in which a few dozen CUs walk the last, coldest pieces.  Cutting
// finer costs every piece its zeroing, pipeline fill and merge; claiming the static pieces from a queue (round 4) moved nothing — the tail is the pieces'
// own length.  Here the unit that is handed out at run time is a ROUND of 128 consecutive chunks (256 KB of the two streams, ~13 us of one CU):
//   * one workgroup per CU for the whole launch; it JOINS a range (state word: joined | done << 16), zeroes its slice, and claims rounds from the range's
//     cursor (thread 0: one memory-side atomic per round, asked for two rounds ahead, handed to the 16 waves through a ring of tagged LDS words) until the
//     cursor is past the range's last round — every worker of a range therefore stops within one round of the others, whatever the chunks cost;
//   * then its slice goes into the range's accumulator in global memory: a worker that was alone from start to end (compare-and-swap 1|0 -> 1|1 on the
//     state word; nobody can join a closed range) stores its words, every other worker adds them with memory-side atomics of op_add; the ROWS of every
//     range — hub and natural — are finished by cob_finish_kernel behind the launch (a natural range's row pass is 15 - 60 us on one CU);
//   * and it takes the next job: first from the launch's job list (range ids in start order, a range listed once per worker the cost model gives it,
//     claimed through a counter), then — the list used up — the range with the most unclaimed rounds per worker (one pass over the cursors), as long as
//     joining still pays (SPLAHIP_COB_DYN_MIN_HUB / _NAT rounds per worker).  No workgroup ever waits for another workgroup: nothing spins on global memory
//     (inside a workgroup a wave that runs a whole turn ahead of wave 0 waits for the ring word wave 0 is about to write).
//   * the last workgroup to leave re-arms the cursors, states and counters for the next call.
// What did NOT work (profiles/r05_notes.md): claims by the WAVE (8 or 16 chunks of its own per claim) — correct, 3 x slower: 16 waves x 2 streams walking
// spans of their own 1 KB a step run the HBM at a third of the rate of the same waves reading neighbouring KBs; the atomics themselves are free
// (tools/exp/atomic_bubble.hip: one returning agent-scope atomic per wave and 16 KB of stream costs 1 % unless thousands of waves share its address).
#pragma once

#include "mxv_cob.hpp"

namespace splahip {

    struct CobDynSpan {
        uint32_t first_chunk;
        uint32_t n_chunks;
    };
    struct CobDynArgs {
        const uint32_t*   jobs;  // range ids in start order (a range appears once per worker the cost model gives it)
        const CobDynSpan* spans; // per range: its chunks
        uint32_t*         cursor;// per range: next round to claim (0 between calls)
        uint32_t*         state; // per range: workers joined | workers done << 16 (0 between calls)
        uint32_t*         ctl;   // [0] next job of the list (== gridDim.x between calls), [1] workgroups that have left (0 between calls)
        uint32_t          n_jobs;
        uint32_t          n_ranges;
        const uint16_t*   weight;// per range: modelled cost of one of its chunks in 1/16 of a plain stream chunk's (16 = no gather lines at all)
        uint32_t*         fin_flag;// per range: == seq: its rows are finished by the kernel behind the launch (its workers merged); anything else: done here
        uint32_t          seq;   // this call's number (never 0)
        uint32_t          min_cost_hub;// a workgroup without a listed job joins a hub range that still holds this much unclaimed weighted cost (chunks x weight) per worker ...
        uint32_t          min_cost_nat;// ... a natural range (its merge is a 128 KB slice, its owner then merges too): this much
    };
    constexpr uint32_t kCobDynNone = 0xFFFFFFFFu;

#ifdef SPLAHIP_COB_TIMING
    static __device__ unsigned int g_cob_tn;// visits recorded in g_cob_t (DIAGNOSTIC BUILD)
#endif

    // What a worker does behind its claim loop.  A worker of a NATURAL range that was alone from start to end (compare-and-swap 1|0 -> 1|1 on the state word:
    // nobody can join a closed range) finishes the rows from its LDS slice, like a single piece of the static launch.  Every other worker puts its slice
    // into the range's accumulator in global memory (hub ranges: always; a lone worker stores 16 bytes per lane, workers that share a range add with
    // memory-side atomics of op_add) and the rows are finished by cob_finish_kernel behind the launch — a shared natural range says so in fin_flag.
    template<typename T, int ACC, bool VXM, unsigned TPB>
    __device__ __noinline__ void cob_dyn_range_end(unsigned char* s_raw, uint32_t* s_ctl, const uint32_t q, const bool hub, const CobRange rg, const uint16_t* __restrict__ slot16,
                                                   const uint32_t* __restrict__ hub_row, uint32_t* __restrict__ acc, uint32_t* __restrict__ state, uint32_t* __restrict__ fin_flag,
                                                   uint32_t seq, uint32_t* __restrict__ touched, uint32_t n_rows, uint32_t R, uint32_t acc_stride, uint32_t accumulate,
                                                   const T* __restrict__ mask, T* __restrict__ r, DynBin add, int op_select, T init) {
        using word = typename CobSlice<T, ACC>::word;
        constexpr bool  kWide = sizeof(word) == 8;
        word* const     s_acc = reinterpret_cast<word*>(s_raw);
        uint32_t* const gacc  = acc + (size_t) q * acc_stride;
        if (threadIdx.x == 0) {
            uint32_t expect = 1u;
            // (bit 31: CLOSED for good.  "joined == done" alone does not say so: a worker that is turned back has already counted itself in, and the NEXT arrival then
            // saw joined != done, joined the finished range, merged an empty slice and raised fin_flag — the kernel behind the launch overwrote the lone worker's
            // rows with the identity: the intermittent mismatch of the three-workers-per-range test, tools/exp/exp_r5_dyn_flaky.py)
            s_ctl[3] = __hip_atomic_compare_exchange_strong(state + q, &expect, 0x80000000u | 1u | (1u << 16), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
        }
        __syncthreads();
        const bool solo = s_ctl[3] != 0;
        if (solo && !hub) {
            if constexpr (kWide) {// round the double slice to fp32 words in place
                constexpr uint32_t KC = TPB >= 1024 ? 20 : 24;
                float              f[KC];
#pragma unroll
                for (uint32_t k = 0; k < KC; ++k) {
                    const uint32_t i = threadIdx.x + k * TPB;
                    f[k]             = i < R ? (float) s_acc[i] : 0.0f;
                }
                __syncthreads();
                float* s_f = reinterpret_cast<float*>(s_raw);
#pragma unroll
                for (uint32_t k = 0; k < KC; ++k) {
                    const uint32_t i = threadIdx.x + k * TPB;
                    if (i < R) s_f[i] = f[k];
                }
                __syncthreads();
            }
            cob_row_pass<T, ACC, VXM, TPB, false>(reinterpret_cast<const uint32_t*>(s_raw), rg, slot16, hub_row, touched, n_rows, R, accumulate, mask, r, add, op_select, init);
            return;
        }
        if (solo) {
            uint4* g4 = reinterpret_cast<uint4*>(gacc);
            for (uint32_t i = threadIdx.x; i < R / 4; i += TPB) {
                if constexpr (kWide) g4[i] = make_uint4(to_bits((float) s_acc[4 * i]), to_bits((float) s_acc[4 * i + 1]), to_bits((float) s_acc[4 * i + 2]), to_bits((float) s_acc[4 * i + 3]));
                else
                    g4[i] = reinterpret_cast<const uint4*>(s_raw)[i];
            }
            return;
        }
        for (uint32_t i = threadIdx.x; i < R; i += TPB) {
            uint32_t w;
            if constexpr (kWide) w = to_bits((float) s_acc[i]);
            else
                w = s_acc[i];
            if (w != cob_identity<T, ACC>()) cob_merge_global<T, ACC>(gacc + i, w);
        }
        if (threadIdx.x == 0) {
            if (!hub) __hip_atomic_store(fin_flag + q, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ctl[3] = __hip_atomic_fetch_add(state + q, 1u << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // The rows of every range, behind the launch: blocks [0, hub_blocks): a thread per hub slot (cob_hub_finish_kernel's body); the blocks behind them: a
    // task of at most 2048 consecutive rows of ONE natural range each (tasks: row0, row1, -, range) — skipped unless the range's workers shared it (fin_flag) —,
    // 8 rows per thread: the row pass of mxv_cob.hpp over the range's accumulator, every word set back to the identity by its reader.
    template<typename T, int ACC, bool VXM>
    __global__ __launch_bounds__(256) void cob_finish_kernel(uint32_t hub_blocks, uint32_t n_hub_slots, uint32_t R, uint32_t acc_stride, uint32_t* __restrict__ acc,
                                                             const uint32_t* __restrict__ hub_row, const CobRange* __restrict__ tasks, const uint32_t* __restrict__ fin_flag, uint32_t seq,
                                                             const uint16_t* __restrict__ slot16,
                                                             uint32_t n_rows, const T* __restrict__ mask, T* __restrict__ r, uint32_t* __restrict__ touched, uint32_t accumulate, DynBin add,
                                                             int op_select, T init) {
        if (blockIdx.x < hub_blocks) {
            const uint32_t i = blockIdx.x * 256 + threadIdx.x;
            if (i >= n_hub_slots) return;
            const size_t   ia  = (size_t) (i / R) * acc_stride + i % R;
            const uint32_t w   = acc[ia];
            const uint32_t row = hub_row[i];
            acc[ia]            = cob_identity<T, ACC>();
            if (row == 0xFFFFFFFFu) return;
            cob_hub_row<T, ACC, VXM>(w, row, mask, r, touched, accumulate, add, op_select, init);
            return;
        }
        CobRange       t = tasks[blockIdx.x - hub_blocks];
        const uint32_t q = t.n_pieces;
        if (fin_flag[q] != seq) return;// its lone worker has finished the rows
        t.hub_base       = 0xFFFFFFFFu;
        cob_row_pass<T, ACC, VXM, 256, true, true>(acc + (size_t) q * acc_stride, t, slot16, hub_row, touched, n_rows, R, accumulate, mask, r, add, op_select, init);
    }

    // A workgroup without a listed job: the range with the most unclaimed rounds per worker, if joining it still pays; kCobDynNone otherwise.
    // (agent-scope loads: the cursors and states are written by memory-side atomics of other XCDs.)  Workgroups that look at the same moment see the same
    // numbers: each scales a range's figure by a factor of its own in [0.75, 1] so that they spread over the near-equal candidates.
    template<unsigned TPB>
    __device__ __noinline__ uint32_t cob_dyn_steal(const CobDynArgs d, uint32_t n_hub_ranges, unsigned long long* s_best) {
        if (threadIdx.x == 0) *s_best = 0ull;
        __syncthreads();
        unsigned long long best = 0ull;
        for (uint32_t q = threadIdx.x; q < d.n_ranges; q += TPB) {
            const uint32_t cur = __hip_atomic_load(d.cursor + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t st  = __hip_atomic_load(d.state + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t nc  = d.spans[q].n_chunks;
            if ((unsigned long long) cur * 128ull >= nc) continue;
            const uint32_t joined = st & 0xFFFFu, done = (st >> 16) & 0x7FFFu;
            if ((st >> 31) != 0u || (joined != 0u && joined == done)) continue;// closed (a range without a chunk left is skipped above already)
            const uint32_t active = joined > done ? joined - done : 0u;
            // the range that would finish last: unclaimed chunks x the modelled cost of one of its chunks (a cold range's chunk takes 2 - 3 x a hot one's:
            // its gathers), per worker, this workgroup included
            const unsigned long long per = (unsigned long long) (nc - cur * 128u) * d.weight[q] / (active + 1u);
            if (per < (q < n_hub_ranges ? d.min_cost_hub : d.min_cost_nat)) continue;
            const uint32_t jit = 256u - ((blockIdx.x * 37u + q * 101u) & 63u);
            best               = max(best, ((unsigned long long) min((per * jit) >> 8, 0xFFFFFFFFull) << 32) | (q + 1u));
        }
        for (unsigned off = kWave / 2; off > 0; off >>= 1) best = max(best, (unsigned long long) __shfl_xor(best, off, kWave));
        if (lane_id() == 0 && best) (void) __hip_atomic_fetch_max(s_best, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        const unsigned long long b = *s_best;
        __syncthreads();
        return b ? (uint32_t) (b & 0xFFFFFFFFull) - 1u : kCobDynNone;
    }

    // (one workgroup per CU: the register budget is that of 4 waves per SIMD for every form — the bit forms' 64-register cut of cob_mxv_kernel spills 32 words here)
    template<typename T, class Mult, class Add, int ACC, bool VXM, unsigned TPB, int MODE = 0>
    __global__ __launch_bounds__(TPB, 4) void cob_dyn_kernel(CobArgs a, CobDynArgs d, const T* __restrict__ v, const T* __restrict__ mask, T* __restrict__ r,
                                                                                           Mult mult, Add add, int op_select, T init) {
        using word = typename CobSlice<T, ACC>::word;
        constexpr bool kWide = sizeof(word) == 8;
        extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
        __shared__ uint32_t           s_ctl[4];// [0] joined (1) / turned back (0), [1] the next listed job, [3] scratch of the range's end
        __shared__ unsigned long long s_best;
        __shared__ uint32_t           s_ring[4], s_prog[TPB / kWave];// the claimed rounds' bases (tagged with the turn they are for); the turn every wave is in
        word* const     s_acc   = reinterpret_cast<word*>(s_raw);
        uint32_t* const s_acc32 = reinterpret_cast<uint32_t*>(s_raw);
        const unsigned  lane    = lane_id();
        const uint32_t  cmask   = (1u << a.col_bits) - 1u;
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const cob_u2*   sl2 = reinterpret_cast<const cob_u2*>(a.sl) + lane;
        const uint32_t* dl1 = reinterpret_cast<const uint32_t*>(a.dl) + lane;
        const u4*       ax4 = reinterpret_cast<const u4*>(a.ax) + lane;
        const uint32_t* hb  = a.hdr;
        uint32_t        job = blockIdx.x;
#ifdef SPLAHIP_COB_TIMING
        if (threadIdx.x == 0) s_ctl[2] = 0;
        __syncthreads();
#endif
        for (;;) {
            uint32_t q;
            if (job < d.n_jobs) q = d.jobs[job];
            else {
                q = __builtin_amdgcn_readfirstlane(cob_dyn_steal<TPB>(d, a.n_hub_ranges, &s_best));
                if (q == kCobDynNone) break;
            }
            const bool       nat = q >= a.n_hub_ranges;
            const uint32_t   R   = nat ? a.Rn : a.R;
            const CobDynSpan sp  = d.spans[q];
            uint32_t* const  cur_p    = d.cursor + q;
            [[maybe_unused]] const bool narrow = kWide && nat && a.Rn > a.R;
            // A claim is a ROUND of 128 consecutive chunks (256 KB of the two streams) for the whole workgroup: wave w walks chunks base + w + 16 i, i = 0 .. 7 —
            // the static pieces' interleave, so that the 16 waves read neighbouring KBs at the same time (a wave walking a span of its own, 1 KB a step, ran the
            // streams at a third of the rate: profiles/r05_notes.md).  Thread 0 claims; the bases reach the other waves through a ring of four tagged LDS words.
            constexpr uint32_t kRound   = 128;
            constexpr uint32_t kNoRound = 0x00FFFFFFu;// base field of a claim behind the range's last round (a chunk index is below 2^24)
            const uint32_t     n_rounds  = (sp.n_chunks + kRound - 1u) / kRound;
            const uint32_t     end_chunk = sp.first_chunk + sp.n_chunks;
            auto               ring_word = [&](uint32_t body, uint32_t round) -> uint32_t {
                return 0x80000000u | ((body & 0x7Fu) << 24) | (round < n_rounds ? sp.first_chunk + round * kRound : kNoRound);
            };
            // (the address goes through a register the compiler cannot see through: with a provably uniform address its atomic optimizer rewrites the add into
            // a wave-level reduction whose broadcast reads — and waits for — the result right behind the instruction)
            uint32_t opaque_zero;
            asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
            uint32_t* const cur_pv = cur_p + opaque_zero;
            // join, claim the first two rounds; the next listed job is asked for now and read when this range is done
            if (threadIdx.x == 0) {
                const uint32_t old = __hip_atomic_fetch_add(d.state + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t rd0 = __hip_atomic_fetch_add(cur_pv, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t nxt = job < d.n_jobs ? __hip_atomic_fetch_add(d.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : d.n_jobs;
                s_ctl[0]           = (old >> 31) != 0u || ((old & 0xFFFFu) != 0u && (old & 0xFFFFu) == ((old >> 16) & 0x7FFFu)) ? 0u : 1u;// closed: its last worker has taken the row pass
                s_ctl[1]           = nxt;
                s_ring[0] = ring_word(0, rd0), s_ring[1] = ring_word(1, rd0 + 1u), s_ring[2] = 0u, s_ring[3] = 0u;
            }
            if (lane == 0) s_prog[threadIdx.x / kWave] = 0u;
#ifdef SPLAHIP_COB_TIMING
            unsigned long long t_start = 0, t_zero = 0, t_loop = 0;
            if (threadIdx.x == 0) t_start = __builtin_amdgcn_s_memrealtime();
#endif
            {
                const uint32_t id  = cob_identity<T, ACC>();
                uint4*         s4  = reinterpret_cast<uint4*>(s_raw);
                const uint4    z   = kWide && !narrow ? make_uint4(0u, id, 0u, id) : make_uint4(id, id, id, id);
                const uint32_t n16 = (R + kCobSpare) * (narrow ? 4u : (uint32_t) sizeof(word)) / 16;
                for (uint32_t i = threadIdx.x; i < n16; i += TPB) s4[i] = z;
            }
            __syncthreads();
            const bool     joined   = __builtin_amdgcn_readfirstlane(s_ctl[0]) != 0u;
            const uint32_t next_job = __builtin_amdgcn_readfirstlane(s_ctl[1]);
#ifdef SPLAHIP_COB_TIMING
            if (threadIdx.x == 0) t_zero = __builtin_amdgcn_s_memrealtime();
#endif
            const uint32_t pad_chunk = a.pad_chunk + (nat ? 1u : 0u);
            const uint32_t wv        = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
            // chunk at position `pos` (0 .. 7) of the round that starts at `base`; it goes on in a VECTOR register: the chunk's base column then arrives
            // through the vector-memory queue like the two streams (a scalar load would have to be waited for together with the LDS atomics)
            auto chunk_at = [&](uint32_t base, uint32_t pos) -> uint32_t {
                const uint32_t c = base + wv + 16u * pos;
                const uint32_t k = base != kNoRound && c < end_chunk ? c : pad_chunk;
                uint32_t       kv;
                asm volatile("v_mov_b32 %0, %1" : "=v"(kv) : "s"(k));
                return kv;
            };
            // the base of round `body` from the ring (written by wave 0 a whole turn earlier: the loop below; a wave that runs ahead of wave 0 waits here)
            auto ring_read = [&](uint32_t body) -> uint32_t {
                const uint32_t want = 0x80u | (body & 0x7Fu);
                uint32_t       x;
                for (uint32_t spins = 0;; ++spins) {
                    x = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile uint32_t*>(&s_ring[body & 3u]));
                    if ((x >> 24) == want) break;
                    if (spins > (1u << 24)) __builtin_trap();// (seconds: wave 0 is lost — fail the launch rather than hang the device)
                    __builtin_amdgcn_s_sleep(1);
                }
                return x & 0x00FFFFFFu;
            };
            uint32_t base_cur = ring_read(0);
            [[maybe_unused]] uint32_t walked = 0;
            u4       E[4], X[4];
            cob_u2   S[4];
            uint32_t D[4];
            uint32_t B[4];
            T        G[2][4];
            u4       W[2];
            uint32_t M[2];
            auto load_set = [&](int qs, uint32_t k) {
                S[qs] = __builtin_nontemporal_load(sl2 + (size_t) k * (kCobChunk / 4));
                D[qs] = __builtin_nontemporal_load(dl1 + (size_t) k * (kCobChunk / 4));
                if constexpr (MODE != 2) X[qs] = __builtin_nontemporal_load(ax4 + (size_t) k * (kCobChunk / 4));
                B[qs] = hb[k];
            };
            auto decode_set = [&](int qs) {// (a padding / bridging slot reads the chunk's base column: column offset 0)
                cob_u4 e = cob_decode(S[qs], D[qs]);
                e.x = (e.x >> 16) >= R ? e.x & 0xFFFF0000u : e.x, e.y = (e.y >> 16) >= R ? e.y & 0xFFFF0000u : e.y;
                e.z = (e.z >> 16) >= R ? e.z & 0xFFFF0000u : e.z, e.w = (e.w >> 16) >= R ? e.w & 0xFFFF0000u : e.w;
                E[qs] = e;
            };
            auto gather_set = [&](int qs, int gq) {
                if constexpr (MODE == 0) decode_set(qs);
                if constexpr (MODE == 0) {
                    G[gq][0] = v[B[qs] + (E[qs].x & cmask)];
                    G[gq][1] = v[B[qs] + (E[qs].y & cmask)];
                    G[gq][2] = v[B[qs] + (E[qs].z & cmask)];
                    G[gq][3] = v[B[qs] + (E[qs].w & cmask)];
                } else {
                    const uint32_t c0 = B[qs] + (E[qs].x & cmask), c1 = B[qs] + (E[qs].y & cmask), c2 = B[qs] + (E[qs].z & cmask), c3 = B[qs] + (E[qs].w & cmask);
                    const uint32_t m  = ((W[gq].x >> (c0 & 31)) & 1u) | (((W[gq].y >> (c1 & 31)) & 1u) << 1) | (((W[gq].z >> (c2 & 31)) & 1u) << 2) | (((W[gq].w >> (c3 & 31)) & 1u) << 3);
                    M[gq]    = m;
                    G[gq][0] = v[(m & 1u) ? c0 : B[qs]];
                    G[gq][1] = v[(m & 2u) ? c1 : B[qs]];
                    G[gq][2] = v[(m & 4u) ? c2 : B[qs]];
                    G[gq][3] = v[(m & 8u) ? c3 : B[qs]];
                }
            };
            auto bits_set = [&](int qs, int wq) {
                decode_set(qs);
                W[wq].x = a.vbits[(B[qs] + (E[qs].x & cmask)) >> 5];
                W[wq].y = a.vbits[(B[qs] + (E[qs].y & cmask)) >> 5];
                W[wq].z = a.vbits[(B[qs] + (E[qs].z & cmask)) >> 5];
                W[wq].w = a.vbits[(B[qs] + (E[qs].w & cmask)) >> 5];
            };
            auto add_set = [&](int qs, int gq) {
                const u4 e = E[qs];
                u4       x;
                if constexpr (MODE == 2) x = {a.x_uniform, a.x_uniform, a.x_uniform, a.x_uniform};
                else
                    x = X[qs];
                const T g0 = G[gq][0], g1 = G[gq][1], g2 = G[gq][2], g3 = G[gq][3];
                if constexpr (MODE != 0) {
                    const uint32_t m = M[gq];
                    if (m & 1u) cob_accumulate<T, ACC>(s_acc + (e.x >> a.col_bits), mult(from_bits<T>(x.x), g0));
                    if (m & 2u) cob_accumulate<T, ACC>(s_acc + (e.y >> a.col_bits), mult(from_bits<T>(x.y), g1));
                    if (m & 4u) cob_accumulate<T, ACC>(s_acc + (e.z >> a.col_bits), mult(from_bits<T>(x.z), g2));
                    if (m & 8u) cob_accumulate<T, ACC>(s_acc + (e.w >> a.col_bits), mult(from_bits<T>(x.w), g3));
                } else if constexpr (VXM) {
                    if constexpr (kWide) {
                        if (narrow) {
                            cob_accumulate4<T, COB_FADD>(s_acc32, e.x >> a.col_bits, e.y >> a.col_bits, e.z >> a.col_bits, e.w >> a.col_bits, mult(g0, from_bits<T>(x.x)),
                                                         mult(g1, from_bits<T>(x.y)), mult(g2, from_bits<T>(x.z)), mult(g3, from_bits<T>(x.w)));
                            return;
                        }
                    }
                    cob_accumulate4<T, ACC>(s_acc, e.x >> a.col_bits, e.y >> a.col_bits, e.z >> a.col_bits, e.w >> a.col_bits, mult(g0, from_bits<T>(x.x)),
                                            mult(g1, from_bits<T>(x.y)), mult(g2, from_bits<T>(x.z)), mult(g3, from_bits<T>(x.w)));
                } else {
                    if constexpr (kWide) {
                        if (narrow) {
                            cob_accumulate4<T, COB_FADD>(s_acc32, e.x >> a.col_bits, e.y >> a.col_bits, e.z >> a.col_bits, e.w >> a.col_bits, mult(from_bits<T>(x.x), g0),
                                                         mult(from_bits<T>(x.y), g1), mult(from_bits<T>(x.z), g2), mult(from_bits<T>(x.w), g3));
                            return;
                        }
                    }
                    cob_accumulate4<T, ACC>(s_acc, e.x >> a.col_bits, e.y >> a.col_bits, e.z >> a.col_bits, e.w >> a.col_bits, mult(from_bits<T>(x.x), g0),
                                            mult(from_bits<T>(x.y), g1), mult(from_bits<T>(x.z), g2), mult(from_bits<T>(x.w), g3));
                }
            };
            if (joined && base_cur != kNoRound) {
                // The four-deep software pipeline of cob_mxv_kernel (streams three steps ahead, gathers one step ahead, LDS atomics of the current step), eight
                // steps per turn = one round.  Step k of turn b requests the streams of position k + 3: positions 3 .. 7 of round b, then 0 .. 2 of round b + 1,
                // whose base is read from the ring before step 5.  Wave 0 keeps the ring filled: at the top of turn b it claims round b + 2 (one returning
                // atomic per 256 KB of stream) and reads the result before step 4 of the SAME turn — 28 vector-memory instructions later, nothing waits for it
                // (the empty asm pins the read there: the compiler would otherwise read — and wait for — the claim right behind the atomic) — then writes
                // ring[(b + 2) & 3] once every wave has left turn b - 3 (s_prog: the slot's previous reader).  A round behind the range's last one ends the
                // loop: every real chunk has been added by then, three requests of the padding chunk are left in flight.
                load_set(0, chunk_at(base_cur, 0));
                load_set(1, chunk_at(base_cur, 1));
                load_set(2, chunk_at(base_cur, 2));
                if constexpr (MODE != 0) {
                    bits_set(0, 0);
                    bits_set(1, 1);
                }
                gather_set(0, 0);
                for (uint32_t body = 0;; ++body) {
                    uint32_t pend = 0;
                    if (wv == 0u) {
                        if (lane == 0) pend = __hip_atomic_fetch_add(cur_pv, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (lane == 0) *reinterpret_cast<volatile uint32_t*>(&s_prog[wv]) = body;
                    uint32_t base_next = kNoRound;
#ifdef SPLAHIP_COB_TIMING
                    if (wv == 0u) walked += min(kRound, end_chunk - base_cur);
#endif
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (k == 4) {
                            if (wv == 0u) {
                                asm volatile("" : "+v"(pend));
                                const uint32_t rd = __builtin_amdgcn_readfirstlane(pend);
                                if (body >= 2u) {// the slot held round body - 2, read by every wave before step 5 of turn body - 3: all of them have left that turn?
                                    for (uint32_t spins = 0;; ++spins) {
                                        const uint32_t pw = lane < TPB / kWave ? *reinterpret_cast<volatile uint32_t*>(&s_prog[lane]) : body;
                                        if (__ballot(pw + 2u < body) == 0ull) break;
                                        if (spins > (1u << 24)) __builtin_trap();// (seconds: a wave of this workgroup is lost — fail the launch rather than hang the device)
                                        __builtin_amdgcn_s_sleep(1);
                                    }
                                }
                                if (lane == 0) *reinterpret_cast<volatile uint32_t*>(&s_ring[(body + 2u) & 3u]) = ring_word(body + 2u, rd);
                            }
                        }
                        if (k == 5) base_next = ring_read(body + 1u);
                        load_set((k + 3) & 3, k < 5 ? chunk_at(base_cur, (uint32_t) k + 3u) : chunk_at(base_next, (uint32_t) k - 5u));
                        gather_set((k + 1) & 3, (k + 1) & 1);
                        if constexpr (MODE != 0) bits_set((k + 2) & 3, (k + 2) & 1);
                        add_set(k & 3, k & 1);
                    }
                    base_cur = base_next;
                    if (base_cur == kNoRound) break;
                }
            }
            __syncthreads();
#ifdef SPLAHIP_COB_TIMING
            if (threadIdx.x == 0) t_loop = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0) s_ctl[2] = walked;
#endif
            if (joined) {
                const CobRange rg = a.ranges[q];
                if constexpr (kWide) {
                    if (narrow)
                        cob_dyn_range_end<T, COB_FADD, VXM, TPB>(s_raw, s_ctl, q, !nat, rg, a.slot16, a.hub_row, a.acc, d.state, d.fin_flag, d.seq, a.touched, a.n_rows, R, a.acc_stride,
                                                                 a.accumulate, mask, r, add, op_select, init);
                    else
                        cob_dyn_range_end<T, ACC, VXM, TPB>(s_raw, s_ctl, q, !nat, rg, a.slot16, a.hub_row, a.acc, d.state, d.fin_flag, d.seq, a.touched, a.n_rows, R, a.acc_stride,
                                                            a.accumulate, mask, r, add, op_select, init);
                } else
                    cob_dyn_range_end<T, ACC, VXM, TPB>(s_raw, s_ctl, q, !nat, rg, a.slot16, a.hub_row, a.acc, d.state, d.fin_flag, d.seq, a.touched, a.n_rows, R, a.acc_stride,
                                                        a.accumulate, mask, r, add, op_select, init);
            }
            __syncthreads();// (the LDS slice and s_ctl are reused by the next job)
#ifdef SPLAHIP_COB_TIMING
            if (threadIdx.x == 0) {
                const unsigned int at = atomicAdd(&g_cob_tn, 1u);
                if (at < 8192) {
                    uint32_t xcc;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                    g_cob_t[0][at] = t_start, g_cob_t[1][at] = t_zero, g_cob_t[2][at] = t_loop, g_cob_t[3][at] = __builtin_amdgcn_s_memrealtime();
                    g_cob_t[4][at] = xcc, g_cob_t[5][at] = blockIdx.x | ((unsigned long long) (job < d.n_jobs ? 0u : 1u) << 32) | ((unsigned long long) (joined ? 1u : 0u) << 33);
                    g_cob_t[6][at] = s_ctl[2], g_cob_t[7][at] = ((unsigned long long) q << 32) | 0u;
                }
                s_ctl[2] = 0;
            }
            __syncthreads();
#endif
            job = next_job;
        }
        // leave: the last workgroup re-arms the launch's words for the next call (every other workgroup's atomics on them have returned: each was a
        // returning atomic of its thread 0 or of its waves' lane 0, consumed before the workgroup came here)
        __syncthreads();
        if (threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_ctl[0] = __hip_atomic_fetch_add(d.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
        }
        __syncthreads();
        if (s_ctl[0]) {
            for (uint32_t i = threadIdx.x; i < d.n_ranges; i += TPB) {
                __hip_atomic_store(d.cursor + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(d.state + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (threadIdx.x == 0) {
                __hip_atomic_store(d.ctl, gridDim.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(d.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

}// namespace splahip
