// mxm.hip — the two SpGEMM operators of the hot path on gfx950.
//
// splahip_mxmT_masked : R = (A * B^T) .mask
//   semantics: reference Algo_mxmT_masked_cpu<T>::execute, src/cpu/cpu_mxmT_masked.hpp:87-125 (the parity oracle): for every
//   STORED mask entry (i, j, m) in row order: r = init; if select(m): two-pointer merge of the column-sorted rows A[i,:] and
//   B[j,:], r = add(r, mult(a, b)) on equal columns (ascending column); the entry is emitted iff r != init.
//   entry point replaced: Algo_mxmT_masked_cl<T>::execute, src/opencl/cl_mxmT_masked.hpp:60-127 (one lane per mask entry,
//   which copies the mask structure and stores init explicitly — a different output structure; parity is with the CPU).
//   Here: a lane per mask entry does the merge for light entries; heavy entries (long rows) are queued and done by one
//   wavefront each (lanes take contiguous pieces of the shorter row, binary-search the longer row, in-order shuffle
//   reduction); then keep-flags -> scan -> compact into CSR.  Work = sum over selected mask entries of the two row lengths.
//
// splahip_mxm : R = A * B  (unmasked)
//   semantics: reference Algo_mxm_cpu<T>::execute, src/cpu/cpu_mxm.hpp:85-114: Gustavson row by row over a dense accumulator
//   pre-filled with init: R[i,j] = fold over k in the order of A[i,:] of add(., mult(a_ik, b_kj)) starting at init; entries
//   equal to init are dropped; columns ascending.  No accelerated version exists in the reference.
//   Here: ESC — expand the products of a block of rows in (A-entry order, B-row order), STABLE radix sort by (row, column)
//   (rocPRIM), fold every run in order from init, drop == init, compact.  Row blocks bound the expanded size.  The stable
//   sort keeps every output's products in the reference's fold order, so the result is bit-exact for every op (fp32 PLUS
//   included) when runs are folded sequentially; runs longer than 64 products use an in-order tree (associative ops).
#include "block_utils.hpp"
#include "common.hpp"
#include "jit_hooks.hpp"
#include "ops.hpp"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace splahip {

    // Scoped device memory: stream-ordered, from the device's default pool (GB-sized hipMalloc / hipFree pairs per call cost more than
    // the kernels).  The two result arrays are handed to the caller (release()), who returns them with splahip_free.
    static thread_local hipStream_t tl_stream = nullptr;// the stream of the running mxm / mxmT call
    struct DevBuf {
        void*       p      = nullptr;
        hipStream_t s      = nullptr;
        bool        pooled = false;
        ~DevBuf() { drop(); }
        void drop() {
            if (p && !cached && !arena) {
                if (handoff) (void) handoff_release(p);
                else
                    (void) (pooled ? hipFreeAsync(p, s) : hipFree(p));
            }
            p      = nullptr;
            cached = handoff = arena = false;
        }
        hipError_t alloc(size_t bytes) {
            drop();
            s      = tl_stream;
            pooled = true;
            if (scratch_active()) {// the calling thread's scratch arena (common.hpp): given back when the call's ScratchScope ends
                arena = true;
                return scratch_take(&p, bytes ? bytes : 4);
            }
            return device_malloc_async(&p, bytes ? bytes : 4, s);
        }
        bool arena = false;
        hipError_t alloc_plain(size_t bytes) {// a result array: handed to the caller with release(), returned with splahip_free
            drop();
            pooled = false;
            handoff = true;
            return handoff_alloc(&p, bytes);
        }
        bool handoff = false;
        // GB-sized workspaces of mxm: kept between calls in a per-thread, grow-only cache (slot = which buffer) — the stream-ordered
        // pool's behaviour for buffers of this size is erratic (the same call: 0.2 ms or 2.6 s).  Released by splahip_mxm_trim().
        hipError_t alloc_cached(unsigned slot, size_t bytes) {
            drop();
            SlotCache& sc = cache();
            std::lock_guard<std::mutex> g(sc.m);
            auto& c = sc.slots;
            if (c.size() <= slot) c.resize(slot + 1);
            if (c[slot].bytes < bytes) {
                if (c[slot].p) (void) hipFree(c[slot].p);
                c[slot] = {nullptr, 0};
                register_trim(&DevBuf::trim);// (idempotent) the cache goes back to the driver when any allocation of the library runs out of memory
                hipError_t e = hipMalloc(&c[slot].p, bytes);
                if (e == hipErrorOutOfMemory) {// (the release-everything retry outside this cache's lock: it trims every thread's cache, this one included)
                    (void) hipGetLastError();
                    sc.m.unlock();
                    release_idle_memory();
                    sc.m.lock();
                    if (c.size() <= slot) c.resize(slot + 1);
                    e = hipMalloc(&c[slot].p, bytes);
                }
                if (e != hipSuccess) return e;
                c[slot].bytes = bytes;
            }
            p      = c[slot].p;
            cached = true;
            return hipSuccess;
        }
        struct Slot {
            void*  p;
            size_t bytes;
        };
        // (one per host thread, listed process-wide like the caches of runtime.hip: a trim frees the workspaces of every thread that is not inside a call, and
        // deletes the caches of threads that have exited — the thread_local holder's destructor only marks its cache an orphan, it makes no HIP call)
        struct SlotCache {
            std::mutex        m;
            std::vector<Slot> slots;
            int               busy   = 0;// > 0 while an mxm / mxmT call of the owning thread is using the cached workspaces: a trim must not free them
            bool              orphan = false;
        };
        struct SlotRegistry {
            std::mutex              m;
            std::vector<SlotCache*> all;
        };
        static SlotRegistry& registry() {
            static SlotRegistry* r = new SlotRegistry;
            return *r;
        }
        struct SlotHolder {
            SlotCache* c;
            SlotHolder() : c(new SlotCache) {
                auto& r = registry();
                std::lock_guard<std::mutex> g(r.m);
                r.all.push_back(c);
            }
            ~SlotHolder() {
                std::lock_guard<std::mutex> g(c->m);
                c->orphan = true;
            }
        };
        static SlotCache& cache() {
            static thread_local SlotHolder h;
            return *h.c;
        }
        static void busy_add(int d) {
            SlotCache& c = cache();
            std::lock_guard<std::mutex> g(c.m);
            c.busy += d;
        }
        static void trim() {
            auto& r = registry();
            std::lock_guard<std::mutex> rg(r.m);
            for (size_t k = 0; k < r.all.size();) {
                SlotCache* c    = r.all[k];
                bool       drop = false;
                {
                    std::lock_guard<std::mutex> g(c->m);
                    if (c->busy <= 0) {
                        for (auto& sl : c->slots)
                            if (sl.p) (void) hipFree(sl.p);
                        c->slots.clear();
                        drop = c->orphan;
                    }
                }
                if (drop) {
                    delete c;
                    r.all.erase(r.all.begin() + (long) k);
                } else
                    ++k;
            }
        }
        bool cached = false;
        void* release() {
            void* q = p;
            p       = nullptr;
            cached = handoff = arena = false;
            return q;
        }
        template<typename U>
        U* as() { return static_cast<U*>(p); }
    };

    // SPLAHIP_MXM_TRACE=1: host-side wall time of the phases of one mxm call on stderr (diagnostic)
    struct PhaseTrace {
        bool                                  on = knob_env("SPLAHIP_MXM_TRACE") != nullptr;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        void mark(const char* what) {
            if (!on) return;
            (void) hipDeviceSynchronize();
            const auto t1 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[mxm] %-28s %9.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        }
    };

    static uint bit_width(unsigned long long x) {
        uint b = 0;
        while (x) {
            ++b;
            x >>= 1;
        }
        return b ? b : 1;
    }

    template<typename T>
    __device__ __forceinline__ bool differs(T a, T b) { return a != b; }// the reference's `r != I` (float compare for F)

    __device__ __forceinline__ uint row_of_entry(const uint* __restrict__ Ap, uint n_rows, uint k) {
        uint lo = 0, hi = n_rows;// last row with Ap[row] <= k
        while (hi - lo > 1) {
            const uint mid = lo + ((hi - lo) >> 1);
            if (Ap[mid] <= k) lo = mid;
            else
                hi = mid;
        }
        return lo;
    }

    // generic flag compaction helpers: count per 1024-chunk, write with ranks
    __global__ __launch_bounds__(256) void flags_count_kernel(const uint* __restrict__ flag, uint n, uint* __restrict__ chunk) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            cnt   = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u)
            if (first + u < n) cnt += flag[first + u] ? 1u : 0u;
        uint total;
        (void) block_exclusive_sum<256>(cnt, s_tmp, total);
        if (threadIdx.x == 0) chunk[blockIdx.x] = total;
    }
    // rank[i] = number of set flags before i (for every i, plus rank[n] = total)
    __global__ __launch_bounds__(256) void flags_rank_kernel(const uint* __restrict__ flag, uint n, const uint* __restrict__ chunk, uint n_chunks,
                                                             uint* __restrict__ rank) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            f[4];
        uint            cnt = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            f[u] = (first + u < n && flag[first + u]) ? 1u : 0u;
            cnt += f[u];
        }
        uint total;
        uint pos = chunk[blockIdx.x] + block_exclusive_sum<256>(cnt, s_tmp, total);
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            if (first + u < n) rank[first + u] = pos;
            pos += f[u];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) rank[n] = chunk[n_chunks];
    }
    static hipError_t flags_exclusive_rank(const uint* flag, uint n, uint* rank, hipStream_t s) {
        if (n == 0) return hipMemsetAsync(rank, 0, sizeof(uint), s);
        const uint n_chunks = (n + 1023u) / 1024u;
        DevBuf     chunk;
        hipError_t e = chunk.alloc(sizeof(uint) * (size_t) (n_chunks + 1));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(flags_count_kernel, dim3(n_chunks), dim3(256), 0, s, flag, n, chunk.as<uint>());
        hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, chunk.as<uint>(), n_chunks);
        hipLaunchKernelGGL(flags_rank_kernel, dim3(n_chunks), dim3(256), 0, s, flag, n, chunk.as<uint>(), n_chunks, rank);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        return e;
    }

    // =============================================================================================================
    // mxmT_masked
    // =============================================================================================================
    constexpr uint kHeavy = 192;// len(A row) + len(B row) above which a mask entry is handled by a wavefront
    // row form of the heavy entries (mxmT_row_kernel)
    constexpr uint kRowCap     = 4095;// longest A row a wavefront stages in its LDS (4096 words with the padding)
    constexpr uint kRowSmall   = 1023;// rows up to here use the small LDS layout (8 KB per wavefront, 20 wavefronts per CU)
    constexpr uint kRowPiece   = 512; // cost units (about a wave step each) per work item
    constexpr uint kRowPending = 2;   // keep[e]: the entry waits for the row kernel
    constexpr uint kRowHits    = 128; // matches recorded per wavefront before their value words are fetched
    constexpr uint kRowQueue   = 128; // filter candidates waiting for the exact search
    __host__ __device__ constexpr uint row_lds_words(uint cap) { return (cap + 1) + (cap + 1) / 2 + 2 * kRowQueue + 2 * kRowHits; }// columns, filter bits, queue, matches
    __host__ __device__ __forceinline__ bool row_streams(uint len_a, uint len_b) { return len_b <= 16u * len_a; }
    __host__ __device__ __forceinline__ uint row_entry_cost(uint len_a, uint len_b) {
        return 1u + (row_streams(len_a, len_b) ? (len_b + 63u) / 64u : (len_a + 3u) / 4u);
    }
    // block merge of two rows of comparable length (mxmT_merge_kernel)
    constexpr uint kMergeTile = 512;// words of the longer row staged per step (256 / 512 / 1024: merge kernel 8.6 / 8.0 / 10.5 ms on the R-MAT 20 triangle form)
    constexpr uint kMergeHits = 128;// matches recorded per wavefront before their value words are fetched
    __host__ __device__ __forceinline__ bool merge_streams(uint len_a, uint len_b, uint factor) {
        const uint mn = len_a < len_b ? len_a : len_b, mx = len_a < len_b ? len_b : len_a;
        return mx <= factor * mn;
    }
    __global__ void len_sum_kernel(const uint* __restrict__ x, uint n, uint* __restrict__ chunk);
    __global__ void len_scan_kernel(const uint* __restrict__ x, uint n, const uint* __restrict__ chunk, uint n_chunks, uint* __restrict__ out);

    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(256) void mxmT_entry_kernel(uint n_rows, uint n_entries,
                                                             const uint* __restrict__ Mp, const uint* __restrict__ Mj, const T* __restrict__ Mx,
                                                             const uint* __restrict__ Ap, const uint* __restrict__ Aj, const T* __restrict__ Ax,
                                                             const uint* __restrict__ Bp, const uint* __restrict__ Bj, const T* __restrict__ Bx,
                                                             T* __restrict__ val, uint* __restrict__ keep,
                                                             uint* __restrict__ heavy_count, uint* __restrict__ heavy_list,
                                                             Mult mult, Add add, int op_select, T init, uint row_cap, uint* __restrict__ cost,
                                                             uint merge_on, uint merge_factor, uint heavy_from, uint* __restrict__ owner_key, uint* __restrict__ erow) {
        const uint e = blockIdx.x * 256 + threadIdx.x;// (lanes behind the last entry stay for the wave-wide list appends below)
        T    r    = init;
        bool done = true;
        uint c    = 0, which = 0, row_i = 0;// which: 1 = search kernel's list (owner form: the owner list), 2 = block merge's list
        if (e < n_entries && eval_sel(op_select, Mx[e])) {
            const uint i  = row_of_entry(Mp, n_rows, e);
            const uint j  = Mj[e];
            uint       ka = Ap[i], ea = Ap[i + 1];
            uint       kb = Bp[j], eb = Bp[j + 1];
            if ((ea - ka) + (eb - kb) > heavy_from && ea > ka && eb > kb) {
                if (owner_key != nullptr) {// owner form (mxmT_owner_kernel): the entry belongs to the LONGER of its two rows — A[i, :] (key i) or B[j, :] (key n_rows + j)
                    which  = 1;
                    row_i  = (ea - ka) >= (eb - kb) ? i : n_rows + j;
                    erow[e] = i;
                } else if (ea - ka <= row_cap) {// row form (mxmT_row_kernel): A[i, :] staged in LDS once for a run of the row's entries
                    keep[e] = kRowPending;
                    c       = row_entry_cost(ea - ka, eb - kb);
                } else if (merge_on && merge_streams(ea - ka, eb - kb, merge_factor)) {// rows of comparable length: the block merge (mxmT_merge_kernel)
                    which = 2;
                } else {
                    which = 1;
                }
                if (owner_key == nullptr) row_i = i;
                done = false;
            } else {
                while (ka < ea && kb < eb) {
                    const uint ca = Aj[ka], cb = Bj[kb];
                    if (ca == cb) {
                        r = add(r, mult(Ax[ka], Bx[kb]));
                        ++ka;
                        ++kb;
                    } else if (ca < cb) {
                        ++ka;
                    } else {
                        ++kb;
                    }
                }
            }
        }
        // the two lists: ONE 64-bit atomic per WORKGROUP (low word: the search kernel's count, high word: the block merge's) — same-address
        // atomics retire at ~5 ns each, and two per wavefront were 2.5 ms of this kernel; slots by rank among the workgroup's appenders
        {
            __shared__ uint               s_cnt[2][256 / kWave];
            __shared__ unsigned long long s_base;
            const unsigned long long      m1 = __ballot(which == 1), m2 = __ballot(which == 2);
            if (lane_id() == 0) {
                s_cnt[0][wave_id()] = (uint) __popcll(m1);
                s_cnt[1][wave_id()] = (uint) __popcll(m2);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned long long add = 0;
                for (uint w = 0; w < 256 / kWave; ++w) add += (unsigned long long) s_cnt[0][w] | ((unsigned long long) s_cnt[1][w] << 32);
                s_base = add ? atomicAdd(reinterpret_cast<unsigned long long*>(heavy_count), add) : 0ull;
            }
            __syncthreads();
            if (which != 0) {
                const uint               l = which - 1u;
                const unsigned long long m = l ? m2 : m1;
                uint                     rank = (uint) (s_base >> (32u * l)) + __builtin_amdgcn_mbcnt_hi((uint) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint) m, 0u));
                for (uint w = 0; w < wave_id(); ++w) rank += s_cnt[l][w];
                // ONE buffer of n_entries pairs: the search kernel's list grows from its front, the block merge's from its back
                const uint slot          = l == 0 ? rank : n_entries - 1u - rank;
                heavy_list[2 * slot]     = e;
                heavy_list[2 * slot + 1] = row_i;// the row travels with the entry: the heavy kernels need not search Mp again (20 dependent loads)
            }
        }
        if (e >= n_entries) return;
        if (done) {
            val[e]  = r;
            keep[e] = differs(r, init) ? 1u : 0u;
        }
        if (cost) cost[e] = c;
    }

    // ---- heavy entries, ROW FORM.  All heavy entries (i, j) of a mask row share A[i, :]; the per-entry kernel below searches every
    // column of the shorter row in the longer one with DEPENDENT global loads (R-MAT 20 triangle form: 7.1 G L1-to-L2 requests and 61 GB
    // of fabric traffic per call — profiles/r02_pmc_tc_mxmT.json: the L2's request rate is what bounds it).  Here a WAVEFRONT takes a
    // work item = (row i, a run of its heavy entries worth ~kRowPiece wave steps) and, once per item, builds in its own LDS
    //   * the columns of A[i, :] (sorted, padded to a power of two with 0xFFFFFFFF), and
    //   * a one-hash FILTER of them, 16 bits per column slot (a column outside the row passes it with probability <= 1/16).
    // Then, per entry (the 64 entries of a batch fetch j and B's row bounds lane-parallel — one dependent chain per batch, not per
    // entry):
    //   * len(B[j, :]) <= 16 len(A[i, :]):  B[j, :] STREAMS by, 4 x 64 coalesced column words per step; a word costs one LDS read (its
    //     filter bit).  The few that pass are queued (ballot compaction) and searched exactly, 64 at a time, by a branch-free binary
    //     search over the LDS columns — so the 12-level search runs on full wavefronts of candidates, not on every streamed word;
    //   * else: the lanes take the LDS-resident columns of A[i, :] and search each in B[j, :] (global).
    // Matches are only RECORDED (position in A[i, :], position in B); their value words are fetched kRowHits at a time with all loads of
    // a batch in flight.  No workgroup barrier anywhere (a workgroup's wavefronts are independent); items are cut by a prefix sum of
    // per-entry costs so a hub row's 20,000 entries spread over many wavefronts.  Needs a commutative, associative op_add (the queue
    // reorders the products); anything else stays with the per-entry kernel.
    __device__ __forceinline__ uint pow2_above(uint x) {// smallest power of two > x
        return 2u << (31u - __builtin_clz(x | 1u));
    }
    template<typename T, class Mult, class Add, uint CAP, uint WAVES>
    __global__ __launch_bounds__(WAVES* kWave) void mxmT_row_kernel(uint n_rows, uint item0, uint n_items, const uint* __restrict__ piece_first,
                                                                    const uint* __restrict__ item_row,
                                                                    const uint* __restrict__ cost_before, const uint* __restrict__ Mp, const uint* __restrict__ Mj,
                                                                    const uint* __restrict__ Ap, const uint* __restrict__ Aj, const T* __restrict__ Ax,
                                                                    const uint* __restrict__ Bp, const uint* __restrict__ Bj, const T* __restrict__ Bx,
                                                                    T* __restrict__ val, uint* __restrict__ keep, Mult mult, Add add_raw, T init) {
        const Reassoc<Add> add(add_raw);// (ops.hpp) partial folds start at products
        extern __shared__ uint s_row[];
        const uint      wv = threadIdx.x / kWave, lane = lane_id();
        if (blockIdx.x * WAVES + wv >= n_items) return;// (whole wavefronts leave: nothing below synchronises across wavefronts)
        const uint  item  = item0 + blockIdx.x * WAVES + wv;
        uint* const cols  = s_row + wv * row_lds_words(CAP);
        uint* const bits  = cols + (CAP + 1);
        uint* const queue = bits + (CAP + 1) / 2;
        uint* const hits  = queue + 2 * kRowQueue;
        const uint  i     = item_row[item];
        const uint m0 = Mp[i], m1 = Mp[i + 1];
        const uint c0 = cost_before[m0] + (item - piece_first[i]) * kRowPiece, c1 = min(cost_before[m1], c0 + kRowPiece);
        auto       first_at = [&](uint c) {// first entry of the row whose cost interval starts at or after c
            uint lo = m0, hi = m1;
            while (lo < hi) {
                const uint mid = lo + ((hi - lo) >> 1);
                if (cost_before[mid] < c) lo = mid + 1;
                else
                    hi = mid;
            }
            return lo;
        };
        const uint e0 = first_at(c0), e1 = c1 == cost_before[m1] ? m1 : first_at(c1);
        const uint sa = Ap[i], len_a = Ap[i + 1] - sa;
        const uint pad = pow2_above(len_a);// > len_a: the last word is padding, so a lower bound always exists
        const uint filter_mask = 16u * pad - 1u;// a filter of 16 pad bits, indexed by the column's low bits (no multiply in the stream loop)
        auto       filter_bit  = [&](uint col) -> uint { return col & filter_mask; };
        for (uint t = lane; t < pad / 2; t += kWave) bits[t] = 0;
        for (uint t = lane; t < pad; t += kWave) cols[t] = t < len_a ? Aj[sa + t] : 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint t = lane; t < len_a; t += kWave) {
            const uint h = filter_bit(cols[t]);
            atomicOr(&bits[h >> 5], 1u << (h & 31u));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        auto find = [&](uint col) -> uint {// lower bound of col among cols[0, pad)
            uint pos = 0;
            for (uint step = pad >> 1; step != 0; step >>= 1)
                if (cols[pos + step - 1] < col) pos += step;
            return pos;
        };
        auto rank_in = [&](unsigned long long m) -> uint { return __builtin_amdgcn_mbcnt_hi((uint) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint) m, 0u)); };

        for (uint eb0 = e0; eb0 < e1; eb0 += kWave) {
            // this batch's entries, a lane each: pending?  j, B's row bounds
            uint my_sb = 0, my_eb = 0;
            bool my_pending = false;
            if (eb0 + lane < e1 && keep[eb0 + lane] == kRowPending) {
                const uint j = Mj[eb0 + lane];
                my_sb        = Bp[j];
                my_eb        = Bp[j + 1];
                my_pending   = true;
            }
            unsigned long long todo = __ballot(my_pending);
            while (todo != 0ull) {
                const uint l = (uint) __ffsll((long long) todo) - 1u;
                todo &= todo - 1ull;
                const uint e  = eb0 + l;
                const uint sb = __builtin_amdgcn_readlane(my_sb, l), eb = __builtin_amdgcn_readlane(my_eb, l);
                T          acc = init;
                bool       has = false;
                uint       n_hits = 0, n_queue = 0;// (uniform)
                auto flush = [&]() {// fetch the value words of the recorded matches, fold their products
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    for (uint h0 = 0; h0 < n_hits; h0 += kWave) {
                        const uint h = h0 + lane;
                        if (h < n_hits) {
                            const T p = mult(Ax[sa + hits[2 * h]], Bx[hits[2 * h + 1]]);
                            acc       = has ? add(acc, p) : p;
                            has       = true;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    n_hits = 0;
                };
                auto record = [&](bool hit, uint ia, uint kb) {
                    const unsigned long long m = __ballot(hit);
                    if (m == 0ull) return;
                    const uint cnt = (uint) __popcll(m);
                    if (n_hits + cnt > kRowHits) flush();
                    if (hit) {
                        const uint slot    = n_hits + rank_in(m);
                        hits[2 * slot]     = ia;
                        hits[2 * slot + 1] = kb;
                    }
                    n_hits += cnt;
                };
                auto search_top = [&](uint count) {// exact search of the newest `count` (<= 64) queued candidates
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const uint q   = n_queue - count + lane;
                    bool       hit = false;
                    uint       pos = 0, kb = 0;
                    if (lane < count) {
                        const uint col = queue[2 * q];
                        kb             = queue[2 * q + 1];
                        pos            = find(col);
                        hit            = cols[pos] == col;
                    }
                    n_queue -= count;
                    record(hit, pos, kb);
                };
                if (row_streams(len_a, eb - sb)) {
                    constexpr uint U = 4;
                    uint           nxt[U];
                    auto           load_step = [&](uint base) {// (a position behind the row's end reads the row's last word: no branch, and masked below)
#pragma unroll
                        for (uint u = 0; u < U; ++u) nxt[u] = Bj[min(base + u * kWave + lane, eb - 1u)];
                    };
                    load_step(sb);
                    for (uint base = sb; base < eb; base += U * kWave) {
                        uint col[U], word[U];
#pragma unroll
                        for (uint u = 0; u < U; ++u) col[u] = nxt[u];
                        if (base + U * kWave < eb) load_step(base + U * kWave);// the next step's words travel while this step is filtered
#pragma unroll
                        for (uint u = 0; u < U; ++u) word[u] = bits[filter_bit(col[u]) >> 5];
#pragma unroll
                        for (uint u = 0; u < U; ++u) {
                            const uint k    = base + u * kWave + lane;
                            const bool cand = k < eb && ((word[u] >> (filter_bit(col[u]) & 31u)) & 1u) != 0;
                            const unsigned long long m = __ballot(cand);
                            if (m != 0ull) {
                                if (cand) {
                                    const uint slot     = n_queue + rank_in(m);
                                    queue[2 * slot]     = col[u];
                                    queue[2 * slot + 1] = k;
                                }
                                n_queue += (uint) __popcll(m);
                                if (n_queue >= kWave) search_top(kWave);
                            }
                        }
                    }
                    if (n_queue) search_top(n_queue);
                } else {
                    for (uint t0 = 0; t0 < len_a; t0 += kWave) {
                        const uint t   = t0 + lane;
                        bool       hit = false;
                        uint       lo  = sb;
                        if (t < len_a) {
                            const uint col = cols[t];
                            uint       hi  = eb;
                            while (lo < hi) {
                                const uint mid = lo + ((hi - lo) >> 1);
                                if (Bj[mid] < col) lo = mid + 1;
                                else
                                    hi = mid;
                            }
                            hit = lo < eb && Bj[lo] == col;
                        }
                        record(hit, t, lo);
                    }
                }
                flush();
                wave_reduce_ordered(acc, has, add);
                if (lane == 0) {
                    const T r = has ? add(init, acc) : init;
                    val[e]    = r;
                    keep[e]   = differs(r, init) ? 1u : 0u;
                }
            }
        }
    }
    // work items of a mask row in its size class (0: A row <= kRowSmall, 1: up to kRowCap): its pending cost in pieces of kRowPiece.
    // pieces[c * n_rows + i]; the prefix sum over both halves gives class 0's items first.
    __global__ __launch_bounds__(256) void mxmT_row_pieces_kernel(uint n_rows, const uint* __restrict__ Mp, const uint* __restrict__ Ap,
                                                                  const uint* __restrict__ cost_before, uint* __restrict__ pieces) {
        const uint i = blockIdx.x * 256 + threadIdx.x;
        if (i >= n_rows) return;
        const uint n = (cost_before[Mp[i + 1]] - cost_before[Mp[i]] + kRowPiece - 1) / kRowPiece;
        const uint c = (Ap[i + 1] - Ap[i]) <= kRowSmall ? 0u : 1u;
        pieces[c * n_rows + i]        = n;
        pieces[(1u - c) * n_rows + i] = 0;
    }

    __global__ __launch_bounds__(256) void mxmT_row_items_kernel(uint n_rows, const uint* __restrict__ piece_first, uint* __restrict__ item_row) {
        const uint v = blockIdx.x * 256 + threadIdx.x;// class * n_rows + row
        if (v >= 2 * n_rows) return;
        const uint i = v < n_rows ? v : v - n_rows;
        for (uint t = piece_first[v]; t < piece_first[v + 1]; ++t) item_row[t] = i;
    }

    // ---- heavy entries whose two rows are of comparable length (longer <= 128 x shorter): BLOCK MERGE, one wavefront per entry.
    // The per-entry search kernel below is bound by the L2's request rate: every probe of every binary search is its own request
    // (R-MAT 20 triangle form: 4.35 G requests per call even with galloping).  Here both rows are read ONCE, coalesced: 64 words of the
    // shorter row sit in the lanes, 512 words of the longer row in the wavefront's LDS tile; every lane whose word is <= the tile's last
    // word finds its lower bound in the tile (8 LDS reads, branch-free) — those lanes are a prefix, the shorter row being sorted; when
    // the whole block is placed the next 64 words of the shorter row are loaded, otherwise the next tile of the longer row (and the
    // block is shifted by the number of placed words).  A step therefore retires 64 words of one row or 256 of the other; requests are
    // the rows' lines, once.  Matches are recorded (position in each row) and their value words fetched kMergeHits at a time; products
    // fold lane-privately for a commutative op_add, in ascending column order otherwise (an in-order wave reduction per step).  (64 entries
    // per wavefront with their indices and row bounds fetched lane-parallel was tried: 10.4 against 9.7 ms — one entry per wavefront hides
    // the metadata chain behind other wavefronts better than it is amortised.)
    template<class Add>
    __device__ __forceinline__ int merge_op_id(const Add& add) {
        return binop_id(add);
    }
    // may an entry's value be computed from the number of matches when all products are equal?  PLUS / BXOR count (integer types: n x p wraps
    // like n additions), MIN / MAX / BOR / BAND / FIRST / SECOND are idempotent (p (+) p = p bit for bit); the logical ops normalise their
    // operands (LOR(5, 5) = 1) and a float sum of n equal terms rounds differently from n x p: those keep the match list
    inline bool merge_counts(int op_add, int type) {
        switch (op_add) {
            case SPLAHIP_PLUS:
            case SPLAHIP_BXOR: return type != SPLAHIP_FLOAT;
            case SPLAHIP_BOR:
            case SPLAHIP_BAND: return type != SPLAHIP_FLOAT;
            case SPLAHIP_MIN:
            case SPLAHIP_MAX:
            case SPLAHIP_FIRST:
            case SPLAHIP_SECOND: return true;
            default: return false;
        }
    }
    // smallest and largest value word of an array (equal <=> every stored value is the same word)
    // (n_ptr: where the device holds the array's length — a row pointer's last word — when the host does not know it; nullptr: n_host)
    __global__ __launch_bounds__(256) void value_range_kernel(uint n_host, const uint* __restrict__ n_ptr, const uint* __restrict__ x, uint* __restrict__ out) {
        const uint n  = n_ptr ? *n_ptr : n_host;
        uint       lo = 0xFFFFFFFFu, hi = 0;
        // 16 bytes per lane and load, four loads in flight (one word per lane and turn ran this pass at 0.65 TB/s: 96 us for the 63 MB of the R-MAT 20 triangle form);
        // the words before the first 16-byte boundary and behind the last whole quad go one by one
        const uint head = min(n, (uint) ((16u - ((uintptr_t) x & 15u)) & 15u) / 4u), quads = (n - head) / 4u;
        const uint4* __restrict__ x4 = reinterpret_cast<const uint4*>(x + head);
        const uint   tid = blockIdx.x * 256 + threadIdx.x, nt = gridDim.x * 256;
        auto         take = [&](uint4 a) {
            lo = min(min(lo, a.x), min(a.y, min(a.z, a.w)));
            hi = max(max(hi, a.x), max(a.y, max(a.z, a.w)));
        };
        uint q = tid;
        for (; q + 3 * nt < quads; q += 4 * nt) {
            const uint4 a = x4[q], b = x4[q + nt], c = x4[q + 2 * nt], d = x4[q + 3 * nt];
            take(a), take(b), take(c), take(d);
        }
        for (; q < quads; q += nt) take(x4[q]);
        for (uint k = tid; k < head; k += nt) lo = min(lo, x[k]), hi = max(hi, x[k]);
        for (uint k = head + 4u * quads + tid; k < n; k += nt) lo = min(lo, x[k]), hi = max(hi, x[k]);
        lo = ~wave_max(~lo);
        hi = wave_max(hi);
        // (one pair of atomics per WORKGROUP: same-address atomics serialise — a pair per wavefront of this 1024-workgroup grid cost 115 us per call)
        __shared__ uint s_lo[256 / kWave], s_hi[256 / kWave];
        if (lane_id() == 0) s_lo[wave_id()] = lo, s_hi[wave_id()] = hi;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint w = 1; w < 256 / kWave; ++w) lo = min(lo, s_lo[w]), hi = max(hi, s_hi[w]);
            atomicMin(&out[0], lo);
            atomicMax(&out[1], hi);
        }
    }

    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(256) void mxmT_merge_kernel(const uint* __restrict__ count, const uint* __restrict__ list, uint list_last, const uint* __restrict__ Mj,
                                                             const uint* __restrict__ Ap, const uint* __restrict__ Aj, const T* __restrict__ Ax,
                                                             const uint* __restrict__ Bp, const uint* __restrict__ Bj, const T* __restrict__ Bx,
                                                             T* __restrict__ val, uint* __restrict__ keep, Mult mult, Add add_raw, T init, uint commutative,
                                                             uint count_mode, uint a_word, uint b_word) {
        const Reassoc<Add> add(add_raw);// (ops.hpp) partial folds start at products
        const T pc = mult(from_bits<T>(a_word), from_bits<T>(b_word));// (count_mode: the one product)
        // count_mode: every stored value of A is one word and every stored value of B is one word (adjacency matrices: the triangle count), so
        // every product is pc and the entry's value follows from the NUMBER of matches — no match list, no value fetches (host: merge_counts())
        __shared__ uint s_tile[256 / kWave][kMergeTile];
        __shared__ uint s_hits[256 / kWave][2 * kMergeHits];
        const uint      wv = threadIdx.x / kWave, lane = lane_id();
        const uint      w  = blockIdx.x * (256 / kWave) + wv;
        if (w >= *count) return;// (whole wavefronts leave: nothing below synchronises across wavefronts)
        uint* const tile = s_tile[wv];
        uint* const hits = s_hits[wv];
        const uint  e = list[2 * (list_last - w)], i = list[2 * (list_last - w) + 1], j = Mj[e];// (this list grows from the buffer's back)
        const uint  sa = Ap[i], ea = Ap[i + 1], sb = Bp[j], eb = Bp[j + 1];
        const bool  a_short = (ea - sa) <= (eb - sb);// S = shorter row (in the lanes), L = longer row (in the tile); products keep mult(a, b)
        const uint* Sj = a_short ? Aj : Bj;
        const uint* Lj = a_short ? Bj : Aj;
        const uint  se = a_short ? ea : eb, le = a_short ? eb : ea;
        uint        ks = a_short ? sa : sb, kl = a_short ? sb : sa;
        T           acc = init, tot = init;
        bool        has = false, tot_has = false;
        uint        n_hits = 0;// (uniform)
        auto product = [&](uint ps, uint pl) -> T { return a_short ? mult(Ax[ps], Bx[pl]) : mult(Ax[pl], Bx[ps]); };
        auto flush   = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (uint h0 = 0; h0 < n_hits; h0 += kWave) {
                const uint h = h0 + lane;
                if (h < n_hits) {
                    const T p = product(hits[2 * h], hits[2 * h + 1]);
                    acc       = has ? add(acc, p) : p;
                    has       = true;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            n_hits = 0;
        };
        // every load is a PREFETCH: the block after this one and the tile after this one are requested a step before they can be needed
        // (a block shifted past its placed words is put together from the current and the prefetched block with two lane permutes)
        constexpr uint kTileRegs = kMergeTile / kWave;
        auto load_block = [&](uint k) -> uint { return k + lane < se ? Sj[k + lane] : 0xFFFFFFFFu; };
        auto load_tile  = [&](uint k, uint (&reg)[kTileRegs]) {
#pragma unroll
            for (uint u = 0; u < kTileRegs; ++u) {
                const uint t = k + u * kWave + lane;
                reg[u]       = t < le ? Lj[t] : 0xFFFFFFFFu;
            }
        };
        auto store_tile = [&](const uint (&reg)[kTileRegs]) {
#pragma unroll
            for (uint u = 0; u < kTileRegs; ++u) tile[u * kWave + lane] = reg[u];
        };
        uint s_col = load_block(ks), s_next = load_block(ks + kWave);
        uint t_next[kTileRegs];
        load_tile(kl, t_next);
        store_tile(t_next);
        load_tile(kl + kMergeTile, t_next);
        while (true) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint n_l   = min(kMergeTile, le - kl);
            const uint l_max = tile[n_l - 1];
            const bool valid = ks + lane < se;
            const bool placed = valid && s_col <= l_max;
            uint       pos    = 0;
#pragma unroll
            for (uint step = kMergeTile >> 1; step != 0; step >>= 1)
                if (tile[pos + step - 1] < s_col) pos += step;
            const bool hit = placed && tile[pos] == s_col;
            if (count_mode) n_hits += (uint) __popcll(__ballot(hit));
            else if (commutative) {
                const unsigned long long m = __ballot(hit);
                if (m != 0ull) {
                    const uint cnt = (uint) __popcll(m);
                    if (n_hits + cnt > kMergeHits) flush();
                    if (hit) {
                        const uint slot    = n_hits + __builtin_amdgcn_mbcnt_hi((uint) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint) m, 0u));
                        hits[2 * slot]     = ks + lane;
                        hits[2 * slot + 1] = kl + pos;
                    }
                    n_hits += cnt;
                }
            } else {// ascending column order: this step's lanes in order, behind what earlier steps gave
                T    x = init;
                bool v = hit;
                if (hit) x = product(ks + lane, kl + pos);
                wave_reduce_ordered(x, v, add);
                if (v) tot = tot_has ? add(tot, x) : x;
                tot_has = tot_has || v;
            }
            const uint n_valid = min((uint) kWave, se - ks), n_placed = (uint) __popcll(__ballot(placed));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");// (the tile is rewritten below)
            if (n_placed == n_valid) {// the block is placed: the next 64 words of the shorter row, same tile
                ks += kWave;
                if (ks >= se) break;
                s_col  = s_next;
                s_next = load_block(ks + kWave);
            } else {// words beyond the tile remain: the next tile, the block shifted past the placed words
                kl += kMergeTile;
                if (kl >= le) break;
                const uint src = lane + n_placed;// word `src` of the 128 words (current block, prefetched block)
                const uint lo_part = __shfl(s_col, (int) (src & (kWave - 1u)), kWave), hi_part = __shfl(s_next, (int) (src & (kWave - 1u)), kWave);
                s_col = src < kWave ? lo_part : hi_part;
                ks += n_placed;
                if (n_placed != 0) s_next = load_block(ks + kWave);
                store_tile(t_next);
                load_tile(kl + kMergeTile, t_next);
            }
        }
        if (count_mode) {// n_hits equal products folded: n x pc for PLUS, parity for BXOR, pc itself for the idempotent ops (merge_counts())
            tot_has = n_hits != 0;
            tot     = pc;
            if constexpr (std::is_integral<T>::value) {
                const int op = merge_op_id(add);
                if (op == SPLAHIP_PLUS) tot = from_bits<T>(n_hits * to_bits(pc));
                if (op == SPLAHIP_BXOR) tot = (n_hits & 1u) ? pc : from_bits<T>(0u);
            }
        } else if (commutative) {
            flush();
            wave_reduce_ordered(acc, has, add);
            tot     = acc;
            tot_has = has;
        }
        if (lane == 0) {
            const T r = tot_has ? add(init, tot) : init;
            val[e]    = r;
            keep[e]   = differs(r, init) ? 1u : 0u;
        }
    }

    // ---- heavy entries, OWNER FORM (count mode: every stored value of A one word, every stored value of B one word — the triangle count).
    // The block merge reads BOTH rows of every entry: len(A[i, :]) + len(B[j, :]) words, 18.4 G on the R-MAT 20 triangle form (~75 GB through the caches,
    // 9.4 ms).  Here an entry belongs to the LONGER of its two rows — its owner — and only the SHORTER one is read: 2.2 G words on the same input.  The
    // heavy entries are sorted by owner (key: i for A[i, :], n_rows + j for B[j, :]); a workgroup takes 1024 consecutive sorted entries and, per run of equal
    // owners, sets the owner's columns in a BITMAP over the inner dimension in LDS (n_cols / 8 bytes: up to 1.2 M columns), then its 16 wavefronts walk the
    // run's entries: the other row streams by, 64 coalesced column words a step, a word costs one LDS read and the matches are counted by ballot; the run's
    // bits are cleared by walking the owner's columns again.  An entry's value follows from the number of matches (merge_counts()).  The entry's bounds are
    // fetched lane-parallel for the 1024 entries at once; a wavefront requests the first 64 words of its NEXT entry before it counts the current one.
    // (reference: Algo_mxmT_masked_cpu's two-pointer merge, src/cpu/cpu_mxmT_masked.hpp:100-118: the same matches, counted instead of folded.)
    constexpr uint kOwnerChunk = 1024;
    // largest column id + 1 referred to by a CSR matrix of n_rows rows whose rows are column-sorted (the last entry of every row)
    __global__ __launch_bounds__(256) void max_col_kernel(uint n_rows, const uint* __restrict__ Xp, const uint* __restrict__ Xj, uint* __restrict__ out) {
        uint m = 0;
        for (uint i = blockIdx.x * 256 + threadIdx.x; i < n_rows; i += gridDim.x * 256) {
            const uint e = Xp[i + 1];
            if (e > Xp[i]) m = max(m, Xj[e - 1] + 1u);
        }
        m = wave_max(m);
        __shared__ uint s_m[256 / kWave];
        if (lane_id() == 0) s_m[wave_id()] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint w = 1; w < 256 / kWave; ++w) m = max(m, s_m[w]);
            if (m) atomicMax(out, m);
        }
    }
    // (key, entry) pairs of the owner list from the (entry, key) pairs the per-entry kernel appended
    __global__ __launch_bounds__(256) void owner_split_kernel(uint n, const uint* __restrict__ list, uint* __restrict__ key, uint* __restrict__ ent) {
        const uint k = blockIdx.x * 256 + threadIdx.x;
        if (k < n) ent[k] = list[2 * k], key[k] = list[2 * k + 1];
    }
    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(1024) void mxmT_owner_kernel(uint n_owned, const uint* __restrict__ key, const uint* __restrict__ ent, const uint* __restrict__ erow, uint n_rows,
                                                              uint n_words, const uint* __restrict__ Mj, const uint* __restrict__ Ap, const uint* __restrict__ Aj,
                                                              const uint* __restrict__ Bp, const uint* __restrict__ Bj, T* __restrict__ val, uint* __restrict__ keep, Mult mult, Add add_raw,
                                                              T init, uint a_word, uint b_word) {
        const Reassoc<Add> add(add_raw);
        const T            pc = mult(from_bits<T>(a_word), from_bits<T>(b_word));// the one product
        extern __shared__ uint s_dyn[];
        uint* const s_bits = s_dyn;                 // n_words
        uint* const s_beg  = s_dyn + n_words;       // per entry of the chunk: first word of the row that streams
        uint* const s_len  = s_beg + kOwnerChunk;   //                         its length | bit 31: the row that streams is B's (the owner is A's)
        uint* const s_key  = s_len + kOwnerChunk;   //                         owner key
        uint* const s_ent  = s_key + kOwnerChunk;   //                         the entry itself
        uint* const s_val  = s_ent + kOwnerChunk;   //                         its result (written by the wavefront that counted it, stored at the end of the chunk)
        uint* const s_run  = s_val + kOwnerChunk;   // first entry of every run of equal owners in the chunk
        uint* const s_tmp  = s_run + kOwnerChunk;   // scan scratch (kOwnerChunk / 64 + 2 words)
        const uint  tid = threadIdx.x, lane = lane_id(), wv = wave_id();
        for (uint w = tid; w < n_words; w += 1024) s_bits[w] = 0;
        for (uint p0 = blockIdx.x * kOwnerChunk; p0 < n_owned; p0 += gridDim.x * kOwnerChunk) {
            const uint n_here = min(kOwnerChunk, n_owned - p0);
            __syncthreads();// (the arrays of the chunk before this one are free; first turn: the bitmap is clean)
            if (tid < n_here) {// the bounds of every entry of the chunk at once: one dependent chain per chunk
                const uint k = key[p0 + tid], my_e = ent[p0 + tid];
                const uint i = erow[my_e], j = Mj[my_e];
                const bool a_owns = k < n_rows;// the owner is A[i, :]: B[j, :] streams
                const uint b = a_owns ? Bp[j] : Ap[i], e2 = a_owns ? Bp[j + 1] : Ap[i + 1];
                s_beg[tid] = b, s_len[tid] = (e2 - b) | (a_owns ? 0x80000000u : 0u), s_key[tid] = k, s_ent[tid] = my_e;
            }
            __syncthreads();
            // the runs of equal owners: every entry that differs from its predecessor starts one; their positions, in order, by a prefix sum over the chunk
            uint n_runs;
            {
                const bool     head = tid < n_here && (tid == 0 || s_key[tid] != s_key[tid - 1]);
                const uint     at   = block_exclusive_sum<kOwnerChunk>(head ? 1u : 0u, s_tmp, n_runs);
                if (head) s_run[at] = tid;
                __syncthreads();
            }
            for (uint rr = 0; rr < n_runs; ++rr) {
                const uint r0 = s_run[rr], r1 = rr + 1 < n_runs ? s_run[rr + 1] : n_here;
                const uint k  = s_key[r0];
                const bool  a_owns = k < n_rows;
                const uint  orow = a_owns ? k : k - n_rows;
                const uint* Op = a_owns ? Ap : Bp;
                const uint* Oj = a_owns ? Aj : Bj;
                const uint* Sj = a_owns ? Bj : Aj;
                const uint  ob = Op[orow], oe = Op[orow + 1];
                for (uint q = ob + tid; q < oe; q += 1024) {
                    const uint c = Oj[q];
                    (void) __hip_atomic_fetch_or(s_bits + (c >> 5), 1u << (c & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __syncthreads();
                // wavefront wv walks the run's entries r0 + wv, r0 + wv + 16, ... as ONE sequence of 64-word steps (entry after entry, a row's steps in order) with
                // SIX steps' (twelve) loads in flight: with one workgroup per CU (the bitmap) sixteen wavefronts are all there is to cover the memory latency.  The step
                // generator is BRANCH-FREE — every step loads (a step behind the wavefront's last entry re-reads word 0), its bookkeeping is selects over two LDS
                // reads — so that the loads sit in straight-line code: with a load inside a branch the compiler waits for every earlier load before each one.
                constexpr uint kStep = 2 * kWave;// words per step: two coalesced loads per lane (a row's median length is ~90 words: most entries are one step)
                uint n_total = 0;// the wavefront's steps in this run
                for (uint t = r0 + wv; t < r1; t += 16) n_total += ((s_len[t] & 0x7FFFFFFFu) + kStep - 1u) / kStep;
                constexpr uint kNoStep = 0x7FFFFFFFu;
                uint           g_t = r0 + wv, g_q = 0, g_b = s_beg[min(g_t, n_here - 1u)], g_len = s_len[min(g_t, n_here - 1u)] & 0x7FFFFFFFu;
                constexpr int  D = 6;
                uint           W0[D], W1[D], F[D], hits = 0;
                bool           P0[D], P1[D];
                auto           issue = [&](int k) {
                    const bool valid = g_t < r1, in0 = valid && g_q + lane < g_len, in1 = valid && g_q + kWave + lane < g_len, last = g_q + kStep >= g_len;
                    W0[k]            = Sj[in0 ? g_b + g_q + lane : 0u];
                    W1[k]            = Sj[in1 ? g_b + g_q + kWave + lane : 0u];
                    P0[k] = in0, P1[k] = in1;
                    F[k]             = valid ? (g_t | (last ? 0x80000000u : 0u)) : kNoStep;
                    const uint t2    = valid && last ? g_t + 16u : g_t;
                    const uint at    = min(t2, n_here - 1u);
                    const uint nb2 = s_beg[at], nl2 = s_len[at] & 0x7FFFFFFFu;
                    g_q   = last ? 0u : g_q + kStep;
                    g_b   = last ? nb2 : g_b;
                    g_len = last ? nl2 : g_len;
                    g_t   = t2;
                };
#pragma unroll
                for (int k = 0; k < D; ++k) issue(k);
                for (uint st = 0; st < n_total; st += D) {
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const uint w0 = W0[k], w1 = W1[k];
                        const bool h0 = P0[k] && ((s_bits[w0 >> 5] >> (w0 & 31)) & 1u), h1 = P1[k] && ((s_bits[w1 >> 5] >> (w1 & 31)) & 1u);
                        hits += (uint) __popcll(__ballot(h0)) + (uint) __popcll(__ballot(h1));
                        if (F[k] >> 31) {// the entry's last step: `hits` equal products folded — n x pc for PLUS, parity for BXOR, pc itself for the idempotent ops (merge_counts())
                            if (lane == 0) {
                                T tot = pc;
                                if constexpr (std::is_integral<T>::value) {
                                    const int op = merge_op_id(add);
                                    if (op == SPLAHIP_PLUS) tot = from_bits<T>(hits * to_bits(pc));
                                    if (op == SPLAHIP_BXOR) tot = (hits & 1u) ? pc : from_bits<T>(0u);
                                }
                                // (the result waits in LDS for the end of the chunk: no global store among the loads)
                                s_val[F[k] & 0x7FFFFFFFu] = to_bits(hits ? add(init, tot) : init);
                            }
                            hits = 0;
                        }
                        issue(k);
                    }
                }
                __syncthreads();
                for (uint q = ob + tid; q < oe; q += 1024) s_bits[Oj[q] >> 5] = 0;
                __syncthreads();
            }
            __syncthreads();
            if (tid < n_here) {// the chunk's results
                const T r       = from_bits<T>(s_val[tid]);
                val[s_ent[tid]] = r;
                keep[s_ent[tid]] = differs(r, init) ? 1u : 0u;
            }
        }
    }

    // one wavefront per heavy mask entry.  TREE: lanes take contiguous pieces of the shorter row (ascending column), look every
    // column up in the longer row by binary search, fold their matches in order, then an in-order shuffle reduction:
    // the products are combined in ascending column order = the merge's order (needs an associative op_add).
    // !TREE: lane 0 runs the plain sequential merge.
    template<typename T, class Mult, class Add, bool TREE>
    __global__ __launch_bounds__(256) void mxmT_heavy_kernel(uint n_rows, const uint* __restrict__ heavy_count, const uint* __restrict__ heavy_list,
                                                             const uint* __restrict__ Mp, const uint* __restrict__ Mj,
                                                             const uint* __restrict__ Ap, const uint* __restrict__ Aj, const T* __restrict__ Ax,
                                                             const uint* __restrict__ Bp, const uint* __restrict__ Bj, const T* __restrict__ Bx,
                                                             T* __restrict__ val, uint* __restrict__ keep, Mult mult, Add add_raw, T init) {
        const Reassoc<Add> add(add_raw);// (ops.hpp) partial folds start at products
        const uint w = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
        if (w >= *heavy_count) return;
        const uint lane = lane_id();
        const uint e    = heavy_list[2 * w];
        const uint i    = heavy_list[2 * w + 1];
        const uint j    = Mj[e];
        const uint sa = Ap[i], ea = Ap[i + 1], sb = Bp[j], eb = Bp[j + 1];
        T          r = init;
        if constexpr (TREE) {
            // S = shorter row (scanned), L = longer row (searched); the product keeps mult(a, b) argument order
            const bool  a_short = (ea - sa) <= (eb - sb);
            const uint* Sj = a_short ? Aj : Bj;
            const uint* Lj = a_short ? Bj : Aj;
            const uint  ss = a_short ? sa : sb, se = a_short ? ea : eb;
            const uint  ls = a_short ? sb : sa, le = a_short ? eb : ea;
            const uint  c  = (se - ss + kWave - 1) / kWave;
            const uint  ks = ss + lane * c;
            const uint  ke = min(ks + c, se);
            T           part  = init;
            bool        valid = false;
            uint        lo    = ls;// columns ascend, so the search window only moves right
            for (uint k = ks; k < ke; ++k) {
                const uint col = Sj[k];
                uint       hi  = le;
                if (k != ks) {// behind the lane's first column the next match is near: gallop from the last lower bound (steps ~ 2 log2(gap), and
                              // the probes stay in the lines the previous search touched) instead of halving the whole rest of the row
                    uint step = 4;
                    hi        = lo;
                    while (hi < le && Lj[hi] < col) {
                        lo = hi + 1;
                        hi = min(le, hi + step);
                        step <<= 1;
                    }
                }
                while (lo < hi) {
                    const uint mid = lo + ((hi - lo) >> 1);
                    if (Lj[mid] < col) lo = mid + 1;
                    else
                        hi = mid;
                }
                if (lo < le && Lj[lo] == col) {
                    const T p = a_short ? mult(Ax[k], Bx[lo]) : mult(Ax[lo], Bx[k]);
                    part      = valid ? add(part, p) : p;
                    valid     = true;
                }
            }
            wave_reduce_ordered(part, valid, add);
            if (valid) r = add(init, part);
        } else {
            uint ka = sa, kb = sb;
            while (ka < ea && kb < eb) {
                const uint ca = Aj[ka], cb = Bj[kb];
                if (ca == cb) {
                    r = add(r, mult(Ax[ka], Bx[kb]));
                    ++ka;
                    ++kb;
                } else if (ca < cb) {
                    ++ka;
                } else {
                    ++kb;
                }
            }
        }
        if (lane == 0) {
            val[e]  = r;
            keep[e] = differs(r, init) ? 1u : 0u;
        }
    }

    __global__ __launch_bounds__(256) void mxmT_compact_kernel(uint n_entries, const uint* __restrict__ Mj, const uint* __restrict__ val,
                                                               const uint* __restrict__ keep, const uint* __restrict__ rank,
                                                               uint* __restrict__ Rj, uint* __restrict__ Rx) {
        const uint e = blockIdx.x * 256 + threadIdx.x;
        if (e < n_entries && keep[e]) {
            Rj[rank[e]] = Mj[e];
            Rx[rank[e]] = val[e];
        }
    }
    __global__ __launch_bounds__(256) void rowptr_from_rank_kernel(uint n_rows, const uint* __restrict__ Mp, const uint* __restrict__ rank, uint* __restrict__ Rp) {
        const uint i = blockIdx.x * 256 + threadIdx.x;
        if (i <= n_rows) Rp[i] = rank[Mp[i]];
    }

    // out[k] = x[0] + ... + x[k - 1] for k in [0, n]
    static hipError_t values_exclusive_scan(const uint* x, uint n, uint* out, hipStream_t s) {
        if (n == 0) return hipMemsetAsync(out, 0, sizeof(uint), s);
        const uint n_chunks = (n + 1023u) / 1024u;
        DevBuf     chunk;
        hipError_t e = chunk.alloc(sizeof(uint) * (size_t) (n_chunks + 1));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(len_sum_kernel, dim3(n_chunks), dim3(256), 0, s, x, n, chunk.as<uint>());
        hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, chunk.as<uint>(), n_chunks);
        hipLaunchKernelGGL(len_scan_kernel, dim3(n_chunks), dim3(256), 0, s, x, n, chunk.as<uint>(), n_chunks, out);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);// (the chunk sums go out of scope)
        return e;
    }

    template<typename T>
    int mxmT_typed(uint n_rows, uint n_entries, const uint* Mp, const uint* Mj, const void* Mx, const uint* Ap, const uint* Aj, const void* Ax,
                   const uint* Bp, const uint* Bj, const void* Bx, uint* Rp, uint** Rj, void** Rx, uint* h_nnz,
                   int op_mult, int op_add, int op_select, uint32_t init_bits, hipStream_t s, hipFunction_t jit_entry = nullptr) {
        struct Busy {
            Busy() { DevBuf::busy_add(1); }
            ~Busy() { DevBuf::busy_add(-1); }
        } busy_scope;// the cached workspaces are in use until this call returns
        ScratchScope scratch_scope(s);// (before every DevBuf: the workspaces below come from the thread's scratch arena and go back when the call ends)

        *Rj = nullptr, *Rx = nullptr, *h_nnz = 0;
        if (n_entries == 0) {
            SPLAHIP_TRY(hipMemsetAsync(Rp, 0, sizeof(uint) * ((size_t) n_rows + 1), s));
            return SPLAHIP_OK;
        }
        const T init = from_bits<T>(init_bits);
        DevBuf  val, keep, rank, hcount, hlist, outj, outx;
        SPLAHIP_TRY(val.alloc(sizeof(uint) * (size_t) n_entries));
        SPLAHIP_TRY(keep.alloc(sizeof(uint) * (size_t) n_entries));
        SPLAHIP_TRY(rank.alloc(sizeof(uint) * ((size_t) n_entries + 1)));
        SPLAHIP_TRY(hcount.alloc(2 * sizeof(uint)));// one 64-bit counter: low word = entries for the galloping search kernel, high word = for the block merge
        SPLAHIP_TRY(hlist.alloc(sizeof(uint) * 2 * (size_t) n_entries));
        SPLAHIP_TRY(hipMemsetAsync(hcount.p, 0, 2 * sizeof(uint), s));
        if (jit_entry) {// user-defined ops (jit.hip): the run-time-compiled per-entry kernel computes every entry; the compaction below is shared
            void* pv = val.p;
            void* pk = keep.p;
            uint  ib = init_bits;
            void* args[] = {&n_rows, &n_entries, &Mp, &Mj, &Mx, &Ap, &Aj, &Ax, &Bp, &Bj, &Bx, &pv, &pk, &ib};
            SPLAHIP_TRY(hipModuleLaunchKernel(jit_entry, (n_entries + 255u) / 256u, 1, 1, 256, 1, 1, 0, s, args, nullptr));
        }
        // the block merge regroups the products like the tree of the search kernel: same condition.  SPLAHIP_MXMT_MERGE=0: search kernel only
        const char* merge_env = knob_env("SPLAHIP_MXMT_MERGE");
        const bool  merge_form = fold_is_reorderable(op_add, TypeId<T>::value) && !(merge_env && merge_env[0] == '0');
        const char* heavy_env  = knob_env("SPLAHIP_MXMT_HEAVY");// len(A row) + len(B row) above which an entry leaves the lane-per-entry merge
        const uint  heavy_from = heavy_env ? (uint) std::strtoul(heavy_env, nullptr, 10) : kHeavy;
        const char* factor_env   = knob_env("SPLAHIP_MXMT_MERGE_FACTOR");// longer row <= factor x shorter row: block merge (default 128; 32 / 128 / all = 13.6 / 13.1 / 13.5 ms on the R-MAT 20 triangle form)
        const uint  merge_factor = factor_env ? (uint) std::strtoul(factor_env, nullptr, 10) : 128u;

        // the row form (mxmT_row_kernel) regroups and reorders the products: op_add must be associative and commutative.  OPT-IN with
        // SPLAHIP_MXMT_ROW=1: on the R-MAT 20 triangle form it is bit-identical and, so far, no faster than the galloping per-entry kernel
        // (35.5 ms against 33.1 ms per call; rows <= 1023: 18.4 ms, longer rows 11.0 ms — 12 M entries of ~650 streamed words each leave the
        // per-entry fixed costs, one round trip for the matches' value words and an in-order reduction, as large as the stream itself)
        const char* row_env  = knob_env("SPLAHIP_MXMT_ROW");
        const bool  row_form = !jit_entry && fold_is_reorderable(op_add, TypeId<T>::value) && binop_is_commutative(op_add) && row_env && row_env[0] == '1';
        DevBuf      cost, cost_before, pieces, piece_first;
        if (row_form) {
            SPLAHIP_TRY(cost.alloc(sizeof(uint) * (size_t) n_entries));
            SPLAHIP_TRY(cost_before.alloc(sizeof(uint) * ((size_t) n_entries + 1)));
            SPLAHIP_TRY(pieces.alloc(sizeof(uint) * 2 * (size_t) n_rows));
            SPLAHIP_TRY(piece_first.alloc(sizeof(uint) * (2 * (size_t) n_rows + 1)));
        }
        // count mode (every stored value of A one word and every stored value of B one word: the triangle count) is looked for BEFORE the entries are classified:
        // it sends every heavy entry to the owner form (mxmT_owner_kernel).  The boundary passes no entry counts for A and B, so the check runs when they are
        // known anyway: nnz(A) = Ap[n_rows], and B is either the same matrix as A or the mask.  SPLAHIP_MXMT_COUNT=0 / SPLAHIP_MXMT_OWNER=0 switch the two off.
        uint count_mode = 0, a_word = 0, b_word = 0, owner_words = 0;
        {
            const char* cenv = knob_env("SPLAHIP_MXMT_COUNT");
            const char* oenv = knob_env("SPLAHIP_MXMT_OWNER");
            const bool  b_is_a = Bp == Ap && Bx == Ax, b_is_m = Bp == Mp && Bx == Mx;
            // (the probe is two passes over the operands' values and a host synchronisation — not for a mask of a few thousand entries, whose merges
            // cost less than that whatever their form: count mode only changes HOW the equal products are folded, never the result)
            const char* fenv       = knob_env("SPLAHIP_MXMT_COUNT_FROM");
            const uint  count_from = fenv && *fenv ? (uint) std::strtoul(fenv, nullptr, 10) : 16384u;
            if (!jit_entry && !row_form && merge_form && merge_counts(op_add, TypeId<T>::value) && (b_is_a || b_is_m) && !(cenv && cenv[0] == '0') && n_entries >= count_from) {
                DevBuf rng;
                SPLAHIP_TRY(rng.alloc(5 * sizeof(uint)));
                const uint init_rng[5] = {0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u, 0u};
                SPLAHIP_TRY(hipMemcpyAsync(rng.p, init_rng, sizeof(init_rng), hipMemcpyHostToDevice, s));
                {// (nnz(A) is read on the device: Ap[n_rows]; B is A — one pass serves both — or the mask, whose entry count the boundary passes)
                    hipLaunchKernelGGL(value_range_kernel, dim3(512), dim3(256), 0, s, 0u, Ap + n_rows, static_cast<const uint*>(Ax), rng.as<uint>());
                    if (!b_is_a) hipLaunchKernelGGL(value_range_kernel, dim3(512), dim3(256), 0, s, n_entries, static_cast<const uint*>(nullptr), static_cast<const uint*>(Bx), rng.as<uint>() + 2);
                    hipLaunchKernelGGL(max_col_kernel, dim3(256), dim3(256), 0, s, n_rows, Ap, Aj, rng.as<uint>() + 4);
                    if (!b_is_a) hipLaunchKernelGGL(max_col_kernel, dim3(256), dim3(256), 0, s, n_rows, Bp, Bj, rng.as<uint>() + 4);
                    uint h_rng[5] = {0, 1, 0, 1, 0};
                    SPLAHIP_TRY(hipMemcpyAsync(h_rng, rng.p, sizeof(h_rng), hipMemcpyDeviceToHost, s));
                    SPLAHIP_TRY(hipStreamSynchronize(s));
                    if (b_is_a) h_rng[2] = h_rng[0], h_rng[3] = h_rng[1];
                    if (h_rng[0] == h_rng[1] && h_rng[2] == h_rng[3]) {
                        count_mode = 1, a_word = h_rng[0], b_word = h_rng[2];
                        const uint words = (h_rng[4] + 31u) / 32u;// bitmap over the inner dimension; + the chunk's three arrays: it has to fit the LDS
                        if (!(oenv && oenv[0] == '0') && n_rows < (1u << 30) && ((size_t) words + 6 * kOwnerChunk + 32) * 4 <= 159u * 1024u) owner_words = std::max(words, 1u);
                    }
                }
            }
        }
        DevBuf okey, oent, okey_s, oent_s, erow;
        if (owner_words) SPLAHIP_TRY(erow.alloc(sizeof(uint) * (size_t) n_entries));
        if (!jit_entry)
            hipLaunchKernelGGL((mxmT_entry_kernel<T, DynBin, DynBin>), dim3((n_entries + 255) / 256), dim3(256), 0, s, n_rows, n_entries, Mp, Mj,
                               static_cast<const T*>(Mx), Ap, Aj, static_cast<const T*>(Ax), Bp, Bj, static_cast<const T*>(Bx), val.as<T>(),
                               keep.as<uint>(), hcount.as<uint>(), hlist.as<uint>(), DynBin(op_mult), DynBin(op_add), op_select, init, row_form ? kRowCap : 0u,
                               row_form ? cost.as<uint>() : nullptr, merge_form ? 1u : 0u, merge_factor, heavy_from, owner_words ? hlist.as<uint>() : nullptr,
                               owner_words ? erow.as<uint>() : nullptr);
        SPLAHIP_TRY(hipGetLastError());
        uint h_heavy[2] = {0, 0}, h_first[2] = {0, 0};
        if (row_form) {
            SPLAHIP_TRY(values_exclusive_scan(cost.as<uint>(), n_entries, cost_before.as<uint>(), s));
            hipLaunchKernelGGL(mxmT_row_pieces_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, n_rows, Mp, Ap, cost_before.as<uint>(), pieces.as<uint>());
            SPLAHIP_TRY(values_exclusive_scan(pieces.as<uint>(), 2 * n_rows, piece_first.as<uint>(), s));
            SPLAHIP_TRY(hipMemcpyAsync(&h_first[0], piece_first.as<uint>() + n_rows, sizeof(uint), hipMemcpyDeviceToHost, s));
            SPLAHIP_TRY(hipMemcpyAsync(&h_first[1], piece_first.as<uint>() + 2 * (size_t) n_rows, sizeof(uint), hipMemcpyDeviceToHost, s));
        }
        SPLAHIP_TRY(hipMemcpyAsync(h_heavy, hcount.p, 2 * sizeof(uint), hipMemcpyDeviceToHost, s));
        SPLAHIP_TRY(hipStreamSynchronize(s));
        if (owner_words && h_heavy[0]) {// the owner form: sort the heavy entries by owner, then one pass
            const uint n_owned = h_heavy[0];
            SPLAHIP_TRY(okey.alloc(sizeof(uint) * (size_t) n_owned));
            SPLAHIP_TRY(oent.alloc(sizeof(uint) * (size_t) n_owned));
            SPLAHIP_TRY(okey_s.alloc(sizeof(uint) * (size_t) n_owned));
            SPLAHIP_TRY(oent_s.alloc(sizeof(uint) * (size_t) n_owned));
            hipLaunchKernelGGL(owner_split_kernel, dim3((n_owned + 255) / 256), dim3(256), 0, s, n_owned, hlist.as<uint>(), okey.as<uint>(), oent.as<uint>());
            size_t     tb   = 0;
            const uint bits = std::max(1u, bit_width(2ull * n_rows));
            SPLAHIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, okey.as<uint>(), okey_s.as<uint>(), oent.as<uint>(), oent_s.as<uint>(), (size_t) n_owned, 0u, bits, s));
            DevBuf st;
            SPLAHIP_TRY(st.alloc(tb ? tb : 4));
            SPLAHIP_TRY(rocprim::radix_sort_pairs(st.p, tb, okey.as<uint>(), okey_s.as<uint>(), oent.as<uint>(), oent_s.as<uint>(), (size_t) n_owned, 0u, bits, s));
            const size_t lds = ((size_t) owner_words + 6 * kOwnerChunk + 32) * 4;
            int          dev = 0, cus = 256;
            (void) hipGetDevice(&dev);
            (void) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            const uint per_cu = (uint) std::max<size_t>(1, std::min<size_t>(2, (160u * 1024u) / lds));// 1024-thread workgroups: at most two per CU
            const uint grid   = std::min((n_owned + kOwnerChunk - 1) / kOwnerChunk, (uint) cus * per_cu * 4u);
            SPLAHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mxmT_owner_kernel<T, DynBin, DynBin>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL((mxmT_owner_kernel<T, DynBin, DynBin>), dim3(grid), dim3(1024), lds, s, n_owned, okey_s.as<uint>(), oent_s.as<uint>(), erow.as<uint>(), n_rows, owner_words, Mj,
                               Ap, Aj, Bp, Bj, val.as<T>(), keep.as<uint>(), DynBin(op_mult), DynBin(op_add), init, a_word, b_word);
            SPLAHIP_TRY(hipGetLastError());
            h_heavy[0] = 0;// (served: nothing is left for the search kernel)
        }
        if (h_heavy[1]) {
            // the plus-times semiring (the triangle count, SpGEMM proper) gets compile-time functors: the runtime-switched ones cost a switch per
            // product and per step of the final reduction
            if (op_mult == SPLAHIP_MULT && op_add == SPLAHIP_PLUS)
                hipLaunchKernelGGL((mxmT_merge_kernel<T, StaBin<SPLAHIP_MULT>, StaBin<SPLAHIP_PLUS>>), dim3((h_heavy[1] + 3) / 4), dim3(256), 0, s, hcount.as<uint>() + 1,
                                   hlist.as<uint>(), n_entries - 1u, Mj, Ap, Aj, static_cast<const T*>(Ax), Bp, Bj, static_cast<const T*>(Bx), val.as<T>(), keep.as<uint>(),
                                   StaBin<SPLAHIP_MULT>(), StaBin<SPLAHIP_PLUS>(), init, 1u, count_mode, a_word, b_word);
            else
                hipLaunchKernelGGL((mxmT_merge_kernel<T, DynBin, DynBin>), dim3((h_heavy[1] + 3) / 4), dim3(256), 0, s, hcount.as<uint>() + 1, hlist.as<uint>(), n_entries - 1u, Mj,
                                   Ap, Aj, static_cast<const T*>(Ax), Bp, Bj, static_cast<const T*>(Bx), val.as<T>(), keep.as<uint>(), DynBin(op_mult), DynBin(op_add), init,
                                   binop_is_commutative(op_add) ? 1u : 0u, count_mode, a_word, b_word);
            SPLAHIP_TRY(hipGetLastError());
        }
        DevBuf item_row;
        if (h_first[1]) {
            SPLAHIP_TRY(item_row.alloc(sizeof(uint) * (size_t) h_first[1]));
            hipLaunchKernelGGL(mxmT_row_items_kernel, dim3((2 * n_rows + 255) / 256), dim3(256), 0, s, n_rows, piece_first.as<uint>(), item_row.as<uint>());
        }
        if (h_first[0]) {// rows up to kRowSmall: 8 KB of LDS per wavefront
            constexpr uint W = 4, kLds = W * row_lds_words(kRowSmall) * 4;
            hipLaunchKernelGGL((mxmT_row_kernel<T, DynBin, DynBin, kRowSmall, W>), dim3((h_first[0] + W - 1) / W), dim3(W * kWave), kLds, s, n_rows, 0u, h_first[0],
                               piece_first.as<uint>(), item_row.as<uint>(), cost_before.as<uint>(), Mp, Mj, Ap, Aj, static_cast<const T*>(Ax), Bp, Bj, static_cast<const T*>(Bx),
                               val.as<T>(), keep.as<uint>(), DynBin(op_mult), DynBin(op_add), init);
            SPLAHIP_TRY(hipGetLastError());
        }
        if (h_first[1] > h_first[0]) {// rows up to kRowCap: 26 KB per wavefront, two wavefronts per workgroup
            constexpr uint W = 2, kLds = W * row_lds_words(kRowCap) * 4;
            hipLaunchKernelGGL((mxmT_row_kernel<T, DynBin, DynBin, kRowCap, W>), dim3((h_first[1] - h_first[0] + W - 1) / W), dim3(W * kWave), kLds, s, n_rows, h_first[0],
                               h_first[1] - h_first[0], piece_first.as<uint>() + n_rows, item_row.as<uint>(), cost_before.as<uint>(), Mp, Mj, Ap, Aj, static_cast<const T*>(Ax), Bp, Bj,
                               static_cast<const T*>(Bx), val.as<T>(), keep.as<uint>(), DynBin(op_mult), DynBin(op_add), init);
            SPLAHIP_TRY(hipGetLastError());
        }
        if (h_heavy[0]) {
            const uint grid = (h_heavy[0] + 3) / 4;
            if (fold_is_reorderable(op_add, TypeId<T>::value))
                hipLaunchKernelGGL((mxmT_heavy_kernel<T, DynBin, DynBin, true>), dim3(grid), dim3(256), 0, s, n_rows, hcount.as<uint>(), hlist.as<uint>(),
                                   Mp, Mj, Ap, Aj, static_cast<const T*>(Ax), Bp, Bj, static_cast<const T*>(Bx), val.as<T>(), keep.as<uint>(),
                                   DynBin(op_mult), DynBin(op_add), init);
            else
                hipLaunchKernelGGL((mxmT_heavy_kernel<T, DynBin, DynBin, false>), dim3(grid), dim3(256), 0, s, n_rows, hcount.as<uint>(), hlist.as<uint>(),
                                   Mp, Mj, Ap, Aj, static_cast<const T*>(Ax), Bp, Bj, static_cast<const T*>(Bx), val.as<T>(), keep.as<uint>(),
                                   DynBin(op_mult), DynBin(op_add), init);
            SPLAHIP_TRY(hipGetLastError());
        }
        SPLAHIP_TRY(flags_exclusive_rank(keep.as<uint>(), n_entries, rank.as<uint>(), s));
        uint nnz = 0;
        SPLAHIP_TRY(hipMemcpyAsync(&nnz, rank.as<uint>() + n_entries, sizeof(uint), hipMemcpyDeviceToHost, s));
        SPLAHIP_TRY(hipStreamSynchronize(s));
        SPLAHIP_TRY(outj.alloc_plain(sizeof(uint) * (size_t) nnz));
        SPLAHIP_TRY(outx.alloc_plain(sizeof(uint) * (size_t) nnz));
        hipLaunchKernelGGL(mxmT_compact_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, s, n_entries, Mj, val.as<uint>(), keep.as<uint>(),
                           rank.as<uint>(), outj.as<uint>(), outx.as<uint>());
        hipLaunchKernelGGL(rowptr_from_rank_kernel, dim3((n_rows + 1 + 255) / 256), dim3(256), 0, s, n_rows, Mp, rank.as<uint>(), Rp);
        SPLAHIP_TRY(hipGetLastError());
        SPLAHIP_TRY(hipStreamSynchronize(s));
        *h_nnz = nnz;
        *Rj    = static_cast<uint*>(outj.release());
        *Rx    = outx.release();
        return SPLAHIP_OK;
    }

    // =============================================================================================================
    // mxm (ESC)
    // =============================================================================================================
    using u64 = unsigned long long;

    // len[e] = nnz(B[Aj[e], :]) for the A entries of a row block
    __global__ __launch_bounds__(256) void mxm_len_kernel(uint e0, uint n, const uint* __restrict__ Aj, const uint* __restrict__ Bp, uint* __restrict__ len) {
        const uint t = blockIdx.x * 256 + threadIdx.x;
        if (t < n) {
            const uint k = Aj[e0 + t];
            len[t]       = Bp[k + 1] - Bp[k];
        }
    }
    // products per row of A (for the row blocking): rowprod[i] = sum over entries of row i of nnz(B[k,:])
    __global__ __launch_bounds__(256) void mxm_rowprod_kernel(uint n_rows, const uint* __restrict__ Ap, const uint* __restrict__ Aj,
                                                              const uint* __restrict__ Bp, u64* __restrict__ rowprod) {
        // a lane per row for the short rows; a row of more than 64 entries is then summed by its whole wavefront, lanes striding its entries
        // (one lane walking a 60,000-entry hub row alone was 3.3 ms of every mxm call)
        const uint i     = blockIdx.x * 256 + threadIdx.x;
        const uint first = i < n_rows ? Ap[i] : 0u, last = i < n_rows ? Ap[i + 1] : 0u;
        const bool wide  = last - first > kWave;
        u64        sum   = 0;
        if (!wide)
            for (uint e = first; e < last; ++e) {
                const uint k = Aj[e];
                sum += Bp[k + 1] - Bp[k];
            }
        unsigned long long todo = __ballot(wide);
        while (todo != 0ull) {
            const uint l = (uint) __ffsll((long long) todo) - 1u;
            todo &= todo - 1ull;
            const uint f = __builtin_amdgcn_readlane(first, l), t = __builtin_amdgcn_readlane(last, l);
            u64        part = 0;
            for (uint e = f + lane_id(); e < t; e += kWave) {
                const uint k = Aj[e];
                part += Bp[k + 1] - Bp[k];
            }
#pragma unroll
            for (uint off = kWave / 2; off != 0; off >>= 1) part += __shfl_down(part, off, kWave);
            part = __shfl(part, 0, kWave);
            if (lane_id() == l) sum = part;
        }
        if (i < n_rows) rowprod[i] = sum;
    }

    __global__ __launch_bounds__(256) void len_sum_kernel(const uint* __restrict__ x, uint n, uint* __restrict__ chunk) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            sum   = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u)
            if (first + u < n) sum += x[first + u];
        uint total;
        (void) block_exclusive_sum<256>(sum, s_tmp, total);
        if (threadIdx.x == 0) chunk[blockIdx.x] = total;
    }
    __global__ __launch_bounds__(256) void len_scan_kernel(const uint* __restrict__ x, uint n, const uint* __restrict__ chunk, uint n_chunks, uint* __restrict__ out) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            val[4];
        uint            sum = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            val[u] = first + u < n ? x[first + u] : 0;
            sum += val[u];
        }
        uint total;
        uint pos = chunk[blockIdx.x] + block_exclusive_sum<256>(sum, s_tmp, total);
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            if (first + u < n) out[first + u] = pos;
            pos += val[u];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = chunk[n_chunks];
    }

    template<typename T, typename K, class Mult>
    __global__ __launch_bounds__(256) void mxm_expand_kernel(uint n_products, uint e0, uint n_entries, const uint* __restrict__ offs, uint n_rows,
                                                             const uint* __restrict__ Ap, const uint* __restrict__ Aj, const T* __restrict__ Ax,
                                                             const uint* __restrict__ Bp, const uint* __restrict__ Bj, const T* __restrict__ Bx,
                                                             Mult mult, uint r0, uint col_bits, K* __restrict__ key, T* __restrict__ val) {
        const uint p = blockIdx.x * 256 + threadIdx.x;
        if (p >= n_products) return;
        uint lo = 0, hi = n_entries;// entry t of the block with offs[t] <= p < offs[t+1]
        while (hi - lo > 1) {
            const uint mid = lo + ((hi - lo) >> 1);
            if (offs[mid] <= p) lo = mid;
            else
                hi = mid;
        }
        const uint e  = e0 + lo;
        const uint i  = row_of_entry(Ap, n_rows, e);
        const uint k  = Aj[e];
        const uint kb = Bp[k] + (p - offs[lo]);
        key[p]        = (K(i - r0) << col_bits) | K(Bj[kb]);// compact (row in block, column): only these bits are sorted; K = 32-bit when they fit
        val[p]        = mult(Ax[e], Bx[kb]);
    }

    template<typename K>
    __global__ __launch_bounds__(256) void mxm_head_kernel(const K* __restrict__ key, uint n, uint* __restrict__ head) {
        const uint p = blockIdx.x * 256 + threadIdx.x;
        if (p < n) head[p] = (p == 0 || key[p - 1] != key[p]) ? 1u : 0u;
    }

    constexpr uint kRunShortMxm = 64;

    // one lane per product; the lanes that start a run fold it in order from init.  The run's later products sit in the following
    // lanes of the same wave (avg run length ~2), so they are fetched with shuffles; a run that leaves the wave continues from memory,
    // a run longer than kRunShortMxm is queued for a wavefront of its own (mxm_long_run_kernel).
    template<typename T, typename K, class Add>
    __global__ __launch_bounds__(256) void mxm_run_kernel(const K* __restrict__ key, const T* __restrict__ val, uint n, const uint* __restrict__ head,
                                                          const uint* __restrict__ run_rank, K* __restrict__ run_key, T* __restrict__ run_val,
                                                          uint* __restrict__ run_keep, uint* __restrict__ long_count, uint* __restrict__ long_pos,
                                                          Add add, T init) {
        const uint p      = blockIdx.x * 256 + threadIdx.x;
        const uint lane   = lane_id();
        const bool inside = p < n;
        const K    kk     = inside ? key[p] : K(0);
        const T    mine   = inside ? val[p] : T();
        const bool lead   = inside && head[p] != 0;
        T          acc    = add(init, mine);
        uint       len    = 1;     // products folded so far
        bool       open   = lead;  // still extending the run
        bool       spill  = false; // the run continues beyond this wave
        for (uint d = 1; d < kWave; ++d) {
            if (__ballot(open) == 0ull) break;
            const K    nk   = __shfl_down(kk, d, kWave);
            const T    nv   = __shfl_down(mine, d, kWave);
            const bool next = __shfl_down((int) inside, d, kWave) != 0 && lane + d < kWave;
            if (open) {
                if (lane + d >= kWave) {// ran off the wave: whether the run continues is decided from memory below
                    spill = true;
                    open  = false;
                } else if (next && nk == kk && len < kRunShortMxm) {
                    acc = add(acc, nv);
                    ++len;
                } else {
                    open = false;
                    if (next && nk == kk) spill = true;// hit the length cap inside the wave
                }
            }
        }
        if (open) spill = true;// (lane 0 with a run covering the whole wave)
        if (!lead) return;
        uint q = p + len;
        if (spill) {
            while (q < n && q - p < kRunShortMxm && key[q] == kk) {
                acc = add(acc, val[q]);
                ++q;
            }
        }
        const uint run = run_rank[p];
        run_key[run]   = kk;
        if (q < n && q - p >= kRunShortMxm && key[q] == kk) {
            const uint slot = atomicAdd(long_count, 1u);
            long_pos[slot]  = p;
        } else {
            run_val[run]  = acc;
            run_keep[run] = differs(acc, init) ? 1u : 0u;
        }
    }

    template<typename T, typename K, class Add, bool TREE>
    __global__ __launch_bounds__(256) void mxm_long_run_kernel(const K* __restrict__ key, const T* __restrict__ val, uint n,
                                                               const uint* __restrict__ long_count, const uint* __restrict__ long_pos,
                                                               const uint* __restrict__ run_rank, T* __restrict__ run_val, uint* __restrict__ run_keep,
                                                               Add add_raw, T init) {
        const Reassoc<Add> add(add_raw);// (ops.hpp) partial folds start at products
        const uint w = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
        if (w >= *long_count) return;
        const uint lane = lane_id();
        const uint p    = long_pos[w];
        const K    kk   = key[p];
        uint       lo = p + 1, hi = n;
        while (lo < hi) {
            const uint mid = lo + ((hi - lo) >> 1);
            if (key[mid] == kk) lo = mid + 1;
            else
                hi = mid;
        }
        const uint e = lo;
        T          acc;
        if constexpr (TREE) {
            const uint c  = (e - p + kWave - 1) / kWave;
            const uint ks = p + lane * c;
            const uint ke = min(ks + c, e);
            bool       valid = ks < e;
            T          part  = init;
            if (valid) {
                part = val[ks];
                for (uint q = ks + 1; q < ke; ++q) part = add(part, val[q]);
            }
            wave_reduce_ordered(part, valid, add);
            acc = add(init, part);
        } else {
            acc = init;
            if (lane == 0)
                for (uint q = p; q < e; ++q) acc = add(acc, val[q]);
        }
        if (lane == 0) {
            run_val[run_rank[p]]  = acc;
            run_keep[run_rank[p]] = differs(acc, init) ? 1u : 0u;
        }
    }

    // kept runs -> (row, col, value) appended at out_base + rank
    template<typename K>
    __global__ __launch_bounds__(256) void mxm_emit_kernel(uint n_runs, const K* __restrict__ run_key, const uint* __restrict__ run_val,
                                                           const uint* __restrict__ run_keep, const uint* __restrict__ rank, uint r0, uint col_bits,
                                                           uint* __restrict__ out_row, uint* __restrict__ out_col, uint* __restrict__ out_val) {
        const uint t = blockIdx.x * 256 + threadIdx.x;
        if (t < n_runs && run_keep[t]) {
            const uint o = rank[t];
            out_row[o]   = r0 + uint(run_key[t] >> col_bits);
            out_col[o]   = uint(run_key[t] & ((K(1) << col_bits) - K(1)));
            out_val[o]   = run_val[t];
        }
    }
    // Rp[i] = first output entry whose row >= i (rows ascending)
    __global__ __launch_bounds__(256) void rowptr_from_rows_kernel(uint n_rows, uint nnz, const uint* __restrict__ rows, uint* __restrict__ Rp) {
        const uint i = blockIdx.x * 256 + threadIdx.x;
        if (i > n_rows) return;
        uint lo = 0, hi = nnz;
        while (lo < hi) {
            const uint mid = lo + ((hi - lo) >> 1);
            if (rows[mid] < i) lo = mid + 1;
            else
                hi = mid;
        }
        Rp[i] = lo;
    }

    // fp32 NaN init: the reference keeps a column when `R_tmp[col] != I` (src/cpu/cpu_mxm.hpp:109-113), which holds for EVERY column when I is a NaN —
    // its result is dense, untouched columns carrying I.  The sparse product (every touched entry survives the same test) is spread over dense rows.
    __global__ __launch_bounds__(256) void mxm_dense_fill_kernel(uint n_rows, uint n_cols, uint init_bits, uint* __restrict__ Rp, uint* __restrict__ Rj,
                                                                 uint* __restrict__ Rx) {
        const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x, n = (size_t) n_rows * n_cols;
        if (t <= n_rows) Rp[t] = (uint) (t * n_cols);
        if (t < n) {
            Rj[t] = (uint) (t % n_cols);
            Rx[t] = init_bits;
        }
    }
    __global__ __launch_bounds__(256) void mxm_dense_scatter_kernel(uint n_rows, uint n_cols, const uint* __restrict__ Sp, const uint* __restrict__ Sj,
                                                                    const uint* __restrict__ Sx, uint* __restrict__ Rx) {
        const uint i = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
        if (i >= n_rows) return;
        for (uint k = Sp[i] + lane_id(); k < Sp[i + 1]; k += kWave) Rx[(size_t) i * n_cols + Sj[k]] = Sx[k];
    }

    struct Piece {
        void *row, *col, *val;
        uint  n;
    };

    template<typename T>
    int mxm_typed(uint a_rows, uint b_cols, const uint* Ap, const uint* Aj, const void* Ax, const uint* Bp, const uint* Bj, const void* Bx,
                  uint* Rp, uint** Rj, void** Rx, uint* h_nnz, u64* h_products, int op_mult, int op_add, uint32_t init_bits,
                  u64 block_products, hipStream_t s, hipFunction_t jit_expand = nullptr, hipFunction_t jit_run = nullptr) {
        struct Busy {
            Busy() { DevBuf::busy_add(1); }
            ~Busy() { DevBuf::busy_add(-1); }
        } busy_scope;// the cached workspaces are in use until this call returns

        *Rj = nullptr, *Rx = nullptr, *h_nnz = 0;
        if (h_products) *h_products = 0;
        const T    init = from_bits<T>(init_bits);
        PhaseTrace trace;
        // products per row -> host -> row blocks of at most `block_products` products (a single row may exceed it)
        std::vector<uint> h_Ap(a_rows + 1);
        std::vector<u64>  h_rowprod(a_rows);
        {
            DevBuf rowprod;
            SPLAHIP_TRY(rowprod.alloc(sizeof(u64) * (size_t) a_rows));
            hipLaunchKernelGGL(mxm_rowprod_kernel, dim3((a_rows + 255) / 256), dim3(256), 0, s, a_rows, Ap, Aj, Bp, rowprod.as<u64>());
            SPLAHIP_TRY(hipGetLastError());
            SPLAHIP_TRY(hipMemcpyAsync(h_rowprod.data(), rowprod.p, sizeof(u64) * (size_t) a_rows, hipMemcpyDeviceToHost, s));
            SPLAHIP_TRY(hipMemcpyAsync(h_Ap.data(), Ap, sizeof(uint) * ((size_t) a_rows + 1), hipMemcpyDeviceToHost, s));
            SPLAHIP_TRY(hipStreamSynchronize(s));
        }
        std::vector<Piece> pieces;
        std::vector<char>  piece_in_arena;// per piece: carved from the cached arena (not freed) or pool-allocated
        auto               free_pieces = [&]() {
            for (size_t k = 0; k < pieces.size(); ++k) {
                if (k < piece_in_arena.size() && piece_in_arena[k]) continue;// carved from the cached arena
                (void) hipFreeAsync(pieces[k].row, s);
                (void) hipFreeAsync(pieces[k].col, s);
                (void) hipFreeAsync(pieces[k].val, s);
            }
            pieces.clear();
        };
        // row blocks: [r0, r1) with at most block_products products (a single row may exceed it)
        struct Block {
            uint r0, r1, e0, n_entries, n_products;
        };
        std::vector<Block> blocks;
        u64                total_products = 0;
        uint               max_entries = 0, max_products = 0;
        for (uint r0 = 0; r0 < a_rows;) {
            u64  prod = 0;
            uint r1   = r0;
            while (r1 < a_rows && (r1 == r0 || prod + h_rowprod[r1] <= block_products)) prod += h_rowprod[r1++];
            total_products += prod;
            if (prod > 0xFFFFFFF0ull) return fail(SPLAHIP_INVALID_ARGUMENT, "a single row expands to %llu products (> 2^32)", prod);
            if (prod > 0) {
                blocks.push_back({r0, r1, h_Ap[r0], h_Ap[r1] - h_Ap[r0], (uint) prod});
                max_entries  = std::max(max_entries, h_Ap[r1] - h_Ap[r0]);
                max_products = std::max(max_products, (uint) prod);
            }
            r0 = r1;
        }
        trace.mark("row products + blocking");
        // one workspace for every block (allocating and freeing GB-sized buffers per block dominated the first version)
        DevBuf len, offs, chunk, key_a, key_b, val_a, val_b, tmp, head, run_rank, run_key, run_val, run_keep, lcount, lpos, keep_rank;
        size_t tmp_bytes = 0;
        const uint col_bits = std::max(1u, bit_width(b_cols ? b_cols - 1 : 0));
        uint       max_span = 1;
        for (const Block& b : blocks) max_span = std::max(max_span, b.r1 - b.r0);
        const uint end_bit_max = col_bits + std::max(1u, bit_width(max_span - 1));
        if (!blocks.empty()) {
            const size_t np1 = (size_t) max_products + 1;
            SPLAHIP_TRY(len.alloc_cached(0, sizeof(uint) * (size_t) max_entries));
            SPLAHIP_TRY(offs.alloc_cached(1, sizeof(uint) * ((size_t) max_entries + 1)));
            SPLAHIP_TRY(chunk.alloc_cached(2, sizeof(uint) * ((size_t) (max_entries + 1023u) / 1024u + 2)));
            SPLAHIP_TRY(key_a.alloc_cached(3, sizeof(u64) * (size_t) max_products));
            SPLAHIP_TRY(key_b.alloc_cached(4, sizeof(u64) * (size_t) max_products));
            SPLAHIP_TRY(val_a.alloc_cached(5, sizeof(uint) * (size_t) max_products));
            SPLAHIP_TRY(val_b.alloc_cached(6, sizeof(uint) * (size_t) max_products));
            SPLAHIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key_a.as<u64>(), key_b.as<u64>(), val_a.as<uint>(), val_b.as<uint>(),
                                                  max_products, 0, end_bit_max, s));
            SPLAHIP_TRY(tmp.alloc_cached(7, tmp_bytes));
            SPLAHIP_TRY(head.alloc_cached(8, sizeof(uint) * (size_t) max_products));
            SPLAHIP_TRY(run_rank.alloc_cached(9, sizeof(uint) * np1));
            SPLAHIP_TRY(run_key.alloc_cached(10, sizeof(u64) * (size_t) max_products));
            SPLAHIP_TRY(run_val.alloc_cached(11, sizeof(uint) * (size_t) max_products));
            SPLAHIP_TRY(run_keep.alloc_cached(12, sizeof(uint) * (size_t) max_products));
            SPLAHIP_TRY(lcount.alloc_cached(13, sizeof(uint)));
            SPLAHIP_TRY(lpos.alloc_cached(14, sizeof(uint) * ((size_t) max_products / kRunShortMxm + 1)));
            SPLAHIP_TRY(keep_rank.alloc_cached(15, sizeof(uint) * np1));
        }
        // output pieces of the row blocks are carved from one cached arena (3 words per kept entry); its size follows the largest
        // result seen so far, a piece that does not fit comes from the pool instead
        static thread_local size_t arena_want = 0;
        DevBuf arena;
        size_t arena_words = 0, arena_used = 0;
        if (!blocks.empty() && arena_want) {
            SPLAHIP_TRY(arena.alloc_cached(16, sizeof(uint) * arena_want));
            arena_words = arena_want;
        }
        trace.mark("workspace allocation");
        // The result's row pointers and *h_nnz are 32-bit (reference include/spla/config.hpp:56: uint): the entry count is summed in 64 bits and a
        // result that cannot be indexed is refused BEFORE anything is sized from a wrapped count (R-MAT scale 19 x itself already holds 3.5 G
        // entries).  SPLAHIP_MXM_MAX_NNZ lowers the limit (tests).
        u64       total_out = 0;
        const u64 nnz_limit = [] {
            const char* e = knob_env("SPLAHIP_MXM_MAX_NNZ");
            const u64   v = e ? std::strtoull(e, nullptr, 10) : 0;
            return v ? std::min<u64>(v, 0xFFFFFFF0ull) : 0xFFFFFFF0ull;
        }();
        int status = SPLAHIP_OK;
        for (const Block& blk : blocks) {
            if (trace.on && &blk == &blocks.back()) trace.mark("all blocks but the last");
            const uint e0 = blk.e0, n_entries = blk.n_entries, n_products = blk.n_products;
            auto run_block = [&](auto key_tag) -> int {
                using K = decltype(key_tag);// 32-bit keys when (row in block, column) fits: a third less sort traffic
                const uint n_chunks = (n_entries + 1023u) / 1024u;
                hipLaunchKernelGGL(mxm_len_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, s, e0, n_entries, Aj, Bp, len.as<uint>());
                hipLaunchKernelGGL(len_sum_kernel, dim3(n_chunks), dim3(256), 0, s, len.as<uint>(), n_entries, chunk.as<uint>());
                hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, chunk.as<uint>(), n_chunks);
                hipLaunchKernelGGL(len_scan_kernel, dim3(n_chunks), dim3(256), 0, s, len.as<uint>(), n_entries, chunk.as<uint>(), n_chunks, offs.as<uint>());
                if (jit_expand) {// user-defined ops (jit.hip): 64-bit keys, the run-time-compiled expand kernel
                    uint  np = n_products, ee = e0, ne = n_entries, nr = a_rows, r0 = blk.r0, cb = col_bits;
                    void *po = offs.p, *pk = key_a.p, *pv = val_a.p;
                    void* args[] = {&np, &ee, &ne, &po, &nr, &Ap, &Aj, &Ax, &Bp, &Bj, &Bx, &r0, &cb, &pk, &pv};
                    SPLAHIP_TRY(hipModuleLaunchKernel(jit_expand, (n_products + 255u) / 256u, 1, 1, 256, 1, 1, 0, s, args, nullptr));
                } else
                    hipLaunchKernelGGL((mxm_expand_kernel<T, K, DynBin>), dim3((n_products + 255) / 256), dim3(256), 0, s, n_products, e0, n_entries,
                                       offs.as<uint>(), a_rows, Ap, Aj, static_cast<const T*>(Ax), Bp, Bj, static_cast<const T*>(Bx), DynBin(op_mult),
                                       blk.r0, col_bits, key_a.as<K>(), val_a.as<T>());
                SPLAHIP_TRY(hipGetLastError());
                if (trace.on && &blk == &blocks.back()) trace.mark("  last block: expand");
                // stable sort by (row, column): equal keys keep the expand order = the reference's fold order
                // only the bits a (row in block, column) key can have are sorted: ~col_bits + log2(rows in the block) instead of 64
                size_t     tb      = tmp_bytes;
                const uint end_bit = col_bits + std::max(1u, bit_width(blk.r1 - blk.r0 - 1));
                SPLAHIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, key_a.as<K>(), key_b.as<K>(), val_a.as<uint>(), val_b.as<uint>(),
                                                      n_products, 0, end_bit, s));
                if (trace.on && &blk == &blocks.back()) trace.mark("  last block: sort");
                // runs
                hipLaunchKernelGGL(mxm_head_kernel<K>, dim3((n_products + 255) / 256), dim3(256), 0, s, key_b.as<K>(), n_products, head.as<uint>());
                SPLAHIP_TRY(flags_exclusive_rank(head.as<uint>(), n_products, run_rank.as<uint>(), s));
                uint n_runs = 0;
                SPLAHIP_TRY(hipMemcpyAsync(&n_runs, run_rank.as<uint>() + n_products, sizeof(uint), hipMemcpyDeviceToHost, s));
                SPLAHIP_TRY(hipMemsetAsync(lcount.p, 0, sizeof(uint), s));
                SPLAHIP_TRY(hipStreamSynchronize(s));
                if (jit_run) {// a lane per run, the whole run in order (no queue of long runs)
                    uint  np = n_products, ib = init_bits;
                    void *pk = key_b.p, *pv = val_b.p, *ph = head.p, *pr = run_rank.p, *qk = run_key.p, *qv = run_val.p, *qe = run_keep.p;
                    void* args[] = {&pk, &pv, &np, &ph, &pr, &qk, &qv, &qe, &ib};
                    SPLAHIP_TRY(hipModuleLaunchKernel(jit_run, (n_products + 255u) / 256u, 1, 1, 256, 1, 1, 0, s, args, nullptr));
                } else
                    hipLaunchKernelGGL((mxm_run_kernel<T, K, DynBin>), dim3((n_products + 255) / 256), dim3(256), 0, s, key_b.as<K>(), val_b.as<T>(), n_products,
                                       head.as<uint>(), run_rank.as<uint>(), run_key.as<K>(), run_val.as<T>(), run_keep.as<uint>(), lcount.as<uint>(),
                                       lpos.as<uint>(), DynBin(op_add), init);
                SPLAHIP_TRY(hipGetLastError());
                uint h_long = 0;
                SPLAHIP_TRY(hipMemcpyAsync(&h_long, lcount.p, sizeof(uint), hipMemcpyDeviceToHost, s));
                SPLAHIP_TRY(hipStreamSynchronize(s));
                if (h_long) {
                    const uint grid = (h_long + 3) / 4;
                    if (fold_is_reorderable(op_add, TypeId<T>::value))
                        hipLaunchKernelGGL((mxm_long_run_kernel<T, K, DynBin, true>), dim3(grid), dim3(256), 0, s, key_b.as<K>(), val_b.as<T>(), n_products,
                                           lcount.as<uint>(), lpos.as<uint>(), run_rank.as<uint>(), run_val.as<T>(), run_keep.as<uint>(), DynBin(op_add), init);
                    else
                        hipLaunchKernelGGL((mxm_long_run_kernel<T, K, DynBin, false>), dim3(grid), dim3(256), 0, s, key_b.as<K>(), val_b.as<T>(), n_products,
                                           lcount.as<uint>(), lpos.as<uint>(), run_rank.as<uint>(), run_val.as<T>(), run_keep.as<uint>(), DynBin(op_add), init);
                    SPLAHIP_TRY(hipGetLastError());
                }
                if (trace.on && &blk == &blocks.back()) trace.mark("  last block: runs");
                // drop == init, compact
                SPLAHIP_TRY(flags_exclusive_rank(run_keep.as<uint>(), n_runs, keep_rank.as<uint>(), s));
                uint n_out = 0;
                SPLAHIP_TRY(hipMemcpyAsync(&n_out, keep_rank.as<uint>() + n_runs, sizeof(uint), hipMemcpyDeviceToHost, s));
                SPLAHIP_TRY(hipStreamSynchronize(s));
                if (n_out) {
                    DevBuf     orow, ocol, oval;
                    const bool fits = arena_used + 3 * (size_t) n_out <= arena_words;
                    if (fits) {
                        uint* base = arena.as<uint>() + arena_used;
                        orow.p = base, ocol.p = base + n_out, oval.p = base + 2 * (size_t) n_out;
                        orow.cached = ocol.cached = oval.cached = true;// not owned: part of the arena
                        arena_used += 3 * (size_t) n_out;
                    } else {
                        SPLAHIP_TRY(orow.alloc(sizeof(uint) * (size_t) n_out));
                        SPLAHIP_TRY(ocol.alloc(sizeof(uint) * (size_t) n_out));
                        SPLAHIP_TRY(oval.alloc(sizeof(uint) * (size_t) n_out));
                    }
                    piece_in_arena.push_back(fits ? 1 : 0);
                    hipLaunchKernelGGL(mxm_emit_kernel<K>, dim3((n_runs + 255) / 256), dim3(256), 0, s, n_runs, run_key.as<K>(), run_val.as<uint>(),
                                       run_keep.as<uint>(), keep_rank.as<uint>(), blk.r0, col_bits, orow.as<uint>(), ocol.as<uint>(), oval.as<uint>());
                    SPLAHIP_TRY(hipGetLastError());
                    SPLAHIP_TRY(hipStreamSynchronize(s));
                    pieces.push_back({orow.release(), ocol.release(), oval.release(), n_out});
                    total_out += n_out;
                    if (total_out > nnz_limit)
                        return fail(SPLAHIP_NOT_IMPLEMENTED, "the product holds more than %llu entries: not representable with 32-bit row pointers", (unsigned long long) nnz_limit);
                }
                if (trace.on && &blk == &blocks.back()) trace.mark("  last block: compact + emit");
                return SPLAHIP_OK;
            };
            status = (!jit_expand && col_bits + std::max(1u, bit_width(blk.r1 - blk.r0 - 1)) <= 32u) ? run_block(uint{}) : run_block(u64{});
            if (status != SPLAHIP_OK) break;
        }
        if (status != SPLAHIP_OK) {
            free_pieces();
            return status;
        }
        trace.mark("all row blocks");
        // concatenate the pieces (row blocks are in row order) and derive Rp
        DevBuf rows, outj, outx;
        hipError_t e = rows.alloc_cached(17, sizeof(uint) * (size_t) total_out);
        if (e == hipSuccess) e = outj.alloc_plain(sizeof(uint) * (size_t) total_out);
        if (e == hipSuccess) e = outx.alloc_plain(sizeof(uint) * (size_t) total_out);
        size_t off = 0;
        for (auto& pc : pieces) {
            if (e != hipSuccess) break;
            e = hipMemcpyAsync(rows.as<uint>() + off, pc.row, sizeof(uint) * (size_t) pc.n, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(outj.as<uint>() + off, pc.col, sizeof(uint) * (size_t) pc.n, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(outx.as<uint>() + off, pc.val, sizeof(uint) * (size_t) pc.n, hipMemcpyDeviceToDevice, s);
            off += pc.n;
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(rowptr_from_rows_kernel, dim3((a_rows + 1 + 255) / 256), dim3(256), 0, s, a_rows, (uint) total_out, rows.as<uint>(), Rp);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        trace.mark("output allocation + assembly");
        free_pieces();
        trace.mark("free pieces");
        if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm assembly failed: %s", hipGetErrorString(e));
        arena_want = std::max(arena_want, 3 * (size_t) total_out + 64);
        *h_nnz     = (uint) total_out;
        if (h_products) *h_products = total_products;
        *Rj = static_cast<uint*>(outj.release());
        *Rx = outx.release();
        return SPLAHIP_OK;
    }

}// namespace splahip

namespace splahip {
    // jit_hooks.hpp: the drivers above with the op-dependent kernels replaced by run-time-compiled ones (4-byte elements of any type: uint32_t stands in)
    int mxmT_jit(uint32_t n_rows, uint32_t n_entries, const uint32_t* Mp, const uint32_t* Mj, const void* Mx, const uint32_t* Ap, const uint32_t* Aj, const void* Ax,
                 const uint32_t* Bp, const uint32_t* Bj, const void* Bx, uint32_t* Rp, uint32_t** Rj, void** Rx, uint32_t* h_nnz, uint32_t init_bits, hipFunction_t entry,
                 hipStream_t s) {
        tl_stream = s;
        return mxmT_typed<uint32_t>(n_rows, n_entries, Mp, Mj, Mx, Ap, Aj, Ax, Bp, Bj, Bx, Rp, Rj, Rx, h_nnz, SPLAHIP_FIRST, SPLAHIP_FIRST, SPLAHIP_ALWAYS, init_bits, s, entry);
    }
    int mxm_jit(uint32_t a_rows, uint32_t b_cols, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const uint32_t* Bp, const uint32_t* Bj, const void* Bx, uint32_t* Rp,
                uint32_t** Rj, void** Rx, uint32_t* h_nnz, uint32_t init_bits, hipFunction_t expand, hipFunction_t run, hipStream_t s) {
        tl_stream = s;
        if (a_rows == 0) {
            *Rj = nullptr, *Rx = nullptr, *h_nnz = 0;
            return SPLAHIP_OK;
        }
        return mxm_typed<uint32_t>(a_rows, b_cols, Ap, Aj, Ax, Bp, Bj, Bx, Rp, Rj, Rx, h_nnz, nullptr, SPLAHIP_FIRST, SPLAHIP_FIRST, init_bits, 1ull << 26, s, expand, run);
    }
}// namespace splahip

namespace splahip {
    void vxm_workspaces_trim();// vxm.hip: the small-frontier push form's per-thread accumulators
    void mxm_pinned_trim();// mxm_spa.hip: the pinned staging chunks of every host thread that is not inside a call
    int mxm_spa(int type, uint32_t a_rows, uint32_t b_cols, const uint32_t* Ap, const uint32_t* Aj, const void* Ax, const uint32_t* Bp, const uint32_t* Bj, const void* Bx, uint32_t* Rp,
                uint32_t** Rj, void** Rx, uint32_t* h_nnz, unsigned long long* h_products, int op_mult, int op_add, uint32_t init_bits, hipStream_t s);
}
using namespace splahip;

extern "C" {

int splahip_mxmT_masked(int type, uint32_t n_rows, uint32_t mask_nnz,
                        const uint32_t* d_Mp, const uint32_t* d_Mj, const void* d_Mx,
                        const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax,
                        const uint32_t* d_Bp, const uint32_t* d_Bj, const void* d_Bx,
                        uint32_t* d_Rp, uint32_t** d_Rj, void** d_Rx, uint32_t* h_nnz,
                        int op_mult, int op_add, int op_select, uint32_t init_bits, void* stream) {
    SPLAHIP_REQUIRE(d_Rp && d_Rj && d_Rx && h_nnz, "null output argument");
    SPLAHIP_REQUIRE(binop_valid_for(op_mult, type) && binop_valid_for(op_add, type), "op not defined for this type");
    SPLAHIP_REQUIRE(op_select >= 0 && op_select < SPLAHIP_SELOP_COUNT, "unknown op_select");
    SPLAHIP_REQUIRE(d_Mp && d_Ap && d_Bp, "null row pointer array");
    op_add = canonical_add(type, op_add, init_bits);// fp32 MIN / MAX from a NaN init: the reference's fold keeps init (ops.hpp)
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    tl_stream     = s;
    switch (type) {
        case SPLAHIP_INT: return mxmT_typed<int32_t>(n_rows, mask_nnz, d_Mp, d_Mj, d_Mx, d_Ap, d_Aj, d_Ax, d_Bp, d_Bj, d_Bx, d_Rp, d_Rj, d_Rx, h_nnz, op_mult, op_add, op_select, init_bits, s);
        case SPLAHIP_UINT: return mxmT_typed<uint32_t>(n_rows, mask_nnz, d_Mp, d_Mj, d_Mx, d_Ap, d_Aj, d_Ax, d_Bp, d_Bj, d_Bx, d_Rp, d_Rj, d_Rx, h_nnz, op_mult, op_add, op_select, init_bits, s);
        case SPLAHIP_FLOAT: return mxmT_typed<float>(n_rows, mask_nnz, d_Mp, d_Mj, d_Mx, d_Ap, d_Aj, d_Ax, d_Bp, d_Bj, d_Bx, d_Rp, d_Rj, d_Rx, h_nnz, op_mult, op_add, op_select, init_bits, s);
    }
    return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
}

int splahip_mxm(int type, uint32_t a_rows, uint32_t b_cols,
                const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax,
                const uint32_t* d_Bp, const uint32_t* d_Bj, const void* d_Bx,
                uint32_t* d_Rp, uint32_t** d_Rj, void** d_Rx, uint32_t* h_nnz, uint64_t* h_products,
                int op_mult, int op_add, uint32_t init_bits, uint64_t block_products, void* stream) {
    SPLAHIP_REQUIRE(d_Rp && d_Rj && d_Rx && h_nnz, "null output argument");
    SPLAHIP_REQUIRE(binop_valid_for(op_mult, type) && binop_valid_for(op_add, type), "op not defined for this type");
    SPLAHIP_REQUIRE(d_Ap && d_Bp, "null row pointer array");
    op_add = canonical_add(type, op_add, init_bits);// fp32 MIN / MAX from a NaN init: the reference's fold keeps init (ops.hpp)
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    tl_stream     = s;
    if (a_rows == 0) {
        *d_Rj = nullptr, *d_Rx = nullptr, *h_nnz = 0;
        return SPLAHIP_OK;
    }
    if (block_products == 0) block_products = 1ull << 27;// 128 Mi products per row block (~5 GiB of sort buffers)
    unsigned long long prod = 0;
    int                rc;
    const bool nan_init = type == SPLAHIP_FLOAT && (init_bits & 0x7fffffffu) > 0x7f800000u;// -> a dense result, see mxm_dense_fill_kernel
    if (nan_init && (uint64_t) a_rows * b_cols > 0xFFFFFFF0ull)
        return fail(SPLAHIP_NOT_IMPLEMENTED, "a NaN init makes the reference's result dense (%u x %u entries): not representable with 32-bit row pointers", a_rows, b_cols);
    // row classes with LDS accumulators (mxm_spa.hip) for every op_add that has an LDS atomic; -1: not applicable -> the ESC form below
    // (a deterministic call's fp32 sums: the ESC form folds every output entry's products in (row of A, entry of B) order)
    const bool fixed_order = deterministic_calls() && type == SPLAHIP_FLOAT && op_add == SPLAHIP_PLUS;
    rc = nan_init || fixed_order ? -1 : mxm_spa(type, a_rows, b_cols, d_Ap, d_Aj, d_Ax, d_Bp, d_Bj, d_Bx, d_Rp, d_Rj, d_Rx, h_nnz, &prod, op_mult, op_add, init_bits, s);
    if (rc != -1) {
        if (h_products) *h_products = prod;
        return rc;
    }
    switch (type) {
        case SPLAHIP_INT: rc = mxm_typed<int32_t>(a_rows, b_cols, d_Ap, d_Aj, d_Ax, d_Bp, d_Bj, d_Bx, d_Rp, d_Rj, d_Rx, h_nnz, &prod, op_mult, op_add, init_bits, block_products, s); break;
        case SPLAHIP_UINT: rc = mxm_typed<uint32_t>(a_rows, b_cols, d_Ap, d_Aj, d_Ax, d_Bp, d_Bj, d_Bx, d_Rp, d_Rj, d_Rx, h_nnz, &prod, op_mult, op_add, init_bits, block_products, s); break;
        case SPLAHIP_FLOAT: rc = mxm_typed<float>(a_rows, b_cols, d_Ap, d_Aj, d_Ax, d_Bp, d_Bj, d_Bx, d_Rp, d_Rj, d_Rx, h_nnz, &prod, op_mult, op_add, init_bits, block_products, s); break;
        default: return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
    }
    if (h_products) *h_products = prod;
    if (rc == SPLAHIP_OK && nan_init && b_cols != 0) {
        const size_t n = (size_t) a_rows * b_cols;
        DevBuf       sp, dj, dx;
        hipError_t   e = sp.alloc(sizeof(uint) * ((size_t) a_rows + 1));
        if (e == hipSuccess) e = dj.alloc_plain(sizeof(uint) * n);
        if (e == hipSuccess) e = dx.alloc_plain(sizeof(uint) * n);
        if (e == hipSuccess) e = hipMemcpyAsync(sp.p, d_Rp, sizeof(uint) * ((size_t) a_rows + 1), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(mxm_dense_fill_kernel, dim3((uint) ((std::max<size_t>(n, a_rows + 1) + 255) / 256)), dim3(256), 0, s, a_rows, b_cols, init_bits, d_Rp,
                               dj.as<uint>(), dx.as<uint>());
            if (*h_nnz)
                hipLaunchKernelGGL(mxm_dense_scatter_kernel, dim3((a_rows + 3) / 4), dim3(256), 0, s, a_rows, b_cols, sp.as<uint>(), *d_Rj,
                                   static_cast<const uint*>(*d_Rx), dx.as<uint>());
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (*d_Rj) (void) splahip_free(*d_Rj);
        if (*d_Rx) (void) splahip_free(*d_Rx);
        *d_Rj = nullptr, *d_Rx = nullptr, *h_nnz = 0;
        if (e != hipSuccess) return fail(SPLAHIP_ERROR, "mxm (dense result of a NaN init) failed: %s", hipGetErrorString(e));
        *d_Rj  = static_cast<uint32_t*>(dj.release());
        *d_Rx  = dx.release();
        *h_nnz = (uint32_t) n;
    }
    return rc;
}

int splahip_mxm_trim(void) {
    DevBuf::trim();
    scratch_trim();
    stream_caches_trim();
    mxm_pinned_trim();
    vxm_workspaces_trim();
    handoff_trim();
    (void) hipDeviceSynchronize();// (what the pool still holds for queued work stays; everything else goes back)
    pool_trim();
    return SPLAHIP_OK;
}
}
