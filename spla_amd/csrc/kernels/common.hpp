// common.hpp — error plumbing shared by the .hip translation units of libspla_amd.so
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../../include/spla_hip.h"

namespace splahip {

    // thread-local last error text, read by splahip_last_error()
    std::string& last_error();
    int          fail(int status, const char* fmt, ...);

#define SPLAHIP_TRY(expr)                                                                                       \
    do {                                                                                                        \
        hipError_t _e = (expr);                                                                                 \
        if (_e != hipSuccess)                                                                                   \
            return ::splahip::fail(SPLAHIP_ERROR, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define SPLAHIP_REQUIRE(cond, msg)                                                                              \
    do {                                                                                                        \
        if (!(cond)) return ::splahip::fail(SPLAHIP_INVALID_ARGUMENT, "invalid argument: %s (%s:%d)", msg, __FILE__, __LINE__); \
    } while (0)

    inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

    // An environment knob (SPLAHIP_...: A/B switches and tuning values for tests and experiments), read from the process environment ONCE and cached
    // (runtime.hip): a call's dispatch costs a hash lookup, not a scan of the environment.  splahip_env_refresh() drops the cache (tests change knobs mid-process).
    // A few result words of a call brought to the host: a one-wave kernel writes them into pinned (mapped) memory of the calling host thread and the host POLLS
    // for them (runtime.hip: at most SPLAHIP_MAILBOX_POLL_US, then it synchronises the stream).  hipMemcpyAsync to a pageable destination costs a staged blit (~15 us
    // per call in the kernel trace of spla_Algorithm_bfs, two per level), and hipStreamSynchronize returns ~10 us after the last kernel; this is one small launch
    // and a spin on host memory.  n <= 16 words; h_dst any host memory.  Returns a hipError_t.
    hipError_t fetch_words(void* h_dst, const void* d_src, uint32_t n_words, hipStream_t stream);
    // The mailbox itself, for a kernel that posts its words on its own: slot k = sequence number << 32 | word, ONE 8-byte store per word — the host takes a slot as
    // valid when it carries the number it waits for, so no fence orders the slots (a system-scope release in the middle of a kernel would first write the L2's dirty
    // lines back).  mailbox_begin: the calling thread's mailbox and the number to post with; mailbox_end: wait for n words (poll, then synchronise) and copy them out.
    struct MailboxTicket {
        unsigned long long* dev = nullptr;// device view of the calling thread's 16 slots
        uint32_t            seq = 0;
    };
    hipError_t mailbox_begin(MailboxTicket* t);
    hipError_t mailbox_end(void* h_dst, uint32_t n_words, const MailboxTicket& t, hipStream_t stream);
    __device__ __forceinline__ void mailbox_store(unsigned long long* mailbox, uint32_t slot, uint32_t mail_seq, uint32_t word) {
        __hip_atomic_store(mailbox + slot, ((unsigned long long) mail_seq << 32) | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const char* knob_env(const char* name);
    // the calling host thread asked for run-to-run identical fp32 sums (splahip_set_deterministic; the Descriptor's deterministic flag): the forms whose
    // fp32 PLUS merges partial sums in arrival order (column-ordered blocks, the LDS row classes of mxm) stand aside for the fixed-order kernels
    bool deterministic_calls();
    int  deterministic_exchange(int on);// returns the previous setting
    // the calling host thread asked for the REFERENCE'S fold order (splahip_set_reference_order; the Descriptor's reference_order flag): an fp32 PLUS fold of
    // mxv / vxm-pull then adds every row's products strictly left to right from `init` in entry order — the reference's own loop, src/cpu/cpu_mxv.hpp:94-99 —
    // and the result is bit-identical to the reference CPU backend.  Implies deterministic_calls().
    bool reference_order_calls();
    int  reference_order_exchange(int on);// returns the previous setting
    void        knob_cache_clear();

    // make sure a device has been selected and is a gfx950; returns SPLAHIP_OK or an error status
    int ensure_device();

    constexpr unsigned WAVE = 64;

    // Result arrays that a call hands to its caller (mxm / mxmT) are plain hipMalloc allocations — the stream-ordered pool is erratic
    // for multi-GB buffers — tracked (and cached when returned) in runtime.hip; splahip_free recognises them.
    // Every cache of idle device memory this library keeps (returned hand-off blocks, mxm's grow-only workspaces, the stream-ordered pool's
    // free list) is given back to the driver when an allocation reports out-of-memory, and the allocation is tried once more: in a process that
    // shares the GPU with another allocator (torch's caching allocator in bench.py / the tests / the N > 1 path) tens of GB could otherwise sit
    // idle here while a later request fails.  Translation units with a cache of their own register a trim callback.
    void       release_idle_memory();
    void       register_trim(void (*fn)());
    hipError_t device_malloc(void** p, size_t bytes);                           // hipMalloc with release-and-retry
    hipError_t device_malloc_async(void** p, size_t bytes, hipStream_t stream);// hipMallocAsync with release-and-retry
    hipError_t handoff_alloc(void** p, size_t bytes);
    bool       handoff_release(void* p);// true: p was a hand-off allocation and has been taken back (cached or freed)
    void       handoff_trim();          // free the cached hand-off blocks

    // SCRATCH ARENA of the blocking calls (mxm, mxmT): device blocks of the calling host thread, kept between calls and handed out by bumping an offset.  A call
    // opens a ScratchScope, takes its ~20 workspaces with scratch_take and gives them all back when the scope ends (the offset returns to where the scope found
    // it; scopes nest; the destructor synchronises the stream, so nothing queued still uses the memory).  Why: hipFreeAsync costs the host 25 us for a small buffer
    // and 140 - 170 us for one of >= 16 MB (tools/exp/exp_r5_alloc_cost.py) — 1.5 - 3 ms per call with the GPU idle.  The arena grows by whole blocks and is
    // melted into one block when the outermost scope ends; scratch_trim() (also run when any allocation of the library runs out of memory, and by
    // splahip_mxm_trim) frees it.  SPLAHIP_SCRATCH_ARENA=0: scratch_active() stays false and the callers use the stream-ordered pool.
    struct ScratchScope {
        explicit ScratchScope(hipStream_t stream);
        ~ScratchScope();
        ScratchScope(const ScratchScope&)            = delete;
        ScratchScope& operator=(const ScratchScope&) = delete;

      private:
        hipStream_t s_;
        size_t      block_, off_, held_;
        bool        on_;
    };
    bool       scratch_active();                    // inside a ScratchScope of this thread (and not paused)
    hipError_t scratch_take(void** p, size_t bytes);// 256-byte aligned; valid until the innermost open scope ends
    void       scratch_trim();                      // frees the arena of every host thread that has no scope open (and those of threads that have exited)
    void       stream_caches_trim();                // frees every host thread's cache of released stream-ordered blocks (below)
    void       pool_trim();                         // gives the stream-ordered pool's unused blocks back to the driver
    struct ScratchPause {                           // a nested call that allocates in a loop keeps to the pool
        ScratchPause();
        ~ScratchPause();
    };

    // Scoped, STREAM-ORDERED scratch memory (hipMallocAsync / hipFreeAsync on the call's stream): taken from the device's
    // default memory pool, whose release threshold runtime.hip raises so that steady-state calls neither reach the driver
    // nor synchronise the device the way hipMalloc / hipFree do.  Everything that touches the buffer must run on `stream`.
    // Between the pool and the caller sits a small per-thread cache of released blocks, keyed by stream (stream_block_alloc / stream_block_free, runtime.hip): a
    // block released on stream s is handed to the next request of a similar size ON THE SAME STREAM — ordered behind its previous user exactly as the pool's own
    // reuse is — without going through hipFreeAsync, which costs the host 25 us for a small block and 140 - 170 us for one of >= 16 MB
    // (tools/exp/exp_r5_alloc_cost.py; a push step of 0.8 ms released 32 MB per call).  At most 24 blocks / 4 GiB per thread; SPLAHIP_STREAM_CACHE=0: off.
    hipError_t stream_block_alloc(void** p, size_t bytes, hipStream_t stream);
    void       stream_block_free(void* p, hipStream_t stream);
    struct DevMem {
        void*       p = nullptr;
        hipStream_t s = nullptr;
        DevMem()               = default;
        DevMem(const DevMem&)  = delete;
        DevMem& operator=(const DevMem&) = delete;
        ~DevMem() { release(); }
        void release() {
            if (p) stream_block_free(p, s);
            p = nullptr;
        }
        hipError_t alloc(hipStream_t stream, size_t bytes) {
            release();
            s = stream;
            return stream_block_alloc(&p, bytes ? bytes : 4, stream);
        }
        template<typename U>
        U* as() { return static_cast<U*>(p); }
    };

}// namespace splahip
