// runtime.hip — device selection, memory helpers and error reporting of the kernel-level C ABI (include/spla_hip.h).
// Replaces what the reference's OpenCL accelerator object does for its kernels: device pick and buffer
// create/read/write (reference src/opencl/cl_accelerator.cpp:60-190, src/opencl/cl_format_dense_vec.hpp:40-88).
#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

namespace splahip {

    // cached environment knobs (common.hpp: knob_env).  The returned pointers stay valid for the life of the process: a value is INTERNED (a set of strings
    // that only grows), the cache maps a knob's name to its interned value, and a refresh re-reads the environment into new map entries without freeing any
    // string — a host thread inside a dispatch may still hold a pointer it was given before another thread refreshed.
    namespace {
        std::mutex& knob_mutex() {
            static std::mutex m;
            return m;
        }
        std::unordered_map<std::string, const char*>& knob_map() {
            static std::unordered_map<std::string, const char*> m;
            return m;
        }
        const char* knob_intern(const char* value) {
            static std::unordered_map<std::string, std::unique_ptr<std::string>> pool;// (node-based and never erased: the strings do not move)
            auto it = pool.find(value);
            if (it == pool.end()) it = pool.emplace(value, std::make_unique<std::string>(value)).first;
            return it->second->c_str();
        }
    }// namespace
    const char* knob_env(const char* name) {
        std::lock_guard<std::mutex> lock(knob_mutex());
        auto&                       m  = knob_map();
        auto                        it = m.find(name);
        if (it == m.end()) {
            const char* e = std::getenv(name);
            it            = m.emplace(name, e ? knob_intern(e) : nullptr).first;
        }
        return it->second;
    }
    namespace {
        thread_local int tl_deterministic   = 0;
        thread_local int tl_reference_order = 0;
    }
    bool deterministic_calls() { return tl_deterministic != 0 || tl_reference_order != 0; }
    bool reference_order_calls() { return tl_reference_order != 0; }
    int  reference_order_exchange(int on) {
        const int before   = tl_reference_order;
        tl_reference_order = on ? 1 : 0;
        return before;
    }
    int  deterministic_exchange(int on) {
        const int before = tl_deterministic;
        tl_deterministic = on ? 1 : 0;
        return before;
    }
    namespace {
        struct PokeWords {
            uint32_t w[16];
        };
        // up to 16 words handed over as kernel ARGUMENTS: no staging copy, no wait — the launch has read them when it returns
        __global__ __launch_bounds__(64) void poke_kernel(uint32_t* __restrict__ dst, PokeWords p, uint32_t n) {
            uint32_t x = 0;
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) x = threadIdx.x == k ? p.w[k] : x;
            if (threadIdx.x < n) dst[threadIdx.x] = x;
        }
        // words [0, n) of src into the host's mailbox slots, each with the call's sequence number (common.hpp: mailbox_store)
        __global__ __launch_bounds__(64) void mailbox_kernel(const uint32_t* __restrict__ src, unsigned long long* __restrict__ dst, uint32_t n, uint32_t seq) {
            if (threadIdx.x < n) mailbox_store(dst, threadIdx.x, seq, src[threadIdx.x]);
        }
        struct Mailbox {// (128 bytes of pinned memory per host thread, kept for the life of the process: a thread_local's destructor may run after the HIP runtime's)
            unsigned long long* host = nullptr;
            unsigned long long* dev  = nullptr;
            uint32_t            seq  = 0;
        };
    }// namespace
    // A few words of device memory for a host decision (a frontier's size, a result's entry count): a one-wave kernel copies them into mapped host memory.  The host
    // does not wait for the STREAM (hipStreamSynchronize returns ~10 us after the last kernel: a barrier packet, its completion signal, the runtime's bookkeeping) but
    // for the WORDS: it polls the sequence number the kernel writes behind them, for at most SPLAHIP_MAILBOX_POLL_US (default 100; 0 = always synchronise) — the case of a
    // traversal level's few short kernels — and synchronises the stream if they have not arrived by then (a long product: no point in burning a core).
    namespace {
        Mailbox& thread_mailbox() {
            static thread_local Mailbox box;
            return box;
        }
    }// namespace
    hipError_t mailbox_begin(MailboxTicket* t) {
        Mailbox& box = thread_mailbox();
        if (!box.host) {
            hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&box.host), 16 * sizeof(unsigned long long), hipHostMallocMapped);
            if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&box.dev), box.host, 0);
            if (e != hipSuccess) {
                box.host = nullptr;
                return e;
            }
            for (int k = 0; k < 16; ++k) box.host[k] = 0;
        }
        t->seq = ++box.seq ? box.seq : ++box.seq;
        t->dev = box.dev;
        return hipSuccess;
    }
    hipError_t mailbox_end(void* h_dst, uint32_t n_words, const MailboxTicket& t, hipStream_t stream) {
        Mailbox&       box     = thread_mailbox();
        const char*    pe      = knob_env("SPLAHIP_MAILBOX_POLL_US");
        const unsigned poll_us = pe ? (unsigned) std::strtoul(pe, nullptr, 10) : 100u;
        uint32_t*      out     = static_cast<uint32_t*>(h_dst);
        // slot k is there when it carries this call's number; the slots arrive in any order
        auto arrived = [&]() {
            for (uint32_t k = 0; k < n_words; ++k) {
                const unsigned long long w = __atomic_load_n(box.host + k, __ATOMIC_ACQUIRE);
                if ((uint32_t) (w >> 32) != t.seq) return false;
                out[k] = (uint32_t) w;
            }
            return true;
        };
        if (poll_us) {
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(poll_us);
            for (;;) {
                for (int k = 0; k < 64; ++k)
                    if (arrived()) return hipSuccess;
                if (std::chrono::steady_clock::now() >= t_end) break;
            }
        }
        const hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        return arrived() ? hipSuccess : hipErrorUnknown;// (not there after the stream has drained: the launch that was to post them did not run)
    }
    hipError_t fetch_words(void* h_dst, const void* d_src, uint32_t n_words, hipStream_t stream) {
        if (n_words == 0) return hipStreamSynchronize(stream);
        if (n_words > 16) return hipErrorInvalidValue;
        MailboxTicket t;
        hipError_t    e = mailbox_begin(&t);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(64), 0, stream, static_cast<const uint32_t*>(d_src), t.dev, n_words, t.seq);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        return mailbox_end(h_dst, n_words, t, stream);
    }
    void knob_cache_clear() {
        std::lock_guard<std::mutex> lock(knob_mutex());
        knob_map().clear();
    }

    std::string& last_error() {
        static thread_local std::string text;
        return text;
    }

    int fail(int status, const char* fmt, ...) {
        char    buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        last_error() = buf;
        return status;
    }

    namespace {
        std::mutex g_lock;
        int        g_device = -1;// device selected through splahip_init (or adopted from the current HIP context)
        char       g_arch[256];
        char       g_name[256];
    }// namespace

    static void handoff_cache_limit(size_t bytes);
    static int  select_device(int device) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
            (void) hipGetLastError();
            return fail(SPLAHIP_NO_DEVICE, "no HIP device visible: the MI355X backend has no CPU fallback");
        }
        if (device < 0 || device >= count) return fail(SPLAHIP_NO_DEVICE, "device index %d out of range [0,%d)", device, count);
        hipDeviceProp_t prop;
        SPLAHIP_TRY(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(SPLAHIP_NO_DEVICE, "device %d is %s; this library only carries gfx950 (MI355X) code objects", device, prop.gcnArchName);
        SPLAHIP_TRY(hipSetDevice(device));
        {// keep freed scratch (common.hpp DevMem) in the default pool instead of handing it back at every synchronisation
            hipMemPool_t pool = nullptr;
            if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess && pool) {
                uint64_t keep = ~0ull;
                (void) hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            } else {
                (void) hipGetLastError();
            }
        }
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b) handoff_cache_limit(free_b / 4);
            else
                (void) hipGetLastError();
        }
        g_device = device;
        std::snprintf(g_arch, sizeof(g_arch), "%s", prop.gcnArchName);
        std::snprintf(g_name, sizeof(g_name), "%s", prop.name);
        return SPLAHIP_OK;
    }

    // Hand-off allocations (result arrays of mxm / mxmT, multi-GB): plain hipMalloc blocks with a small cache of returned blocks, so
    // that a steady-state loop (call, consume, splahip_free, call again) neither reaches the driver nor depends on the stream-ordered
    // pool, whose latency for buffers of this size is erratic.  A returned block is reused for a request of at least half
    // its size; at most kHandoffCacheBytes stay cached (splahip_mxm_trim() drops them).
    namespace {
        std::mutex                 g_trim_lock;
        std::vector<void (*)()>    g_trim_fns;
    }// namespace
    void register_trim(void (*fn)()) {
        std::lock_guard<std::mutex> guard(g_trim_lock);
        for (auto f : g_trim_fns)
            if (f == fn) return;
        g_trim_fns.push_back(fn);
    }
    void pool_trim() {// the stream-ordered pool's unused blocks (its release threshold is raised at start-up: it gives nothing back on its own)
        int          dev  = 0;
        hipMemPool_t pool = nullptr;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) (void) hipMemPoolTrimTo(pool, 0);
        (void) hipGetLastError();
    }
    void release_idle_memory() {
        (void) hipDeviceSynchronize();// nothing queued may still use a block that goes back to the driver
        handoff_trim();
        std::vector<void (*)()> fns;
        {
            std::lock_guard<std::mutex> guard(g_trim_lock);
            fns = g_trim_fns;
        }
        for (auto f : fns) f();
        pool_trim();
    }
    hipError_t device_malloc(void** p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipErrorOutOfMemory) {
            (void) hipGetLastError();
            release_idle_memory();
            e = hipMalloc(p, bytes);
        }
        return e;
    }
    hipError_t device_malloc_async(void** p, size_t bytes, hipStream_t stream) {
        hipError_t e = hipMallocAsync(p, bytes, stream);
        if (e == hipErrorOutOfMemory) {
            (void) hipGetLastError();
            release_idle_memory();
            e = hipMallocAsync(p, bytes, stream);
        }
        return e;
    }

    // ---- per-thread cache of released stream-ordered blocks (common.hpp: DevMem) ---------------------------------------------------------------------------
    namespace {
        struct StreamBlock {
            void*       p;
            size_t      cap;
            hipStream_t s;
        };
        // Every host thread owns one cache; all of them are listed in a process-wide registry so that an out-of-memory retry (release_idle_memory) and
        // splahip_mxm_trim can give back the blocks of EVERY thread, and so that the blocks of a thread that has exited are not lost: the thread_local holder's
        // destructor only marks its cache an orphan (no HIP call from a TLS destructor: it may run after the runtime's), the next trim frees and deletes it.
        // (the caches were thread_locals created with `new`, trimmed only by their own thread and never destroyed.)
        struct StreamCache {
            std::mutex                        m;// the owner's calls and another thread's trim
            std::vector<StreamBlock>          free;// oldest first
            std::unordered_map<void*, size_t> live;// handed out: capacity
            size_t                            free_bytes = 0;
            bool                              orphan     = false;
        };
        template<class C>
        struct CacheRegistry {
            std::mutex      m;
            std::vector<C*> all;
        };
        CacheRegistry<StreamCache>& stream_registry() {
            static CacheRegistry<StreamCache>* r = new CacheRegistry<StreamCache>;// (never destroyed: a late thread may still register)
            return *r;
        }
        struct StreamCacheHolder {
            StreamCache* c;
            StreamCacheHolder() : c(new StreamCache) {
                auto& r = stream_registry();
                std::lock_guard<std::mutex> g(r.m);
                r.all.push_back(c);
            }
            ~StreamCacheHolder() {
                std::lock_guard<std::mutex> g(c->m);
                c->orphan = true;
            }
        };
        StreamCache& stream_cache() {
            static thread_local StreamCacheHolder h;
            return *h.c;
        }
        constexpr size_t kStreamCacheBlocks = 24, kStreamCacheBytes = size_t(4) << 30, kStreamCacheBlockMax = size_t(1) << 30;
        bool stream_cache_enabled() {// (process-start-only knob: INTEGRATION.md)
            static const bool on = [] {
                const char* e = knob_env("SPLAHIP_STREAM_CACHE");
                return !(e && e[0] == '0');
            }();
            return on;
        }
        // every thread's released blocks go back to the driver (run by release_idle_memory behind a device synchronisation, and by splahip_mxm_trim); the caches
        // of threads that have exited are deleted
        void stream_cache_trim() {
            auto& r = stream_registry();
            std::lock_guard<std::mutex> rg(r.m);
            for (size_t k = 0; k < r.all.size();) {
                StreamCache* c    = r.all[k];
                bool         drop = false;
                {
                    std::lock_guard<std::mutex> g(c->m);
                    for (StreamBlock& b : c->free) (void) hipFree(b.p);// (hipFree, not hipFreeAsync(b.s): the stream a block was released on may be gone by now)
                    c->free.clear();
                    c->free_bytes = 0;
                    drop          = c->orphan && c->live.empty();
                }
                if (drop) {
                    delete c;
                    r.all.erase(r.all.begin() + (long) k);
                } else
                    ++k;
            }
        }
    }// namespace
    void stream_caches_trim() { stream_cache_trim(); }
    hipError_t stream_block_alloc(void** p, size_t bytes, hipStream_t stream) {
        bytes = (std::max<size_t>(bytes, 4) + 255) & ~size_t(255);
        if (!stream_cache_enabled()) return device_malloc_async(p, bytes, stream);
        StreamCache& c = stream_cache();
        {
            std::lock_guard<std::mutex> g(c.m);
            size_t                      best = c.free.size();
            for (size_t k = 0; k < c.free.size(); ++k) {
                const StreamBlock& b = c.free[k];
                if (b.s == stream && b.cap >= bytes && b.cap <= 4 * bytes + (size_t(1) << 20) && (best == c.free.size() || b.cap < c.free[best].cap)) best = k;
            }
            if (best != c.free.size()) {
                *p = c.free[best].p;
                c.live[*p] = c.free[best].cap;
                c.free_bytes -= c.free[best].cap;
                c.free.erase(c.free.begin() + (long) best);
                return hipSuccess;
            }
        }
        const hipError_t e = device_malloc_async(p, bytes, stream);// (outside the cache's lock: an out-of-memory retry trims every cache, this one included)
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> g(c.m);
            c.live[*p] = bytes;
        }
        return e;
    }
    void stream_block_free(void* p, hipStream_t stream) {
        if (!p) return;
        if (!stream_cache_enabled()) {
            (void) hipFreeAsync(p, stream);
            return;
        }
        StreamCache& c = stream_cache();
        register_trim(&stream_cache_trim);// (idempotent)
        std::lock_guard<std::mutex> g(c.m);
        const auto                  it = c.live.find(p);
        if (it == c.live.end() || it->second > kStreamCacheBlockMax) {// (not ours, or too large to keep)
            if (it != c.live.end()) c.live.erase(it);
            (void) hipFreeAsync(p, stream);
            return;
        }
        const size_t cap = it->second;
        c.live.erase(it);
        c.free.push_back({p, cap, stream});
        c.free_bytes += cap;
        while (c.free.size() > kStreamCacheBlocks || c.free_bytes > kStreamCacheBytes) {// the oldest blocks go back to the pool (rare in steady state; hipFree waits for the
            (void) hipFree(c.free.front().p);                                            // device: the stream the block was released on may no longer exist)
            c.free_bytes -= c.free.front().cap;
            c.free.erase(c.free.begin());
        }
    }

    // ---- scratch arena (common.hpp) ----------------------------------------------------------------------------------------------------------------------
    namespace {
        struct ArenaBlock {
            unsigned char* p;
            size_t         cap;
        };
        struct Arena {
            std::mutex              m;// the owner's takes and another thread's trim (the owner's scope bookkeeping — depth, cur, off — is its own)
            std::vector<ArenaBlock> blocks;
            size_t                  cur = 0, off = 0;// the next byte handed out: blocks[cur].p + off
            int                     depth = 0, paused = 0;
            size_t                  held = 0, peak = 0;// bytes handed out and not yet returned; the most held at once since the last trim
            bool                    orphan = false;
        };
        CacheRegistry<Arena>& arena_registry() {
            static CacheRegistry<Arena>* r = new CacheRegistry<Arena>;
            return *r;
        }
        struct ArenaHolder {
            Arena* a;
            ArenaHolder() : a(new Arena) {
                auto& r = arena_registry();
                std::lock_guard<std::mutex> g(r.m);
                r.all.push_back(a);
            }
            ~ArenaHolder() {
                std::lock_guard<std::mutex> g(a->m);
                a->orphan = true;
            }
        };
        Arena& scratch_arena() {
            static thread_local ArenaHolder h;// (the holder's destructor makes no HIP call; the arena of a thread that has exited is freed by the next trim)
            return *h.a;
        }
        bool scratch_enabled() {// (process-start-only knob: INTEGRATION.md)
            static const bool on = [] {
                const char* e = knob_env("SPLAHIP_SCRATCH_ARENA");
                return !(e && e[0] == '0');
            }();
            return on;
        }
        void arena_free_blocks(Arena& a) {// (a.m held; no scope open)
            for (ArenaBlock& b : a.blocks)
                if (b.p) (void) hipFree(b.p);
            a.blocks.clear();
            a.cur = a.off = 0;
        }
    }// namespace
    bool scratch_active() {
        const Arena& a = scratch_arena();
        return a.depth > 0 && a.paused == 0;
    }
    // Every thread's arena that has no scope open goes back to the driver — the caller's own, other threads' (an out-of-memory retry on one thread must be able to
    // reclaim what idle threads hold) and those of threads that have exited (deleted).  The remembered peak is forgotten too: the call after a trim sizes its first
    // block from what IT needs, not from the largest call the thread ever made.
    void scratch_trim() {
        auto& r = arena_registry();
        std::lock_guard<std::mutex> rg(r.m);
        for (size_t k = 0; k < r.all.size();) {
            Arena* a    = r.all[k];
            bool   drop = false;
            {
                std::lock_guard<std::mutex> g(a->m);
                if (a->depth <= 0) {
                    arena_free_blocks(*a);
                    a->peak = 0;
                    drop    = a->orphan;
                }
            }
            if (drop) {
                delete a;
                r.all.erase(r.all.begin() + (long) k);
            } else
                ++k;
        }
    }
    hipError_t scratch_take(void** p, size_t bytes) {
        Arena& a = scratch_arena();
        *p       = nullptr;
        if (a.depth <= 0) return hipErrorInvalidValue;
        bytes = (std::max<size_t>(bytes, 4) + 255) & ~size_t(255);
        for (;;) {
            size_t cap = 0;
            {
                std::lock_guard<std::mutex> g(a.m);
                if (a.cur < a.blocks.size() && a.blocks[a.cur].cap - a.off >= bytes) {
                    *p = a.blocks[a.cur].p + a.off;
                    a.off += bytes;
                    a.held += bytes;
                    a.peak = std::max(a.peak, a.held);
                    return hipSuccess;
                }
                if (a.cur + 1 < a.blocks.size()) {// the next block the arena already owns
                    ++a.cur;
                    a.off = 0;
                    continue;
                }
                // a new block: at least what the largest call since the last trim held (so a steady-state call fits one block), at least 32 MB
                cap = std::max({bytes, a.peak + a.peak / 8, size_t(32) << 20});
            }
            // (allocated outside the arena's lock: an out-of-memory retry trims every idle arena.)  The generous size first, plainly; then just what is asked for,
            // plainly; only then the release-everything retry — a failed `peak`-sized request must not cost a device synchronisation and every cache of the
            // library when `bytes` would have fit
            void* q = nullptr;
            register_trim(&scratch_trim);// (idempotent)
            size_t     got = cap;
            hipError_t e   = hipMalloc(&q, cap);
            if (e != hipSuccess && cap > bytes) (void) hipGetLastError(), got = bytes, e = hipMalloc(&q, bytes);
            if (e != hipSuccess) (void) hipGetLastError(), got = bytes, e = device_malloc(&q, bytes);
            if (e != hipSuccess) return e;
            std::lock_guard<std::mutex> g(a.m);
            a.blocks.push_back({static_cast<unsigned char*>(q), got});
            a.cur = a.blocks.size() - 1;
            a.off = 0;
        }
    }
    ScratchScope::ScratchScope(hipStream_t stream) : s_(stream), block_(0), off_(0), held_(0), on_(scratch_enabled()) {
        if (!on_) return;
        Arena& a = scratch_arena();
        std::lock_guard<std::mutex> g(a.m);
        block_ = a.cur, off_ = a.off, held_ = a.held;
        ++a.depth;
    }
    ScratchScope::~ScratchScope() {
        if (!on_) return;
        Arena& a = scratch_arena();
        (void) hipStreamSynchronize(s_);// nothing queued may still use what goes back
        std::lock_guard<std::mutex> g(a.m);
        a.cur = block_, a.off = off_, a.held = held_;
        if (--a.depth == 0) {
            a.cur = a.off = a.held = 0;
            if (a.blocks.size() > 1) arena_free_blocks(a);// melted into one block of `peak` bytes by the next call's first take
        }
    }
    ScratchPause::ScratchPause() { ++scratch_arena().paused; }
    ScratchPause::~ScratchPause() { --scratch_arena().paused; }

    namespace {
        // returned hand-off blocks kept for reuse: at most a quarter of the memory that was free when the device was selected, and at most 48 GiB
        size_t                            kHandoffCacheBytes = size_t(48) << 30;
        std::mutex                        g_handoff_lock;
        std::unordered_map<void*, size_t> g_handoff_live;  // handed out: capacity
        std::vector<std::pair<void*, size_t>> g_handoff_free;// returned, reusable
        size_t                            g_handoff_free_bytes = 0;
    }// namespace
    static void handoff_cache_limit(size_t bytes) { kHandoffCacheBytes = std::min(kHandoffCacheBytes, bytes); }
    hipError_t handoff_alloc(void** p, size_t bytes) {
        if (bytes == 0) bytes = 4;
        {
            std::lock_guard<std::mutex> guard(g_handoff_lock);
            size_t best = g_handoff_free.size();
            for (size_t k = 0; k < g_handoff_free.size(); ++k) {
                const size_t cap = g_handoff_free[k].second;
                if (cap >= bytes && cap / 2 <= bytes && (best == g_handoff_free.size() || cap < g_handoff_free[best].second)) best = k;
            }
            if (best != g_handoff_free.size()) {
                *p = g_handoff_free[best].first;
                g_handoff_live[*p] = g_handoff_free[best].second;
                g_handoff_free_bytes -= g_handoff_free[best].second;
                g_handoff_free.erase(g_handoff_free.begin() + (long) best);
                return hipSuccess;
            }
        }
        hipError_t e = device_malloc(p, bytes);
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> guard(g_handoff_lock);
            g_handoff_live[*p] = bytes;
        }
        return e;
    }
    bool handoff_release(void* p) {
        size_t cap = 0;
        {
            std::lock_guard<std::mutex> guard(g_handoff_lock);
            auto                        it = g_handoff_live.find(p);
            if (it == g_handoff_live.end()) return false;
            cap = it->second;
            g_handoff_live.erase(it);
        }
        (void) hipDeviceSynchronize();// like hipFree: nothing queued may still read the block when it is reused
        std::lock_guard<std::mutex> guard(g_handoff_lock);
        if (g_handoff_free_bytes + cap <= kHandoffCacheBytes) {
            g_handoff_free.emplace_back(p, cap);
            g_handoff_free_bytes += cap;
        } else {
            (void) hipFree(p);
        }
        return true;
    }
    void handoff_trim() {
        std::lock_guard<std::mutex> guard(g_handoff_lock);
        for (auto& b : g_handoff_free) (void) hipFree(b.first);
        g_handoff_free.clear();
        g_handoff_free_bytes = 0;
    }

    int ensure_device() {
        std::lock_guard<std::mutex> guard(g_lock);
        if (g_device >= 0) return SPLAHIP_OK;
        // adopt whatever device the process already made current (e.g. torch.cuda.set_device(LOCAL_RANK))
        int current = 0;
        if (hipGetDevice(&current) != hipSuccess) {
            (void) hipGetLastError();
            return fail(SPLAHIP_NO_DEVICE, "no HIP device visible: the MI355X backend has no CPU fallback");
        }
        return select_device(current);
    }

}// namespace splahip

using namespace splahip;

extern "C" {

int splahip_version(void) { return 100; }

int splahip_set_deterministic(int on) { return splahip::deterministic_exchange(on); }
int splahip_set_reference_order(int on) { return splahip::reference_order_exchange(on); }

namespace {
    // a read-only stream the way the kernels of this library read theirs: 16 bytes per lane, non-temporal, a 1024-thread workgroup per CU walking a contiguous span,
    // four loads in flight per wavefront (tools/exp/stream_bench.hip, form A; profiles/r03_stream_bench.txt)
    typedef uint32_t stream_u4 __attribute__((ext_vector_type(4)));
    __global__ __launch_bounds__(1024) void read_stream_kernel(const stream_u4* __restrict__ src, size_t words_per_wg, uint32_t* __restrict__ out) {
        const unsigned   lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const stream_u4* base = src + (size_t) blockIdx.x * words_per_wg;
        const size_t     steps = words_per_wg / 1024;// 1 KB pieces per wavefront
        uint32_t         acc   = 0;
        for (size_t s = 0; s + 4 <= steps; s += 4) {
            stream_u4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = __builtin_nontemporal_load(base + ((s + u) * 16 + wave) * 64 + lane);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
        }
        if (acc == 0x9E3779B9u) out[blockIdx.x] = acc;// (keeps the loads; the buffer is zeroed, so nothing is stored)
    }
}// namespace

int splahip_read_stream_gbs(size_t bytes, int reps, double* gbs, void* stream) {
    SPLAHIP_REQUIRE(gbs != nullptr && reps > 0, "null argument");
    *gbs   = 0.0;
    int st = splahip::ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t  s    = splahip::as_stream(stream);
    const size_t wgs  = 256;
    const size_t per  = std::max<size_t>(bytes / 16 / wgs / 4096, 1) * 4096;// 16-byte words per workgroup: whole rounds of 4 x 1 KB per wavefront
    const size_t real = per * wgs * 16;
    void *       buf = nullptr, *out = nullptr;
    SPLAHIP_TRY(hipMalloc(&buf, real));
    hipError_t e = hipMalloc(&out, wgs * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(buf, 0, real, s);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float best = 0.0f;
    for (int k = 0; e == hipSuccess && k < reps + 2; ++k) {
        (void) hipEventRecord(e0, s);
        hipLaunchKernelGGL(read_stream_kernel, dim3((unsigned) wgs), dim3(1024), 0, s, static_cast<const stream_u4*>(buf), per, static_cast<uint32_t*>(out));
        (void) hipEventRecord(e1, s);
        e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (k >= 2 && ms > 0.0f && (best == 0.0f || ms < best)) best = ms;// (two warm-up launches; the fastest of the rest)
    }
    if (e0) (void) hipEventDestroy(e0);
    if (e1) (void) hipEventDestroy(e1);
    (void) hipFree(buf);
    if (out) (void) hipFree(out);
    if (e != hipSuccess) return splahip::fail(SPLAHIP_ERROR, "read stream: %s", hipGetErrorString(e));
    if (best > 0.0f) *gbs = (double) real / ((double) best * 1e-3) / 1e9;
    return SPLAHIP_OK;
}

int splahip_env_refresh(void) {
    splahip::knob_cache_clear();
    return SPLAHIP_OK;
}

int splahip_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void) hipGetLastError();
        return 0;
    }
    return count;
}

int splahip_init(int device) {
    std::lock_guard<std::mutex> guard(g_lock);
    return select_device(device);
}

int splahip_device_info(char* buffer, int length) {
    if (!buffer || length <= 0) return fail(SPLAHIP_INVALID_ARGUMENT, "null buffer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) {
        std::snprintf(buffer, (size_t) length, "no device");
        return st;
    }
    std::snprintf(buffer, (size_t) length, "HIP device %d: %s (%s)", g_device, g_name, g_arch);
    return SPLAHIP_OK;
}

const char* splahip_last_error(void) { return last_error().c_str(); }

int splahip_malloc(void** d_ptr, size_t bytes) {
    SPLAHIP_REQUIRE(d_ptr != nullptr, "d_ptr is null");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    *d_ptr = nullptr;
    // from the device's default memory pool, ordered on the default stream: hipMalloc / hipFree pairs synchronise the device and cost
    // ~0.1 ms each, which dominated short-lived objects (a BFS creates and drops a dozen vectors)
    SPLAHIP_TRY(device_malloc_async(d_ptr, bytes ? bytes : 4, nullptr));
    return SPLAHIP_OK;
}

int splahip_free(void* d_ptr) {
    if (!d_ptr) return SPLAHIP_OK;
    if (handoff_release(d_ptr)) return SPLAHIP_OK;// a result array of mxm / mxmT: plain allocation
    SPLAHIP_TRY(hipFreeAsync(d_ptr, nullptr));// reusable once the work queued on the default stream so far has finished
    return SPLAHIP_OK;
}

int splahip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream) {
    if (bytes == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_dst && h_src, "null pointer");
    // a few words (a traversal's source vertex, a scalar): a copy from pageable memory is a staging copy + a stream synchronisation, ~18 us; as kernel arguments ~5 us of
    // the host and nothing to wait for
    if (bytes <= sizeof(PokeWords) && bytes % 4 == 0 && reinterpret_cast<uintptr_t>(d_dst) % 4 == 0) {
        PokeWords p;
        std::memset(&p, 0, sizeof(p));
        std::memcpy(&p, h_src, bytes);
        hipLaunchKernelGGL(poke_kernel, dim3(1), dim3(64), 0, as_stream(stream), static_cast<uint32_t*>(d_dst), p, (uint32_t) (bytes / 4));
        SPLAHIP_TRY(hipGetLastError());
        return SPLAHIP_OK;
    }
    SPLAHIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    // pageable host memory may be reused by the caller right after return
    SPLAHIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return SPLAHIP_OK;
}

int splahip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream) {
    if (bytes == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(h_dst && d_src, "null pointer");
    SPLAHIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    SPLAHIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return SPLAHIP_OK;
}

int splahip_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream) {
    if (bytes == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_dst && d_src, "null pointer");
    SPLAHIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return SPLAHIP_OK;
}

int splahip_memset32(void* d_dst, uint32_t word, size_t count, void* stream) {
    if (count == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_dst, "null pointer");
    SPLAHIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_dst), (int) word, count, as_stream(stream)));
    return SPLAHIP_OK;
}

int splahip_stream_sync(void* stream) {
    SPLAHIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return SPLAHIP_OK;
}
}
