// dist.hip — the 1-D row-block decomposition of the masked SpMV behind the kernel-level C ABI (no reference counterpart:
// the reference is single-device).  One process per GPU; rank p owns a contiguous, nnz-balanced block of rows (its CSR slice, the
// matching slices of mask / r) and slice p of the input vector under the same cuts.  One exchange per product: the all-gather of v.
//
//   splahip_comm   an RCCL communicator over the ranks of the node (RCCL is loaded at run time, torch's copy first when the process has one:
//                  one RCCL per process).  all_gather: ncclAllGather on the padded layout [world][max_rows] (RCCL has no all-gatherv), or the
//                  DIRECT form — a grouped ncclSend / ncclRecv pair per peer — which on the full xGMI mesh moves every slice over its own link
//                  instead of relaying it around a ring (ring all-gather is per-link bound, ~7x slower inside a node).
//   splahip_shard  this rank's row block cut by COLUMN OWNER into a local part (columns of this rank's own slice: needs only v_local) and a
//                  remote part (everything else: needs the gathered vector), each with its plan and column-ordered-blocks copy.
//   splahip_dist_mxv_masked   exchange on the communicator's stream  ||  local part on the caller's stream; then the remote part folds onto r.
//                  r[i] = select(mask[i]) ? add(add(init, local fold), remote fold) : init — for an associative + commutative op_add the
//                  reference's result (cpu_mxv.hpp:88-103): bit-identical for int / uint, rounding-only for fp32 PLUS.  Semirings the blocks
//                  cannot serve (no LDS atomic, early exit) take splahip_shard_mxv_whole after the exchange: the unsplit block, any op.
#include "block_utils.hpp"
#include "common.hpp"
#include "ops.hpp"
#include "plan.hpp"

#include <rccl/rccl.h>
#include <rocprim/device/device_scan.hpp>

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

using namespace splahip;

struct splahip_comm_s {
    ncclComm_t  comm   = nullptr;
    int         world  = 1;
    int         rank   = 0;
    hipStream_t stream = nullptr;// the exchange runs here, beside the caller's compute stream
    hipEvent_t  ready  = nullptr;// input slice ready on the compute stream
    hipEvent_t  done   = nullptr;// gathered vector complete on the exchange stream
};

struct splahip_shard_s {
    int       world = 1, rank = 0;
    uint32_t  n_local = 0, max_rows = 0, nnz = 0, nnz_loc = 0, nnz_rem = 0;
    // the unsplit block (the caller's arrays, not owned) and its plan
    const uint32_t* Ap = nullptr;
    const uint32_t* Aj = nullptr;
    const void*     Ax = nullptr;
    splahip_plan    plan_whole = nullptr;
    // the two column parts (owned): local columns rebased to this rank's slice [0, max_rows), remote columns in the padded layout
    uint32_t *LAp = nullptr, *LAj = nullptr, *LAx = nullptr, *RAp = nullptr, *RAj = nullptr, *RAx = nullptr;
    splahip_plan plan_loc = nullptr, plan_rem = nullptr;
    bool         split_ok = false;// both parts carry a column-ordered-blocks copy (or are empty)
    bool         whole_form = false;// splahip_dist_mxv_masked prefers the UNSPLIT block over its own column-ordered blocks (exchange first, then one product)
};

// Sparsity-aware exchange of the input vector (a BFS frontier, the changed distances of an SSSP step): positions (+ values) of the entries that
// differ from the fill value travel instead of the whole slice.  Buffers of one rank's message: [count | m positions in the padded layout | m values],
// m = the caller's bound on any rank's count for this exchange (<= capacity).  Two receive buffers alternate: the one of the previous exchange lists
// the positions to clear.
struct splahip_sx_s {
    int       world = 1, rank = 0;
    uint32_t  max_rows = 0, capacity = 0;
    uint32_t* d_send    = nullptr;// 1 + 2 capacity words
    uint32_t* d_recv[2] = {nullptr, nullptr};// world x (1 + 2 capacity) words each
    uint32_t  prev_m = 0, prev_with_values = 0;
    int       cur = 0;   // buffer the next exchange receives into
    bool      has_prev = false;
    uint32_t* d_overflow = nullptr;// set by pack when this rank's count exceeded m
};

namespace {

    // ---- RCCL, loaded at run time ---------------------------------------------------------------------------------------------------------
    struct Rccl {
        void* lib = nullptr;
        ncclResult_t (*GetUniqueId)(ncclUniqueId*)                                                            = nullptr;
        ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int)                                     = nullptr;
        ncclResult_t (*CommDestroy)(ncclComm_t)                                                               = nullptr;
        ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)        = nullptr;
        ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)               = nullptr;
        ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                     = nullptr;
        ncclResult_t (*GroupStart)()                                                                          = nullptr;
        ncclResult_t (*GroupEnd)()                                                                            = nullptr;
        const char* (*GetErrorString)(ncclResult_t)                                                           = nullptr;
        bool ok = false;
    };
    Rccl& rccl() {
        static Rccl       r;
        static std::once_flag once;
        std::call_once(once, [] {
            const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : names)// a copy the process already holds (PyTorch-ROCm bundles one) comes first
                if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
            for (const char* n : names) {
                if (r.lib) break;
                r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            }
            if (!r.lib) return;
            auto sym = [&](const char* s) { return dlsym(r.lib, s); };
            r.GetUniqueId    = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
            r.CommInitRank   = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
            r.CommDestroy    = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
            r.AllGather      = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
            r.Send           = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
            r.Recv           = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
            r.GroupStart     = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
            r.GroupEnd       = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
            r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
        });
        return r;
    }
#define SPLAHIP_NCCL(expr)                                                                                                                  \
    do {                                                                                                                                    \
        ncclResult_t _r = (expr);                                                                                                           \
        if (_r != ncclSuccess)                                                                                                              \
            return fail(SPLAHIP_ERROR, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString ? rccl().GetErrorString(_r) : "?", __FILE__, __LINE__); \
    } while (0)

    // ---- column split of a row block ---------------------------------------------------------------------------------------------------------
    __global__ __launch_bounds__(256) void split_flag_kernel(uint32_t nnz, const uint32_t* __restrict__ Aj, uint32_t lo, uint32_t hi, uint32_t* __restrict__ flag) {
        const uint32_t k = blockIdx.x * 256 + threadIdx.x;
        if (k < nnz) {
            const uint32_t c = Aj[k];
            flag[k]          = (c >= lo && c < hi) ? 1u : 0u;
        }
    }
    // a stable partition: entry k goes to position pos[k] of the local part or k - pos[k] of the remote part (pos = exclusive scan of the flags)
    __global__ __launch_bounds__(256) void split_scatter_kernel(uint32_t nnz, const uint32_t* __restrict__ Aj, const uint32_t* __restrict__ Ax, const uint32_t* __restrict__ flag,
                                                                const uint32_t* __restrict__ pos, uint32_t lo, uint32_t* __restrict__ LAj, uint32_t* __restrict__ LAx,
                                                                uint32_t* __restrict__ RAj, uint32_t* __restrict__ RAx) {
        const uint32_t k = blockIdx.x * 256 + threadIdx.x;
        if (k >= nnz) return;
        const uint32_t p = pos[k];
        if (flag[k]) {
            LAj[p] = Aj[k] - lo;
            LAx[p] = Ax[k];
        } else {
            RAj[k - p] = Aj[k];
            RAx[k - p] = Ax[k];
        }
    }
    // row pointers of both parts from the scan: LAp[i] = pos[Ap[i]], RAp[i] = Ap[i] - LAp[i]  (pos has nnz + 1 entries)
    __global__ __launch_bounds__(256) void split_rowptr_kernel(uint32_t n_rows, const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ pos, uint32_t* __restrict__ LAp,
                                                               uint32_t* __restrict__ RAp) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i <= n_rows) {
            const uint32_t a = Ap[i], l = pos[a];
            LAp[i] = l;
            RAp[i] = a - l;
        }
    }

    // ---- sparse exchange ------------------------------------------------------------------------------------------------------------------------
    // entries of the slice that differ from `fill`: appended (order free) behind one atomic per wavefront; msg[0] = count, positions in the padded layout
    __global__ __launch_bounds__(256) void sx_pack_kernel(uint32_t n_local, const uint32_t* __restrict__ v, uint32_t fill, uint32_t base, uint32_t m, int with_values,
                                                          uint32_t* __restrict__ msg, uint32_t* __restrict__ overflow) {
        const uint32_t           i    = blockIdx.x * 256 + threadIdx.x;
        const bool               live = i < n_local && v[i] != fill;
        const unsigned long long b    = __ballot(live);
        if (b == 0ull) return;
        uint32_t first = 0;
        if (lane_id() == 0) first = atomicAdd(&msg[0], (uint32_t) __popcll(b));
        first = __shfl(first, 0, kWave);
        if (!live) return;
        const uint32_t slot = first + __builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0u));
        if (slot >= m) {
            *overflow = 1u;// the caller's bound was too small: the exchange is incomplete and says so
            return;
        }
        msg[1 + slot] = base + i;
        if (with_values) msg[1 + m + slot] = v[i];
    }
    // rank p's message sits at recv + p * words: positions [0, min(count, m)) are written (`clear`: with the fill value — the previous round's)
    __global__ __launch_bounds__(256) void sx_scatter_kernel(uint32_t world, uint32_t words, uint32_t m, const uint32_t* __restrict__ recv, int with_values, uint32_t value,
                                                             uint32_t* __restrict__ v_full, uint32_t limit) {
        const uint32_t k = blockIdx.x * 256 + threadIdx.x;
        const uint32_t p = blockIdx.y;
        if (p >= world || k >= m) return;
        const uint32_t* msg = recv + (size_t) p * words;
        if (k >= min(msg[0], m)) return;
        const uint32_t pos = msg[1 + k];
        if (pos < limit) v_full[pos] = with_values ? msg[1 + m + k] : value;
    }

    void shard_release(splahip_shard_s* s) {
        if (!s) return;
        for (splahip_plan p : {s->plan_whole, s->plan_loc, s->plan_rem})
            if (p) (void) splahip_csr_plan_destroy(p);
        for (void* p : {(void*) s->LAp, (void*) s->LAj, (void*) s->LAx, (void*) s->RAp, (void*) s->RAj, (void*) s->RAx})
            if (p) (void) hipFree(p);
        delete s;
    }

    bool part_served(splahip_plan plan, uint32_t nnz_part, int type, int op_add, uint32_t n_cols) { return nnz_part == 0 || cob_serves(plan, type, op_add, n_cols); }

}// namespace

extern "C" {

int splahip_dist_available(void) { return rccl().ok ? 1 : 0; }

int splahip_dist_get_unique_id(void* id128) {
    SPLAHIP_REQUIRE(id128 != nullptr, "id buffer is null");
    if (!rccl().ok) return fail(SPLAHIP_NOT_IMPLEMENTED, "RCCL is not loadable in this process (librccl.so)");
    ncclUniqueId id;
    SPLAHIP_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return SPLAHIP_OK;
}

int splahip_dist_comm_create(splahip_comm* out, int world, int rank, const void* id128) {
    SPLAHIP_REQUIRE(out != nullptr && id128 != nullptr, "null argument");
    SPLAHIP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank / world out of range");
    if (!rccl().ok) return fail(SPLAHIP_NOT_IMPLEMENTED, "RCCL is not loadable in this process (librccl.so)");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    auto* c  = new splahip_comm_s();
    c->world = world;
    c->rank  = rank;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(SPLAHIP_ERROR, "ncclCommInitRank failed: %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void) splahip_dist_comm_destroy(c);
        return fail(SPLAHIP_ERROR, "communicator streams / events: %s", hipGetErrorString(e));
    }
    *out = c;
    return SPLAHIP_OK;
}

int splahip_dist_comm_destroy(splahip_comm c) {
    if (!c) return SPLAHIP_OK;
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    if (c->comm && rccl().ok) (void) rccl().CommDestroy(c->comm);
    if (c->ready) (void) hipEventDestroy(c->ready);
    if (c->done) (void) hipEventDestroy(c->done);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
    return SPLAHIP_OK;
}

int splahip_dist_comm_info(splahip_comm c, int* world, int* rank) {
    SPLAHIP_REQUIRE(c != nullptr, "communicator is null");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return SPLAHIP_OK;
}

// every rank contributes `words` 4-byte words; d_recv receives world x words in rank order.  direct != 0: one grouped send / receive pair per peer
int splahip_dist_all_gather(splahip_comm c, const void* d_send, void* d_recv, size_t words, int direct, void* stream) {
    SPLAHIP_REQUIRE(c != nullptr && d_send != nullptr && d_recv != nullptr, "null argument");
    hipStream_t s = as_stream(stream);
    if (c->world == 1) {
        if (d_recv != d_send) SPLAHIP_TRY(hipMemcpyAsync(d_recv, d_send, 4 * words, hipMemcpyDeviceToDevice, s));
        return SPLAHIP_OK;
    }
    if (!direct) {
        SPLAHIP_NCCL(rccl().AllGather(d_send, d_recv, words, ncclUint32, c->comm, s));
        return SPLAHIP_OK;
    }
    uint32_t* recv = static_cast<uint32_t*>(d_recv);
    SPLAHIP_TRY(hipMemcpyAsync(recv + (size_t) c->rank * words, d_send, 4 * words, hipMemcpyDeviceToDevice, s));
    SPLAHIP_NCCL(rccl().GroupStart());
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        SPLAHIP_NCCL(rccl().Send(d_send, words, ncclUint32, p, c->comm, s));
        SPLAHIP_NCCL(rccl().Recv(recv + (size_t) p * words, words, ncclUint32, p, c->comm, s));
    }
    SPLAHIP_NCCL(rccl().GroupEnd());
    return SPLAHIP_OK;
}

// ---- the shard ------------------------------------------------------------------------------------------------------------------------------
int splahip_shard_create(splahip_shard* out, int world, int rank, uint32_t n_local, uint32_t max_rows, uint32_t nnz, const uint32_t* d_Ap,
                         const uint32_t* d_Aj_padded, const void* d_Ax, void* stream) {
    SPLAHIP_REQUIRE(out != nullptr, "shard is null");
    SPLAHIP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank / world out of range");
    SPLAHIP_REQUIRE(n_local <= max_rows, "n_local exceeds the padded slice length");
    SPLAHIP_REQUIRE((uint64_t) world * max_rows < (1ull << 31), "the padded layout must stay below 2^31 columns");
    SPLAHIP_REQUIRE(d_Ap != nullptr || n_local == 0, "d_Ap is null");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s   = as_stream(stream);
    auto*       sh  = new splahip_shard_s();
    sh->world = world, sh->rank = rank, sh->n_local = n_local, sh->max_rows = max_rows, sh->nnz = nnz;
    sh->Ap = d_Ap, sh->Aj = d_Aj_padded, sh->Ax = d_Ax;
    const uint32_t n_cols_padded = world > 1 ? (uint32_t) world * max_rows : max_rows;
    auto           bail          = [&](int code) {
        shard_release(sh);
        return code;
    };
    st = splahip_csr_plan_create(&sh->plan_whole, n_local, nnz, d_Ap, stream);
    if (st != SPLAHIP_OK) return bail(st);
    // Which form splahip_dist_mxv_masked runs.  Default: the two-part form (exchange hidden under the local-column part).  SPLAHIP_DIST_FORM=whole: the exchange
    // first, then ONE product over the unsplit block's own column-ordered blocks.  At the driver's N = 2 size (R-MAT 24, two blocks of 131 M nnz x 16.8 M columns)
    // the attach's launch model gives 176 us per part and 342 us for the unsplit block (two ranks sharing one GPU measured twice that: 338 + 358 against 641 us);
    // with ~220 us of exchange over one xGMI link, max(220, 176) + 176 beats 220 + 342: the two-part form stays the default at every rank count.
    {
        const char* fe         = knob_env("SPLAHIP_DIST_FORM");
        const bool  want_whole = fe && fe[0] == 'w';
        if (want_whole && nnz != 0 && n_local != 0) {
            st = splahip_csr_plan_attach_cob(sh->plan_whole, n_cols_padded, d_Ap, d_Aj_padded, d_Ax, 0, 0, stream);
            if (st != SPLAHIP_OK) return bail(st);
            sh->whole_form = sh->plan_whole->cob != nullptr;
        }
    }
    if (nnz == 0 || n_local == 0) {
        sh->split_ok = true;
        *out         = sh;
        return SPLAHIP_OK;
    }
    // flags -> exclusive scan -> stable partition
    uint32_t *flag = nullptr, *pos = nullptr;
    void*     tmp  = nullptr;
    hipError_t e   = device_malloc(reinterpret_cast<void**>(&flag), sizeof(uint32_t) * ((size_t) nnz + 1));
    if (e == hipSuccess) e = device_malloc(reinterpret_cast<void**>(&pos), sizeof(uint32_t) * ((size_t) nnz + 1));
    const uint32_t lo = (uint32_t) rank * max_rows, hi = lo + max_rows;
    size_t         tb = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(split_flag_kernel, dim3((nnz + 255) / 256), dim3(256), 0, s, nnz, d_Aj_padded, lo, hi, flag);
        e = hipMemsetAsync(flag + nnz, 0, 4, s);
    }
    if (e == hipSuccess) e = rocprim::exclusive_scan(nullptr, tb, flag, pos, 0u, (size_t) nnz + 1, rocprim::plus<uint32_t>(), s);
    if (e == hipSuccess) e = device_malloc(&tmp, tb ? tb : 4);
    if (e == hipSuccess) e = rocprim::exclusive_scan(tmp, tb, flag, pos, 0u, (size_t) nnz + 1, rocprim::plus<uint32_t>(), s);
    uint32_t n_loc = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&n_loc, pos + nnz, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    sh->nnz_loc = n_loc, sh->nnz_rem = nnz - n_loc;
    auto own = [&](uint32_t** q, size_t words) {
        if (e == hipSuccess) e = device_malloc(reinterpret_cast<void**>(q), sizeof(uint32_t) * (words ? words : 1));
    };
    own(&sh->LAp, (size_t) n_local + 1), own(&sh->RAp, (size_t) n_local + 1);
    own(&sh->LAj, sh->nnz_loc), own(&sh->LAx, sh->nnz_loc), own(&sh->RAj, sh->nnz_rem), own(&sh->RAx, sh->nnz_rem);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(split_scatter_kernel, dim3((nnz + 255) / 256), dim3(256), 0, s, nnz, d_Aj_padded, static_cast<const uint32_t*>(d_Ax), flag, pos, lo, sh->LAj,
                           sh->LAx, sh->RAj, sh->RAx);
        hipLaunchKernelGGL(split_rowptr_kernel, dim3((n_local + 256) / 256), dim3(256), 0, s, n_local, d_Ap, pos, sh->LAp, sh->RAp);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    for (void* p : {(void*) flag, (void*) pos, tmp})
        if (p) (void) hipFree(p);
    if (e != hipSuccess) {
        fail(SPLAHIP_ERROR, "shard split failed: %s", hipGetErrorString(e));
        return bail(SPLAHIP_ERROR);
    }
    st = splahip_csr_plan_create(&sh->plan_loc, n_local, sh->nnz_loc, sh->LAp, stream);
    if (st == SPLAHIP_OK) st = splahip_csr_plan_create(&sh->plan_rem, n_local, sh->nnz_rem, sh->RAp, stream);
    if (st == SPLAHIP_OK && sh->nnz_loc) st = splahip_csr_plan_attach_cob(sh->plan_loc, max_rows, sh->LAp, sh->LAj, sh->LAx, 0, 0, stream);
    if (st == SPLAHIP_OK && sh->nnz_rem) st = splahip_csr_plan_attach_cob(sh->plan_rem, n_cols_padded, sh->RAp, sh->RAj, sh->RAx, 0, 0, stream);
    if (st != SPLAHIP_OK) return bail(st);
    sh->split_ok = (sh->nnz_loc == 0 || sh->plan_loc->cob) && (sh->nnz_rem == 0 || sh->plan_rem->cob);
    *out         = sh;
    return SPLAHIP_OK;
}

int splahip_shard_destroy(splahip_shard s) {
    shard_release(s);
    return SPLAHIP_OK;
}

int splahip_shard_info(splahip_shard s, uint32_t* nnz_local_cols, uint32_t* nnz_remote_cols, int* split_ok) {
    SPLAHIP_REQUIRE(s != nullptr, "shard is null");
    if (nnz_local_cols) *nnz_local_cols = s->nnz_loc;
    if (nnz_remote_cols) *nnz_remote_cols = s->nnz_rem;
    if (split_ok) *split_ok = s->split_ok ? 1 : 0;
    return SPLAHIP_OK;
}

// may the two-part form serve this call (else: splahip_shard_mxv_whole after the exchange)
// 1: splahip_dist_mxv_masked runs the unsplit block (exchange first) although the two-part form could serve the call
int splahip_shard_prefers_whole(splahip_shard s, int type, int op_add) {
    if (!s || !s->whole_form) return 0;
    const uint32_t n_cols_padded = s->world > 1 ? (uint32_t) s->world * s->max_rows : s->max_rows;
    return cob_serves(s->plan_whole, type, op_add, n_cols_padded) ? 1 : 0;
}

int splahip_shard_split_serves(splahip_shard s, int type, int op_add, int early_exit) {
    if (!s || !s->split_ok || early_exit) return 0;
    const uint32_t n_cols_padded = s->world > 1 ? (uint32_t) s->world * s->max_rows : s->max_rows;
    return part_served(s->plan_loc, s->nnz_loc, type, op_add, s->max_rows) && part_served(s->plan_rem, s->nnz_rem, type, op_add, n_cols_padded) ? 1 : 0;
}

// phase A: the local-column part.  Needs only this rank's own slice of v (max_rows words, padded); writes every row of r.
int splahip_shard_mxv_local(splahip_shard s, int type, const void* d_mask, const void* d_v_local, void* d_r, int op_mult, int op_add, int op_select, uint32_t init_bits,
                            void* stream) {
    SPLAHIP_REQUIRE(s != nullptr, "shard is null");
    if (!splahip_shard_split_serves(s, type, canonical_add(type, op_add, init_bits), 0))// (fp32 MIN / MAX from a NaN init is FIRST, ops.hpp: the unsplit block)
        return fail(SPLAHIP_NOT_IMPLEMENTED, "the two-part form does not serve this op_add: use splahip_shard_mxv_whole");
    if (s->n_local == 0) return SPLAHIP_OK;
    if (s->nnz_loc == 0)// no local-column entry: every row is init (the remote part folds onto it)
        return splahip_memset32(d_r, init_bits, s->n_local, stream);
    return splahip_mxv_masked(s->plan_loc, type, s->n_local, s->max_rows, s->LAp, s->LAj, s->LAx, d_mask, d_v_local, d_r, op_mult, op_add, op_select, init_bits, 0, stream);
}

// phase B: the remote-column part folds onto r; needs the gathered vector (padded layout)
int splahip_shard_mxv_remote(splahip_shard s, int type, const void* d_mask, const void* d_v_full, void* d_r, int op_mult, int op_add, int op_select, void* stream) {
    SPLAHIP_REQUIRE(s != nullptr, "shard is null");
    if (!splahip_shard_split_serves(s, type, op_add, 0)) return fail(SPLAHIP_NOT_IMPLEMENTED, "the two-part form does not serve this op_add: use splahip_shard_mxv_whole");
    if (s->n_local == 0 || s->nnz_rem == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(((reinterpret_cast<uintptr_t>(d_r) | reinterpret_cast<uintptr_t>(d_mask)) & 15u) == 0, "r and mask must be 16-byte aligned");
    return cob_mxv_accumulate(s->plan_rem, type, s->RAp, d_mask, d_v_full, d_r, op_mult, op_add, op_select, stream);
}

// the unsplit block over the gathered vector: every op, early exit included (what round 1 did for every call)
int splahip_shard_mxv_whole(splahip_shard s, int type, const void* d_mask, const void* d_v_full, void* d_r, int op_mult, int op_add, int op_select, uint32_t init_bits,
                            int early_exit, void* stream) {
    SPLAHIP_REQUIRE(s != nullptr, "shard is null");
    const uint32_t n_cols_padded = s->world > 1 ? (uint32_t) s->world * s->max_rows : s->max_rows;
    return splahip_mxv_masked(s->plan_whole, type, s->n_local, n_cols_padded, s->Ap, s->Aj, s->Ax, d_mask, d_v_full, d_r, op_mult, op_add, op_select, init_bits, early_exit,
                              stream);
}

// One sharded product: d_v_local = this rank's slice (max_rows words, padded with anything), d_v_full = world x max_rows words of workspace that
// receives the gathered vector, d_r / d_mask = this rank's n_local rows.  The exchange runs on the communicator's stream while the local part
// runs on `stream`; the remote part waits for it.  *overlapped = 1 when the two-part form ran.
int splahip_dist_mxv_masked(splahip_comm c, splahip_shard s, int type, const void* d_mask, const void* d_v_local, void* d_v_full, void* d_r, int op_mult, int op_add,
                            int op_select, uint32_t init_bits, int early_exit, int direct, int* overlapped, void* stream) {
    SPLAHIP_REQUIRE(c != nullptr && s != nullptr, "null communicator / shard");
    SPLAHIP_REQUIRE(c->world == s->world && c->rank == s->rank, "communicator and shard disagree on rank / world");
    hipStream_t cs = as_stream(stream);
    const uint32_t n_cols_all = s->world > 1 ? (uint32_t) s->world * s->max_rows : s->max_rows;
    op_add             = canonical_add(type, op_add, init_bits);// fp32 MIN / MAX from a NaN init: FIRST (ops.hpp) -> the unsplit block
    // (the remote part's row pass moves r and mask in 16-byte loads: a call whose r / mask is not 16-byte aligned runs the unsplit block, whose kernels take any alignment)
    const bool aligned = ((reinterpret_cast<uintptr_t>(d_r) | reinterpret_cast<uintptr_t>(d_mask)) & 15u) == 0;
    const bool     two = aligned && splahip_shard_split_serves(s, type, op_add, early_exit) != 0 && !(s->whole_form && cob_serves(s->plan_whole, type, op_add, n_cols_all));
    if (overlapped) *overlapped = two ? 1 : 0;
    // the exchange starts as soon as the input slice is ready on the compute stream
    SPLAHIP_TRY(hipEventRecord(c->ready, cs));
    SPLAHIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
    int st = splahip_dist_all_gather(c, d_v_local, d_v_full, s->max_rows, direct, c->stream);
    if (st != SPLAHIP_OK) return st;
    SPLAHIP_TRY(hipEventRecord(c->done, c->stream));
    if (two) {
        st = splahip_shard_mxv_local(s, type, d_mask, d_v_local, d_r, op_mult, op_add, op_select, init_bits, stream);
        if (st != SPLAHIP_OK) return st;
        SPLAHIP_TRY(hipStreamWaitEvent(cs, c->done, 0));
        return splahip_shard_mxv_remote(s, type, d_mask, d_v_full, d_r, op_mult, op_add, op_select, stream);
    }
    SPLAHIP_TRY(hipStreamWaitEvent(cs, c->done, 0));
    return splahip_shard_mxv_whole(s, type, d_mask, d_v_full, d_r, op_mult, op_add, op_select, init_bits, early_exit, stream);
}

// ---- sparsity-aware exchange -------------------------------------------------------------------------------------------------------------------
int splahip_dist_sx_create(splahip_sx* out, int world, int rank, uint32_t max_rows, uint32_t capacity) {
    SPLAHIP_REQUIRE(out != nullptr, "null argument");
    SPLAHIP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank / world out of range");
    SPLAHIP_REQUIRE(capacity >= 1 && capacity <= max_rows, "capacity must be in [1, max_rows]");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    auto* x     = new splahip_sx_s();
    x->world    = world, x->rank = rank, x->max_rows = max_rows, x->capacity = capacity;
    const size_t words = 1 + 2 * (size_t) capacity;
    hipError_t   e     = device_malloc(reinterpret_cast<void**>(&x->d_send), 4 * words);
    for (int b = 0; b < 2 && e == hipSuccess; ++b) e = device_malloc(reinterpret_cast<void**>(&x->d_recv[b]), 4 * words * (size_t) world);
    if (e == hipSuccess) e = device_malloc(reinterpret_cast<void**>(&x->d_overflow), 4);
    if (e == hipSuccess) e = hipMemset(x->d_overflow, 0, 4);
    if (e != hipSuccess) {
        (void) splahip_dist_sx_destroy(x);
        return fail(SPLAHIP_ERROR, "sparse exchange buffers: %s", hipGetErrorString(e));
    }
    *out = x;
    return SPLAHIP_OK;
}
int splahip_dist_sx_destroy(splahip_sx x) {
    if (!x) return SPLAHIP_OK;
    for (void* p : {(void*) x->d_send, (void*) x->d_recv[0], (void*) x->d_recv[1], (void*) x->d_overflow})
        if (p) (void) hipFree(p);
    delete x;
    return SPLAHIP_OK;
}
// phase 1: this rank's message for an exchange whose per-rank count is bounded by m (0: the capacity)
int splahip_dist_sx_pack(splahip_sx x, const void* d_v_local, uint32_t n_local, uint32_t fill_bits, uint32_t m, int with_values, void* stream) {
    SPLAHIP_REQUIRE(x != nullptr && (d_v_local != nullptr || n_local == 0), "null argument");
    SPLAHIP_REQUIRE(n_local <= x->max_rows, "n_local exceeds the padded slice length");
    if (m == 0 || m > x->capacity) m = x->capacity;
    hipStream_t s = as_stream(stream);
    SPLAHIP_TRY(hipMemsetAsync(x->d_send, 0, 4, s));
    if (n_local)
        hipLaunchKernelGGL(sx_pack_kernel, dim3((n_local + 255) / 256), dim3(256), 0, s, n_local, static_cast<const uint32_t*>(d_v_local), fill_bits,
                           x->world > 1 ? (uint32_t) x->rank * x->max_rows : 0u, m, with_values, x->d_send, x->d_overflow);
    SPLAHIP_TRY(hipGetLastError());
    return SPLAHIP_OK;
}
// the message buffers of the running exchange, for a host that brings its own transport: this rank's message (1 + (1 or 2) m words are meaningful) and
// where rank p's message must land (d_recv + p * words_per_rank)
int splahip_dist_sx_buffers(splahip_sx x, uint32_t m, int with_values, void** d_send, void** d_recv, size_t* words_per_rank) {
    SPLAHIP_REQUIRE(x != nullptr, "null argument");
    if (m == 0 || m > x->capacity) m = x->capacity;
    if (d_send) *d_send = x->d_send;
    if (d_recv) *d_recv = x->d_recv[x->cur];
    if (words_per_rank) *words_per_rank = 1 + (with_values ? 2 : 1) * (size_t) m;
    return SPLAHIP_OK;
}
// phase 3: the previous exchange's positions go back to the fill value, this exchange's are written (values from the messages, or `value_bits` for all)
int splahip_dist_sx_scatter(splahip_sx x, void* d_v_full, uint32_t fill_bits, uint32_t m, int with_values, uint32_t value_bits, void* stream) {
    SPLAHIP_REQUIRE(x != nullptr && d_v_full != nullptr, "null argument");
    if (m == 0 || m > x->capacity) m = x->capacity;
    hipStream_t    s     = as_stream(stream);
    const uint32_t limit = x->world > 1 ? (uint32_t) x->world * x->max_rows : x->max_rows;
    if (x->has_prev) {
        const uint32_t pm = x->prev_m, pw = 1 + (x->prev_with_values ? 2 : 1) * pm;
        hipLaunchKernelGGL(sx_scatter_kernel, dim3((pm + 255) / 256, x->world), dim3(256), 0, s, (uint32_t) x->world, pw, pm, x->d_recv[x->cur ^ 1], 0, fill_bits,
                           static_cast<uint32_t*>(d_v_full), limit);
    }
    const uint32_t words = 1 + (with_values ? 2 : 1) * m;
    hipLaunchKernelGGL(sx_scatter_kernel, dim3((m + 255) / 256, x->world), dim3(256), 0, s, (uint32_t) x->world, words, m, x->d_recv[x->cur], with_values, value_bits,
                       static_cast<uint32_t*>(d_v_full), limit);
    SPLAHIP_TRY(hipGetLastError());
    x->prev_m = m, x->prev_with_values = with_values ? 1u : 0u, x->has_prev = true;
    x->cur ^= 1;
    return SPLAHIP_OK;
}
// 1 when some pack since the last query met more entries than its bound (blocking: reads one word)
int splahip_dist_sx_overflowed(splahip_sx x, int* overflowed) {
    SPLAHIP_REQUIRE(x != nullptr && overflowed != nullptr, "null argument");
    uint32_t h = 0;
    SPLAHIP_TRY(hipMemcpy(&h, x->d_overflow, 4, hipMemcpyDeviceToHost));
    if (h) SPLAHIP_TRY(hipMemset(x->d_overflow, 0, 4));
    *overflowed = h ? 1 : 0;
    return SPLAHIP_OK;
}
// the three phases with RCCL between them, all on `stream`: d_v_full (world x max_rows words, at the fill value before the first exchange) then holds
// what the dense all-gather of the slices would
int splahip_dist_exchange_sparse(splahip_comm c, splahip_sx x, const void* d_v_local, uint32_t n_local, uint32_t fill_bits, uint32_t m, int with_values,
                                 uint32_t value_bits, void* d_v_full, int direct, void* stream) {
    SPLAHIP_REQUIRE(c != nullptr && x != nullptr, "null communicator / exchange");
    SPLAHIP_REQUIRE(c->world == x->world && c->rank == x->rank, "communicator and exchange disagree on rank / world");
    if (m == 0 || m > x->capacity) m = x->capacity;
    int st = splahip_dist_sx_pack(x, d_v_local, n_local, fill_bits, m, with_values, stream);
    if (st != SPLAHIP_OK) return st;
    const size_t words = 1 + (with_values ? 2 : 1) * (size_t) m;
    st                 = splahip_dist_all_gather(c, x->d_send, x->d_recv[x->cur], words, direct, stream);
    if (st != SPLAHIP_OK) return st;
    return splahip_dist_sx_scatter(x, d_v_full, fill_bits, m, with_values, value_bits, stream);
}
}
