// vector_ops.hip — the three vector operations that sit between two hot-path calls in every BFS / SSSP iteration
// (SURVEY.md 8f rank 1), so a traversal loop never leaves the device:
//
//   splahip_v_assign_masked_*  semantics: reference Algo_v_assign_masked_cpu, src/cpu/cpu_v_assign.hpp:68-132
//                              (replaces Algo_v_assign_masked_cl, src/opencl/cl_v_assign.hpp:96-148)
//   splahip_v_eadd_fdb_*       semantics: reference Algo_v_eadd_fdb_cpu, src/cpu/cpu_v_eadd_fdb.hpp:71-139
//                              (replaces Algo_v_eadd_fdb_cl, src/opencl/cl_v_eadd_fdb.hpp:110-163)
//   splahip_v_count_mf_dense   semantics: reference Algo_v_count_mf_cpu::execute_dense, src/cpu/cpu_v_count_mf.hpp:99-108
//                              (replaces Algo_v_count_mf_cl, src/opencl/cl_v_count_mf.hpp:99-108)
//
// and the remainder of the PageRank step and the triangle-count tail:
//
//   splahip_v_eadd_dense/_sparse  semantics: reference Algo_v_eadd_cpu, src/cpu/cpu_v_eadd.hpp:70-157
//   splahip_reduce                semantics: reference Algo_v_reduce_cpu / Algo_m_reduce_cpu: a fold of the stored values from
//                                 the scalar s (src/cpu/cpu_v_reduce.hpp:69-117, src/cpu/cpu_m_reduce.hpp:134-156)
//
// All are element-wise, HBM-bound streaming kernels (4-16 bytes per element).  The sparse forms assume the COO operand
// holds every index at most once (always true for vectors produced by this library and by Dok->Coo / Dense->Coo
// conversions; the reference's sequential loop would apply a duplicate twice).  The sparse feedback of v_eadd_fdb is
// compacted order-preservingly, so it lists the changed entries in the operand's stored order like the reference.
#include "block_utils.hpp"
#include "common.hpp"
#include "scan_post.hpp"
#include "jit_hooks.hpp"
#include "ops.hpp"

#include <algorithm>

namespace splahip {

    template<typename T>
    __global__ __launch_bounds__(256) void assign_dense_kernel(uint n, T* __restrict__ r, const T* __restrict__ mask, T value, DynBin assign, int op_select) {
        const uint i = blockIdx.x * 256 + threadIdx.x;
        if (i < n && eval_sel(op_select, mask[i])) r[i] = assign(r[i], value);
    }
    template<typename T>
    __global__ __launch_bounds__(256) void assign_sparse_kernel(uint nvals, T* __restrict__ r, const uint* __restrict__ mi, const T* __restrict__ mx,
                                                                T value, DynBin assign, int op_select) {
        const uint k = blockIdx.x * 256 + threadIdx.x;
        if (k < nvals && eval_sel(op_select, mx[k])) {
            const uint i = mi[k];
            r[i]         = assign(r[i], value);
        }
    }

    template<typename T>
    __global__ __launch_bounds__(256) void count_mf_kernel(uint n, const T* __restrict__ v, T fill, unsigned long long* __restrict__ count) {
        __shared__ uint s_tmp[256 / kWave + 1];
        uint            cnt = 0;
        for (uint i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) cnt += v[i] != fill ? 1u : 0u;
        uint total;
        (void) block_exclusive_sum<256>(cnt, s_tmp, total);
        if (threadIdx.x == 0 && total) atomicAdd(count, (unsigned long long) total);// one atomic per workgroup: 8 K same-address atomics cost ~90 us
    }

    template<typename T>
    __global__ __launch_bounds__(256) void eadd_fdb_dense_kernel(uint n, T* __restrict__ r, const T* __restrict__ v, T* __restrict__ fdb, T fdb_fill, DynBin op) {
        const uint i = blockIdx.x * 256 + threadIdx.x;
        if (i >= n) return;
        const T prev = r[i];
        const T next = op(prev, v[i]);
        r[i]         = next;
        fdb[i]       = prev != next ? next : fdb_fill;
    }

    // sparse form, pass 1: apply, remember which entries changed
    template<typename T>
    __global__ __launch_bounds__(256) void eadd_fdb_sparse_apply_kernel(uint nvals, T* __restrict__ r, const uint* __restrict__ vi, const T* __restrict__ vx,
                                                                        uint* __restrict__ changed, T* __restrict__ newval, DynBin op) {
        const uint k = blockIdx.x * 256 + threadIdx.x;
        if (k >= nvals) return;
        const uint i    = vi[k];
        const T    prev = r[i];
        const T    next = op(prev, vx[k]);
        r[i]            = next;
        changed[k]      = prev != next ? 1u : 0u;
        newval[k]       = next;
    }
    __global__ __launch_bounds__(256) void changed_count_kernel(const uint* __restrict__ flag, uint n, uint* __restrict__ chunk) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            cnt   = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u)
            if (first + u < n) cnt += flag[first + u];
        uint total;
        (void) block_exclusive_sum<256>(cnt, s_tmp, total);
        if (threadIdx.x == 0) chunk[blockIdx.x] = total;
    }
    __global__ __launch_bounds__(256) void changed_write_kernel(const uint* __restrict__ flag, const uint* __restrict__ vi, const uint* __restrict__ newval, uint n,
                                                                const uint* __restrict__ chunk, uint* __restrict__ fi, uint* __restrict__ fx) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        bool            keep[4];
        uint            cnt = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            keep[u] = first + u < n && flag[first + u] != 0;
            cnt += keep[u] ? 1u : 0u;
        }
        uint total;
        uint pos = chunk[blockIdx.x] + block_exclusive_sum<256>(cnt, s_tmp, total);
#pragma unroll
        for (uint u = 0; u < 4; ++u)
            if (keep[u]) {
                fi[pos] = vi[first + u];
                fx[pos] = newval[first + u];
                ++pos;
            }
    }

    template<typename T>
    int eadd_fdb_sparse(uint nvals, void* r, const uint* vi, const void* vx, uint* fi, void* fx, uint* h_count, int op, hipStream_t s) {
        *h_count = 0;
        if (nvals == 0) return SPLAHIP_OK;
        const uint n_chunks = (nvals + 1023u) / 1024u;
        DevMem     mem;// changed[nvals] | newval[nvals] | chunk[n_chunks + 1]
        SPLAHIP_TRY(mem.alloc(s, sizeof(uint) * (2 * (size_t) nvals + n_chunks + 1)));
        uint* scratch = mem.as<uint>();
        uint* changed = scratch;
        uint* newval  = scratch + nvals;
        uint* chunk   = newval + nvals;
        hipLaunchKernelGGL(eadd_fdb_sparse_apply_kernel<T>, dim3((nvals + 255) / 256), dim3(256), 0, s, nvals, static_cast<T*>(r), vi,
                           static_cast<const T*>(vx), changed, reinterpret_cast<T*>(newval), DynBin(op));
        hipLaunchKernelGGL(changed_count_kernel, dim3(n_chunks), dim3(256), 0, s, changed, nvals, chunk);
        ScanPost sp;
        (void) scan_launch(chunk, n_chunks, s, &sp);// (the count reaches the host from the scan: scan_post.hpp)
        hipLaunchKernelGGL(changed_write_kernel, dim3(n_chunks), dim3(256), 0, s, changed, vi, newval, nvals, chunk, fi, static_cast<uint*>(fx));
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = scan_total(h_count, chunk + n_chunks, sp, s);
        if (e != hipSuccess) return fail(SPLAHIP_ERROR, "v_eadd_fdb (sparse) failed: %s", hipGetErrorString(e));
        return SPLAHIP_OK;
    }

    // ---- v_eadd -------------------------------------------------------------------------------------------------------
    template<typename T>
    __global__ __launch_bounds__(256) void eadd_dense_kernel(uint n, T* __restrict__ r, const T* __restrict__ u, const T* __restrict__ v, DynBin op) {
        const uint i = blockIdx.x * 256 + threadIdx.x;
        if (i < n) r[i] = op(u[i], v[i]);
    }

    // first position in the ascending list a[0..n) whose value is >= key
    __device__ __forceinline__ uint lower_bound_u32(const uint* __restrict__ a, uint n, uint key) {
        uint lo = 0, hi = n;
        while (lo < hi) {
            const uint mid = lo + ((hi - lo) >> 1);
            if (a[mid] < key) lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    }
    // sparse (+) sparse, both index-sorted without duplicates: the merge of the reference (cpu_v_eadd.hpp:96-127) as a
    // rank computation.  Pass 1: per entry its rank in the other list and whether the index is shared.
    __global__ __launch_bounds__(256) void eadd_rank_kernel(uint na, const uint* __restrict__ ai, uint nb, const uint* __restrict__ bi,
                                                            uint* __restrict__ rank, uint* __restrict__ shared_flag) {
        const uint k = blockIdx.x * 256 + threadIdx.x;
        if (k >= na) return;
        const uint key = ai[k];
        const uint lb  = lower_bound_u32(bi, nb, key);
        rank[k]        = lb;
        shared_flag[k] = (lb < nb && bi[lb] == key) ? 1u : 0u;
    }
    // exclusive prefix sums of the shared flags, in place (flag[k] -> shared entries before k), count/scan/write style
    __global__ __launch_bounds__(256) void flag_chunk_count_kernel(const uint* __restrict__ flag, uint n, uint* __restrict__ chunk) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            cnt   = 0;
#pragma unroll
        for (uint e = 0; e < 4; ++e)
            if (first + e < n) cnt += flag[first + e];
        uint total;
        (void) block_exclusive_sum<256>(cnt, s_tmp, total);
        if (threadIdx.x == 0) chunk[blockIdx.x] = total;
    }
    // pass 2 for the u side: every u entry is emitted; position = k + (v entries before it) - (shared entries before it)
    template<typename T>
    __global__ __launch_bounds__(256) void eadd_emit_u_kernel(uint nu, const uint* __restrict__ ui, const T* __restrict__ ux, const T* __restrict__ vx, T v_fill,
                                                              const uint* __restrict__ rank, const uint* __restrict__ flag, const uint* __restrict__ chunk,
                                                              uint* __restrict__ ri, T* __restrict__ rx, DynBin op) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            f[4], cnt = 0;
#pragma unroll
        for (uint e = 0; e < 4; ++e) {
            f[e] = first + e < nu ? flag[first + e] : 0u;
            cnt += f[e];
        }
        uint total;
        uint before = chunk[blockIdx.x] + block_exclusive_sum<256>(cnt, s_tmp, total);
#pragma unroll
        for (uint e = 0; e < 4; ++e) {
            const uint k = first + e;
            if (k < nu) {
                const uint pos = k + rank[k] - before;
                ri[pos]        = ui[k];
                rx[pos]        = op(ux[k], f[e] ? vx[rank[k]] : v_fill);
            }
            before += f[e];
        }
    }
    // pass 2 for the v side: only entries whose index u does not hold
    template<typename T>
    __global__ __launch_bounds__(256) void eadd_emit_v_kernel(uint nv, const uint* __restrict__ vi, const T* __restrict__ vx, T u_fill,
                                                              const uint* __restrict__ rank, const uint* __restrict__ flag, const uint* __restrict__ chunk,
                                                              uint* __restrict__ ri, T* __restrict__ rx, DynBin op) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            f[4], cnt = 0;
#pragma unroll
        for (uint e = 0; e < 4; ++e) {
            f[e] = first + e < nv ? flag[first + e] : 0u;
            cnt += f[e];
        }
        uint total;
        uint before = chunk[blockIdx.x] + block_exclusive_sum<256>(cnt, s_tmp, total);
#pragma unroll
        for (uint e = 0; e < 4; ++e) {
            const uint k = first + e;
            if (k < nv && !f[e]) {
                const uint pos = k + rank[k] - before;
                ri[pos]        = vi[k];
                rx[pos]        = op(u_fill, vx[k]);
            }
            before += f[e];
        }
    }

    template<typename T>
    int eadd_sparse(uint nu, const uint* ui, const void* ux, uint32_t u_fill, uint nv, const uint* vi, const void* vx, uint32_t v_fill, uint* ri, void* rx,
                    uint* h_count, int op, hipStream_t s) {
        const uint cu = (nu + 1023u) / 1024u, cv = (nv + 1023u) / 1024u;
        DevMem     mem;// rank_u[nu] flag_u[nu] chunk_u[cu+1] rank_v[nv] flag_v[nv] chunk_v[cv+1]
        SPLAHIP_TRY(mem.alloc(s, sizeof(uint) * (2 * (size_t) nu + 2 * (size_t) nv + cu + cv + 2)));
        uint* scratch = mem.as<uint>();
        uint *rank_u = scratch, *flag_u = rank_u + nu, *chunk_u = flag_u + nu, *rank_v = chunk_u + cu + 1, *flag_v = rank_v + nv, *chunk_v = flag_v + nv;
        if (nu) {
            hipLaunchKernelGGL(eadd_rank_kernel, dim3((nu + 255) / 256), dim3(256), 0, s, nu, ui, nv, vi, rank_u, flag_u);
            hipLaunchKernelGGL(flag_chunk_count_kernel, dim3(cu), dim3(256), 0, s, flag_u, nu, chunk_u);
            hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, chunk_u, cu);
            hipLaunchKernelGGL(eadd_emit_u_kernel<T>, dim3(cu), dim3(256), 0, s, nu, ui, static_cast<const T*>(ux), static_cast<const T*>(vx), from_bits<T>(v_fill),
                               rank_u, flag_u, chunk_u, ri, static_cast<T*>(rx), DynBin(op));
        }
        if (nv) {
            hipLaunchKernelGGL(eadd_rank_kernel, dim3((nv + 255) / 256), dim3(256), 0, s, nv, vi, nu, ui, rank_v, flag_v);
            hipLaunchKernelGGL(flag_chunk_count_kernel, dim3(cv), dim3(256), 0, s, flag_v, nv, chunk_v);
            hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, chunk_v, cv);
            hipLaunchKernelGGL(eadd_emit_v_kernel<T>, dim3(cv), dim3(256), 0, s, nv, vi, static_cast<const T*>(vx), from_bits<T>(u_fill), rank_v, flag_v, chunk_v, ri,
                               static_cast<T*>(rx), DynBin(op));
        }
        hipError_t e      = hipGetLastError();
        uint       shared = 0;
        if (e == hipSuccess) e = fetch_words(&shared, chunk_v + cv, nv ? 1 : 0, s);
        if (e != hipSuccess) return fail(SPLAHIP_ERROR, "v_eadd (sparse) failed: %s", hipGetErrorString(e));
        *h_count = nu + nv - shared;
        return SPLAHIP_OK;
    }

    // ---- reduce ---------------------------------------------------------------------------------------------------------
    // Fold of x[0..n) from s.  Associative + commutative ops: every workgroup folds a grid-strided share (16-byte loads),
    // one more workgroup folds the partials in block order and applies s on the left — a fixed tree, so results repeat
    // bit for bit from run to run.  Other ops (MINUS, DIV, MINUS_POW2, FIRST, SECOND): the reference's strict left fold by one lane.
    constexpr uint kReduceBlocks = 1024;

    template<typename T>
    __device__ __forceinline__ void block_fold(T& acc, bool& valid, const Reassoc<DynBin>& op, T* s_val, int* s_ok) {
        wave_reduce_ordered(acc, valid, op);
        if (lane_id() == 0) {
            s_val[wave_id()] = acc;
            s_ok[wave_id()]  = valid ? 1 : 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            bool ok = false;
            T    a  = acc;
            for (uint w = 0; w < 256 / kWave; ++w)
                if (s_ok[w]) {
                    a  = ok ? op(a, s_val[w]) : s_val[w];
                    ok = true;
                }
            acc   = a;
            valid = ok;
        }
    }
    template<typename T>
    __global__ __launch_bounds__(256) void reduce_partial_kernel(uint n, const T* __restrict__ x, T* __restrict__ partial, uint* __restrict__ partial_ok, DynBin op_raw) {
        const Reassoc<DynBin> op(op_raw);// (ops.hpp) the tree's partials start at elements: fp32 MIN / MAX drop NaN elements like the fold from `s` does
        __shared__ T   s_val[256 / kWave];
        __shared__ int s_ok[256 / kWave];
        T              acc   = T();
        bool           valid = false;
        const uint     nq    = n / 4;
        for (uint q = blockIdx.x * 256 + threadIdx.x; q < nq; q += gridDim.x * 256) {
            const uint4 w = *reinterpret_cast<const uint4*>(x + 4 * (size_t) q);
            const T     e[4] = {from_bits<T>(w.x), from_bits<T>(w.y), from_bits<T>(w.z), from_bits<T>(w.w)};
            T           g    = op(op(op(e[0], e[1]), e[2]), e[3]);
            acc              = valid ? op(acc, g) : g;
            valid            = true;
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {// the up to 3 trailing elements
            const T g = x[4 * (size_t) nq + threadIdx.x];
            acc       = valid ? op(acc, g) : g;
            valid     = true;
        }
        block_fold(acc, valid, op, s_val, s_ok);
        if (threadIdx.x == 0) {
            partial[blockIdx.x]    = acc;
            partial_ok[blockIdx.x] = valid ? 1u : 0u;
        }
    }
    template<typename T>
    __global__ __launch_bounds__(256) void reduce_final_kernel(uint n_partial, const T* __restrict__ partial, const uint* __restrict__ partial_ok, T s, T* __restrict__ out, DynBin op_raw,
                                                               unsigned long long* __restrict__ mailbox, uint mail_seq) {
        const Reassoc<DynBin> op(op_raw);
        __shared__ T   s_val[256 / kWave];
        __shared__ int s_ok[256 / kWave];
        T              acc   = T();
        bool           valid = false;
        // thread t folds the contiguous partials [t*per, (t+1)*per): block order is kept
        const uint per = (n_partial + 255u) / 256u;
        for (uint k = threadIdx.x * per; k < min((threadIdx.x + 1) * per, n_partial); ++k)
            if (partial_ok[k]) {
                acc   = valid ? op(acc, partial[k]) : partial[k];
                valid = true;
            }
        block_fold(acc, valid, op, s_val, s_ok);
        if (threadIdx.x == 0) {
            const T res = valid ? op(s, acc) : s;
            out[0]      = res;
            if (mailbox) mailbox_store(mailbox, 0, mail_seq, to_bits(res));// the host waits for this word: posted here, no mailbox launch behind (common.hpp)
        }
    }
    // one wave: 64 lanes fetch a 1024-element chunk (coalesced), lane 0 folds it in order out of LDS
    template<typename T>
    __global__ __launch_bounds__(64) void reduce_sequential_kernel(uint n, const T* __restrict__ x, T s, T* __restrict__ out, DynBin op) {
        __shared__ T s_buf[1024];
        T            sum = s;
        for (uint base = 0; base < n; base += 1024) {
            const uint len = min(1024u, n - base);
            for (uint k = threadIdx.x; k < len; k += 64) s_buf[k] = x[base + k];
            __syncthreads();
            if (threadIdx.x == 0)
                for (uint k = 0; k < len; ++k) sum = op(sum, s_buf[k]);
            __syncthreads();
        }
        if (threadIdx.x == 0) out[0] = sum;
    }

    template<typename T>
    int reduce_typed(uint n, const void* x, uint32_t s_bits, uint32_t* h_result, int op, hipStream_t s) {
        op = canonical_add(TypeId<T>::value, op, s_bits);// fp32 MIN / MAX from a NaN `s`: the reference's fold keeps `s` (ops.hpp)
        const uint blocks  = std::max(1u, std::min(kReduceBlocks, (n / 4 + 255u) / 256u));
        DevMem     mem;// out[1] | partial[blocks] | ok[blocks]
        SPLAHIP_TRY(mem.alloc(s, sizeof(uint32_t) * (1 + 2 * (size_t) blocks)));
        T*    out     = mem.as<T>();
        T*    partial = out + 1;
        uint* ok      = reinterpret_cast<uint*>(partial + blocks);
        MailboxTicket mt;
        bool          posted = false;
        if (fold_is_reorderable(op, TypeId<T>::value) && binop_is_commutative(op)) {
            const char* pe = knob_env("SPLAHIP_SCAN_POST");// (the knob of the compactions' posted counts: scan_post.hpp)
            posted         = !(pe && pe[0] == '0') && mailbox_begin(&mt) == hipSuccess;
            hipLaunchKernelGGL(reduce_partial_kernel<T>, dim3(blocks), dim3(256), 0, s, n, static_cast<const T*>(x), partial, ok, DynBin(op));
            hipLaunchKernelGGL(reduce_final_kernel<T>, dim3(1), dim3(256), 0, s, blocks, partial, ok, from_bits<T>(s_bits), out, DynBin(op), posted ? mt.dev : nullptr, mt.seq);
        } else {
            hipLaunchKernelGGL(reduce_sequential_kernel<T>, dim3(1), dim3(64), 0, s, n, static_cast<const T*>(x), from_bits<T>(s_bits), out, DynBin(op));
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = posted ? mailbox_end(h_result, 1, mt, s) : fetch_words(h_result, out, 1, s);
        if (e != hipSuccess) return fail(SPLAHIP_ERROR, "reduce failed: %s", hipGetErrorString(e));
        return SPLAHIP_OK;
    }

    // ---- user-defined ops (jit_hooks.hpp): the drivers above with the op-dependent kernel replaced by a run-time-compiled one --------------------------
    int eadd_fdb_sparse_jit(uint32_t nvals, void* r, const uint32_t* vi, const void* vx, uint32_t* fi, void* fx, uint32_t* h_count, hipFunction_t apply, hipStream_t s) {
        *h_count = 0;
        if (nvals == 0) return SPLAHIP_OK;
        const uint n_chunks = (nvals + 1023u) / 1024u;
        DevMem     mem;// changed[nvals] | newval[nvals] | chunk[n_chunks + 1]
        SPLAHIP_TRY(mem.alloc(s, sizeof(uint) * (2 * (size_t) nvals + n_chunks + 1)));
        uint* changed = mem.as<uint>();
        uint* newval  = changed + nvals;
        uint* chunk   = newval + nvals;
        uint  n       = nvals;
        void* args[]  = {&n, &r, &vi, &vx, &changed, &newval};
        SPLAHIP_TRY(hipModuleLaunchKernel(apply, (nvals + 255u) / 256u, 1, 1, 256, 1, 1, 0, s, args, nullptr));
        hipLaunchKernelGGL(changed_count_kernel, dim3(n_chunks), dim3(256), 0, s, changed, nvals, chunk);
        ScanPost sp;
        (void) scan_launch(chunk, n_chunks, s, &sp);// (the count reaches the host from the scan: scan_post.hpp)
        hipLaunchKernelGGL(changed_write_kernel, dim3(n_chunks), dim3(256), 0, s, changed, vi, newval, nvals, chunk, fi, static_cast<uint*>(fx));
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = scan_total(h_count, chunk + n_chunks, sp, s);
        if (e != hipSuccess) return fail(SPLAHIP_ERROR, "v_eadd_fdb (sparse, user-defined op) failed: %s", hipGetErrorString(e));
        return SPLAHIP_OK;
    }

    int eadd_sparse_jit(int type, uint32_t nu, const uint32_t* ui, const void* ux, uint32_t u_fill, uint32_t nv, const uint32_t* vi, const void* vx, uint32_t v_fill,
                        uint32_t* ri, void* rx, uint32_t* h_count, hipFunction_t combine, hipStream_t s) {
        *h_count = 0;
        if (nu + nv == 0) return SPLAHIP_OK;
        DevMem mem;// a[nu + nv] | b[nu + nv]: the left and the right operand of every output entry
        SPLAHIP_TRY(mem.alloc(s, sizeof(uint) * 2 * ((size_t) nu + nv)));
        uint* a  = mem.as<uint>();
        uint* b  = a + nu + nv;
        int   st = SPLAHIP_OK;
        uint  count = 0;
        for (int side = 0; side < 2 && st == SPLAHIP_OK; ++side) {
            const int pick = side == 0 ? SPLAHIP_FIRST : SPLAHIP_SECOND;
            void*     dst  = side == 0 ? a : b;
            switch (type) {// (the values are moved, never computed with: the word-sized types share one instantiation's behaviour)
                case SPLAHIP_FLOAT: st = eadd_sparse<float>(nu, ui, ux, u_fill, nv, vi, vx, v_fill, ri, dst, &count, pick, s); break;
                case SPLAHIP_INT: st = eadd_sparse<int32_t>(nu, ui, ux, u_fill, nv, vi, vx, v_fill, ri, dst, &count, pick, s); break;
                default: st = eadd_sparse<uint32_t>(nu, ui, ux, u_fill, nv, vi, vx, v_fill, ri, dst, &count, pick, s); break;
            }
        }
        if (st != SPLAHIP_OK) return st;
        if (count) {
            void* args[] = {&count, &rx, &a, &b};
            SPLAHIP_TRY(hipModuleLaunchKernel(combine, (count + 255u) / 256u, 1, 1, 256, 1, 1, 0, s, args, nullptr));
            SPLAHIP_TRY(hipStreamSynchronize(s));// (the scratch arrays go with this scope)
        }
        *h_count = count;
        return SPLAHIP_OK;
    }

}// namespace splahip


using namespace splahip;

#define SPLAHIP_BY_TYPE(type, CALL_I, CALL_U, CALL_F)                   \
    switch (type) {                                                     \
        case SPLAHIP_INT: CALL_I; break;                                \
        case SPLAHIP_UINT: CALL_U; break;                               \
        case SPLAHIP_FLOAT: CALL_F; break;                              \
        default: return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type"); \
    }

extern "C" {

int splahip_v_assign_masked_dense(int type, uint32_t n, void* d_r, const void* d_mask, uint32_t value_bits, int op_assign, int op_select, void* stream) {
    SPLAHIP_REQUIRE(binop_valid_for(op_assign, type) && op_select >= 0 && op_select < SPLAHIP_SELOP_COUNT, "op not defined for this type");
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_r && d_mask, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    const dim3 grid((n + 255) / 256), block(256);
    hipStream_t s = as_stream(stream);
    SPLAHIP_BY_TYPE(type,
                    hipLaunchKernelGGL(assign_dense_kernel<int32_t>, grid, block, 0, s, n, static_cast<int32_t*>(d_r), static_cast<const int32_t*>(d_mask), from_bits<int32_t>(value_bits), DynBin(op_assign), op_select),
                    hipLaunchKernelGGL(assign_dense_kernel<uint32_t>, grid, block, 0, s, n, static_cast<uint32_t*>(d_r), static_cast<const uint32_t*>(d_mask), from_bits<uint32_t>(value_bits), DynBin(op_assign), op_select),
                    hipLaunchKernelGGL(assign_dense_kernel<float>, grid, block, 0, s, n, static_cast<float*>(d_r), static_cast<const float*>(d_mask), from_bits<float>(value_bits), DynBin(op_assign), op_select))
    SPLAHIP_TRY(hipGetLastError());
    return SPLAHIP_OK;
}

int splahip_v_assign_masked_sparse(int type, uint32_t nvals, void* d_r, const uint32_t* d_mi, const void* d_mx, uint32_t value_bits, int op_assign,
                                   int op_select, void* stream) {
    SPLAHIP_REQUIRE(binop_valid_for(op_assign, type) && op_select >= 0 && op_select < SPLAHIP_SELOP_COUNT, "op not defined for this type");
    if (nvals == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_r && d_mi && d_mx, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    const dim3 grid((nvals + 255) / 256), block(256);
    hipStream_t s = as_stream(stream);
    SPLAHIP_BY_TYPE(type,
                    hipLaunchKernelGGL(assign_sparse_kernel<int32_t>, grid, block, 0, s, nvals, static_cast<int32_t*>(d_r), d_mi, static_cast<const int32_t*>(d_mx), from_bits<int32_t>(value_bits), DynBin(op_assign), op_select),
                    hipLaunchKernelGGL(assign_sparse_kernel<uint32_t>, grid, block, 0, s, nvals, static_cast<uint32_t*>(d_r), d_mi, static_cast<const uint32_t*>(d_mx), from_bits<uint32_t>(value_bits), DynBin(op_assign), op_select),
                    hipLaunchKernelGGL(assign_sparse_kernel<float>, grid, block, 0, s, nvals, static_cast<float*>(d_r), d_mi, static_cast<const float*>(d_mx), from_bits<float>(value_bits), DynBin(op_assign), op_select))
    SPLAHIP_TRY(hipGetLastError());
    return SPLAHIP_OK;
}

int splahip_v_count_mf_dense(int type, uint32_t n, const void* d_v, uint32_t fill_bits, uint32_t* h_count, void* stream) {
    SPLAHIP_REQUIRE(h_count != nullptr, "h_count is null");
    *h_count = 0;
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_v, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t         s = as_stream(stream);
    DevMem              mem;
    SPLAHIP_TRY(mem.alloc(s, sizeof(unsigned long long)));
    unsigned long long* d_count = mem.as<unsigned long long>();
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s);
    const dim3 grid(std::min<uint32_t>((n + 255) / 256, 512)), block(256);
    if (e == hipSuccess) {
        switch (type) {
            case SPLAHIP_INT: hipLaunchKernelGGL(count_mf_kernel<int32_t>, grid, block, 0, s, n, static_cast<const int32_t*>(d_v), from_bits<int32_t>(fill_bits), d_count); break;
            case SPLAHIP_UINT: hipLaunchKernelGGL(count_mf_kernel<uint32_t>, grid, block, 0, s, n, static_cast<const uint32_t*>(d_v), from_bits<uint32_t>(fill_bits), d_count); break;
            default: hipLaunchKernelGGL(count_mf_kernel<float>, grid, block, 0, s, n, static_cast<const float*>(d_v), from_bits<float>(fill_bits), d_count); break;
        }
        e = hipGetLastError();
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = fetch_words(&h, d_count, 2, s);
    if (e != hipSuccess) return fail(SPLAHIP_ERROR, "v_count_mf failed: %s", hipGetErrorString(e));
    *h_count = (uint32_t) h;
    return SPLAHIP_OK;
}

int splahip_v_eadd_fdb_dense(int type, uint32_t n, void* d_r, const void* d_v, void* d_fdb, uint32_t fdb_fill_bits, int op, void* stream) {
    SPLAHIP_REQUIRE(binop_valid_for(op, type), "op not defined for this type");
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_r && d_v && d_fdb, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    const dim3 grid((n + 255) / 256), block(256);
    hipStream_t s = as_stream(stream);
    SPLAHIP_BY_TYPE(type,
                    hipLaunchKernelGGL(eadd_fdb_dense_kernel<int32_t>, grid, block, 0, s, n, static_cast<int32_t*>(d_r), static_cast<const int32_t*>(d_v), static_cast<int32_t*>(d_fdb), from_bits<int32_t>(fdb_fill_bits), DynBin(op)),
                    hipLaunchKernelGGL(eadd_fdb_dense_kernel<uint32_t>, grid, block, 0, s, n, static_cast<uint32_t*>(d_r), static_cast<const uint32_t*>(d_v), static_cast<uint32_t*>(d_fdb), from_bits<uint32_t>(fdb_fill_bits), DynBin(op)),
                    hipLaunchKernelGGL(eadd_fdb_dense_kernel<float>, grid, block, 0, s, n, static_cast<float*>(d_r), static_cast<const float*>(d_v), static_cast<float*>(d_fdb), from_bits<float>(fdb_fill_bits), DynBin(op)))
    SPLAHIP_TRY(hipGetLastError());
    return SPLAHIP_OK;
}

int splahip_v_eadd_fdb_sparse(int type, uint32_t nvals, void* d_r, const uint32_t* d_vi, const void* d_vx, uint32_t* d_fi, void* d_fx,
                              uint32_t* h_count, int op, void* stream) {
    SPLAHIP_REQUIRE(h_count != nullptr, "h_count is null");
    SPLAHIP_REQUIRE(binop_valid_for(op, type), "op not defined for this type");
    *h_count = 0;
    if (nvals == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_r && d_vi && d_vx && d_fi && d_fx, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    switch (type) {
        case SPLAHIP_INT: return eadd_fdb_sparse<int32_t>(nvals, d_r, d_vi, d_vx, d_fi, d_fx, h_count, op, s);
        case SPLAHIP_UINT: return eadd_fdb_sparse<uint32_t>(nvals, d_r, d_vi, d_vx, d_fi, d_fx, h_count, op, s);
        case SPLAHIP_FLOAT: return eadd_fdb_sparse<float>(nvals, d_r, d_vi, d_vx, d_fi, d_fx, h_count, op, s);
    }
    return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
}

int splahip_v_eadd_dense(int type, uint32_t n, void* d_r, const void* d_u, const void* d_v, int op, void* stream) {
    SPLAHIP_REQUIRE(binop_valid_for(op, type), "op not defined for this type");
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_r && d_u && d_v, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    const dim3 grid((n + 255) / 256), block(256);
    hipStream_t s = as_stream(stream);
    SPLAHIP_BY_TYPE(type,
                    hipLaunchKernelGGL(eadd_dense_kernel<int32_t>, grid, block, 0, s, n, static_cast<int32_t*>(d_r), static_cast<const int32_t*>(d_u), static_cast<const int32_t*>(d_v), DynBin(op)),
                    hipLaunchKernelGGL(eadd_dense_kernel<uint32_t>, grid, block, 0, s, n, static_cast<uint32_t*>(d_r), static_cast<const uint32_t*>(d_u), static_cast<const uint32_t*>(d_v), DynBin(op)),
                    hipLaunchKernelGGL(eadd_dense_kernel<float>, grid, block, 0, s, n, static_cast<float*>(d_r), static_cast<const float*>(d_u), static_cast<const float*>(d_v), DynBin(op)))
    SPLAHIP_TRY(hipGetLastError());
    return SPLAHIP_OK;
}

int splahip_v_eadd_sparse(int type, uint32_t nu, const uint32_t* d_ui, const void* d_ux, uint32_t u_fill_bits, uint32_t nv, const uint32_t* d_vi,
                          const void* d_vx, uint32_t v_fill_bits, uint32_t* d_ri, void* d_rx, uint32_t* h_count, int op, void* stream) {
    SPLAHIP_REQUIRE(h_count != nullptr, "h_count is null");
    SPLAHIP_REQUIRE(binop_valid_for(op, type), "op not defined for this type");
    *h_count = 0;
    if (nu == 0 && nv == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE((nu == 0 || (d_ui && d_ux)) && (nv == 0 || (d_vi && d_vx)) && d_ri && d_rx, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    switch (type) {
        case SPLAHIP_INT: return eadd_sparse<int32_t>(nu, d_ui, d_ux, u_fill_bits, nv, d_vi, d_vx, v_fill_bits, d_ri, d_rx, h_count, op, s);
        case SPLAHIP_UINT: return eadd_sparse<uint32_t>(nu, d_ui, d_ux, u_fill_bits, nv, d_vi, d_vx, v_fill_bits, d_ri, d_rx, h_count, op, s);
        case SPLAHIP_FLOAT: return eadd_sparse<float>(nu, d_ui, d_ux, u_fill_bits, nv, d_vi, d_vx, v_fill_bits, d_ri, d_rx, h_count, op, s);
    }
    return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
}

int splahip_reduce(int type, uint32_t n, const void* d_x, uint32_t s_bits, uint32_t* h_result_bits, int op, void* stream) {
    SPLAHIP_REQUIRE(h_result_bits != nullptr, "h_result_bits is null");
    SPLAHIP_REQUIRE(binop_valid_for(op, type), "op not defined for this type");
    *h_result_bits = s_bits;
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_x != nullptr && (reinterpret_cast<uintptr_t>(d_x) & 15u) == 0, "d_x must be a 16-byte aligned device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    switch (type) {
        case SPLAHIP_INT: return reduce_typed<int32_t>(n, d_x, s_bits, h_result_bits, op, s);
        case SPLAHIP_UINT: return reduce_typed<uint32_t>(n, d_x, s_bits, h_result_bits, op, s);
        case SPLAHIP_FLOAT: return reduce_typed<float>(n, d_x, s_bits, h_result_bits, op, s);
    }
    return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
}
}
