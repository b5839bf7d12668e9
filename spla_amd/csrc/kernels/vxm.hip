// vxm.hip — masked sparse-vector x sparse-matrix, r = v^T * M restricted to columns j with select(mask[j]).
//
// Semantics: reference Algo_vxm_masked_cpu<T>::execute, src/cpu/cpu_vxm.hpp:58-128 (the parity oracle): for every
// stored entry (i, v_i) of v IN STORED ORDER, for every (j, x) of row i: the first hit of column j stores mult(v_i, x),
// later hits add(r_j, mult(v_i, x)); `init` is not used; output = index-sorted COO with explicit zeros kept.
// Entry point replaced: Algo_vxm_masked_cl<T>::execute, src/opencl/cl_vxm.hpp:60-173 (count / collect / sort /
// reduce-by-key) — not ported.  Two forms, chosen by the host algorithm from the product count:
//
//   pull  (dense-ish v, e.g. the PageRank step): masked SpMV over the CSR of M^T with the wave-tile kernel in VXM mode
//         (mxv_kernels.hpp): column j folds its entries in ascending i, which is v's stored order whenever v's COO is
//         index-sorted (always, unless the caller build()s an unsorted vector with a non-commutative op_add, in which
//         case the host picks push).  HBM-bound: 8*nnz(M) + 12*n_cols + 4*n_rows bytes.
//   push  (sparse frontier): expand the products of the frontier rows in (stored order of v, row order), STABLE radix sort
//         by column (rocPRIM), fold every column's run in order, compact.  Exact for every op, including fp32 PLUS and
//         non-commutative op_add, because the run order is the reference's fold order.
#include "block_utils.hpp"
#include "common.hpp"
#include "scan_post.hpp"
#include "jit_hooks.hpp"
#include "mxv_first_hit.hpp"
#include "mxv_kernels.hpp"
#include "ops.hpp"
#include "plan.hpp"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <rocprim/device/device_radix_sort.hpp>

namespace splahip {

    using Cfg = MxvCfg;

    // ------------------------------------------------------------------------------------------------------------
    // small device utilities
    // ------------------------------------------------------------------------------------------------------------

    __host__ __device__ inline uint bit_width_u32(uint x) {
        uint b = 0;
        while (x) {
            ++b;
            x >>= 1;
        }
        return b ? b : 1;
    }

    // per-chunk (1024 elements) sums of x[0..n)
    __global__ __launch_bounds__(256) void chunk_sum_kernel(const uint* __restrict__ x, uint n, uint* __restrict__ chunk) {
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
    // out[i] = chunk_offset + exclusive prefix inside the chunk; out[n] = total
    __global__ __launch_bounds__(256) void chunk_scan_apply_kernel(const uint* __restrict__ x, uint n, const uint* __restrict__ chunk,
                                                                   uint n_chunks, uint* __restrict__ out) {
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

    // exclusive scan of n uint32 into out[n+1] (out[n] = total); x and out may not alias
    static hipError_t exclusive_scan_u32(const uint* x, uint n, uint* out, hipStream_t s) {
        const uint n_chunks = (n + 1023u) / 1024u;
        DevMem     chunk;
        hipError_t e = chunk.alloc(s, sizeof(uint) * (size_t) (n_chunks + 1));
        if (e != hipSuccess) return e;
        if (n == 0) return hipMemsetAsync(out, 0, sizeof(uint), s);
        hipLaunchKernelGGL(chunk_sum_kernel, dim3(n_chunks), dim3(256), 0, s, x, n, chunk.as<uint>());
        hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, chunk.as<uint>(), n_chunks);
        hipLaunchKernelGGL(chunk_scan_apply_kernel, dim3(n_chunks), dim3(256), 0, s, x, n, chunk.as<uint>(), n_chunks, out);
        return hipGetLastError();// (no synchronisation: `chunk` is released in stream order)
    }

    // last i in [0, n) with a[i] <= k, for a[0] <= k < a[n]
    __device__ __forceinline__ uint upper_row(const uint* __restrict__ a, uint n, uint k) {
        uint lo = 0, hi = n;
        while (hi - lo > 1) {
            const uint mid = lo + ((hi - lo) >> 1);
            if (a[mid] <= k) lo = mid;
            else
                hi = mid;
        }
        return lo;
    }

    // ------------------------------------------------------------------------------------------------------------
    // transpose (CSR of M^T, rows ascending inside every column)
    // ------------------------------------------------------------------------------------------------------------
    // ------------------------------------------------------------------------------------------------------------
    // probe: number of products of the frontier and whether v's indices are strictly ascending
    // ------------------------------------------------------------------------------------------------------------
    __global__ __launch_bounds__(256) void probe_kernel(uint nvals, const uint* __restrict__ vi, const uint* __restrict__ Ap,
                                                        unsigned long long* __restrict__ products, uint* __restrict__ unsorted) {
        const uint         idx = blockIdx.x * 256 + threadIdx.x;
        unsigned long long len = 0;
        uint               bad = 0;
        if (idx < nvals) {
            const uint i = vi[idx];
            len          = Ap[i + 1] - Ap[i];
            if (idx > 0 && vi[idx - 1] >= i) bad = 1;
        }
        for (uint off = kWave / 2; off > 0; off >>= 1) {
            len += __shfl_xor(len, off, kWave);
            bad |= __shfl_xor(bad, off, kWave);
        }
        if (lane_id() == 0) {
            if (len) atomicAdd(products, len);
            if (bad) atomicOr(unsorted, 1u);
        }
    }

    // vxm has no `init`: a column's fold STARTS at its first product (reference src/cpu/cpu_vxm.hpp:101-107).  Under fp32 MIN / MAX a NaN first
    // product therefore is the result (`b < NaN` is false for every later b), while a NaN anywhere else is never picked.  The re-associating
    // kernels drop every NaN (Reassoc, ops.hpp); the first product's rule is applied on top by whoever knows which product came first.
    template<class Add, typename T>
    __device__ __forceinline__ bool lead_nan_sticks(const Add& add, T first) {
        if constexpr (std::is_same<T, float>::value) {
            const int op = binop_id(add);
            return (op == SPLAHIP_MIN || op == SPLAHIP_MAX) && first != first;
        }
        return false;
    }

    // ------------------------------------------------------------------------------------------------------------
    // pull helpers
    // ------------------------------------------------------------------------------------------------------------
    // fp32 MIN / MAX behind a pull kernel: one lane per touched output column re-evaluates the column's first product (the first present entry of
    // row j of the transposed matrix: v's indices ascend in the pull form, so that is the reference's first) and lets a NaN there stick
    template<class Mult>
    __global__ __launch_bounds__(256) void pull_lead_nan_kernel(uint n_out, const uint* __restrict__ Tp, const uint* __restrict__ Tj, const uint* __restrict__ Tx,
                                                                const uint* __restrict__ present, const float* __restrict__ v, Mult mult,
                                                                float* __restrict__ outv, const uint* __restrict__ touched) {
        const uint j = blockIdx.x * 256 + threadIdx.x;
        if (j >= n_out || !touched[j]) return;
        const uint e = Tp[j + 1];
        for (uint k = Tp[j]; k < e; ++k) {
            const uint i = Tj[k];
            if (present != nullptr && ((present[i >> 5] >> (i & 31u)) & 1u) == 0) continue;
            const float p = mult(v[i], from_bits<float>(Tx[k]));
            if (p != p) outv[j] = p;
            break;
        }
    }
    __global__ __launch_bounds__(256) void scatter_present_kernel(uint nvals, const uint* __restrict__ vi, const uint* __restrict__ vx,
                                                                  uint* __restrict__ dense, uint* __restrict__ present) {
        const uint idx = blockIdx.x * 256 + threadIdx.x;
        if (idx >= nvals) return;
        const uint i = vi[idx];
        dense[i]     = vx[idx];
        atomicOr(&present[i >> 5], 1u << (i & 31u));
    }

    __global__ __launch_bounds__(256) void flag_count_kernel(const uint* __restrict__ flag, uint n, uint* __restrict__ chunk) {
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
    __global__ __launch_bounds__(256) void flag_write_kernel(const uint* __restrict__ flag, const uint* __restrict__ val, uint n,
                                                             const uint* __restrict__ chunk, uint* __restrict__ out_i, uint* __restrict__ out_x) {
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
                out_i[pos] = first + u;
                out_x[pos] = val[first + u];
                ++pos;
            }
    }

    // ------------------------------------------------------------------------------------------------------------
    // push (expand / stable sort / fold runs / compact)
    // ------------------------------------------------------------------------------------------------------------
    __global__ __launch_bounds__(256) void frontier_len_kernel(uint nvals, const uint* __restrict__ vi, const uint* __restrict__ Ap, uint* __restrict__ len) {
        const uint idx = blockIdx.x * 256 + threadIdx.x;
        if (idx < nvals) len[idx] = Ap[vi[idx] + 1] - Ap[vi[idx]];
    }

    template<typename T, class Mult>
    __global__ __launch_bounds__(256) void expand_kernel(uint n_products, uint nvals, const uint* __restrict__ offs,
                                                         const uint* __restrict__ vi, const T* __restrict__ vx,
                                                         const uint* __restrict__ Ap, const uint* __restrict__ Aj, const uint* __restrict__ Ax,
                                                         const T* __restrict__ mask, uint n_cols, int op_select, Mult mult,
                                                         uint* __restrict__ key, T* __restrict__ val) {
        const uint p = blockIdx.x * 256 + threadIdx.x;
        if (p >= n_products) return;
        const uint idx = upper_row(offs, nvals, p);
        const uint k   = Ap[vi[idx]] + (p - offs[idx]);
        const uint j   = Aj[k];
        const bool sel = eval_sel(op_select, mask[j]);
        key[p]         = sel ? j : n_cols;// unselected products sort behind every column
        val[p]         = mult(vx[idx], from_bits<T>(Ax[k]));
    }

    // ---- push, order-free form ---------------------------------------------------------------------------------------
    // For an op_add whose result does not depend on the fold order (associative + commutative, and not a rounding fp32 sum / product)
    // "first hit stores, later hits add" is just a reduction per output column: every product goes straight into a packed
    // (touched, value) word per column with a compare-and-swap — no expand buffer, no sort.  Used for small product counts, where the
    // sort form's ~15 launches are the cost (BFS / SSSP levels with small frontiers); hub columns would serialise a large expansion.
    using u64 = unsigned long long;
    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(256) void push_atomic_kernel(uint n_products, uint nvals, const uint* __restrict__ offs, const uint* __restrict__ vi,
                                                              const T* __restrict__ vx, const uint* __restrict__ Ap, const uint* __restrict__ Aj,
                                                              const uint* __restrict__ Ax, const T* __restrict__ mask, int op_select, Mult mult, Add add,
                                                              u64* __restrict__ acc, uint* __restrict__ nan_seen) {
        const uint p = blockIdx.x * 256 + threadIdx.x;
        if (p >= n_products) return;
        const uint idx = upper_row(offs, nvals, p);
        const uint k   = Ap[vi[idx]] + (p - offs[idx]);
        const uint j   = Aj[k];
        if (!eval_sel(op_select, mask[j])) return;
        const T prod = mult(vx[idx], from_bits<T>(Ax[k]));
        if (lead_nan_sticks(add, prod)) *nan_seen = 1u;// fp32 MIN / MAX: whether a NaN sticks depends on the fold order -> the host redoes the call in the sort form
        u64*    slot = acc + j;
        u64     seen = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (true) {
            const bool touched = (seen >> 32) != 0;
            const T    next    = touched ? add(from_bits<T>((uint) seen), prod) : prod;
            const u64  want    = (1ull << 32) | (u64) to_bits(next);
            if (want == seen) break;// nothing to change (e.g. OR-ing a bit that is set, MIN with a larger value)
            const u64 prev = atomicCAS(slot, seen, want);
            if (prev == seen) break;
            seen = prev;
        }
    }
    // The same expansion with the product -> frontier-entry search taken off the critical chain.  part[b] = the frontier entry that holds product 256 b
    // (push_part_kernel: one 20-step search per WORKGROUP of the expansion, all of them in one small launch); the workgroup then brings the offsets of its
    // entries [part[b], part[b + 1]] into LDS with one coalesced load and every thread searches THERE.  A thread of push_atomic_kernel walked ~26 dependent
    // memory operations (20 of them the search over `offs` in L2); here ~8.  Workgroups whose 256 products span more than kPushSpan entries (runs of empty rows)
    // keep the global search.
    constexpr uint kPushSpan = 768;
    __global__ __launch_bounds__(256) void push_part_kernel(uint n_blocks, uint nvals, const uint* __restrict__ offs, uint* __restrict__ part) {
        const uint b = blockIdx.x * 256 + threadIdx.x;
        if (b < n_blocks) part[b] = upper_row(offs, nvals, b * 256u);
        if (b == n_blocks) part[b] = nvals - 1u;
    }
    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(256) void push_atomic_blocked_kernel(uint n_products, uint nvals, const uint* __restrict__ offs, const uint* __restrict__ part,
                                                                      const uint* __restrict__ vi, const T* __restrict__ vx, const uint* __restrict__ Ap,
                                                                      const uint* __restrict__ Aj, const uint* __restrict__ Ax, const T* __restrict__ mask, int op_select,
                                                                      Mult mult, Add add, u64* __restrict__ acc, uint* __restrict__ nan_seen) {
        __shared__ uint s_offs[kPushSpan];
        const uint      r_lo = part[blockIdx.x], r_hi = part[blockIdx.x + 1], n_r = r_hi - r_lo + 1u;
        const bool      local = n_r <= kPushSpan;// (uniform)
        if (local)
            for (uint i = threadIdx.x; i < n_r; i += 256) s_offs[i] = offs[r_lo + i];
        __syncthreads();
        const uint p = blockIdx.x * 256 + threadIdx.x;
        if (p >= n_products) return;
        uint idx;
        if (local) {
            uint lo = 0, hi = n_r;// largest i with s_offs[i] <= p (s_offs[0] <= 256 b <= p)
            while (hi - lo > 1) {
                const uint mid = lo + ((hi - lo) >> 1);
                if (s_offs[mid] <= p) lo = mid;
                else
                    hi = mid;
            }
            idx = r_lo + lo;
        } else
            idx = upper_row(offs, nvals, p);
        const uint k = Ap[vi[idx]] + (p - (local ? s_offs[idx - r_lo] : offs[idx]));
        const uint j = Aj[k];
        if (!eval_sel(op_select, mask[j])) return;
        const T prod = mult(vx[idx], from_bits<T>(Ax[k]));
        if (lead_nan_sticks(add, prod)) *nan_seen = 1u;// fp32 MIN / MAX: whether a NaN sticks depends on the fold order -> the host redoes the call in the sort form
        u64*    slot = acc + j;
        u64     seen = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (true) {
            const bool touched = (seen >> 32) != 0;
            const T    next    = touched ? add(from_bits<T>((uint) seen), prod) : prod;
            const u64  want    = (1ull << 32) | (u64) to_bits(next);
            if (want == seen) break;// nothing to change (e.g. OR-ing a bit that is set, MIN with a larger value)
            const u64 prev = atomicCAS(slot, seen, want);
            if (prev == seen) break;
            seen = prev;
        }
    }
    __global__ __launch_bounds__(256) void acc_count_kernel(const u64* __restrict__ acc, uint n, uint* __restrict__ chunk) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            cnt   = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u)
            if (first + u < n) cnt += (acc[first + u] >> 32) ? 1u : 0u;
        uint total;
        (void) block_exclusive_sum<256>(cnt, s_tmp, total);
        if (threadIdx.x == 0) chunk[blockIdx.x] = total;
    }
    __global__ __launch_bounds__(256) void acc_write_kernel(const u64* __restrict__ acc, uint n, const uint* __restrict__ chunk, uint* __restrict__ out_i,
                                                            uint* __restrict__ out_x) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        u64             w[4];
        uint            cnt = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            w[u] = first + u < n ? acc[first + u] : 0ull;
            cnt += (w[u] >> 32) ? 1u : 0u;
        }
        uint total;
        uint pos = chunk[blockIdx.x] + block_exclusive_sum<256>(cnt, s_tmp, total);
#pragma unroll
        for (uint u = 0; u < 4; ++u)
            if (w[u] >> 32) {
                out_i[pos] = first + u;
                out_x[pos] = (uint) w[u];
                ++pos;
            }
    }

    // ---- push, small frontiers: probe + expansion + compaction back to back, ONE host synchronisation ---------------------------
    // gate[0..1] = product count (64 bit), gate[2] = "indices not ascending" flag: written by probe_kernel; the kernels behind it
    // run only when the count is within `max_products` (the host then reads count + gate in one copy).  A workgroup per frontier
    // entry walks that row of M and folds the products into the packed (touched, value) words with compare-and-swap; the words
    // live in a cached workspace that is all-zero between calls: the write kernel clears every word it emits.
    __device__ __forceinline__ bool small_gate_open(const uint* __restrict__ gate, uint max_products) {
        return gate[1] == 0u && gate[0] != 0u && gate[0] <= max_products;
    }
    template<typename T, class Mult, class Add>
    __global__ __launch_bounds__(256) void push_direct_kernel(const uint* __restrict__ gate, uint max_products, const uint* __restrict__ vi,
                                                              const T* __restrict__ vx, const uint* __restrict__ Ap, const uint* __restrict__ Aj,
                                                              const uint* __restrict__ Ax, const T* __restrict__ mask, int op_select, Mult mult, Add add,
                                                              u64* __restrict__ acc) {
        // gate[4]: a NaN product under fp32 MIN / MAX was seen — whether it sticks depends on the fold order, the host discards this call's result
        if (!small_gate_open(gate, max_products)) return;
        // blockIdx.y: the row is shared by gridDim.y workgroups (a lone hub of 64 K entries would otherwise be one workgroup's 250 rounds)
        const uint row = vi[blockIdx.x];
        const T    x   = vx[blockIdx.x];
        const uint k1  = Ap[row + 1];
        for (uint k = Ap[row] + blockIdx.y * 256 + threadIdx.x; k < k1; k += gridDim.y * 256) {
            const uint j = Aj[k];
            if (!eval_sel(op_select, mask[j])) continue;
            const T prod = mult(x, from_bits<T>(Ax[k]));
            if (lead_nan_sticks(add, prod)) const_cast<uint*>(gate)[4] = 1u;
            u64*    slot = acc + j;
            u64     seen = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (true) {
                const bool touched = (seen >> 32) != 0;
                const T    next    = touched ? add(from_bits<T>((uint) seen), prod) : prod;
                const u64  want    = (1ull << 32) | (u64) to_bits(next);
                if (want == seen) break;
                const u64 prev = atomicCAS(slot, seen, want);
                if (prev == seen) break;
                seen = prev;
            }
        }
    }
    __global__ __launch_bounds__(256) void acc_count_gated_kernel(const uint* __restrict__ gate, uint max_products, const u64* __restrict__ acc, uint n,
                                                                  uint* __restrict__ chunk) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const bool      open  = small_gate_open(gate, max_products);
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            cnt   = 0;
        if (open) {
#pragma unroll
            for (uint u = 0; u < 4; ++u)
                if (first + u < n) cnt += (acc[first + u] >> 32) ? 1u : 0u;
        }
        uint total;
        (void) block_exclusive_sum<256>(cnt, s_tmp, total);
        if (threadIdx.x == 0) chunk[blockIdx.x] = total;
    }
    // emits the touched columns in ascending order and returns their words to zero
    __global__ __launch_bounds__(256) void acc_write_reset_kernel(const uint* __restrict__ gate, uint max_products, u64* __restrict__ acc, uint n,
                                                                  const uint* __restrict__ chunk, uint* __restrict__ out_i, uint* __restrict__ out_x) {
        __shared__ uint s_tmp[256 / kWave + 1];
        if (!small_gate_open(gate, max_products)) return;
        const uint first = blockIdx.x * 1024u + threadIdx.x * 4u;
        u64        w[4];
        uint       cnt = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            w[u] = first + u < n ? acc[first + u] : 0ull;
            cnt += (w[u] >> 32) ? 1u : 0u;
        }
        uint total;
        uint pos = chunk[blockIdx.x] + block_exclusive_sum<256>(cnt, s_tmp, total);
        if (blockIdx.x == 0 && threadIdx.x == 0) const_cast<uint*>(gate)[3] = chunk[gridDim.x];// the total, next to the gate: one read-back
#pragma unroll
        for (uint u = 0; u < 4; ++u)
            if (w[u] >> 32) {
                out_i[pos]     = first + u;
                out_x[pos]     = (uint) w[u];
                acc[first + u] = 0ull;
                ++pos;
            }
    }

    __global__ __launch_bounds__(256) void head_count_kernel(const uint* __restrict__ key, uint n, uint n_cols, uint* __restrict__ chunk) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            cnt   = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            const uint p = first + u;
            if (p < n && key[p] < n_cols && (p == 0 || key[p - 1] != key[p])) cnt += 1;
        }
        uint total;
        (void) block_exclusive_sum<256>(cnt, s_tmp, total);
        if (threadIdx.x == 0) chunk[blockIdx.x] = total;
    }

    constexpr uint kRunShort = 64;// runs up to this length are folded by the lane that owns their head

    // every run head: write the column id, fold the run in order if it is short, else queue it for the wave kernel
    template<typename T, class Add>
    __global__ __launch_bounds__(256) void run_fold_kernel(const uint* __restrict__ key, const T* __restrict__ val, uint n, uint n_cols,
                                                           const uint* __restrict__ chunk, uint* __restrict__ out_i, T* __restrict__ out_x,
                                                           uint* __restrict__ long_count, uint* __restrict__ long_pos, uint* __restrict__ long_rank,
                                                           Add add) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        bool            head[4];
        uint            cnt = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            const uint p = first + u;
            head[u]      = p < n && key[p] < n_cols && (p == 0 || key[p - 1] != key[p]);
            cnt += head[u] ? 1u : 0u;
        }
        uint total;
        uint rank = chunk[blockIdx.x] + block_exclusive_sum<256>(cnt, s_tmp, total);
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            if (!head[u]) continue;
            const uint p = first + u;
            const uint j = key[p];
            out_i[rank]  = j;
            T    acc     = val[p];
            uint q       = p + 1;
            while (q < n && q - p < kRunShort && key[q] == j) {
                acc = add(acc, val[q]);
                ++q;
            }
            if (q < n && key[q] == j) {// longer than kRunShort: hand over (the whole run is redone in order)
                const uint slot = atomicAdd(long_count, 1u);
                long_pos[slot]  = p;
                long_rank[slot] = rank;
            } else {
                out_x[rank] = acc;
            }
            ++rank;
        }
    }

    // one wavefront per long run: lane-contiguous chunks, in-order segmented shuffle reduction.  ORDERED = false is
    // only used for non-associative op_add: lane 0 folds the run sequentially.
    template<typename T, class Add, bool TREE>
    __global__ __launch_bounds__(256) void long_run_kernel(const uint* __restrict__ key, const T* __restrict__ val, uint n,
                                                           const uint* __restrict__ long_count, const uint* __restrict__ long_pos,
                                                           const uint* __restrict__ long_rank, T* __restrict__ out_x, Add add_raw) {
        const Reassoc<Add> add(add_raw);// (ops.hpp) lane partials start at products
        const uint w = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
        if (w >= *long_count) return;
        const uint lane = lane_id();
        const uint p    = long_pos[w];
        const uint j    = key[p];
        // end of the run: first q > p with key[q] != j (keys are sorted)
        uint lo = p + 1, hi = n;
        while (lo < hi) {
            const uint mid = lo + ((hi - lo) >> 1);
            if (key[mid] == j) lo = mid + 1;
            else
                hi = mid;
        }
        const uint e = lo;
        if constexpr (TREE) {
            const uint c  = (e - p + kWave - 1) / kWave;
            const uint ks = p + lane * c;
            const uint ke = min(ks + c, e);
            bool       valid = ks < e;
            T          part  = T(0);
            if (valid) {
                part = val[ks];
                for (uint q = ks + 1; q < ke; ++q) part = add(part, val[q]);
            }
            wave_reduce_ordered(part, valid, add);
            if (lane == 0) {
                // a column folds from its FIRST product (cpu_vxm.hpp:101-107): under fp32 MIN / MAX a NaN there sticks (`b < NaN` is false)
                if (lead_nan_sticks(add_raw, val[p])) part = val[p];
                out_x[long_rank[w]] = part;
            }
        } else {
            if (lane == 0) {
                T acc = val[p];
                for (uint q = p + 1; q < e; ++q) acc = add(acc, val[q]);
                out_x[long_rank[w]] = acc;
            }
        }
    }

    // ------------------------------------------------------------------------------------------------------------
    // typed drivers
    // ------------------------------------------------------------------------------------------------------------
    template<typename T, class Mult, class Add>
    int pull_typed(splahip_plan plan, uint n_out, uint n_in, const uint* Tp, const uint* Tj, const void* Tx, const void* mask,
                   uint nvals, const uint* vi, const void* vx, int v_is_identity, uint* ri, void* rx, uint* h_count,
                   int op_mult, int op_add, int op_select, hipStream_t s) {
        // workspace kept with the plan: [dense v | present bitmap | out values | touched flags | chunk counts]
        const uint   words    = (n_in + 31u) / 32u;
        const uint   n_chunks = (n_out + 1023u) / 1024u;
        const uint   n_in4    = (n_in + 3u) & ~3u, words4 = (words + 3u) & ~3u, n_out4 = (n_out + 3u) & ~3u;// every sub-array starts 16-byte aligned
        const size_t need     = sizeof(uint) * ((size_t) n_in4 + words4 + 2 * (size_t) n_out4 + n_chunks + 8);
        if (plan->ws_bytes < need) {
            if (plan->d_ws) (void) hipFree(plan->d_ws);
            plan->d_ws = nullptr, plan->ws_bytes = 0;
            SPLAHIP_TRY(hipMalloc(&plan->d_ws, need));
            plan->ws_bytes = need;
        }
        uint* ws_dense   = static_cast<uint*>(plan->d_ws);
        uint* ws_present = ws_dense + n_in4;
        uint* ws_outv    = ws_present + words4;
        uint* ws_touched = ws_outv + n_out4;
        uint* ws_chunk   = ws_touched + n_out4;

        const uint* d_present = nullptr;
        const T*    d_v       = static_cast<const T*>(vx);
        if (!v_is_identity) {
            SPLAHIP_TRY(hipMemsetAsync(ws_present, 0, sizeof(uint) * (size_t) words, s));
            if (nvals)
                hipLaunchKernelGGL(scatter_present_kernel, dim3((nvals + 255) / 256), dim3(256), 0, s, nvals, vi,
                                   static_cast<const uint*>(vx), ws_dense, ws_present);
            d_present = ws_present;
            d_v       = reinterpret_cast<const T*>(ws_dense);
        }
        if (((reinterpret_cast<uintptr_t>(Tj) | reinterpret_cast<uintptr_t>(Tx)) & 15u) != 0)
            return fail(SPLAHIP_INVALID_ARGUMENT, "the transposed CSR arrays must be 16-byte aligned");
        const uint grid   = (plan->n_items + Cfg::kWaves - 1) / Cfg::kWaves;
        // A traversal's pull level (reference src/algorithm.cpp:93-99: BAND / BOR over an adjacency matrix, a frontier of equal values): every product is one
        // word, so the first stored in-neighbour settles an output entry (mxv_first_hit.hpp, VXM mode) — taken when the owner of the transposed matrix said its
        // values are one word (splahip_csr_plan_set_uniform_values) and the caller's product count (splahip_csr_plan_hint_products; the probe every caller of
        // the pull form has run) says the frontier's rows hold at least 1 / SPLAHIP_PULL_FIRST_HIT_DEN (6) of the entries: a walk then ends within a few entries.
        // The launch checks the stored values itself: a vector of several words is folded completely, in order.
        uint32_t       xbits     = 0;
        const uint64_t products  = plan->products_hint;
        plan->products_hint      = 0;
        bool           first_hit = false;
        if constexpr (!std::is_same<T, float>::value) {
            const char*    de  = knob_env("SPLAHIP_PULL_FIRST_HIT_DEN");
            const uint64_t den = de && *de ? std::strtoull(de, nullptr, 10) : 6ull;
            const char*    fe  = knob_env("SPLAHIP_FIRST_HIT");
            first_hit = !v_is_identity && nvals != 0 && !(fe && fe[0] == '0') && first_hit_family(TypeId<T>::value, op_mult, op_add) && uniform_values(plan, n_in, &xbits) &&
                        products != 0 && (den == 0 || products * den >= plan->nnz);
        }
        if (first_hit) {
            if constexpr (!std::is_same<T, float>::value) {
                uint* uf = ws_chunk + n_chunks + 2;// two spare words of the workspace: max(~word), max(word) over the stored values
                SPLAHIP_TRY(hipMemsetAsync(uf, 0, 2 * sizeof(uint), s));
                hipLaunchKernelGGL((vx_range_kernel<0>), dim3(std::min(64u, (nvals + 1023u) / 1024u)), dim3(256), 0, s, nvals, static_cast<const uint*>(vx), uf);
                if (op_mult == SPLAHIP_BAND && op_add == SPLAHIP_BOR)
                    hipLaunchKernelGGL((mxv_first_hit_kernel<T, StaBin<SPLAHIP_BAND>, StaBin<SPLAHIP_BOR>, true>), dim3((n_out + 255) / 256), dim3(256), 0, s, n_out, Tp, Tj,
                                       static_cast<const T*>(mask), d_v, reinterpret_cast<T*>(ws_outv), StaBin<SPLAHIP_BAND>(), StaBin<SPLAHIP_BOR>(), op_select, T(0), xbits, uf,
                                       d_present, ws_touched, 0);
                else
                    hipLaunchKernelGGL((mxv_first_hit_kernel<T, DynBin, DynBin, true>), dim3((n_out + 255) / 256), dim3(256), 0, s, n_out, Tp, Tj, static_cast<const T*>(mask), d_v,
                                       reinterpret_cast<T*>(ws_outv), DynBin(op_mult), DynBin(op_add), op_select, T(0), xbits, uf, d_present, ws_touched, 0);
            }
        }
        // every index of v stored + a sliced-ELL copy on the transposed matrix's plan: the value-form kernel in its vxm mode
        const char* sell_env = knob_env("SPLAHIP_SELL");
        // every index of v stored + a column-ordered-blocks copy on the transposed matrix's plan: its kernel in vxm mode
        // reference-order mode (splahip_set_reference_order): an fp32 sum per output index is folded strictly in the order of v's indices from its first product —
        // the reference's loop, src/cpu/cpu_vxm.hpp:101-113, over the rows of the transposed matrix (mxv_kernels.hpp: ORDERED, spmv_longrow_ordered_kernel)
        const bool ref_order = reference_order_calls() && std::is_same<T, float>::value && op_add == SPLAHIP_PLUS;
        if (first_hit) {// (launched above)
        } else if (ref_order && plan->n_items) {
            hipLaunchKernelGGL((spmv_wavetile_kernel<Cfg, T, DynBin, StaBin<SPLAHIP_PLUS>, true, true>), dim3(grid), dim3(Cfg::kTpb), 0, s, plan->d_item_row, plan->d_item_k0, plan->n_items, Tp, Tj,
                               static_cast<const uint*>(Tx), static_cast<const T*>(mask), d_v, d_present, reinterpret_cast<T*>(ws_outv), ws_touched, reinterpret_cast<T*>(plan->d_partial),
                               plan->d_partial_ok, DynBin(op_mult), StaBin<SPLAHIP_PLUS>(), op_select, T(0), plan->nnz);
            if (plan->n_long > 0)
                hipLaunchKernelGGL((spmv_longrow_ordered_kernel<Cfg, T, DynBin, StaBin<SPLAHIP_PLUS>, true>), dim3(grid), dim3(Cfg::kTpb), 0, s, plan->d_item_row, plan->d_item_k0, plan->n_items, Tp,
                                   Tj, static_cast<const uint*>(Tx), static_cast<const T*>(mask), d_v, d_present, reinterpret_cast<T*>(ws_outv), ws_touched, DynBin(op_mult),
                                   StaBin<SPLAHIP_PLUS>(), op_select, T(0), plan->nnz);
        } else if (v_is_identity && cob_vxm_serves(plan, TypeId<T>::value, op_add, n_in) &&
            ((reinterpret_cast<uintptr_t>(ws_outv) | reinterpret_cast<uintptr_t>(ws_touched) | reinterpret_cast<uintptr_t>(mask)) & 15u) == 0) {
            int st = cob_vxm(plan, TypeId<T>::value, Tp, mask, d_v, ws_outv, ws_touched, op_mult, op_add, op_select, s);
            if (st != SPLAHIP_OK) return st;
        } else if (plan->sell && v_is_identity && plan->sell->n_cols == n_in && !(sell_env && sell_env[0] == '0') &&
            ((reinterpret_cast<uintptr_t>(Tp) | reinterpret_cast<uintptr_t>(mask)) & 15u) == 0) {
            int st = sell_vxm(plan, TypeId<T>::value, Tp, mask, d_v, ws_outv, ws_touched, op_mult, op_add, op_select, s);
            if (st != SPLAHIP_OK) return st;
        } else if (plan->n_items) {
            hipLaunchKernelGGL((spmv_wavetile_kernel<Cfg, T, Mult, Add, true>), dim3(grid), dim3(Cfg::kTpb), 0, s,
                               plan->d_item_row, plan->d_item_k0, plan->n_items, Tp, Tj, static_cast<const uint*>(Tx),
                               static_cast<const T*>(mask), d_v, d_present, reinterpret_cast<T*>(ws_outv), ws_touched,
                               reinterpret_cast<T*>(plan->d_partial), plan->d_partial_ok, Mult(op_mult), Add(op_add), op_select,
                               T(0), plan->nnz);
            if (plan->n_long > 0)
                hipLaunchKernelGGL((spmv_combine_kernel<Cfg, T, Add, true>), dim3((plan->n_items + 255) / 256), dim3(256), 0, s,
                                   plan->d_item_row, plan->d_item_k0, plan->n_items, Tp, static_cast<const T*>(mask), reinterpret_cast<T*>(ws_outv),
                                   ws_touched, reinterpret_cast<const T*>(plan->d_partial), plan->d_partial_ok, Add(op_add),
                                   op_select, T(0));
        }
        SPLAHIP_TRY(hipGetLastError());
        if constexpr (std::is_same<T, float>::value) {
            if (op_add == SPLAHIP_MIN || op_add == SPLAHIP_MAX)
                hipLaunchKernelGGL((pull_lead_nan_kernel<Mult>), dim3((n_out + 255) / 256), dim3(256), 0, s, n_out, Tp, Tj, static_cast<const uint*>(Tx), d_present,
                                   reinterpret_cast<const float*>(d_v), Mult(op_mult), reinterpret_cast<float*>(ws_outv), ws_touched);
        }
        // compaction of the touched columns, ascending
        hipLaunchKernelGGL(flag_count_kernel, dim3(n_chunks), dim3(256), 0, s, ws_touched, n_out, ws_chunk);
        ScanPost sp;
        (void) scan_launch(ws_chunk, n_chunks, s, &sp);// (the count reaches the host from the scan: scan_post.hpp)
        hipLaunchKernelGGL(flag_write_kernel, dim3(n_chunks), dim3(256), 0, s, ws_touched, ws_outv, n_out, ws_chunk, ri, static_cast<uint*>(rx));
        SPLAHIP_TRY(hipGetLastError());
        SPLAHIP_TRY(scan_total(h_count, ws_chunk + n_chunks, sp, s));
        return SPLAHIP_OK;
    }

    // jit_hooks.hpp: the pull form with the per-column fold replaced by a run-time-compiled kernel (user-defined ops, jit.hip): dense image of v + presence
    // bitmap (+ the position of every index in v's stored order when v is not index-sorted) in the plan's workspace, then the same compaction
    // list[1 + i], i < list[0]: the rows of the transposed matrix (output indices) of more than long_from entries, in no particular order
    __global__ __launch_bounds__(256) void vxm_long_rows_kernel(uint n_out, const uint* __restrict__ Tp, uint long_from, uint* __restrict__ list) {
        const uint j = blockIdx.x * 256 + threadIdx.x;
        if (j < n_out && Tp[j + 1] - Tp[j] > long_from) list[1 + atomicAdd(list, 1u)] = j;
    }
    __global__ __launch_bounds__(256) void scatter_vpos_kernel(uint nvals, const uint* __restrict__ vi, uint* __restrict__ vpos) {
        const uint idx = blockIdx.x * 256 + threadIdx.x;
        if (idx < nvals) vpos[vi[idx]] = idx;
    }
    int vxm_pull_jit(splahip_plan plan, uint32_t n_out, uint32_t n_in, const uint32_t* Tp, const uint32_t* Tj, const void* Tx, const void* mask, uint32_t nvals,
                     const uint32_t* vi, const void* vx, int sorted, uint32_t* ri, void* rx, uint32_t* h_count, hipFunction_t column, hipFunction_t column_long, hipStream_t s) {
        const uint   words    = (n_in + 31u) / 32u;
        const uint   n_chunks = (n_out + 1023u) / 1024u;
        const uint   n_in4    = (n_in + 3u) & ~3u, words4 = (words + 3u) & ~3u, n_out4 = (n_out + 3u) & ~3u;
        const size_t need     = sizeof(uint) * (2 * (size_t) n_in4 + words4 + 2 * (size_t) n_out4 + n_chunks + 8);
        if (plan->ws_bytes < need) {
            if (plan->d_ws) (void) hipFree(plan->d_ws);
            plan->d_ws = nullptr, plan->ws_bytes = 0;
            SPLAHIP_TRY(hipMalloc(&plan->d_ws, need));
            plan->ws_bytes = need;
        }
        uint* ws_dense   = static_cast<uint*>(plan->d_ws);
        uint* ws_present = ws_dense + n_in4;
        uint* ws_outv    = ws_present + words4;
        uint* ws_touched = ws_outv + n_out4;
        uint* ws_chunk   = ws_touched + n_out4;
        uint* ws_vpos    = ws_chunk + n_chunks + 8;
        SPLAHIP_TRY(hipMemsetAsync(ws_present, 0, sizeof(uint) * (size_t) words, s));
        hipLaunchKernelGGL(scatter_present_kernel, dim3((nvals + 255) / 256), dim3(256), 0, s, nvals, vi, static_cast<const uint*>(vx), ws_dense, ws_present);
        if (!sorted) hipLaunchKernelGGL(scatter_vpos_kernel, dim3((nvals + 255) / 256), dim3(256), 0, s, nvals, vi, ws_vpos);
        SPLAHIP_TRY(hipGetLastError());
        uint  no = n_out;
        int   so = sorted ? 1 : 0;
        void *pd = ws_dense, *pp = ws_present, *pq = ws_vpos, *po = ws_outv, *pt = ws_touched;
        constexpr uint kLongFrom = 508;// output indices of more entries (index-sorted v): a wavefront each (column_long) instead of one lane
        uint           long_from = sorted && column_long ? kLongFrom : 0xFFFFFFFFu;
        void* args[] = {&no, &Tp, &Tj, &Tx, &mask, &pd, &pp, &pq, &so, &po, &pt, &long_from};
        SPLAHIP_TRY(hipModuleLaunchKernel(column, (n_out + 255u) / 256u, 1, 1, 256, 1, 1, 0, s, args, nullptr));
        DevMem list;// [0] the count, then the listed output indices (kept until the stream has run the kernels below: released stream-ordered)
        if (long_from != 0xFFFFFFFFu) {
            SPLAHIP_TRY(list.alloc(s, sizeof(uint) * ((size_t) n_out + 1)));
            SPLAHIP_TRY(hipMemsetAsync(list.p, 0, sizeof(uint), s));
            hipLaunchKernelGGL(vxm_long_rows_kernel, dim3((n_out + 255u) / 256u), dim3(256), 0, s, n_out, Tp, kLongFrom, list.as<uint>());
            SPLAHIP_TRY(hipGetLastError());
            const uint* rows  = list.as<uint>() + 1;
            const uint* count = list.as<uint>();
            void*       la[]  = {&rows, &count, &Tp, &Tj, &Tx, &mask, &pd, &pp, &po, &pt};
            SPLAHIP_TRY(hipModuleLaunchKernel(column_long, 2048, 1, 1, 256, 1, 1, 0, s, la, nullptr));
        }
        hipLaunchKernelGGL(flag_count_kernel, dim3(n_chunks), dim3(256), 0, s, ws_touched, n_out, ws_chunk);
        ScanPost sp;
        (void) scan_launch(ws_chunk, n_chunks, s, &sp);// (the count reaches the host from the scan: scan_post.hpp)
        hipLaunchKernelGGL(flag_write_kernel, dim3(n_chunks), dim3(256), 0, s, ws_touched, ws_outv, n_out, ws_chunk, ri, static_cast<uint*>(rx));
        SPLAHIP_TRY(hipGetLastError());
        SPLAHIP_TRY(scan_total(h_count, ws_chunk + n_chunks, sp, s));
        return SPLAHIP_OK;
    }

    // all-zero accumulator words kept between calls (grow-only, per host thread like the other cached workspaces; listed process-wide like them — runtime.hip: a
    // trim from any thread frees the workspaces of threads that are not inside a call and deletes those of threads that have exited)
    struct SmallPushWs {
        std::mutex m;
        u64*       acc   = nullptr;
        size_t     words = 0;
        uint*      misc  = nullptr;// gate[4] | chunk[...]
        size_t     misc_words = 0;
        int        busy   = 0;
        bool       orphan = false;
    };
    struct SmallPushRegistry {
        std::mutex                m;
        std::vector<SmallPushWs*> all;
    };
    inline SmallPushRegistry& small_push_registry() {
        static SmallPushRegistry* r = new SmallPushRegistry;
        return *r;
    }
    struct SmallPushHolder {
        SmallPushWs* w;
        SmallPushHolder() : w(new SmallPushWs) {
            auto& r = small_push_registry();
            std::lock_guard<std::mutex> g(r.m);
            r.all.push_back(w);
        }
        ~SmallPushHolder() {
            std::lock_guard<std::mutex> g(w->m);
            w->orphan = true;
        }
    };
    inline SmallPushWs& small_push_ws() {
        static thread_local SmallPushHolder h;
        return *h.w;
    }
    void small_push_trim() {
        auto& r = small_push_registry();
        std::lock_guard<std::mutex> rg(r.m);
        for (size_t k = 0; k < r.all.size();) {
            SmallPushWs* w    = r.all[k];
            bool         drop = false;
            {
                std::lock_guard<std::mutex> g(w->m);
                if (w->busy == 0) {
                    if (w->acc) (void) hipFree(w->acc);
                    if (w->misc) (void) hipFree(w->misc);
                    w->acc = nullptr, w->misc = nullptr, w->words = w->misc_words = 0;
                    drop = w->orphan;
                }
            }
            if (drop) {
                delete w;
                r.all.erase(r.all.begin() + (long) k);
            } else
                ++k;
        }
    }
    struct SmallPushBusy {// (the workspace is in use for the length of a call: the call ends with a host synchronisation)
        SmallPushWs& w;
        explicit SmallPushBusy(SmallPushWs& ws) : w(ws) {
            std::lock_guard<std::mutex> g(w.m);
            ++w.busy;
        }
        ~SmallPushBusy() {
            std::lock_guard<std::mutex> g(w.m);
            --w.busy;
        }
    };

    template<typename T>
    int push_small_typed(uint n_cols, const uint* Ap, const uint* Aj, const void* Ax, const void* mask, uint nvals, const uint* vi, const void* vx, uint* ri,
                         void* rx, uint max_products, uint* h_out /* count, products lo, products hi, unsorted, NaN seen */, int op_mult, int op_add, int op_select,
                         hipStream_t s) {
        SmallPushWs&  ws = small_push_ws();
        SmallPushBusy in_use(ws);
        register_trim(&small_push_trim);// (idempotent)
        const uint    n_chunks = (n_cols + 1023u) / 1024u;
        if (ws.words < n_cols) {
            if (ws.acc) (void) hipFree(ws.acc);
            ws.acc = nullptr, ws.words = 0;
            SPLAHIP_TRY(hipMalloc(reinterpret_cast<void**>(&ws.acc), sizeof(u64) * (size_t) n_cols));
            ws.words = n_cols;
            SPLAHIP_TRY(hipMemsetAsync(ws.acc, 0, sizeof(u64) * (size_t) n_cols, s));
        }
        if (ws.misc_words < (size_t) n_chunks + 12) {
            if (ws.misc) (void) hipFree(ws.misc);
            ws.misc = nullptr, ws.misc_words = 0;
            SPLAHIP_TRY(hipMalloc(reinterpret_cast<void**>(&ws.misc), sizeof(uint) * ((size_t) n_chunks + 12)));
            ws.misc_words = (size_t) n_chunks + 12;
        }
        uint* gate  = ws.misc;// [0..1] products, [2] unsorted, [3] count, [4] NaN product under fp32 MIN / MAX
        uint* chunk = ws.misc + 8;
        SPLAHIP_TRY(hipMemsetAsync(gate, 0, 32, s));
        hipLaunchKernelGGL(probe_kernel, dim3((nvals + 255) / 256), dim3(256), 0, s, nvals, vi, Ap, reinterpret_cast<unsigned long long*>(gate), gate + 2);
        const uint share = std::max(1u, std::min(64u, 16384u / nvals));// workgroups per frontier entry: few entries -> split their rows
        hipLaunchKernelGGL((push_direct_kernel<T, DynBin, DynBin>), dim3(nvals, share), dim3(256), 0, s, gate, max_products, vi, static_cast<const T*>(vx), Ap, Aj,
                           static_cast<const uint*>(Ax), static_cast<const T*>(mask), op_select, DynBin(op_mult), DynBin(op_add), ws.acc);
        hipLaunchKernelGGL(acc_count_gated_kernel, dim3(n_chunks), dim3(256), 0, s, gate, max_products, ws.acc, n_cols, chunk);
        // the count and the gate words reach the host from the scan (scan_post.hpp: slot 0 the count, slots 1 .. 5 gate[0 .. 4] — gate[3] itself is written by the pass behind)
        ScanPost sp;
        (void) scan_launch(chunk, n_chunks, s, &sp, gate, 5);
        hipLaunchKernelGGL(acc_write_reset_kernel, dim3(n_chunks), dim3(256), 0, s, gate, max_products, ws.acc, n_cols, chunk, ri, static_cast<uint*>(rx));
        SPLAHIP_TRY(hipGetLastError());
        uint h[5] = {0, 0, 0, 0, 0};// products lo, hi, unsorted, count, NaN seen
        if (sp.posted) {
            uint w[6] = {0, 0, 0, 0, 0, 0};
            SPLAHIP_TRY(mailbox_end(w, 6, sp.mt, s));
            h[0] = w[1], h[1] = w[2], h[2] = w[3], h[3] = w[0], h[4] = w[5];
        } else
            SPLAHIP_TRY(fetch_words(h, gate, 5, s));
        h_out[0] = h[3], h_out[1] = h[0], h_out[2] = h[1], h_out[3] = h[2], h_out[4] = h[4];
        return SPLAHIP_OK;
    }

    // products up to which the order-free form is used (SPLAHIP_PUSH_ATOMIC_MAX).  Measured at R-MAT 22, a 10^6-entry frontier, 15.4 M products (bench.py --workload
    // vxm-push): sort form 1.76 - 1.83 ms, this form 0.74 - 0.81 ms; the limit was 2^22 until round 4
    constexpr uint kPushAtomicMax = 1u << 26;
    inline bool    push_order_free(int op_add, int type) {
        if (!binop_is_associative(op_add) || !binop_is_commutative(op_add)) return false;
        return !(type == SPLAHIP_FLOAT && (op_add == SPLAHIP_PLUS || op_add == SPLAHIP_MULT));// fp32 sums / products round by order: sort form
    }
    inline bool push_atomic_disabled() {
        const char* e = knob_env("SPLAHIP_PUSH_ATOMIC");// A/B runs and tests: 0 keeps the sort form
        return e && e[0] == '0';
    }

    template<typename T>
    int push_typed(uint n_rows, uint n_cols, const uint* Ap, const uint* Aj, const void* Ax, const void* mask, uint nvals, const uint* vi,
                   const void* vx, uint* ri, void* rx, uint* h_count, int op_mult, int op_add, int op_select, uint64_t products_hint, hipStream_t s) {
        (void) n_rows;
        *h_count = 0;
        if (nvals == 0) return SPLAHIP_OK;
        // ONE stream-ordered allocation and ONE host synchronisation per call (two when the caller did not pass the product count
        // it already got from splahip_vxm_probe): the fixed cost of this form is launches and read-backs, not bytes.
        DevMem head;// len[nvals] | offs[nvals + 1]
        SPLAHIP_TRY(head.alloc(s, sizeof(uint) * (2 * (size_t) nvals + 1)));
        uint* len  = head.as<uint>();
        uint* offs = len + nvals;
        hipLaunchKernelGGL(frontier_len_kernel, dim3((nvals + 255) / 256), dim3(256), 0, s, nvals, vi, Ap, len);
        SPLAHIP_TRY(exclusive_scan_u32(len, nvals, offs, s));
        uint n_products = (uint) products_hint;
        if (products_hint == 0) {
            SPLAHIP_TRY(fetch_words(&n_products, offs + nvals, 1, s));
        }
        if (n_products == 0) return SPLAHIP_OK;

        const char* pm_env     = knob_env("SPLAHIP_PUSH_ATOMIC_MAX");
        const uint  atomic_max = pm_env && *pm_env ? (uint) std::strtoul(pm_env, nullptr, 10) : kPushAtomicMax;
        if (n_products <= atomic_max && push_order_free(op_add, TypeId<T>::value) && !push_atomic_disabled()) {
            const uint n_chunks = (n_cols + 1023u) / 1024u;
            const uint n_blocks = (n_products + 255u) / 256u;
            DevMem     body;// acc[n_cols] (8 bytes each) | chunk[n_chunks + 1] | NaN-seen flag | part[n_blocks + 1]
            SPLAHIP_TRY(body.alloc(s, sizeof(u64) * (size_t) n_cols + sizeof(uint) * ((size_t) n_chunks + 2 + (size_t) n_blocks + 1)));
            u64*  acc   = body.as<u64>();
            uint* chunk = reinterpret_cast<uint*>(acc + n_cols);
            uint* part  = chunk + n_chunks + 2;
            SPLAHIP_TRY(hipMemsetAsync(acc, 0, sizeof(u64) * (size_t) n_cols, s));
            SPLAHIP_TRY(hipMemsetAsync(chunk + n_chunks + 1, 0, sizeof(uint), s));
            const char* pb_env  = knob_env("SPLAHIP_PUSH_BLOCKED");// =0: every thread searches `offs` in global memory (the form of rounds 2 - 4); read per call
            const bool  blocked = !(pb_env && pb_env[0] == '0');
            if (blocked) {
                hipLaunchKernelGGL(push_part_kernel, dim3((n_blocks + 1 + 255) / 256), dim3(256), 0, s, n_blocks, nvals, offs, part);
                hipLaunchKernelGGL((push_atomic_blocked_kernel<T, DynBin, DynBin>), dim3(n_blocks), dim3(256), 0, s, n_products, nvals, offs, part, vi,
                                   static_cast<const T*>(vx), Ap, Aj, static_cast<const uint*>(Ax), static_cast<const T*>(mask), op_select, DynBin(op_mult),
                                   DynBin(op_add), acc, chunk + n_chunks + 1);
            } else
                hipLaunchKernelGGL((push_atomic_kernel<T, DynBin, DynBin>), dim3(n_blocks), dim3(256), 0, s, n_products, nvals, offs, vi,
                                   static_cast<const T*>(vx), Ap, Aj, static_cast<const uint*>(Ax), static_cast<const T*>(mask), op_select, DynBin(op_mult),
                                   DynBin(op_add), acc, chunk + n_chunks + 1);
            hipLaunchKernelGGL(acc_count_kernel, dim3(n_chunks), dim3(256), 0, s, acc, n_cols, chunk);
            hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, chunk, n_chunks);
            hipLaunchKernelGGL(acc_write_kernel, dim3(n_chunks), dim3(256), 0, s, acc, n_cols, chunk, ri, static_cast<uint*>(rx));
            SPLAHIP_TRY(hipGetLastError());
            uint h[2] = {0, 0};// count, NaN seen
            SPLAHIP_TRY(fetch_words(h, chunk + n_chunks, 2, s));
            if (h[1] == 0) {
                *h_count = h[0];
                return SPLAHIP_OK;
            }
            // a NaN product under fp32 MIN / MAX: the sort form below folds in the reference's order
        }

        // stable LSD radix sort by column: equal columns keep the expand order = the reference's fold order
        size_t     tmp_bytes = 0;
        const uint bits      = bit_width_u32(n_cols);
        SPLAHIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint*) nullptr, (uint*) nullptr, (uint*) nullptr, (uint*) nullptr, n_products, 0, bits, s));
        const uint   n_chunks = (n_products + 1023u) / 1024u;
        const uint   max_long = n_products / kRunShort + 1;
        const size_t words    = 4 * (size_t) n_products + (n_chunks + 1) + 1 + 2 * (size_t) max_long;
        const size_t tmp_at   = (sizeof(uint) * words + 255) & ~(size_t) 255;
        DevMem       body;// key_a | key_b | val_a | val_b | chunk | lcount | lpos | lrank | (aligned) sort scratch
        SPLAHIP_TRY(body.alloc(s, tmp_at + tmp_bytes));
        uint* key_a  = body.as<uint>();
        uint* key_b  = key_a + n_products;
        uint* val_a  = key_b + n_products;
        uint* val_b  = val_a + n_products;
        uint* chunk  = val_b + n_products;
        uint* lcount = chunk + n_chunks + 1;
        uint* lpos   = lcount + 1;
        uint* lrank  = lpos + max_long;
        void* tmp    = static_cast<char*>(body.p) + tmp_at;
        hipLaunchKernelGGL((expand_kernel<T, DynBin>), dim3((n_products + 255) / 256), dim3(256), 0, s, n_products, nvals, offs, vi,
                           static_cast<const T*>(vx), Ap, Aj, static_cast<const uint*>(Ax), static_cast<const T*>(mask), n_cols, op_select,
                           DynBin(op_mult), key_a, reinterpret_cast<T*>(val_a));
        SPLAHIP_TRY(hipGetLastError());
        SPLAHIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, key_a, key_b, val_a, val_b, n_products, 0, bits, s));
        // run heads -> output slots; short runs folded in place, long runs by a wavefront each
        SPLAHIP_TRY(hipMemsetAsync(lcount, 0, sizeof(uint), s));
        hipLaunchKernelGGL(head_count_kernel, dim3(n_chunks), dim3(256), 0, s, key_b, n_products, n_cols, chunk);
        hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, chunk, n_chunks);
        hipLaunchKernelGGL((run_fold_kernel<T, DynBin>), dim3(n_chunks), dim3(256), 0, s, key_b, reinterpret_cast<const T*>(val_b), n_products, n_cols,
                           chunk, ri, static_cast<T*>(rx), lcount, lpos, lrank, DynBin(op_add));
        // the long-run pass is launched for the worst case and sizes itself from the device-side counter (no read-back in between)
        const uint grid = (max_long + 3) / 4;
        // (reference-order mode: an fp32 sum's long runs are folded one product after the other too — the expand order is the reference's fold order)
        const bool strict = reference_order_calls() && std::is_same<T, float>::value && op_add == SPLAHIP_PLUS;
        if (fold_is_reorderable(op_add, TypeId<T>::value) && !strict)
            hipLaunchKernelGGL((long_run_kernel<T, DynBin, true>), dim3(grid), dim3(256), 0, s, key_b, reinterpret_cast<const T*>(val_b), n_products, lcount, lpos,
                               lrank, static_cast<T*>(rx), DynBin(op_add));
        else
            hipLaunchKernelGGL((long_run_kernel<T, DynBin, false>), dim3(grid), dim3(256), 0, s, key_b, reinterpret_cast<const T*>(val_b), n_products, lcount, lpos,
                               lrank, static_cast<T*>(rx), DynBin(op_add));
        SPLAHIP_TRY(hipGetLastError());
        SPLAHIP_TRY(fetch_words(h_count, chunk + n_chunks, 1, s));
        return SPLAHIP_OK;
    }

#define SPLAHIP_PULL(MUL, ADD) pull_typed<T, StaBin<MUL>, StaBin<ADD>>(plan, n_out, n_in, Tp, Tj, Tx, mask, nvals, vi, vx, ident, ri, rx, h_count, op_mult, op_add, op_select, s)
    template<typename T>
    int pull_dispatch(splahip_plan plan, uint n_out, uint n_in, const uint* Tp, const uint* Tj, const void* Tx, const void* mask, uint nvals,
                      const uint* vi, const void* vx, int ident, uint* ri, void* rx, uint* h_count, int op_mult, int op_add, int op_select,
                      hipStream_t s) {
        if (op_mult == SPLAHIP_MULT && op_add == SPLAHIP_PLUS) return SPLAHIP_PULL(SPLAHIP_MULT, SPLAHIP_PLUS);
        if (op_mult == SPLAHIP_PLUS && op_add == SPLAHIP_MIN) return SPLAHIP_PULL(SPLAHIP_PLUS, SPLAHIP_MIN);
        if constexpr (!std::is_same<T, float>::value) {
            if (op_mult == SPLAHIP_BAND && op_add == SPLAHIP_BOR) return SPLAHIP_PULL(SPLAHIP_BAND, SPLAHIP_BOR);
        }
        return pull_typed<T, DynBin, DynBin>(plan, n_out, n_in, Tp, Tj, Tx, mask, nvals, vi, vx, ident, ri, rx, h_count, op_mult, op_add, op_select, s);
    }
#undef SPLAHIP_PULL

}// namespace splahip

using namespace splahip;

namespace splahip {
    void vxm_workspaces_trim() { small_push_trim(); }// the small-frontier push form's per-thread accumulators (splahip_mxm_trim, out-of-memory retries)
}// namespace splahip

extern "C" {

int splahip_csr_transpose(uint32_t n_rows, uint32_t n_cols, uint32_t nnz, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax,
                          uint32_t* d_Tp, uint32_t* d_Tj, void* d_Tx, void* stream) {
    // the value words travel untouched: the transpose of matrix_ops.hip with the identity over 4-byte words
    int st = splahip_m_transpose(SPLAHIP_UINT, n_rows, n_cols, nnz, d_Ap, d_Aj, d_Ax, d_Tp, d_Tj, d_Tx, SPLAHIP_IDENTITY, stream);
    if (st != SPLAHIP_OK) return st;
    SPLAHIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return SPLAHIP_OK;
}

int splahip_vxm_probe(uint32_t nvals, const uint32_t* d_vi, const uint32_t* d_Ap, uint64_t* h_products, int* h_sorted, void* stream) {
    SPLAHIP_REQUIRE(h_products && h_sorted, "null output");
    *h_products = 0;
    *h_sorted   = 1;
    if (nvals == 0) return SPLAHIP_OK;
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    DevMem      acc;
    SPLAHIP_TRY(acc.alloc(s, 16));
    SPLAHIP_TRY(hipMemsetAsync(acc.p, 0, 16, s));
    hipLaunchKernelGGL(probe_kernel, dim3((nvals + 255) / 256), dim3(256), 0, s, nvals, d_vi, d_Ap, acc.as<unsigned long long>(),
                       reinterpret_cast<uint*>(acc.as<unsigned long long>() + 1));
    SPLAHIP_TRY(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    SPLAHIP_TRY(fetch_words(h, acc.p, 4, s));
    *h_products = h[0];
    *h_sorted   = (h[1] & 0xffffffffull) ? 0 : 1;
    return SPLAHIP_OK;
}

int splahip_vxm_push_small(int type, uint32_t n_rows, uint32_t n_cols, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, const void* d_mask,
                           uint32_t nvals, const uint32_t* d_vi, const void* d_vx, uint32_t* d_ri, void* d_rx, uint32_t max_products, uint32_t* h_count,
                           uint64_t* h_products, int* h_sorted, int* h_done, int op_mult, int op_add, int op_select, void* stream) {
    (void) n_rows;
    SPLAHIP_REQUIRE(h_count && h_products && h_sorted && h_done, "null output");
    SPLAHIP_REQUIRE(binop_valid_for(op_mult, type) && binop_valid_for(op_add, type), "op not defined for this type");
    SPLAHIP_REQUIRE(push_order_free(op_add, type), "the small-frontier form needs an op_add whose result does not depend on the fold order");
    *h_count = 0, *h_products = 0, *h_sorted = 1, *h_done = 1;
    if (nvals == 0 || n_cols == 0) return SPLAHIP_OK;
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    uint32_t    out[5] = {0, 0, 0, 0, 0};
    switch (type) {
        case SPLAHIP_INT: st = push_small_typed<int32_t>(n_cols, d_Ap, d_Aj, d_Ax, d_mask, nvals, d_vi, d_vx, d_ri, d_rx, max_products, out, op_mult, op_add, op_select, s); break;
        case SPLAHIP_UINT: st = push_small_typed<uint32_t>(n_cols, d_Ap, d_Aj, d_Ax, d_mask, nvals, d_vi, d_vx, d_ri, d_rx, max_products, out, op_mult, op_add, op_select, s); break;
        case SPLAHIP_FLOAT: st = push_small_typed<float>(n_cols, d_Ap, d_Aj, d_Ax, d_mask, nvals, d_vi, d_vx, d_ri, d_rx, max_products, out, op_mult, op_add, op_select, s); break;
        default: return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
    }
    if (st != SPLAHIP_OK) return st;
    *h_products = ((uint64_t) out[2] << 32) | out[1];
    *h_sorted   = out[3] ? 0 : 1;
    *h_done     = (*h_products == 0 || *h_products <= max_products) && out[4] == 0 ? 1 : 0;// (out[4]: a NaN product under fp32 MIN / MAX -> the ordered forms)
    *h_count    = *h_done ? out[0] : 0;
    return SPLAHIP_OK;
}

int splahip_vxm_pull(splahip_plan plan_T, int type, uint32_t n_rows, uint32_t n_cols,
                     const uint32_t* d_Tp, const uint32_t* d_Tj, const void* d_Tx, const void* d_mask,
                     uint32_t nvals, const uint32_t* d_vi, const void* d_vx, int v_is_identity,
                     uint32_t* d_ri, void* d_rx, uint32_t* h_count,
                     int op_mult, int op_add, int op_select, void* stream) {
    SPLAHIP_REQUIRE(plan_T != nullptr && h_count != nullptr, "null argument");
    SPLAHIP_REQUIRE(plan_T->n_rows == n_cols, "plan_T must be the plan of the transposed matrix (n_cols rows)");
    SPLAHIP_REQUIRE(binop_valid_for(op_mult, type) && binop_valid_for(op_add, type), "op not defined for this type");
    SPLAHIP_REQUIRE(fold_is_reorderable(op_add, type) && binop_is_commutative(op_add),
                    "the pull form needs an op_add that may be folded in any order: associative, commutative, not an fp32 MULT (use splahip_vxm_push)");
    *h_count = 0;
    if (n_cols == 0) return SPLAHIP_OK;
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    switch (type) {
        case SPLAHIP_INT: return pull_dispatch<int32_t>(plan_T, n_cols, n_rows, d_Tp, d_Tj, d_Tx, d_mask, nvals, d_vi, d_vx, v_is_identity, d_ri, d_rx, h_count, op_mult, op_add, op_select, s);
        case SPLAHIP_UINT: return pull_dispatch<uint32_t>(plan_T, n_cols, n_rows, d_Tp, d_Tj, d_Tx, d_mask, nvals, d_vi, d_vx, v_is_identity, d_ri, d_rx, h_count, op_mult, op_add, op_select, s);
        case SPLAHIP_FLOAT: return pull_dispatch<float>(plan_T, n_cols, n_rows, d_Tp, d_Tj, d_Tx, d_mask, nvals, d_vi, d_vx, v_is_identity, d_ri, d_rx, h_count, op_mult, op_add, op_select, s);
    }
    return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
}

int splahip_vxm_push(int type, uint32_t n_rows, uint32_t n_cols,
                     const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, const void* d_mask,
                     uint32_t nvals, const uint32_t* d_vi, const void* d_vx,
                     uint32_t* d_ri, void* d_rx, uint32_t* h_count,
                     int op_mult, int op_add, int op_select, void* stream) {
    return splahip_vxm_push_n(type, n_rows, n_cols, d_Ap, d_Aj, d_Ax, d_mask, nvals, d_vi, d_vx, d_ri, d_rx, h_count, op_mult, op_add, op_select, 0, stream);
}

int splahip_vxm_push_n(int type, uint32_t n_rows, uint32_t n_cols,
                       const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, const void* d_mask,
                       uint32_t nvals, const uint32_t* d_vi, const void* d_vx,
                       uint32_t* d_ri, void* d_rx, uint32_t* h_count,
                       int op_mult, int op_add, int op_select, uint64_t n_products, void* stream) {
    SPLAHIP_REQUIRE(h_count != nullptr, "null argument");
    SPLAHIP_REQUIRE(n_products <= 0xFFFFFFF0ull, "product count exceeds 32-bit product indices");
    SPLAHIP_REQUIRE(binop_valid_for(op_mult, type) && binop_valid_for(op_add, type), "op not defined for this type");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t s = as_stream(stream);
    switch (type) {
        case SPLAHIP_INT: return push_typed<int32_t>(n_rows, n_cols, d_Ap, d_Aj, d_Ax, d_mask, nvals, d_vi, d_vx, d_ri, d_rx, h_count, op_mult, op_add, op_select, n_products, s);
        case SPLAHIP_UINT: return push_typed<uint32_t>(n_rows, n_cols, d_Ap, d_Aj, d_Ax, d_mask, nvals, d_vi, d_vx, d_ri, d_rx, h_count, op_mult, op_add, op_select, n_products, s);
        case SPLAHIP_FLOAT: return push_typed<float>(n_rows, n_cols, d_Ap, d_Aj, d_Ax, d_mask, nvals, d_vi, d_vx, d_ri, d_rx, h_count, op_mult, op_add, op_select, n_products, s);
    }
    return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
}
}
