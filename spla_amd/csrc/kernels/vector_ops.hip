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
// All are element-wise, HBM-bound streaming kernels (4-16 bytes per element).  The sparse forms assume the COO operand
// holds every index at most once (always true for vectors produced by this library and by Dok->Coo / Dense->Coo
// conversions; the reference's sequential loop would apply a duplicate twice).  The sparse feedback of v_eadd_fdb is
// compacted order-preservingly, so it lists the changed entries in the operand's stored order like the reference.
#include "block_utils.hpp"
#include "common.hpp"
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
        uint cnt = 0;
        for (uint i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) cnt += v[i] != fill ? 1u : 0u;
        cnt = wave_sum(cnt);
        if (lane_id() == 0 && cnt) atomicAdd(count, (unsigned long long) cnt);
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
        uint*      scratch  = nullptr;// changed[nvals] | newval[nvals] | chunk[n_chunks + 1]
        SPLAHIP_TRY(hipMalloc(reinterpret_cast<void**>(&scratch), sizeof(uint) * (2 * (size_t) nvals + n_chunks + 1)));
        uint* changed = scratch;
        uint* newval  = scratch + nvals;
        uint* chunk   = newval + nvals;
        hipLaunchKernelGGL(eadd_fdb_sparse_apply_kernel<T>, dim3((nvals + 255) / 256), dim3(256), 0, s, nvals, static_cast<T*>(r), vi,
                           static_cast<const T*>(vx), changed, reinterpret_cast<T*>(newval), DynBin(op));
        hipLaunchKernelGGL(changed_count_kernel, dim3(n_chunks), dim3(256), 0, s, changed, nvals, chunk);
        hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, chunk, n_chunks);
        hipLaunchKernelGGL(changed_write_kernel, dim3(n_chunks), dim3(256), 0, s, changed, vi, newval, nvals, chunk, fi, static_cast<uint*>(fx));
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h_count, chunk + n_chunks, sizeof(uint), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void) hipFree(scratch);
        if (e != hipSuccess) return fail(SPLAHIP_ERROR, "v_eadd_fdb (sparse) failed: %s", hipGetErrorString(e));
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
    unsigned long long* d_count = nullptr;
    SPLAHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_count), sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s);
    const dim3 grid(std::min<uint32_t>((n + 255) / 256, 2048)), block(256);
    if (e == hipSuccess) {
        switch (type) {
            case SPLAHIP_INT: hipLaunchKernelGGL(count_mf_kernel<int32_t>, grid, block, 0, s, n, static_cast<const int32_t*>(d_v), from_bits<int32_t>(fill_bits), d_count); break;
            case SPLAHIP_UINT: hipLaunchKernelGGL(count_mf_kernel<uint32_t>, grid, block, 0, s, n, static_cast<const uint32_t*>(d_v), from_bits<uint32_t>(fill_bits), d_count); break;
            default: hipLaunchKernelGGL(count_mf_kernel<float>, grid, block, 0, s, n, static_cast<const float*>(d_v), from_bits<float>(fill_bits), d_count); break;
        }
        e = hipGetLastError();
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d_count, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void) hipFree(d_count);
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
}
