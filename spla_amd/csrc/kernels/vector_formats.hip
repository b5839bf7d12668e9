// vector_formats.hip — device-side conversions between the accelerator vector formats AccDense and AccCoo.
//
// Replaces the reference's cl_dense_vec_to_coo / cl_coo_vec_to_dense (reference src/opencl/cl_format_dense_vec.hpp:90-143,
// cl_format_coo_vec.hpp:149-158, kernels/vector_formats.cl:30,42).  Unlike the reference's atomic_inc compaction
// (which yields UNORDERED COO) the dense->COO compaction here is an order-preserving count / scan / write, so the
// result is index-sorted like the CPU path's (Dense->Dok->Coo sorts, reference src/cpu/cpu_format_dok_vec.hpp:42-69).
// Pure HBM-bound streaming: 4*n bytes read (+ 8*nvals written).
#include "block_utils.hpp"
#include "common.hpp"
#include "scan_post.hpp"
#include "ops.hpp"

namespace splahip {

    template<typename T>
    __device__ __forceinline__ bool stored(T x, T fill) { return x != fill; }// the reference's `!= fill_value` (float compare for F)

    template<typename T>
    __global__ __launch_bounds__(256) void d2c_count_kernel(const T* __restrict__ dense, uint n, T fill, uint* __restrict__ chunk) {
        __shared__ uint s_tmp[256 / kWave + 1];
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        uint            cnt   = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u)
            if (first + u < n) cnt += stored(dense[first + u], fill) ? 1u : 0u;
        uint total;
        (void) block_exclusive_sum<256>(cnt, s_tmp, total);
        if (threadIdx.x == 0) chunk[blockIdx.x] = total;
    }

    template<typename T>
    __global__ __launch_bounds__(256) void d2c_write_kernel(const T* __restrict__ dense, uint n, T fill, const uint* __restrict__ chunk,
                                                            uint* __restrict__ out_i, T* __restrict__ out_x, uint limit) {
        __shared__ uint s_tmp[256 / kWave + 1];
        if (chunk[gridDim.x] > limit) return;// (the caller wants the entries only if they are few: it reads the same count and knows)
        const uint      first = blockIdx.x * 1024u + threadIdx.x * 4u;
        T               x[4];
        bool            keep[4];
        uint            cnt = 0;
#pragma unroll
        for (uint u = 0; u < 4; ++u) {
            keep[u] = first + u < n;
            if (keep[u]) {
                x[u]    = dense[first + u];
                keep[u] = stored(x[u], fill);
            }
            cnt += keep[u] ? 1u : 0u;
        }
        uint total;
        uint pos = chunk[blockIdx.x] + block_exclusive_sum<256>(cnt, s_tmp, total);
#pragma unroll
        for (uint u = 0; u < 4; ++u)
            if (keep[u]) {
                out_i[pos] = first + u;
                out_x[pos] = x[u];
                ++pos;
            }
    }

    __global__ __launch_bounds__(256) void scatter_kernel(uint nvals, const uint* __restrict__ idx, const uint* __restrict__ val, uint* __restrict__ dense) {
        const uint k = blockIdx.x * 256 + threadIdx.x;
        if (k < nvals) dense[idx[k]] = val[k];
    }

    template<typename T>
    int dense_to_coo(uint n, const void* d_dense, uint32_t fill_bits, void* d_idx, void* d_val, uint32_t* h_count, hipStream_t s, uint limit = 0xFFFFFFFFu) {
        const uint n_chunks = (n + 1023u) / 1024u;
        const T    fill = from_bits<T>(fill_bits);
        DevMem     mem;
        SPLAHIP_TRY(mem.alloc(s, sizeof(uint) * (size_t) (n_chunks + 1)));
        uint*      d_chunk = mem.as<uint>();
        hipLaunchKernelGGL(d2c_count_kernel<T>, dim3(n_chunks), dim3(256), 0, s, static_cast<const T*>(d_dense), n, fill, d_chunk);
        ScanPost sp;
        (void) scan_launch(d_chunk, n_chunks, s, &sp);// (the count reaches the host from the scan: scan_post.hpp)
        hipLaunchKernelGGL(d2c_write_kernel<T>, dim3(n_chunks), dim3(256), 0, s, static_cast<const T*>(d_dense), n, fill, d_chunk,
                           static_cast<uint*>(d_idx), static_cast<T*>(d_val), limit);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = scan_total(h_count, d_chunk + n_chunks, sp, s);
        if (e != hipSuccess) return fail(SPLAHIP_ERROR, "dense_to_coo failed: %s", hipGetErrorString(e));
        return SPLAHIP_OK;
    }

}// namespace splahip

using namespace splahip;

extern "C" {

int splahip_vec_dense_to_coo_limited(int type, uint32_t n, const void* d_dense, uint32_t fill_bits,
                                     uint32_t* d_idx, void* d_val, uint32_t limit, uint32_t* h_count, void* stream) {
    SPLAHIP_REQUIRE(h_count != nullptr, "h_count is null");
    *h_count = 0;
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_dense && d_idx && d_val, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    switch (type) {
        case SPLAHIP_INT: return dense_to_coo<int32_t>(n, d_dense, fill_bits, d_idx, d_val, h_count, as_stream(stream), limit);
        case SPLAHIP_UINT: return dense_to_coo<uint32_t>(n, d_dense, fill_bits, d_idx, d_val, h_count, as_stream(stream), limit);
        case SPLAHIP_FLOAT: return dense_to_coo<float>(n, d_dense, fill_bits, d_idx, d_val, h_count, as_stream(stream), limit);
    }
    return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
}

int splahip_vec_dense_to_coo(int type, uint32_t n, const void* d_dense, uint32_t fill_bits,
                             uint32_t* d_idx, void* d_val, uint32_t* h_count, void* stream) {
    return splahip_vec_dense_to_coo_limited(type, n, d_dense, fill_bits, d_idx, d_val, 0xFFFFFFFFu, h_count, stream);
}

int splahip_vec_coo_to_dense(uint32_t n, uint32_t nvals, const uint32_t* d_idx, const void* d_val, uint32_t fill_bits,
                             void* d_dense, void* stream) {
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_dense, "null device pointer");
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    SPLAHIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_dense), (int) fill_bits, n, as_stream(stream)));
    if (nvals) {
        SPLAHIP_REQUIRE(d_idx && d_val, "null device pointer");
        hipLaunchKernelGGL(scatter_kernel, dim3((nvals + 255) / 256), dim3(256), 0, as_stream(stream), nvals, d_idx,
                           static_cast<const uint32_t*>(d_val), static_cast<uint32_t*>(d_dense));
        SPLAHIP_TRY(hipGetLastError());
    }
    return SPLAHIP_OK;
}
}
