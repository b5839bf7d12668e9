// matrix_ops.hip — exec_m_transpose and exec_v_map on the device (the cached transpose a dense-frontier
// vxm wants, with the unary op the reference applies on the way).
//
//   splahip_m_transpose    semantics: reference Algo_m_transpose_cpu::execute_csr, src/cpu/cpu_m_transpose.hpp:141-185 —
//                          R = op(M)^T as CSR; inside a row of R the entries keep the row-major order of M (ascending source
//                          row), exactly what the reference's counting sort produces
//   splahip_map            semantics: reference Algo_v_map_cpu, src/cpu/cpu_v_map.hpp:70-121 — out[i] = op(x[i]) over a dense
//                          image or over the stored values of a sparse one (the caller copies the indices)
//
// The transpose is a STABLE least-significant-digit radix sort of the entries by column (rocprim, ceil(log2 n_cols) key bits),
// carrying (source row, value bits) as one 8-byte payload so that no pass gathers: a 4-byte gather per entry costs a whole
// 128-byte request on this chip, a streamed 12-byte record does not.  HBM-bound: per radix pass 24 B per entry.
#include "common.hpp"
#include "ops.hpp"

#include <rocprim/device/device_radix_sort.hpp>

namespace splahip {

    constexpr uint kEntriesPerThread = 8;

    // payload[k] = (row of entry k) << 32 | value bits.  A workgroup owns a tile of 2048 consecutive entries and its lanes stride across
    // it (coalesced value reads and payload writes); the row of an entry = the last row with Ap[row] <= k, by a binary search inside the
    // tile's own row range (a few steps over row pointers the whole workgroup keeps re-reading, i.e. cache hits).
    __global__ __launch_bounds__(256) void transpose_payload_kernel(uint n_rows, uint nnz, const uint* __restrict__ Ap, const uint* __restrict__ Ax,
                                                                    unsigned long long* __restrict__ payload) {
        const unsigned long long tile = (unsigned long long) blockIdx.x * (256 * kEntriesPerThread);
        if (tile >= nnz) return;
        const uint t0 = (uint) tile;
        const uint t1 = (uint) min((unsigned long long) nnz, tile + 256 * kEntriesPerThread) - 1u;// the tile's last entry
        auto row_of = [&](uint k, uint lo, uint hi) {// Ap[lo] <= k < Ap[hi]
            while (hi - lo > 1) {
                const uint mid = lo + ((hi - lo) >> 1);
                if (Ap[mid] <= k) lo = mid;
                else
                    hi = mid;
            }
            return lo;
        };
        const uint r0 = row_of(t0, 0, n_rows);
        const uint r1 = row_of(t1, r0, n_rows);
#pragma unroll
        for (uint i = 0; i < kEntriesPerThread; ++i) {
            const unsigned long long kk = tile + threadIdx.x + 256 * i;// (64-bit: a tile may end at nnz = 2^32 - 1)
            if (kk <= t1) {
                const uint k = (uint) kk;
                payload[k]   = ((unsigned long long) row_of(k, r0, r1 + 1) << 32) | Ax[k];
            }
        }
    }

    template<typename T>
    __global__ __launch_bounds__(256) void transpose_emit_kernel(uint nnz, const unsigned long long* __restrict__ payload, uint* __restrict__ Rj,
                                                                 T* __restrict__ Rx, int op) {
        const uint p = blockIdx.x * 256 + threadIdx.x;
        if (p >= nnz) return;
        const unsigned long long w = payload[p];
        Rj[p]                      = (uint) (w >> 32);
        const uint bits            = (uint) w;
        T          x;
        __builtin_memcpy(&x, &bits, 4);
        Rx[p] = eval_un(op, x);
    }

    // Rp[j] = number of sorted keys < j, j in [0, n_cols]
    __global__ __launch_bounds__(256) void transpose_offsets_kernel(uint n_cols, uint nnz, const uint* __restrict__ keys, uint* __restrict__ Rp) {
        const uint j = blockIdx.x * 256 + threadIdx.x;
        if (j > n_cols) return;
        uint lo = 0, hi = nnz;
        while (lo < hi) {
            const uint mid = lo + ((hi - lo) >> 1);
            if (keys[mid] < j) lo = mid + 1;
            else
                hi = mid;
        }
        Rp[j] = lo;
    }

    template<typename T>
    __global__ __launch_bounds__(256) void map_kernel(uint n, const T* __restrict__ x, T* __restrict__ out, int op) {
        const uint i = blockIdx.x * 256 + threadIdx.x;
        if (i < n) out[i] = eval_un(op, x[i]);
    }

    static uint key_bits(uint n) {
        uint b = 1;
        while (b < 32 && (1ull << b) < n) ++b;
        return b;
    }

}// namespace splahip

using namespace splahip;

extern "C" int splahip_m_transpose(int type, uint32_t n_rows, uint32_t n_cols, uint32_t nnz, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax,
                                     uint32_t* d_Rp, uint32_t* d_Rj, void* d_Rx, int op_apply, void* stream) {
    if (int st = ensure_device()) return st;
    SPLAHIP_REQUIRE(type >= SPLAHIP_INT && type <= SPLAHIP_FLOAT, "element type");
    SPLAHIP_REQUIRE(unop_valid_for(op_apply, type), "unary op for this element type");
    SPLAHIP_REQUIRE(d_Rp && (nnz == 0 || (d_Ap && d_Aj && d_Ax && d_Rj && d_Rx)), "null array");
    hipStream_t s = as_stream(stream);
    if (nnz == 0) {
        SPLAHIP_TRY(hipMemsetAsync(d_Rp, 0, (size_t(n_cols) + 1) * 4, s));
        return SPLAHIP_OK;
    }
    DevMem payload_in, payload_out, keys_out, tmp;
    SPLAHIP_TRY(payload_in.alloc(s, size_t(nnz) * 8));
    SPLAHIP_TRY(payload_out.alloc(s, size_t(nnz) * 8));
    SPLAHIP_TRY(keys_out.alloc(s, size_t(nnz) * 4));
    const uint threads = (nnz + kEntriesPerThread - 1) / kEntriesPerThread;
    transpose_payload_kernel<<<(threads + 255) / 256, 256, 0, s>>>(n_rows, nnz, d_Ap, static_cast<const uint*>(d_Ax), payload_in.as<unsigned long long>());
    const uint bits  = key_bits(n_cols);
    size_t     bytes = 0;
    SPLAHIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, d_Aj, keys_out.as<uint32_t>(), payload_in.as<unsigned long long>(), payload_out.as<unsigned long long>(),
                                          nnz, 0, bits, s));
    SPLAHIP_TRY(tmp.alloc(s, bytes));
    SPLAHIP_TRY(rocprim::radix_sort_pairs(tmp.p, bytes, d_Aj, keys_out.as<uint32_t>(), payload_in.as<unsigned long long>(), payload_out.as<unsigned long long>(),
                                          nnz, 0, bits, s));
    transpose_offsets_kernel<<<(n_cols + 1 + 255) / 256, 256, 0, s>>>(n_cols, nnz, keys_out.as<uint32_t>(), d_Rp);
    const uint grid = (nnz + 255) / 256;
    switch (type) {
        case SPLAHIP_INT: transpose_emit_kernel<int32_t><<<grid, 256, 0, s>>>(nnz, payload_out.as<unsigned long long>(), d_Rj, static_cast<int32_t*>(d_Rx), op_apply); break;
        case SPLAHIP_UINT: transpose_emit_kernel<uint32_t><<<grid, 256, 0, s>>>(nnz, payload_out.as<unsigned long long>(), d_Rj, static_cast<uint32_t*>(d_Rx), op_apply); break;
        default: transpose_emit_kernel<float><<<grid, 256, 0, s>>>(nnz, payload_out.as<unsigned long long>(), d_Rj, static_cast<float*>(d_Rx), op_apply); break;
    }
    SPLAHIP_TRY(hipGetLastError());
    return SPLAHIP_OK;
}

extern "C" int splahip_map(int type, uint32_t n, const void* d_x, void* d_out, int op, void* stream) {
    if (int st = ensure_device()) return st;
    SPLAHIP_REQUIRE(type >= SPLAHIP_INT && type <= SPLAHIP_FLOAT, "element type");
    SPLAHIP_REQUIRE(unop_valid_for(op, type), "unary op for this element type");
    if (n == 0) return SPLAHIP_OK;
    SPLAHIP_REQUIRE(d_x && d_out, "null array");
    hipStream_t s    = as_stream(stream);
    const uint  grid = (n + 255) / 256;
    switch (type) {
        case SPLAHIP_INT: map_kernel<int32_t><<<grid, 256, 0, s>>>(n, static_cast<const int32_t*>(d_x), static_cast<int32_t*>(d_out), op); break;
        case SPLAHIP_UINT: map_kernel<uint32_t><<<grid, 256, 0, s>>>(n, static_cast<const uint32_t*>(d_x), static_cast<uint32_t*>(d_out), op); break;
        default: map_kernel<float><<<grid, 256, 0, s>>>(n, static_cast<const float*>(d_x), static_cast<float*>(d_out), op); break;
    }
    SPLAHIP_TRY(hipGetLastError());
    return SPLAHIP_OK;
}
