// block_utils.hpp — small wavefront / workgroup primitives (64-wide waves, gfx950) shared by the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace splahip {

    constexpr unsigned kWave = 64;

    __device__ __forceinline__ unsigned lane_id() { return threadIdx.x & (kWave - 1); }
    __device__ __forceinline__ unsigned wave_id() { return threadIdx.x / kWave; }

    // inclusive prefix sum across the 64 lanes of a wave
    // (DPP row shifts and row broadcasts: six VALU adds — the ds_bpermute form, __shfl_up, was six LDS-crossbar round trips of ~60 cycles each)
    __device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x111, 0xF, 0xF, false);// row_shr:1
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x112, 0xF, 0xF, false);// row_shr:2
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x114, 0xF, 0xF, false);// row_shr:4
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x118, 0xF, 0xF, false);// row_shr:8
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x142, 0xA, 0xF, false);// row_bcast:15 into rows 1 and 3
        x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x143, 0xC, 0xF, false);// row_bcast:31 into rows 2 and 3
        return x;
    }

    // sum across the 64 lanes of a wave, result valid in every lane
    __device__ __forceinline__ uint32_t wave_sum(uint32_t x) {// (the scan's last lane, read back as a scalar: no LDS crossbar)
        return (uint32_t) __builtin_amdgcn_readlane((int) wave_inclusive_sum(x), (int) kWave - 1);
    }

    __device__ __forceinline__ uint32_t wave_max(uint32_t x) {// (an inclusive max-scan on the same DPP pattern; 0 is the identity of an unsigned max)
        x = max(x, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x111, 0xF, 0xF, false));
        x = max(x, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x112, 0xF, 0xF, false));
        x = max(x, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x114, 0xF, 0xF, false));
        x = max(x, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x118, 0xF, 0xF, false));
        x = max(x, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x142, 0xA, 0xF, false));
        x = max(x, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x143, 0xC, 0xF, false));
        return (uint32_t) __builtin_amdgcn_readlane((int) x, (int) kWave - 1);
    }

    // Exclusive prefix sum of one value per thread across a workgroup of NT threads (NT multiple of 64, <= 1024).
    // `total` receives the workgroup sum.  `s_tmp` must hold NT/64 + 1 words of LDS.  Contains barriers.
    template<unsigned NT>
    __device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t x, uint32_t* s_tmp, uint32_t& total) {
        constexpr unsigned NW   = NT / kWave;
        const unsigned     lane = lane_id();
        const unsigned     wave = wave_id();
        const uint32_t     inc  = wave_inclusive_sum(x);
        if (lane == kWave - 1) s_tmp[wave] = inc;
        __syncthreads();
        // every wave scans the NW wave totals itself (one LDS read, one wave scan): thread 0 walking them one by one — 16 dependent LDS round trips with a
        // barrier on either side, 1023 threads waiting — was ~2.5 K cycles per call, 5 K of the 26 K cycles of a block step of mxm (round 5)
        const uint32_t wt   = lane < NW ? s_tmp[lane] : 0u;
        const uint32_t winc = wave_inclusive_sum(wt);
        const uint32_t base = (uint32_t) __builtin_amdgcn_readlane((int) (winc - wt), __builtin_amdgcn_readfirstlane((int) wave));
        total               = (uint32_t) __builtin_amdgcn_readlane((int) winc, (int) NW - 1);
        __syncthreads();// s_tmp may be reused by the caller
        return base + inc - x;
    }

    // In-order wave reduction of (valid, value) pairs with an arbitrary associative op: after the call lane 0 holds
    // x_0 (+) x_1 (+) ... (+) x_63 over the valid lanes, combined in lane order (so non-commutative ops are honoured).
    template<typename T, class Add>
    __device__ __forceinline__ void wave_reduce_ordered(T& x, bool& valid, const Add& add) {
#pragma unroll
        for (unsigned off = 1; off < kWave; off <<= 1) {
            const T   y  = __shfl_down(x, off, kWave);
            const int yv = __shfl_down((int) valid, off, kWave);
            const bool in_range = (lane_id() + off) < kWave;
            if (in_range && yv) {
                x     = valid ? add(x, y) : y;
                valid = true;
            }
        }
    }

    // Exclusive scan of `n` per-chunk counts in place by ONE workgroup of NT threads; counts[n] receives the total.
    // (second pass of the count / scan / write compaction pattern used by the plan builder and the format kernels)
    template<unsigned NT>
    __global__ __launch_bounds__(NT) void chunk_scan_kernel(uint32_t* __restrict__ counts, uint32_t n) {
        __shared__ uint32_t s_tmp[NT / kWave + 1];
        __shared__ uint32_t s_carry;
        if (threadIdx.x == 0) s_carry = 0;
        __syncthreads();
        for (uint32_t base = 0; base < n; base += NT) {
            const uint32_t i = base + threadIdx.x;
            const uint32_t x = i < n ? counts[i] : 0;
            uint32_t       total;
            const uint32_t ex    = block_exclusive_sum<NT>(x, s_tmp, total);
            const uint32_t carry = s_carry;
            if (i < n) counts[i] = carry + ex;
            __syncthreads();
            if (threadIdx.x == 0) s_carry = carry + total;
            __syncthreads();
        }
        if (threadIdx.x == 0) counts[n] = s_carry;
    }

}// namespace splahip
