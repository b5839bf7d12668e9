// Calibration of FETCH_SIZE for 4-byte gathers (round 3): does an L2 miss of a 4-byte load fetch the whole 128-byte line or a 64-byte sector?
// One workgroup (one XCD's L2), a table far smaller than that L2, every line touched once:
//   gather_first_half : word 0 of every 128-B line                          -> N requests
//   gather_both_halves: word 0 of every line, then (after a barrier and s_waitcnt) word 16 of the same lines
//                       -> still N requests if a miss brings the line, 2 N if it brings a 64-B sector
//   stream_read       : the same table with 16-B loads per lane, coalesced   -> the guide's case: 128-B requests tallied as 64 B
// Run each under rocprofv3 --pmc FETCH_SIZE (tools/gpu_sector_calib.sh) and compare FETCH_SIZE * 1024 with lines * 128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void gather_first_half(const unsigned* __restrict__ t, unsigned n_lines, unsigned* out) {
    unsigned acc = 0;
    for (unsigned l = threadIdx.x; l < n_lines; l += 256) acc += t[(size_t) l * 32];
    if (acc == 0xdeadbeef) out[0] = acc;
}
__global__ __launch_bounds__(256) void gather_both_halves(const unsigned* __restrict__ t, unsigned n_lines, unsigned* out) {
    unsigned acc = 0;
    for (unsigned l = threadIdx.x; l < n_lines; l += 256) acc += t[(size_t) l * 32];
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (unsigned l = threadIdx.x; l < n_lines; l += 256) acc += t[(size_t) l * 32 + 16];
    if (acc == 0xdeadbeef) out[0] = acc;
}
__global__ __launch_bounds__(256) void gather_second_half_only(const unsigned* __restrict__ t, unsigned n_lines, unsigned* out) {
    unsigned acc = 0;
    for (unsigned l = threadIdx.x; l < n_lines; l += 256) acc += t[(size_t) l * 32 + 16];
    if (acc == 0xdeadbeef) out[0] = acc;
}
__global__ __launch_bounds__(256) void stream_read(const uint4* __restrict__ t, unsigned n_lines, unsigned* out) {
    unsigned acc = 0;
    for (unsigned i = threadIdx.x; i < n_lines * 8; i += 256) acc += t[i].x;
    if (acc == 0xdeadbeef) out[0] = acc;
}
// many workgroups, random lines of a large table (the SpMV's pattern): one 4-byte word per line
__global__ __launch_bounds__(256) void gather_random(const unsigned* __restrict__ t, unsigned n_lines, unsigned per_thread, unsigned* out) {
    unsigned acc = 0, x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (unsigned i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        acc += t[(size_t) ((x >> 8) % n_lines) * 32 + ((x >> 3) & 31u)];
    }
    if (acc == 0xdeadbeef) out[0] = acc;
}

int main() {
    const unsigned n_lines = 8192;// 1 MB
    unsigned *t = nullptr, *out = nullptr, *big = nullptr;
    const unsigned big_lines = 1u << 22;// 512 MB: past the L2s and the Infinity Cache
    if (hipMalloc(&t, (size_t) n_lines * 128) != hipSuccess || hipMalloc(&out, 64) != hipSuccess || hipMalloc(&big, (size_t) big_lines * 128) != hipSuccess) return 1;
    hipMemset(t, 0, (size_t) n_lines * 128);
    hipMemset(big, 0, (size_t) big_lines * 128);
    hipDeviceSynchronize();
    // a 768 MB memset between the cases pushes the table out of the L2s and the Infinity Cache
    auto flush = [&]() { hipMemset(big, 1, (size_t) big_lines * 128); hipDeviceSynchronize(); };
    for (int rep = 0; rep < 3; ++rep) {
        flush(); hipLaunchKernelGGL(gather_first_half, dim3(1), dim3(256), 0, 0, t, n_lines, out); hipDeviceSynchronize();
        flush(); hipLaunchKernelGGL(gather_both_halves, dim3(1), dim3(256), 0, 0, t, n_lines, out); hipDeviceSynchronize();
        flush(); hipLaunchKernelGGL(gather_second_half_only, dim3(1), dim3(256), 0, 0, t, n_lines, out); hipDeviceSynchronize();
        flush(); hipLaunchKernelGGL(stream_read, dim3(1), dim3(256), 0, 0, reinterpret_cast<const uint4*>(t), n_lines, out); hipDeviceSynchronize();
        flush(); hipLaunchKernelGGL(gather_random, dim3(4096), dim3(256), 0, 0, big, big_lines, 16u, out); hipDeviceSynchronize();
    }
    std::printf("lines %u (x128 B = %u KB); gather_random: %u loads\n", n_lines, n_lines / 8, 4096u * 256u * 16u);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 2;
}
