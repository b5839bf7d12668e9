// atomic_bubble.hip — what does a RETURNING global atomic cost a CU that is streaming?  (round 5: the run-time schedule of the blocks kernel claimed its
// work with one memory-side atomic per wave and 16 KB of stream and ran 3 x slower than the static pieces: profiles/r05_notes.md)
//
// Every wave of a 1024-thread workgroup (one per CU) streams 16 B per lane, 4 loads in flight, and every `every` loads issues ONE atomic add from lane 0
// whose result it reads `every` loads later:
//   A = 0 no atomic                         A = 1 agent scope, returning, one address per 16 workgroups       A = 2 workgroup scope (performed at the XCD's L2), returning, one address per XCD
//   A = 3 agent scope, returning, an address per wave (no contention)        A = 4 agent scope, NOT returning, shared address
//   A = 5 scalar s_atomic_add glc, one address per XCD                      A = 6 system scope, returning, one address per 16 workgroups
//   A = 7 / 8 no atomic, but every wave walks its own span / 8 KB runs (what a wave-level claim makes of the stream) instead of the workgroup's waves side by side, 1 KB each
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/bin/atomic_bubble tools/exp/atomic_bubble.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template<int A, int EVERY>
__global__ __launch_bounds__(1024) void bubble_kernel(const u4* __restrict__ src, size_t words_per_wg, uint32_t* __restrict__ ctr, uint32_t* __restrict__ out) {
    constexpr int  NW   = 16, U = 4;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u4*      base = src + (size_t) blockIdx.x * words_per_wg;
    const size_t   steps = words_per_wg / (64 * NW);
    uint32_t       xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    uint32_t opaque_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    uint32_t* p;
    if (A == 1 || A == 4 || A == 6) p = ctr + (blockIdx.x / 16) * 64;// 256-byte apart
    else if (A == 2 || A == 5) p = ctr + xcc * 64;
    else
        p = ctr + (blockIdx.x * NW + wave) * 64;
    uint32_t* const pv = p + opaque_zero;
    uint32_t        acc = 0, pend = 0;
    [[maybe_unused]] uint32_t spend = 0;
    for (size_t s = 0; s + U * EVERY <= steps; s += U * EVERY) {
        if (A == 1 || A == 3) {
            if (lane == 0) pend = __hip_atomic_fetch_add(pv, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (A == 2) {
            if (lane == 0) pend = __hip_atomic_fetch_add(pv, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (A == 6) {
            if (lane == 0) pend = __hip_atomic_fetch_add(pv, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (A == 4) {
            if (lane == 0) (void) __hip_atomic_fetch_add(pv, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (A == 5) {
            const uint64_t q = ((uint64_t) __builtin_amdgcn_readfirstlane((uint32_t) ((uint64_t) p >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t) (uint64_t) p);
            spend            = 1;
            asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(spend) : "s"(q) : "memory");
        }
#pragma unroll
        for (int e = 0; e < EVERY; ++e) {
            u4 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                r[u] = A == 7 ? __builtin_nontemporal_load(base + ((size_t) wave * steps + s + e * U + u) * 64 + lane)// every wave walks a span of its own, 1 KB a step
                     : A == 8 ? __builtin_nontemporal_load(base + ((((s + e * U + u) >> 3) * NW + wave) * 8 + ((s + e * U + u) & 7)) * 64 + lane)// 8 KB runs per wave, the workgroup's runs side by side (the octs of the run-time schedule)
                              : __builtin_nontemporal_load(base + ((s + e * U + u) * NW + wave) * 64 + lane);
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
        }
        if (A == 1 || A == 2 || A == 3 || A == 6) {
            asm volatile("" : "+v"(pend));
            acc += __builtin_amdgcn_readfirstlane(pend);
        } else if (A == 5) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(spend)::"memory");
            acc += spend;
        }
    }
    if (acc == 0x12345678u) out[1] = acc;
}

template<int A, int EVERY>
void run(const u4* src, size_t bytes, uint32_t* ctr, uint32_t* out, const char* what) {
    const int    wgs          = 256;
    const size_t words_per_wg = bytes / 16 / wgs / 1024 * 1024;
    hipEvent_t   a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ms;
    CK(hipMemset(ctr, 0, 4096 * 64 * 4));
    for (int it = 0; it < 7; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((bubble_kernel<A, EVERY>), dim3(wgs), dim3(1024), 0, 0, src, words_per_wg, ctr, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float t;
        CK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    // the counters: did every atomic land?
    std::vector<uint32_t> h(4096 * 64);
    CK(hipMemcpy(h.data(), ctr, h.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    for (uint32_t x : h) sum += x;
    const unsigned long long expect = (A == 0 || A >= 7) ? 0 : 7ull * wgs * 16 * (words_per_wg / 1024 / (4 * EVERY));
    printf("  A=%d every %2d KB : %8.1f us  %6.2f TB/s   atomics counted %llu of %llu %s   (%s)\n", A, EVERY * 4, ms[3] * 1e3, (double) words_per_wg * 16 * wgs / (ms[3] * 1e-3) / 1e12, sum, expect,
           sum == expect ? "ok" : "MISMATCH", what);
    fflush(stdout);
}

int main() {
    const size_t bytes = (size_t) 1 << 30;
    u4*          src;
    uint32_t *   ctr, *out;
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(src, 1, bytes));
    CK(hipMalloc(&ctr, 4096 * 64 * 4));
    CK(hipMalloc(&out, 64));
    run<0, 4>(src, bytes, ctr, out, "no atomic");
    run<1, 4>(src, bytes, ctr, out, "agent scope, returning, address per 16 workgroups");
    run<1, 16>(src, bytes, ctr, out, "agent scope, returning, address per 16 workgroups");
    run<3, 4>(src, bytes, ctr, out, "agent scope, returning, address per wave");
    run<2, 4>(src, bytes, ctr, out, "workgroup scope, returning, address per XCD");
    run<2, 1>(src, bytes, ctr, out, "workgroup scope, returning, address per XCD");
    run<4, 4>(src, bytes, ctr, out, "agent scope, not returning, address per 16 workgroups");
    run<6, 4>(src, bytes, ctr, out, "system scope, returning, address per 16 workgroups");
    run<5, 4>(src, bytes, ctr, out, "scalar s_atomic_add glc, address per XCD");
    run<5, 1>(src, bytes, ctr, out, "scalar s_atomic_add glc, address per XCD");
    run<7, 4>(src, bytes, ctr, out, "no atomic; every wave walks a contiguous span of its own");
    run<8, 4>(src, bytes, ctr, out, "no atomic; every wave walks 8 KB runs, the workgroup's 16 runs side by side");
    return 0;
}
