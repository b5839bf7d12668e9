// stream_bench.hip — what read rate can a CDNA4 kernel reach on a stream, and what do random 4-byte gathers beside it cost?
// (round 3 experiment behind the cob kernel's design: DESIGN.md 4.9; results in profiles/r03_stream_bench.txt)
//
//   A  read-only stream, 16 B per lane, U loads in flight per wave, W waves per workgroup, B workgroups per CU
//   B  the same stream + G gather instructions per 2 KB of stream, each lane gathering one word of a table of T bytes at a random
//      (per-lane pre-computed) position: T = 256 KB (L2), 16 MB (Infinity Cache), 64 MB
//   C  scalar-cache prefetch: one wave per workgroup walks the stream ahead of the vector loads with s_load_dwordx2 per 128-byte line
//      (does a line pulled through the scalar cache's path land in L2 early enough to shorten the vector loads' stay in the L1's miss queue?)
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/bin/stream_bench tools/exp/stream_bench.hip
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

// every workgroup reads a contiguous span of `span` 16-byte words per lane-slot; waves interleave 1 KB pieces
// S = lanes that share one 128-byte line of the table in a gather instruction (64 / S lines per instruction; the cob kernel averages ~7 lines)
template<int TPB, int U, int G, int PF, int S>
__global__ __launch_bounds__(TPB) void stream_kernel(const u4* __restrict__ src, size_t words_per_wg, const uint32_t* __restrict__ table, uint32_t tmask,
                                                     uint32_t* __restrict__ out, int pf_ahead) {
    constexpr int NW    = TPB / 64;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u4*      base = src + (size_t) blockIdx.x * words_per_wg;
    const size_t   steps = words_per_wg / (64 * NW);// 1 KB pieces per wave
    uint32_t       acc = 0;
    if (PF && wave >= NW - PF) {
        // the prefetch waves: one scalar load per 128-byte line of the whole workgroup's span; they simply run at their own pace from the start
        // (they have nothing else to do), which is the best case for the question asked
        const uint64_t p0 = (uint64_t) base;
        const size_t   lines = words_per_wg * 16 / 128;
        uint32_t       s = 0;
        for (size_t l = (size_t) (wave - (NW - PF)) * 8; l < lines; l += 8 * PF) {
            const uint64_t qv = p0 + l * 128;
            const uint64_t q  = ((uint64_t) __builtin_amdgcn_readfirstlane((uint32_t) (qv >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t) qv);
            uint32_t a0, a1, a2, a3, a4, a5, a6, a7;
            asm volatile("s_load_dword %0, %8, 0x0\n s_load_dword %1, %8, 0x80\n s_load_dword %2, %8, 0x100\n s_load_dword %3, %8, 0x180\n"
                         "s_load_dword %4, %8, 0x200\n s_load_dword %5, %8, 0x280\n s_load_dword %6, %8, 0x300\n s_load_dword %7, %8, 0x380\n s_waitcnt lgkmcnt(0)"
                         : "=&s"(a0), "=&s"(a1), "=&s"(a2), "=&s"(a3), "=&s"(a4), "=&s"(a5), "=&s"(a6), "=&s"(a7)
                         : "s"(q)
                         : "memory");
            s ^= a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
        }
        if (s == 0x12345678u) out[0] = s;
        return;
    }
    constexpr int NWV = NW - PF;// waves that stream
    if (wave >= NWV) return;
    const size_t my_steps = (steps * NW) / NWV;// the streaming waves share the whole span
    uint32_t     rnd      = (blockIdx.x * TPB + (threadIdx.x & ~63u) + lane / S) * 2654435761u + 12345u;
    for (size_t s = 0; s + U <= my_steps; s += U) {
        u4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = __builtin_nontemporal_load(base + ((s + u) * NWV + wave) * 64 + lane);
        uint32_t g[G > 0 ? G * U / 2 + 1 : 1];
        if (G > 0) {
#pragma unroll
            for (int k = 0; k < G * U / 2; ++k) {// G gathers per 2 KB of stream
                rnd  = rnd * 1664525u + 1013904223u;
                g[k] = table[((rnd >> 8) & tmask & ~31u) + (lane & 31)];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
        if (G > 0) {
#pragma unroll
            for (int k = 0; k < G * U / 2; ++k) acc += g[k];
        }
    }
    if (acc == 0x12345678u) out[1] = acc;
}

template<int TPB, int U, int G, int PF, int S = 8>
float run(const u4* src, size_t bytes, int wgs, const uint32_t* table, uint32_t tmask, uint32_t* out, int per_cu_hint) {
    const size_t words_per_wg = bytes / 16 / wgs / (64 * (TPB / 64)) * (64 * (TPB / 64));
    hipEvent_t   a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int it = 0; it < 7; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_kernel<TPB, U, G, PF, S>), dim3(wgs), dim3(TPB), 0, 0, src, words_per_wg, table, tmask, out, 0);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float t;
        CK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double tb = (double) words_per_wg * 16 * wgs / (ms[3] * 1e-3) / 1e12;
    printf("  TPB %4d  wgs %5d  U %d  G %d (%2d lines each)  PF %d  table %8u KB : %8.1f us  %6.2f TB/s of stream", TPB, wgs, U, G, 64 / S, PF, (tmask + 1) / 256, ms[3] * 1e3, tb);
    if (G > 0) {
        const double gi = (double) words_per_wg * 16 * wgs / 2048 * G;// gather instructions
        printf("   %.1f G gather lines/s + %.1f G stream lines/s", gi * (64 / S) / (ms[3] * 1e-3) / 1e9, (double) words_per_wg * 16 * wgs / 128 / (ms[3] * 1e-3) / 1e9);
    }
    printf("\n");
    fflush(stdout);
    return ms[3];
}

int main() {
    const size_t bytes = (size_t) 1 << 30;
    u4*          src;
    uint32_t *   table, *out;
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(src, 1, bytes));
    CK(hipMalloc(&table, (size_t) 64 << 20));
    CK(hipMemset(table, 0, (size_t) 64 << 20));
    CK(hipMalloc(&out, 64));
    printf("A: read-only stream of 1 GiB (16 B per lane, non-temporal)\n");
    run<1024, 4, 0, 0>(src, bytes, 256, table, 0, out, 1);
    run<1024, 8, 0, 0>(src, bytes, 256, table, 0, out, 1);
    run<1024, 4, 0, 0>(src, bytes, 512, table, 0, out, 2);
    run<512, 4, 0, 0>(src, bytes, 1024, table, 0, out, 4);
    run<256, 4, 0, 0>(src, bytes, 2048, table, 0, out, 8);
    run<256, 8, 0, 0>(src, bytes, 2048, table, 0, out, 8);
    run<256, 4, 0, 0>(src, bytes, 8192, table, 0, out, 8);
    printf("B: stream + G random gathers per 2 KB of stream (cob: 4 per chunk)\n");
    for (uint32_t tb : {16u, 20u, 22u, 24u}) {// table words: 2^16 (256 KB), 2^20 (4 MB), 2^22 (16 MB), 2^24 (64 MB)
        run<1024, 4, 4, 0, 8>(src, bytes, 256, table, (1u << tb) - 1, out, 1);
        run<1024, 4, 4, 0, 16>(src, bytes, 256, table, (1u << tb) - 1, out, 1);
        run<1024, 4, 4, 0, 4>(src, bytes, 256, table, (1u << tb) - 1, out, 1);
        run<512, 4, 4, 0, 8>(src, bytes, 512, table, (1u << tb) - 1, out, 2);
    }
    printf("C: one scalar-prefetch wave per workgroup (s_load_dword per 128-byte line), 15 streaming waves\n");
    run<1024, 4, 0, 1>(src, bytes, 256, table, 0, out, 1);
    run<1024, 4, 0, 2>(src, bytes, 256, table, 0, out, 1);
    run<1024, 4, 0, 4>(src, bytes, 256, table, 0, out, 1);
    run<1024, 4, 4, 2>(src, bytes, 256, table, (1u << 16) - 1, out, 1);
    run<1024, 4, 4, 4>(src, bytes, 256, table, (1u << 16) - 1, out, 1);
    run<1024, 4, 4, 2>(src, bytes, 256, table, (1u << 22) - 1, out, 1);
    run<1024, 4, 4, 4>(src, bytes, 256, table, (1u << 22) - 1, out, 1);
    return 0;
}
