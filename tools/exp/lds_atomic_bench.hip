// LDS atomic throughput on gfx950: ds_add_f32 / ds_add_u32 / ds_min_i32 / plain RMW / plain store, by address pattern.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/bin/lds_atomic_bench tools/exp/lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int R = 16384;
constexpr int ITER = 128;

// mode: 0 ds_add_f32, 1 ds_add_u32, 2 ds_min_i32, 3 plain store, 4 plain read-add-write, 5 ds_add_rtn_f32 (value used), 6 ds_or_b32, 7 ds_add_f32 via 2 half-wave passes
template<int MODE>
__global__ __launch_bounds__(1024) void bench(const unsigned* __restrict__ idx, unsigned* __restrict__ out, int n_idx) {
    __shared__ unsigned s[R + 4];
    for (int i = threadIdx.x; i < R + 4; i += 1024) s[i] = 0;
    __syncthreads();
    // every thread walks ITER indices; the index stream is per-thread contiguous chunks preloaded to registers in groups of 8
    const unsigned* p = idx + ((size_t) blockIdx.x * 1024 + threadIdx.x) * ITER % n_idx;
    float acc = 0.f;
    for (int it = 0; it < ITER; it += 8) {
        unsigned a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = p[it + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned* q = s + a[j];
            if constexpr (MODE == 0) (void) __hip_atomic_fetch_add(reinterpret_cast<float*>(q), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (MODE == 1) (void) __hip_atomic_fetch_add(q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (MODE == 2) (void) __hip_atomic_fetch_min(reinterpret_cast<int*>(q), (int) a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (MODE == 3) *reinterpret_cast<volatile unsigned*>(q) = a[j];
            else if constexpr (MODE == 4) {
                volatile float* f = reinterpret_cast<volatile float*>(q);
                *f = *f + 1.0f;
            } else if constexpr (MODE == 5) acc += __hip_atomic_fetch_add(reinterpret_cast<float*>(q), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (MODE == 6) (void) __hip_atomic_fetch_or(q, a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (MODE == 7) {// float add by compare-and-swap on the bits
                unsigned old = *reinterpret_cast<volatile unsigned*>(q);
                for (;;) {
                    unsigned want = __float_as_uint(__uint_as_float(old) + 1.0f);
                    unsigned seen = old;
                    if (__hip_atomic_compare_exchange_strong(q, &seen, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                    old = seen;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[blockIdx.x % R] + (unsigned) acc;
}

template<int MODE>
float run(const unsigned* d_idx, unsigned* d_out, int n_idx, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    bench<MODE><<<blocks, 1024>>>(d_idx, d_out, n_idx);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) bench<MODE><<<blocks, 1024>>>(d_idx, d_out, n_idx);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    const int blocks = 512;
    const size_t n_idx = (size_t) blocks * 1024 * ITER;
    std::vector<unsigned> h(n_idx);
    unsigned *d_idx, *d_out;
    hipMalloc(&d_idx, n_idx * 4);
    hipMalloc(&d_out, blocks * 4);
    const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_min_i32", "plain store", "plain read+add+write", "ds_add_rtn_f32", "ds_or_b32", "float add by CAS loop"};
    for (int pat = 0; pat < 4; ++pat) {
        // address patterns per wave instruction (64 consecutive threads, same j): 0 = conflict-free (lane), 1 = uniform random in R, 2 = random but 8 lanes share
        // each address, 3 = all 64 lanes the same address
        for (size_t t = 0; t < (size_t) blocks * 1024; ++t)
            for (int k = 0; k < ITER; ++k) {
                unsigned lane = t & 63, a;
                if (pat == 0) a = (unsigned) ((lane + 64 * ((t / 64 + k) % (R / 64))));
                else if (pat == 1) a = (unsigned) (rand() % R);
                else if (pat == 2) a = (unsigned) (((t / 64 * 131 + k * 17) % (R / 8)) * 8 + (lane % 8));
                else a = (unsigned) ((t / 64 + k) % R);
                if (pat == 2) a = (unsigned) ((((t / 64) * 131 + k * 17 + (lane / 8)) * 2654435761u) % R);// 8 lanes share one address
                h[t * ITER + k] = a;
            }
        hipMemcpy(d_idx, h.data(), n_idx * 4, hipMemcpyHostToDevice);
        const char* pn[] = {"conflict-free (address = lane + 64 k)", "uniform random over 16 Ki words", "8 lanes per address", "64 lanes one address"};
        printf("pattern: %s\n", pn[pat]);
        float ms[8] = {run<0>(d_idx, d_out, n_idx, blocks), run<1>(d_idx, d_out, n_idx, blocks), run<2>(d_idx, d_out, n_idx, blocks), run<3>(d_idx, d_out, n_idx, blocks),
                       run<4>(d_idx, d_out, n_idx, blocks), run<5>(d_idx, d_out, n_idx, blocks), run<6>(d_idx, d_out, n_idx, blocks), run<7>(d_idx, d_out, n_idx, blocks)};
        for (int m = 0; m < 8; ++m) {
            // lane-ops per clock per CU at 2.1 GHz nominal, 256 CUs
            double ops = (double) n_idx;
            printf("  %-22s %8.3f ms   %6.2f lane-ops/clk/CU (at 2.1 GHz)\n", names[m], ms[m], ops / (ms[m] * 1e-3) / 256 / 2.1e9);
        }
    }
    return 0;
}
