// LDS accumulate throughput on gfx950 with addresses computed in registers (no global traffic): what one CU sustains.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int R = 16384;
constexpr int ITER = 512;// steps of 4 entries per lane
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// MODE 0 ds_add_f32, 1 ds_add_u32, 2 ds_min_i32, 3 plain RMW (batched 4), 4 CAS float add (batched 4 + retry), 5 plain store
// PAT 0 conflict-free (lane-distinct), 1 uniform random, 2 skewed (25 % of the accesses go to 16 hot words)
template<int MODE, int PAT>
__global__ __launch_bounds__(1024) void bench(unsigned* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned s[R + 4];
    for (int i = threadIdx.x; i < R + 4; i += 1024) s[i] = 0;
    __syncthreads();
    unsigned seed = blockIdx.x * 1024 + threadIdx.x;
    const unsigned lane = threadIdx.x & 63;
    for (int it = 0; it < ITER; ++it) {
        unsigned a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned h = hash(seed + (it * 4 + j) * 0x9E3779B9u);
            if (PAT == 0) a[j] = ((h >> 8) % (R / 64)) * 64 + lane;
            else if (PAT == 1) a[j] = h % R;
            else a[j] = (h & 3) == 0 ? ((h >> 4) & 15) : (h >> 8) % R;
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) (void) __hip_atomic_fetch_add(reinterpret_cast<float*>(s + a[j]), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) (void) __hip_atomic_fetch_add(s + a[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) (void) __hip_atomic_fetch_min(reinterpret_cast<int*>(s + a[j]), (int) a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if constexpr (MODE == 3) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = *reinterpret_cast<volatile float*>(s + a[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<volatile float*>(s + a[j]) = o[j] + 1.0f;
        } else if constexpr (MODE == 4) {
            unsigned o[4], seen[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = *reinterpret_cast<volatile unsigned*>(s + a[j]);
            bool again = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                seen[j] = o[j];
                unsigned want = __float_as_uint(__uint_as_float(o[j]) + 1.0f);
                if (!__hip_atomic_compare_exchange_strong(s + a[j], &seen[j], want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) again = true;
            }
            if (again) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (seen[j] != o[j]) {// this one lost: retry until it lands
                        unsigned old = seen[j];
                        for (;;) {
                            unsigned want = __float_as_uint(__uint_as_float(old) + 1.0f), sn = old;
                            if (__hip_atomic_compare_exchange_strong(s + a[j], &sn, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                            old = sn;
                        }
                    }
                }
            }
        } else if constexpr (MODE == 6) {// double add: the slice as R / 2 doubles
            double* d = reinterpret_cast<double*>(s);
#pragma unroll
            for (int j = 0; j < 4; ++j) (void) __hip_atomic_fetch_add(d + (a[j] >> 1), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if constexpr (MODE == 7) {// 64-bit integer add
            unsigned long long* d = reinterpret_cast<unsigned long long*>(s);
#pragma unroll
            for (int j = 0; j < 4; ++j) (void) __hip_atomic_fetch_add(d + (a[j] >> 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if constexpr (MODE == 8) {// packed half add
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 one = {(_Float16) 1.0f, (_Float16) 1.0f};
#pragma unroll
            for (int j = 0; j < 4; ++j) (void) __builtin_amdgcn_ds_atomic_fadd_v2f16(reinterpret_cast<__attribute__((address_space(3))) h2*>((__attribute__((address_space(3))) unsigned*) (s + a[j])), one);
        } else if constexpr (MODE == 9) {// float add WITH its return value used (ds_add_rtn_f32)
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += __hip_atomic_fetch_add(reinterpret_cast<float*>(s + a[j]), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (acc == -12345.f) s[0] = 1;
        } else if constexpr (MODE == 10) {// float add through inline ds_add_f32 (no compiler lowering in between)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned addr = (unsigned) (size_t) (__attribute__((address_space(3))) unsigned*) (s + a[j]);
                asm volatile("ds_add_f32 %0, %1" ::"v"(addr), "v"(1.0f) : "memory");
            }
        } else if constexpr (MODE == 11) {// float max as an atomic (does any fp32 LDS atomic run at rate?)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned addr = (unsigned) (size_t) (__attribute__((address_space(3))) unsigned*) (s + a[j]);
                asm volatile("ds_max_f32 %0, %1" ::"v"(addr), "v"(1.0f) : "memory");
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<volatile unsigned*>(s + a[j]) = a[j];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[blockIdx.x % R];
}
template<int MODE, int PAT>
void run(unsigned* d_out, const char* name) {
    const int blocks = 512;// 2 per CU
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    bench<MODE, PAT><<<blocks, 1024>>>(d_out);
    (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) bench<MODE, PAT><<<blocks, 1024>>>(d_out);
    (void) hipEventRecord(e1);
    (void) hipEventSynchronize(e1);
    float ms;
    (void) hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    double ops = (double) blocks * 1024 * ITER * 4;
    printf("  %-28s %8.3f ms   %6.2f lane-ops/clk/CU (2.1 GHz)   %5.1f cycles per wave instruction\n", name, ms, ops / (ms * 1e-3) / 256 / 2.1e9,
           64.0 / (ops / (ms * 1e-3) / 256 / 2.1e9));
}
template<int PAT>
void all(unsigned* d_out, const char* pn) {
    printf("pattern: %s\n", pn);
    run<5, PAT>(d_out, "plain store");
    run<1, PAT>(d_out, "ds_add_u32");
    run<2, PAT>(d_out, "ds_min_i32");
    run<0, PAT>(d_out, "ds_add_f32");
    run<9, PAT>(d_out, "ds_add_rtn_f32 (value used)");
    run<10, PAT>(d_out, "ds_add_f32 (inline asm)");
    run<11, PAT>(d_out, "ds_max_f32 (inline asm)");
    run<3, PAT>(d_out, "plain read, add, write x4");
    run<4, PAT>(d_out, "CAS float add x4 + retry");
    run<6, PAT>(d_out, "ds_add_f64");
    run<7, PAT>(d_out, "ds_add_u64");
    run<8, PAT>(d_out, "ds_pk_add_f16");
}
int main() {
    unsigned* d_out;
    (void) hipMalloc(&d_out, 4096 * 4);
    all<0>(d_out, "conflict-free (distinct banks per lane)");
    all<1>(d_out, "uniform random over 16 Ki words");
    all<2>(d_out, "skewed: 25 % of the accesses to 16 hot words");
    return 0;
}
