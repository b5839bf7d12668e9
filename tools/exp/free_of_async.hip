// Round 5: may hipFree release a block that came from hipMallocAsync (the block cache of runtime.hip evicts that way)?  Prints the error codes.
#include <hip/hip_runtime.h>
#include <cstdio>
int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    for (int rep = 0; rep < 3; ++rep) {
        void* p = nullptr;
        hipError_t a = hipMallocAsync(&p, 64 << 20, s);
        hipError_t m = hipMemsetAsync(p, 1, 64 << 20, s);
        hipError_t f = hipFree(p);
        std::printf("rep %d: hipMallocAsync %d  memset %d  hipFree %d (%s)\n", rep, (int) a, (int) m, (int) f, hipGetErrorString(f));
    }
    void* q = nullptr;
    hipMallocAsync(&q, 1 << 20, s);
    hipStreamDestroy(s);
    hipError_t f = hipFree(q);
    std::printf("after its stream was destroyed: hipFree %d (%s); last error %d\n", (int) f, hipGetErrorString(f), (int) hipGetLastError());
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    std::printf("free %.1f GiB of %.1f\n", fr / 1073741824.0, tot / 1073741824.0);
    return 0;
}
