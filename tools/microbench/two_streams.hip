// Do two streams of one process run concurrently when the first one's kernel waits (bounded) for the second one's?
// Kernel A polls a flag; kernel B (other stream, enqueued later) sets it.  Reports how long A waited.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void waiter(unsigned long long* flag, unsigned long long* waited) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        if (wall_clock64() - t0 > 50000000ull) break;   // 0.5 s
        __builtin_amdgcn_s_sleep(2);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *waited = wall_clock64() - t0;
}
__global__ void setter(unsigned long long* flag) {
    __hip_atomic_store(flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main() {
    for (int uncached = 0; uncached < 2; ++uncached)
        for (int blocks : {1, 64, 2048}) {
            unsigned long long *flag, *waited, h = 0;
            if (uncached) CK(hipExtMallocWithFlags((void**)&flag, 8, hipDeviceMallocUncached)); else CK(hipMalloc((void**)&flag, 8));
            CK(hipMalloc((void**)&waited, 8));
            CK(hipMemset(flag, 0, 8));
            CK(hipDeviceSynchronize());
            hipStream_t s1, s2;
            CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
            CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            hipLaunchKernelGGL(waiter, dim3(blocks), dim3(256), 0, s1, flag, waited);
            usleep(2000);
            hipLaunchKernelGGL(setter, dim3(1), dim3(64), 0, s2, flag);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&h, waited, 8, hipMemcpyDeviceToHost));
            printf("uncached=%d waiter blocks=%4d: waited %.3f ms\n", uncached, blocks, h / 1e5);
            hipStreamDestroy(s1); hipStreamDestroy(s2); hipFree(flag); hipFree(waited);
        }
    return 0;
}
