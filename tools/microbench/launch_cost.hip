// What a kernel boundary costs as a function of the launch shape: every workgroup spins for `work` shader cycles
// (s_memtime), so per-launch time minus work / clock = dispatch ramp + drain + the gap between dependent launches.
//   hipcc --offload-arch=gfx950 -O2 -o launch_cost launch_cost.hip && ./launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { double pad[88]; };   // ~700 bytes of kernel arguments, as a DevProblem by value

__global__ void spin(unsigned long long work, unsigned long long* out) {
    extern __shared__ double lds[];
    unsigned long long t0, t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (threadIdx.x == 0) lds[0] = 1.;
    do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); } while (t - t0 < work);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t - t0;
}
__global__ void spin_big(Big b, unsigned long long work, unsigned long long* out) {
    extern __shared__ double lds[];
    unsigned long long t0, t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (threadIdx.x == 0) lds[0] = b.pad[threadIdx.x & 7];
    do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); } while (t - t0 < work);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t - t0;
}

// The same with the memory traffic of a half-step's boundary: every workgroup first reads 64 bytes another workgroup
// wrote in the PREVIOUS launch (a walker's row), spins, and at the end writes its own 64 bytes.
__global__ void spin_rw(unsigned long long work, double* rows, int parity, unsigned long long* out) {
    extern __shared__ double lds[];
    unsigned long long t0, t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const unsigned long long w0 = wall_clock64();
    const int n = gridDim.x, i = blockIdx.x, j = (i * 37 + 11) % n;
    double v = 0.;
    if (threadIdx.x < 8) v = rows[((size_t)(parity ^ 1) * n + j) * 8 + threadIdx.x];
    if (threadIdx.x == 0) lds[0] = v;
    do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); } while (t - t0 < work);
    if (threadIdx.x < 8) rows[((size_t)parity * n + i) * 8 + threadIdx.x] = v + 1.;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t - t0; out[1] = wall_clock64() - w0; }
}

int main() {
    unsigned long long* out;
    CK(hipMalloc(&out, 64));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)spin_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int reps = 2000;
    struct Shape { int wgs, threads, lds_kb, big; };
    const Shape shapes[] = {{512, 512, 40, 1}, {512, 512, 40, 0}, {512, 512, 1, 0}, {512, 256, 40, 0}, {256, 512, 40, 0},
                            {256, 1024, 40, 0}, {1024, 256, 20, 0}, {256, 256, 40, 0}, {512, 64, 1, 0}, {1, 64, 1, 0}};
    Big big{};
    for (unsigned long long work : {0ull, 4000ull, 12000ull}) {
        for (const Shape& s : shapes) {
            for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms up
                CK(hipEventRecord(a, st));
                for (int r = 0; r < reps; ++r) {
                    if (s.big) hipLaunchKernelGGL(spin_big, dim3(s.wgs), dim3(s.threads), s.lds_kb * 1024, st, big, work, out);
                    else hipLaunchKernelGGL(spin, dim3(s.wgs), dim3(s.threads), s.lds_kb * 1024, st, work, out);
                }
                CK(hipEventRecord(b, st));
                CK(hipStreamSynchronize(st));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, a, b));
                if (pass == 1)
                    printf("work %6llu cycles  %4d wgs x %4d threads  lds %3d KiB  args %s : %.3f us per launch\n", work, s.wgs,
                           s.threads, s.lds_kb, s.big ? "700 B" : "small", 1e3 * ms / reps);
            }
        }
    }
    double* rows;
    CK(hipMalloc(&rows, 2 * 1024 * 8 * sizeof(double)));
    CK(hipMemset(rows, 0, 2 * 1024 * 8 * sizeof(double)));
    CK(hipFuncSetAttribute((const void*)spin_rw, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (unsigned long long work : {0ull, 4000ull, 6000ull, 8000ull, 10000ull, 12000ull, 16000ull, 24000ull, 48000ull}) {
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(a, st));
            for (int r = 0; r < reps; ++r)
                hipLaunchKernelGGL(spin_rw, dim3(512), dim3(512), 40 * 1024, st, work, rows, r & 1, out);
            CK(hipEventRecord(b, st));
            CK(hipStreamSynchronize(st));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, a, b));
            unsigned long long h[2];
            CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
            if (pass == 1)
                printf("work %6llu cycles + row read/write  512 wgs x 512 threads : %.3f us per launch  (in-kernel: %llu ticks of s_memtime = %.2f us of the 100 MHz clock)\n",
                       work, 1e3 * ms / reps, h[0], h[1] / 100.);
        }
    }
    return 0;
}
