// Throughput of the FP64 / int32 VALU instructions used by the band-sum loop, on a fully occupied gfx950.
// Each wave issues 8 independent chains x ITER iterations of ONE instruction; 8 waves per SIMD are resident.
// Reported: cycles per wave-instruction per SIMD at the measured clock (s_memtime ticks / s_memrealtime 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double seed, int iseed) {
    double a[8];
    int b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i * 0.37 + threadIdx.x * 1e-3; b[i] = iseed + i + threadIdx.x; }
    const double c = 1.0000001, d = 1e-9;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(d));
            if (OP == 3) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b[i]) : "v"(a[i]));
            if (OP == 5) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i] & 1));
            if (OP == 6) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_and_b32 %0, %0, %1" : "+v"(b[i]) : "v"(0x7fffffff));
            if (OP == 8) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(b[i]) : "v"(3));
            if (OP == 9) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 10) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(b[i]));
            if (OP == 11) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(1.0000001f), "v"(1e-9f));
            if (OP == 12) asm volatile("v_mov_b32 %0, %1" : "=v"(b[i]) : "v"(b[(i + 1) & 7]));
            if (OP == 13) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            if (OP == 14) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    if (s == 12345.678) out[0] = s;
}

template <int OP>
double run(const char* name) {
    double* out;
    hipMalloc(&out, 8);
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    // wave-instructions per SIMD = 8 waves * ITER * 8
    const double per_simd = 8.0 * ITER * 8;
    const double ns_per_instr = ms * 1e6 / per_simd;
    printf("%-16s %8.3f ms  %6.3f ns per wave-instr per SIMD  = %5.2f cycles @2.4GHz, %5.2f @2.1GHz\n", name, ms,
           ns_per_instr, ns_per_instr * 2.4, ns_per_instr * 2.1);
    hipFree(out);
    return ns_per_instr;
}

int main() {
    run<0>("v_fma_f64"); run<1>("v_mul_f64"); run<2>("v_add_f64"); run<3>("v_rndne_f64"); run<4>("v_cvt_i32_f64");
    run<5>("v_ldexp_f64"); run<6>("v_rcp_f64"); run<7>("v_and_b32"); run<8>("v_lshl_add_u32"); run<9>("v_min_f64");
    run<10>("v_ashrrev_i32"); run<11>("v_fma_f32"); run<12>("v_mov_b32"); run<13>("v_cvt_f64_i32"); run<14>("v_cmp_gt_f64");
    return 0;
}
