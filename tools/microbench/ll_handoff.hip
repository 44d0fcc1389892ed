// What does a half-step boundary cost when it is NOT a kernel boundary?  512 workgroups of 512 threads (two per CU, as
// k_solo at configs[1]) run `iters` dependent half-steps inside ONE launch: each workgroup waits for the row of a random
// partner of the other colour (two 8-byte {32 data bits, 32-bit tag} granules per double in uncached memory, ring of 8
// generations), checks that every workgroup has finished the half-step before the previous one (plain flags, one load
// per thread, so drift is bounded and the ring cannot be overrun), "works" for W ticks of the 100 MHz clock and posts
// its own row.  Compared with the same chain as one launch per half-step.  Every wait is bounded (0.5 s).
//   hipcc --offload-arch=gfx950 -O2 -o ll_handoff ll_handoff.hip && ./ll_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kCols = 6, kRing = 8;
typedef unsigned long long u64;

__host__ __device__ inline int partner_of(int i, int h, int nb) {
    uint32_t x = (uint32_t)i * 2654435761u ^ (uint32_t)h * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    return (int)(x % (uint32_t)nb);
}
__host__ __device__ inline double mix(double own, double other, int c) { return own * 0.5 + other * 0.25 + (double)(c + 1); }

__device__ inline bool poll(const u64* p, u64 tag, double& v, int* err) {
    const u64 t0 = wall_clock64();
    for (;;) {
        const u64 a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const u64 b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((a >> 32) == tag && (b >> 32) == tag) {
            v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
            return true;
        }
        if (wall_clock64() - t0 > 50000000ull || *(volatile int*)err) { atomicOr(err, 2); v = 0.; return false; }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ inline void post(u64* p, u64 tag, double v) {
    const u64 b = (u64)__double_as_longlong(v);
    __hip_atomic_store(p, (b & 0xffffffffull) | (tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p + 1, (b >> 32) | (tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// board[ring][walker][col][2]; prog[block] = half-steps finished
__global__ __launch_bounds__(512) void persistent(u64* board, u64* prog, int* err, int nb, int iters, int work_ticks,
                                                  int lagged, u64* stats) {
    extern __shared__ double lds[];
    const int i = blockIdx.x, tid = threadIdx.x;
    u64 waited = 0;
    for (int h = 0; h < iters; ++h) {
        const int half = h & 1, wid = half * nb + i, pid = (1 - half) * nb + partner_of(i, h, nb);
        const u64 ptag = h >= 1 ? (u64)h : 0ull, otag = h >= 2 ? (u64)(h - 1) : 0ull;
        const u64 t0 = wall_clock64();
        if (tid < 2 * kCols) {
            const int c = tid % kCols;
            const bool own = tid >= kCols;
            const u64 tag = own ? otag : ptag;
            const int w = own ? wid : pid;
            double v;
            poll(board + 2 * (((size_t)(tag % kRing) * 2 * nb + w) * kCols + c), tag, v, err);
            lds[tid] = v;
        }
        if (lagged) {
            // nobody starts half-step h before everybody has finished h - 2
            for (;;) {
                bool ok = true;
                if (h >= 2 && tid < nb) ok = __hip_atomic_load(prog + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= (u64)(h - 1);
                if (__syncthreads_and(ok || *(volatile int*)err)) break;
                if (wall_clock64() - t0 > 50000000ull) atomicOr(err, 4);
            }
        } else {
            __syncthreads();
        }
        if (tid == 0) waited += wall_clock64() - t0;
        const u64 t1 = wall_clock64();
        while (wall_clock64() - t1 < (u64)work_ticks) __builtin_amdgcn_s_sleep(4);
        __syncthreads();
        if (tid < kCols)
            post(board + 2 * (((size_t)((h + 1) % kRing) * 2 * nb + wid) * kCols + tid), (u64)(h + 1), mix(lds[kCols + tid], lds[tid], tid));
        if (tid == 0) __hip_atomic_store(prog + i, (u64)(h + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
    }
    if (tid == 0) stats[i] = waited;
}

// the same chain, one launch per half-step, rows in ordinary memory
__global__ __launch_bounds__(512) void per_launch(double* X, int nb, int h, int work_ticks) {
    extern __shared__ double lds[];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int half = h & 1, wid = half * nb + i, pid = (1 - half) * nb + partner_of(i, h, nb);
    if (tid < 2 * kCols) lds[tid] = X[(size_t)(tid >= kCols ? wid : pid) * kCols + tid % kCols];
    __syncthreads();
    const u64 t1 = wall_clock64();
    while (wall_clock64() - t1 < (u64)work_ticks) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    if (tid < kCols) X[(size_t)wid * kCols + tid] = mix(lds[kCols + tid], lds[tid], tid);
}

int main() {
    const int nb = 512, iters = 2000;
    const size_t board_bytes = (size_t)kRing * 2 * nb * kCols * 16;
    u64 *board, *prog, *stats;
    int* err;
    double* X;
    CK(hipExtMallocWithFlags((void**)&board, board_bytes, hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void**)&prog, nb * 8, hipDeviceMallocUncached));
    CK(hipMalloc((void**)&err, 4));
    CK(hipMalloc((void**)&stats, nb * 8));
    CK(hipMalloc((void**)&X, 2 * nb * kCols * 8));
    // reference on the host
    std::vector<double> ref(2 * nb * kCols, 0.);
    for (int h = 0; h < iters; ++h)
        for (int i = 0; i < nb; ++i) {
            const int half = h & 1, wid = half * nb + i, pid = (1 - half) * nb + partner_of(i, h, nb);
            for (int c = 0; c < kCols; ++c) ref[wid * kCols + c] = mix(ref[wid * kCols + c], ref[pid * kCols + c], c);
        }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 60 * 1024;   // two workgroups per CU, as k_solo
    CK(hipFuncSetAttribute((const void*)persistent, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)per_launch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persistent, 512, lds));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("resident capacity: %d per CU x %d CUs\n", per_cu, prop.multiProcessorCount);
    if (per_cu * prop.multiProcessorCount < nb) { printf("not all workgroups would be resident: not run\n"); return 1; }
    for (int work : {0, 400, 800}) {
        float ms_pl = 0.f;
        CK(hipMemset(X, 0, 2 * nb * kCols * 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int h = 0; h < iters; ++h) hipLaunchKernelGGL(per_launch, dim3(nb), dim3(512), lds, 0, X, nb, h, work);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms_pl, e0, e1));
        std::vector<double> got(2 * nb * kCols);
        CK(hipMemcpy(got.data(), X, got.size() * 8, hipMemcpyDeviceToHost));
        bool same_pl = got == ref;
        for (int lagged = 0; lagged < 2; ++lagged) {
            CK(hipMemset(board, 0, board_bytes));
            CK(hipMemset(prog, 0, nb * 8));
            CK(hipMemset(err, 0, 4));
            CK(hipDeviceSynchronize());
            float ms = 0.f;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(persistent, dim3(nb), dim3(512), lds, 0, board, prog, err, nb, iters, work, lagged, stats);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
            int herr = 0;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            std::vector<u64> hb(board_bytes / 8), st(nb);
            CK(hipMemcpy(hb.data(), board, board_bytes, hipMemcpyDeviceToHost));
            CK(hipMemcpy(st.data(), stats, nb * 8, hipMemcpyDeviceToHost));
            // final rows: walker of colour (iters-1)&1 carries tag iters, the other colour tag iters-1
            bool same = true;
            for (int w = 0; w < 2 * nb && same; ++w) {
                const u64 tag = (w / nb) == ((iters - 1) & 1) ? iters : iters - 1;
                for (int c = 0; c < kCols; ++c) {
                    const u64* p = hb.data() + 2 * (((size_t)(tag % kRing) * 2 * nb + w) * kCols + c);
                    const u64 bits = (p[0] & 0xffffffffull) | (p[1] << 32);
                    double v; memcpy(&v, &bits, 8);
                    if ((p[0] >> 32) != tag || v != ref[w * kCols + c]) same = false;
                }
            }
            double wsum = 0; for (u64 v : st) wsum += v;
            printf("work %4.1f us: one launch per half-step %.2f us/half-step (chain %s) | persistent%s %.2f us/half-step, "
                   "mean wait %.2f us, err %d, chain %s\n", work / 100., 1e3 * ms_pl / iters, same_pl ? "ok" : "WRONG",
                   lagged ? " + lagged progress check" : "", 1e3 * ms / iters, wsum / nb / iters / 100., herr,
                   same ? "ok" : "WRONG");
        }
    }
    return 0;
}
