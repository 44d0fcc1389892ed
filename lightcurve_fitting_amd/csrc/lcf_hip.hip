// MI355X (gfx950) batched light-curve log-likelihood engine: kernels + C ABI (include/lcf.h).
//
// Data layout in HBM (all float64 unless noted):
//   photometry  t[N], y[N], dy[N], 1/dy[N], int32 pt_filt[N], pt_orig[N], pt_epoch[N]; per filter a FiltDesc
//               -- points ordered by (part, filter), a part being a contiguous range of observation epochs;
//               distinct observation times epoch_t[n_epochs]
//   band tables tab[] = per filter [full | cool | hot] interleaved (a_k, W_k) pairs (the last two Gauss-compressed),
//               each padded to quads
//   walkers     P[n][n_dim] row-major; rows part[n][n_parts + 1] = partial chi^2 sums + log-prior; on the two-kernel
//               paths also derived coefficients coef[n][8] and thermal states therm[n][n_epochs] (1/T, R^2), which
//               the one-launch half-step (k_fused) keeps in LDS
// Work decomposition (k_points / k_fused): workgroup = (walker w, part j) walks the points of part j in chunks of
// 256; lane = one data point.  The exp table and all band tables are staged in LDS once per
// workgroup; lanes of a wave read the same LDS address (broadcast) because neighbouring points share a filter.
// Reductions are wave shuffles + a fixed-order LDS sum: no float atomics anywhere, results are bitwise reproducible
// run to run and independent of the GPU count.  The sampler (k_fused = a whole half-step in one launch; k_step,
// k_draws, k_make_perm), the multi-transient launches (k_*_multi) and the RCCL-driven sharded run live further down;
// the SED engine is in lcf_sed.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "lcf.h"
#include "lcf_device.h"
#include "lcf_host.h"

// Tuning knobs of the likelihood loop (measured on the 1024-walker, 3000-point fit: 2-6 chunks of prefetch and 4-6
// waves per SIMD are within 1 % of each other; 8 waves per SIMD spill and lose 40 %).
#ifndef LCF_KPRE
#define LCF_KPRE 4   // chunks of points whose operands are fetched together
#endif
#ifndef LCF_KPRE_SOLO
#define LCF_KPRE_SOLO 2  // the same in the one-workgroup-per-proposal kernel (4: 3.73e7 walker-steps/s, 3: 3.74e7, 2: 3.78e7)
#endif
#ifndef LCF_FIRST_BLOCK
#define LCF_FIRST_BLOCK 32  // steps in the first block of draw records of a run (the later ones: up to 256)
#endif
#ifndef LCF_HEAD_START
#define LCF_HEAD_START 0  // k_solo: the waves beside the serial head wait 64 x this many cycles before their first loads
#endif
#ifndef LCF_WAVES
#define LCF_WAVES 4  // occupancy the register allocator is asked to keep (waves per SIMD)
#endif

using namespace lcf;

// =================================================================================================================
// kernels
// =================================================================================================================
namespace {

// The value lane `lane` (wave-uniform) holds, in every lane: two v_readlane instead of the LDS round trip of a shuffle
// (the serial heads wait for ~20 of them with nothing else to issue).  Used by the one-workgroup-per-proposal heads; in
// k_fused's step_serial the extra scalar registers tip two instantiations into scratch, so it keeps the shuffles.
__device__ __forceinline__ double lane_value(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}

// v of the lane the DPP control names (a compile-time lane pattern inside a row of 16 lanes), two 32-bit moves
template <int CTRL>
__device__ __forceinline__ double dpp_value(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}

// Sum over the 64 lanes of a wave, the same number in every lane, in a fixed order: inside quads (lane ^ 1, lane ^ 2), the
// two quads of a row half (mirror of 8), the two halves of a row (mirror of 16) -- data-parallel moves between vector
// registers, no LDS round trip as a shuffle is -- then the four rows (r0 + r1) + (r2 + r3) through scalar registers.
__device__ inline double wave_sum(double v) {
    v += dpp_value<0xB1>(v);    // quad_perm [1, 0, 3, 2]
    v += dpp_value<0x4E>(v);    // quad_perm [2, 3, 0, 1]
    v += dpp_value<0x141>(v);   // row_half_mirror
    v += dpp_value<0x140>(v);   // row_mirror
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// One thread per walker: derived coefficients, log-prior, skip flag.
__global__ void k_prepare(const DevProblem pb, int n, const double* __restrict__ P, double* __restrict__ coef,
                          double* __restrict__ lprior, int with_prior) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    const double* p = P + (size_t)w * pb.n_dim;
    double c[kNCoef];
    walker_coefficients(pb, p, c);
    for (int i = 0; i < kNCoef; ++i) coef[(size_t)w * kNCoef + i] = c[i];
    lprior[w] = with_prior ? walker_log_prior(pb, p) : 0.;
}

// The tables of a point's filter (offsets relative to `base`): the full one and up to two Gauss-compressed
// companions -- the "cool" one valid for 1/T <= inv_tmin, and a shorter "hot" one valid for 1/T <= inv_tmin2
// (inv_tmin2 <= inv_tmin; 0 = none).  The shortest valid table is used.
template <class TabPtr>
struct TabSel {
    TabPtr base;                // LDS copy of the first n_lds samples of the table array (or the array itself)
    long long full, cool, hot;  // offset | count << 32 of each table (count 0: the level does not exist)
    double inv_tmin, inv_tmin2;
    const double2* gbase = nullptr;  // the whole array in global memory, when `base` holds only part of it
    int n_lds = 0x7fffffff;
};

__device__ __forceinline__ long long tab_slice(int off, int cnt) {
    return (long long)(unsigned int)off | ((long long)cnt << 32);
}

template <int VARIANT, class TabPtr>
__device__ __forceinline__ double band_sum_at(const TabSel<TabPtr>& ts, bool use_ctab, double invT, const ExpTab et) {
    long long sel = ts.full;
    sel = (use_ctab && invT <= ts.inv_tmin) ? ts.cool : sel;
    sel = (use_ctab && invT <= ts.inv_tmin2) ? ts.hot : sel;
    const int off = (int)sel, cnt = (int)(sel >> 32);
    // a wave with a point whose table is not staged (colder than every compressed level, long tables) reads global
    // memory for all its points: same values
    if (__builtin_amdgcn_ballot_w64(off >= ts.n_lds) != 0) {
        const double2* tab = ts.gbase + off;
        return VARIANT == 0 ? band_sum_ref(tab, cnt, invT) : band_sum_fast(tab, cnt, invT, et);
    }
    const TabPtr tab = ts.base + off;
    return VARIANT == 0 ? band_sum_ref(tab, cnt, invT) : band_sum_fast(tab, cnt, invT, et);
}

// ln S_f(e^u) from the filter's interpolant: interval of u, then Horner's rule on 8 coefficients (four 16-byte reads).
// `lds_at` >= 0: the interpolants are staged in LDS at that byte offset of the workgroup's dynamic LDS (the address is
// formed from the LDS symbol itself, so that the reads are ds_read_b128 and not generic-pointer loads); < 0: global.
// r = the temperature's interval coordinate (thermal_state_log): interval (int)r, position 2 frac(r) - 1 in [-1, 1).
// (Measured and not kept: a row of padding behind every filter's rows in the staged copy, against bank conflicts between
// lanes that hold different filters -- no change for photometry without shared epochs, 26.1 against 25.7 us.)
__device__ __forceinline__ double interp_log_band_sum(const DevProblem& pb, int lds_at, int filt, double r) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int j = (int)r;
    const double s = fma(__builtin_amdgcn_fract(r), 2., -1.);
    double2 q0, q1, q2, q3;
    if (lds_at >= 0) {
        const double2* q = reinterpret_cast<const double2*>(smem + lds_at) + (filt * pb.itab_m * 8 + 8 * j) / 2;
        q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    } else {
        const double2* q = reinterpret_cast<const double2*>(pb.itab + filt * pb.itab_m * 8 + 8 * j);
        q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    }
    double g = fma(q0.x, s, q0.y);
    g = fma(g, s, q1.x);
    g = fma(g, s, q1.y);
    g = fma(g, s, q2.x);
    g = fma(g, s, q2.y);
    g = fma(g, s, q3.x);
    return fma(g, s, q3.y);
}

// Everything one lane does for its data point after the thermal state (1/T, R_bb^2): band sum(s) -> template term.
// STAGED: the band tables (or their compressed levels) are in LDS; the on-the-fly reddening fall-back exists only in
// the unstaged instantiations, so that it costs the staged kernels no registers.
// ITAB: the instantiation may meet log-space states (engines with shared epochs only: with the thermal state inside the
// point loop the interpolated path's registers do not fit next to it).
// The factors a companion-shocking model applies per filter (models.py:909-917): shock factor x the filter's Kasen
// parameter, its SiFTO parameter, its time shift -- from the fourth entry of the filter's descriptor and the proposal.
__device__ __forceinline__ void companion_factors(const double2 fd3, const double* __restrict__ c, const double* __restrict__ p,
                                                  double& kfac, double& sfac, double& dt) {
    const long long ks = __double_as_longlong(fd3.x);   // {kpar, spar}, dtpar
    const int kp = (int)ks, sp = (int)(ks >> 32), dp = (int)__double_as_longlong(fd3.y);
    kfac = c[5] * (kp >= 0 ? p[kp] : 1.);
    sfac = sp >= 0 ? p[sp] : 1.;
    dt = dp >= 0 ? p[dp] : 0.;
}
// shock component x factor + template x factor: ONE expression, so that every path contracts it the same way
__device__ __forceinline__ double companion_combine(double shock, double kfac, double tmpl, double sfac) {
    return shock * kfac + tmpl * sfac;
}

// MODEL > 0: compile-time model of an engine whose interpolants are all proved from their first interval
// (DevProblem::itab_uniform) and which fits no sigma -- the kernels specialised for the benchmark shapes.
template <int VARIANT, bool STAGED, bool ITAB, int MODEL = 0>
__device__ __forceinline__ double point_model(const DevProblem& pb, const double* __restrict__ c,
                                     const double* __restrict__ p, double t_in, int filt, const double2* tbase,
                                     const FiltDesc* fdesc, const ExpTab et, double invT, double pref, int itab_at) {
    const int model = MODEL ? MODEL : pb.model;
    double S = 0., yfit;
    const double2* fd = reinterpret_cast<const double2*>(fdesc) + kFdD2 * filt;  // the filter's descriptor: 16-byte reads
    // Log-space state (x = ln T > 0, pref = ln R_bb^2; see thermal_state_log): the interpolant if every point of the
    // wave is inside its filter's proved range -- a per-wave choice, like the fast / safe band sums, so that a walker's
    // result never depends on how walkers are batched -- else the sample tables after one exponential each.
    bool by_table = false;
    if (ITAB && VARIANT != 0 && (MODEL || pb.use_itab)) {
        const bool log_form = __double2hiint(invT) >= 0;   // (-0.0 and -1/T have the sign bit set)
        bool inside = log_form;
        if ((!MODEL && !pb.itab_uniform) || model == kShockCooling4) {
            const long long fmeta = __double_as_longlong(fd[1].y);   // {float r_min, int ioff} of the filter's interpolant
            // (ShockCooling4 also needs the band sum at 0.74 T: ln 0.74 / h intervals lower)
            const double r_low = model == kShockCooling4 ? invT - 0.3011050927839216 * pb.itab_inv_h : invT;
            inside = log_form && r_low >= (double)__int_as_float((int)fmeta);
        }
        if (__builtin_amdgcn_ballot_w64(!inside) == 0) {
            by_table = true;
        } else if (log_form) {  // a mixed wave: back to linear space
            invT = exp(-fma(invT, 1. / pb.itab_inv_h, pb.itab_u0));
            pref = exp(pref);
        } else {
            invT = -invT;
        }
    }
    if (by_table) {
        double L = interp_log_band_sum(pb, itab_at, filt, invT);
        if (model == kShockCooling4)  // min(blackbody, suppressed blackbody at 0.74 T), models.py:629-631
            L = fmin(L, interp_log_band_sum(pb, itab_at, filt, invT - 0.3011050927839216 * pb.itab_inv_h) + 1.2044203711356864);
        // (|ln R_bb^2| <= 600 and |ln S| < 100: the exponent can be added as an integer)
        yfit = exp_scaled<false>(fma(L, kInvLn2N, pref * kInvLn2N), et);
    } else if (!STAGED && invT > 0. && pb.redden_slow) {
        // ShockCooling3 through tables too long for LDS: the walker's reddening is applied sample by sample to the
        // full table in global memory (libm; a fall-back, not a fast path)
        const long long full = __double_as_longlong(fd[0].x);
        const int off = (int)full, cnt = (int)(full >> 32);
        for (int k = 0; k < cnt; ++k) {
            const double2 aw = pb.tab[off + k];
            S += aw.y * exp2(-c[6] * pb.tab_ext[off + k]) / expm1(aw.x * invT);
        }
    } else if (invT > 0.) {
        const double2 d0 = fd[0], d1 = fd[1], d2 = fd[2];
        const TabSel<const double2*> ts{tbase, __double_as_longlong(d0.x), __double_as_longlong(d0.y),
                                        __double_as_longlong(d1.x), d2.x, d2.y, pb.tab,
                                        STAGED ? pb.n_lds_tab : 0x7fffffff};
        S = band_sum_at<VARIANT>(ts, pb.use_ctab != 0, invT, et);
        if (model == kShockCooling4) {  // models.py:629-631: min(blackbody, suppressed blackbody)
            const double S2 = band_sum_at<VARIANT>(ts, pb.use_ctab != 0, invT * (1. / 0.74), et);
            S = fmin(S, S2 * (1. / (0.74 * 0.74 * 0.74 * 0.74)));
        }
    }
    // pref may be NaN (propagates) or 0 with 1/T == 0.
    if (!by_table) yfit = (pref != pref) ? pref : pref * S;
    if (model == kShockCooling3) yfit *= c[5];  // models.py:495
    if (model >= kCompanion && model <= kCompanion3) {  // models.py:909-917, 977-980, 1040-1045
        double kfac, sfac, dt;
        companion_factors(fd[3], c, p, kfac, sfac, dt);
        // (the stretch as a reciprocal, taken once per column by the column paths: (t - t_peak - dt) / s within a rounding)
        const double x = (t_in - c[3] - dt) * (1. / c[4]);
        double tmpl;
        if (pb.knot_h > 0.) {   // integer-day knots: interval computed; coefficients from LDS where they are staged
            extern __shared__ __align__(16) unsigned char smem[];
            const int row0 = filt * (pb.n_knots - 1) * 2;   // double2 entries in front of the filter's
            if (STAGED && pb.n_spl_lds > 0) {
                const int spl_at = (int)(kLdsHead * sizeof(double) + (pb.n_lds_tab + kFdD2 * pb.n_filters) * sizeof(double2) +
                                         pb.n_itab_lds * sizeof(double));
                tmpl = spline_eval_uniform(pb, reinterpret_cast<const double2*>(smem + spl_at) + row0, x);
            } else {
                tmpl = spline_eval_uniform(pb, reinterpret_cast<const double2*>(pb.spl) + row0, x);
            }
        } else {
            tmpl = spline_eval(pb.knots, pb.n_knots, pb.spl + (size_t)filt * (pb.n_knots - 1) * 4, x, pb.knot_inv_h);
        }
        yfit = companion_combine(yfit, kfac, tmpl, sfac);
    }
    return yfit;
}

// Diagnostic build only (-DLCF_STAMPS, tools/debug/make_stamp_build.py; never shipped): s_memtime stamps of the first 64
// workgroups of k_solo, written by the first lane of wave `W` into a buffer nothing else reads.
#ifdef LCF_STAMPS
__device__ unsigned long long g_wall[2 * 1024 * 2];   // [launch parity][workgroup][entry, exit] of the 100 MHz wall clock
__device__ unsigned long long g_stamps[64 * 16];
// ... and, for wave 0, the ticks since its previous stamp summed over all half-steps of all launches (g_acc) with the
// number of times the stamp was passed (g_cnt): the mean life of a workgroup, phase by phase (resident launches: the
// last half-step alone says little -- it writes the state and the snapshot).
__device__ unsigned long long g_acc[64 * 16], g_cnt[64 * 16], g_last[64];
__shared__ unsigned long long s_stamp_last;   // (wave 0's previous stamp; garbage at kernel entry: implausible deltas are dropped)
#define LCF_STAMP(W, k)                                                                                        \
    do {                                                                                                       \
        unsigned long long t_;                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
        if (blockIdx.x < 64 && threadIdx.x == 64 * (W)) {                                                      \
            g_stamps[blockIdx.x * 16 + (k)] = t_;                                                              \
            if ((W) == 0) {                                                                                    \
                const unsigned long long d_ = t_ - s_stamp_last;                                               \
                if (d_ < (1ull << 24)) {   /* (fire-and-forget adds: nothing here waits for memory) */         \
                    atomicAdd(&g_acc[blockIdx.x * 16 + (k)], d_);                                              \
                    atomicAdd(&g_cnt[blockIdx.x * 16 + (k)], 1ull);                                            \
                }                                                                                              \
                s_stamp_last = t_;                                                                             \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
#elif defined(LCF_PROGRESS)
// Diagnostic build (-DLCF_PROGRESS; never shipped): how often each workgroup has passed each stamp -- where the
// workgroups of a launch that cannot finish are.
__device__ unsigned int g_progress[64 * 16];
#define LCF_STAMP(W, k)                                                                                        \
    do {                                                                                                       \
        if (blockIdx.x < 64 && threadIdx.x == 64 * (W)) atomicAdd(&g_progress[blockIdx.x * 16 + (k)], 1u);    \
    } while (0)
#else
#define LCF_STAMP(W, k) do {} while (0)
#endif

// ---- epoch-major likelihood: one lane per COLUMN (an observation time and its points, DevProblem::em_*) ----------------
// The lane computes the column's thermal state in registers and walks the column's points itself: no hand-off of the
// state through LDS, no barrier between states and points, one interval / position of the temperature for all filters
// of the epoch, and as many independent Horner chains per lane as the epoch has filters.
//
// M filters side by side for one column on the interpolated level: filters f0 .. f0 + M - 1 (wave-uniform: the column
// layout is dense), interval j and position s shared by all of them.  Returns y_fit per filter.  `row` = the lane's
// offset 8 j (doubles) inside a filter's interpolant; a filter's interpolant is 8 itab_m doubles long.
// IN_LDS: the interpolants are staged at byte `lds_at` of the dynamic LDS (the address is formed from the LDS symbol
// itself, so that the reads are ds_read_b128 with immediate offsets and not generic-pointer loads); else global memory.
// (A compile-time choice: with both in one function the two sets of loads share registers, and the LDS reads then wait
// for every outstanding global load -- the column's own photometry included.)
// `lnr2k` = ln R_bb^2 x 256 / ln 2: y_fit = 2^((ln S + ln R_bb^2) 256 / ln 2 / 256), the scaling folded into one FMA.
template <int M, bool IN_LDS>
__device__ __forceinline__ void interp_columns(const DevProblem& pb, int lds_at, int f0, int row, double s, double lnr2k,
                                               const ExpTab et, double (&yfit)[M]) {
    extern __shared__ __align__(16) unsigned char smem[];
    double2 q[M][4];
    const int fstride = pb.itab_m * 8;
    if (IN_LDS) {
        const double2* base = reinterpret_cast<const double2*>(smem + lds_at) + row / 2;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const double2* qa = base + ((f0 + m) * fstride) / 2;
            q[m][0] = qa[0], q[m][1] = qa[1], q[m][2] = qa[2], q[m][3] = qa[3];
        }
    } else {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const double2* qa = reinterpret_cast<const double2*>(pb.itab + (size_t)(f0 + m) * fstride) + row / 2;
            q[m][0] = qa[0], q[m][1] = qa[1], q[m][2] = qa[2], q[m][3] = qa[3];
        }
    }
    double g[M];
#pragma unroll
    for (int m = 0; m < M; ++m) g[m] = fma(q[m][0].x, s, q[m][0].y);
#pragma unroll
    for (int m = 0; m < M; ++m) g[m] = fma(g[m], s, q[m][1].x);
#pragma unroll
    for (int m = 0; m < M; ++m) g[m] = fma(g[m], s, q[m][1].y);
#pragma unroll
    for (int m = 0; m < M; ++m) g[m] = fma(g[m], s, q[m][2].x);
#pragma unroll
    for (int m = 0; m < M; ++m) g[m] = fma(g[m], s, q[m][2].y);
#pragma unroll
    for (int m = 0; m < M; ++m) g[m] = fma(g[m], s, q[m][3].x);
#pragma unroll
    for (int m = 0; m < M; ++m) g[m] = fma(g[m], s, q[m][3].y);
    // (|ln R_bb^2| <= 600 and |ln S| < 100: the exponent can be added as an integer)
#pragma unroll
    for (int m = 0; m < M; ++m) yfit[m] = exp_scaled<false>(fma(g[m], kInvLn2N, lnr2k), et);
}

// M filters of one dense column of a COMPANION-SHOCKING model side by side: the Kasen blackbody through the interpolants
// (interp_columns) and the SiFTO template from the staged splines (integer-day knots), combined with the filters' factors;
// returns the M squared scaled residuals' sum.  The arithmetic is point_model's, operation for operation.
template <int M>
__device__ __forceinline__ double companion_columns(const DevProblem& pb, int itab_at, int spl_at, const double2* fd2,
                                                    const double* __restrict__ p, const double* __restrict__ c, int f0,
                                                    int row, double s, double lnr2k, double t_in, double inv_s,
                                                    const ExpTab et, const double2* __restrict__ yd, int nc) {
    extern __shared__ __align__(16) unsigned char smem[];
    double2 o[M];
#pragma unroll
    for (int m = 0; m < M; ++m) o[m] = yd[(size_t)(f0 + m) * nc];
    double shock[M];
    interp_columns<M, true>(pb, itab_at, f0, row, s, lnr2k, et, shock);
    double acc = 0.;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        double kfac, sfac, dt;
        companion_factors(fd2[kFdD2 * (f0 + m) + 3], c, p, kfac, sfac, dt);
        const double x = (t_in - c[3] - dt) * inv_s;
        const double tmpl = spline_eval_uniform(pb, reinterpret_cast<const double2*>(smem + spl_at) + (f0 + m) * (pb.n_knots - 1) * 2, x);
        const double q = (o[m].x - companion_combine(shock[m], kfac, tmpl, sfac)) * o[m].y;
        acc = fma(q, q, acc);
    }
    return acc;
}

// What a lane needs of its FIRST column before any arithmetic: the observation time and (dense columns of up to kPreK
// filters) the photometry.  None of it depends on the proposal, so the one-launch half-step kernels fetch it ahead of
// the serial head: by the time the coefficients exist the loads have long landed.
constexpr int kPreK = 6;
struct ColumnOperands {
    double t;
    double2 o[kPreK];
    bool have_o;   // (wave-uniform)
};

template <int MODEL = 0>
__device__ __forceinline__ bool columns_fast_kind(const DevProblem& pb, bool by_itab) {
    if (MODEL) return true;   // (what a model-specialised kernel is launched for)
    // dense columns of a power-law model without a fitted sigma and with interpolants proved from their first interval:
    // every point of a wave whose states are all in log space goes through interp_columns
    return by_itab && pb.em_dense && pb.itab_uniform && !pb.use_sigma && (pb.model == kShockCooling || pb.model == kShockCooling2);
}

template <int VARIANT, int MODEL = 0>
__device__ __forceinline__ void fetch_column(const DevProblem& pb, int part, int vtid, ColumnOperands& co) {
    const int c0 = part_entry(pb.part_col0, part), c1 = part_entry(pb.part_col0, part + 1);
    const int cb = c0 + (vtid & ~63), lane = vtid & 63;
    co.have_o = false;
    co.t = 0.;
#pragma unroll
    for (int k = 0; k < kPreK; ++k) co.o[k] = make_double2(0., 0.);
    if (cb >= c1) return;
    const int col = min(cb + lane, c1 - 1);
    co.t = pb.em_t[col];
    if (columns_fast_kind<MODEL>(pb, VARIANT != 0 && pb.use_itab) && pb.em_k == kPreK) {
        co.have_o = true;
#pragma unroll
        for (int k = 0; k < kPreK; ++k) co.o[k] = pb.em_yd[(size_t)k * pb.em_cols + col];
    }
}

// The interpolated points of one dense column (the lane's): sum of squared scaled residuals.
// ... of kPreK filters whose photometry is in registers already (fetch_column): two groups of three, all operands at hand
template <bool IN_LDS>
__device__ __forceinline__ double fast_column_fetched(const DevProblem& pb, int itab_at, int row, double s, double lnr2k,
                                                      const ExpTab et, const double2 (&o)[kPreK]) {
    static_assert(kPreK == 6, "two groups of three");
    double acc = 0.;
#pragma unroll
    for (int f = 0; f < kPreK; f += 3) {
        double yf[3];
        interp_columns<3, IN_LDS>(pb, itab_at, f, row, s, lnr2k, et, yf);
        const double q0 = (o[f].x - yf[0]) * o[f].y, q1 = (o[f + 1].x - yf[1]) * o[f + 1].y;
        const double q2 = (o[f + 2].x - yf[2]) * o[f + 2].y;
        acc = fma(q0, q0, acc);
        acc = fma(q1, q1, acc);
        acc = fma(q2, q2, acc);
        LCF_STAMP(0, 14 + (f > 0));
    }
    return acc;
}
// ... of any number of filters, photometry fetched group by group
template <bool IN_LDS>
__device__ __forceinline__ double fast_column(const DevProblem& pb, int itab_at, int col, int row, double s, double lnr2k,
                                              const ExpTab et) {
    const int nf = pb.em_k, nc = pb.em_cols;
    const double2* yd = pb.em_yd + col;
    double acc = 0.;
    int f = 0;
#pragma unroll 1
    for (; f + 3 <= nf; f += 3) {
        const double2 o0 = yd[(size_t)f * nc], o1 = yd[(size_t)(f + 1) * nc], o2 = yd[(size_t)(f + 2) * nc];
        double yf[3];
        interp_columns<3, IN_LDS>(pb, itab_at, f, row, s, lnr2k, et, yf);
        const double q0 = (o0.x - yf[0]) * o0.y, q1 = (o1.x - yf[1]) * o1.y, q2 = (o2.x - yf[2]) * o2.y;
        acc = fma(q0, q0, acc);
        acc = fma(q1, q1, acc);
        acc = fma(q2, q2, acc);
    }
    if (f + 2 <= nf) {
        const double2 o0 = yd[(size_t)f * nc], o1 = yd[(size_t)(f + 1) * nc];
        double yf[2];
        interp_columns<2, IN_LDS>(pb, itab_at, f, row, s, lnr2k, et, yf);
        const double q0 = (o0.x - yf[0]) * o0.y, q1 = (o1.x - yf[1]) * o1.y;
        acc = fma(q0, q0, acc);
        acc = fma(q1, q1, acc);
        f += 2;
    }
    if (f < nf) {
        const double2 o0 = yd[(size_t)f * nc];
        double yf[1];
        interp_columns<1, IN_LDS>(pb, itab_at, f, row, s, lnr2k, et, yf);
        const double q0 = (o0.x - yf[0]) * o0.y;
        acc = fma(q0, q0, acc);
    }
    return acc;
}

// One column point by point: any model, fitted sigma, states outside the interpolants, ragged columns.  `dense`: the
// filter index is the loop counter (wave-uniform, scalar table loads).  The ballots inside point_model see the lanes
// that have a point at this position of their column only.
template <int VARIANT, bool LDS_TAB, int MODEL>
__device__ __forceinline__ double general_column(const DevProblem& pb, const double* __restrict__ p,
                                                  const double* __restrict__ c, const double2* tbase, const FiltDesc* fdesc,
                                                  const ExpTab et, int itab_at, double t_in, int col, bool live, double x,
                                                  double pr) {
    const int nc = pb.em_cols;
    double acc = 0.;
    auto one_point = [&](int k, int filt) {
        const double2 o = pb.em_yd[(size_t)k * nc + col];
        const double yfit = point_model<VARIANT, LDS_TAB, true, MODEL>(pb, c, p, t_in, filt, tbase, fdesc, et, x, pr, itab_at);
        const double r = o.x - yfit;   // models.py:121-135
        if (!MODEL && pb.use_sigma) {
            const double dy = o.y;
            const double su = p[pb.n_dim - 1] * (pb.sigma_abs ? pb.sigma_unit_abs : dy);
            const double var = fma(dy, dy, su * su);
            acc += log(kTwoPi * var) + r * r / var;
        } else {
            const double q = r * o.y;
            acc = fma(q, q, acc);
        }
    };
    if (MODEL || pb.em_dense) {
#pragma unroll 1
        for (int k = 0; k < pb.em_k; ++k)
            if (live) one_point(k, k);
    } else {
#pragma unroll 1
        for (int k = 0; k < pb.em_k; ++k) {
            const int filt = pb.em_filt[(size_t)k * nc + col];
            if (live && filt >= 0) one_point(k, filt);
        }
    }
    return acc;
}

// The same for a MODEL-SPECIALISED kernel, out of line: a wave of such a kernel gets here when one of its states is
// outside the interpolants (a phase before the explosion, a temperature outside 2..256 kK) -- rare in a fit, and inlined the
// sample-table code would sit in the kernel's instruction stream and register budget.  Tables, descriptors and the exp
// table are read from global memory (the numbers their staged copies hold), the coefficients through a plain pointer:
// no LDS symbol in here.
template <int VARIANT, int MODEL>
__device__ __attribute__((noinline)) double cold_column(const DevProblem* pbp, const double* c_mem, const double* p, double t_in,
                                                        int col, bool live, double x, double pr) {
    const DevProblem& pb = *pbp;
    double c[kNCoef];
#pragma unroll
    for (int k = 0; k < kNCoef; ++k) c[k] = c_mem[k];
    return general_column<VARIANT, false, MODEL>(pb, p, c, pb.tab, pb.f_desc, ExpTab{pb.exp2tab}, -1, t_in, col, live, x, pr);
}

// ... computing the state of the column itself (for the caller that found the wave outside the fast path before it had one)
template <int VARIANT, int MODEL>
__device__ __attribute__((noinline)) double cold_column_with_state(const DevProblem* pbp, const double* c_mem, const double* p,
                                                                   double t_in, int col, bool live) {
    const DevProblem& pb = *pbp;
    double c[kNCoef], x, pr;
#pragma unroll
    for (int k = 0; k < kNCoef; ++k) c[k] = c_mem[k];
    const ExpTab et{pb.exp2tab};
    thermal_state_log<MODEL>(pb, c, t_in, x, pr, et);
    return general_column<VARIANT, false, MODEL>(pb, p, c, pb.tab, pb.f_desc, et, -1, t_in, col, live, x, pr);
}

// The whole column phase of a lane of a MODEL-specialised kernel in the shape the benchmarks have -- a power-law model,
// kPreK filters, the lane's one column already fetched (fetch_column), interpolants in LDS: straight-line code, the five
// coefficients the state needs as vector registers (no scalar copies, nothing hoisted).  The arithmetic is epochs_loop's
// fast branch, operation for operation.  Returns false (wave-uniform) where a state of the wave is outside the
// interpolants: the caller then takes the general path for the column.
template <int MODEL>
__device__ __forceinline__ bool lean_column(const DevProblem& pb, const double* __restrict__ sc, const ColumnOperands& first,
                                            const ExpTab et, int itab_at, double& acc) {
    const double t = first.t - sc[0];
    double u, lp, lL;
    powerlaw_log_state(pb, sc[3], sc[6], sc[7], t, et, u, lp, lL);
    const double r = (u - pb.itab_u0) * pb.itab_inv_h;
    // exactly where thermal_state_log leaves a log-space pair (lL NaN makes lp NaN: every comparison fails)
    const bool ok = t > 0. && r >= 0. && r < (double)pb.itab_m && lp >= -600. && lp <= 600.;
    if (__builtin_amdgcn_ballot_w64(!ok) != 0) return false;
    const int row = 8 * (int)r;
    const double s = fma(__builtin_amdgcn_fract(r), 2., -1.);
    acc = fast_column_fetched<true>(pb, itab_at, row, s, lp * kInvLn2N, et, first.o);
    return true;
}

// chi^2 share of virtual thread `vtid` (0 .. kBlock-1) of part `part` for one walker: parameters p, coefficients c.
// Virtual wave v = vtid / 64 owns the columns part_col0 + 64 v + lane + 256 m, m = 0, 1, ...: whichever kernel walks a
// part (k_points, k_fused, k_solo, k_pop) and whichever physical wave plays virtual wave v, the lane sums, the wave
// sums (xor butterfly) and the part's sum (w0 + w1) + (w2 + w3) are the same numbers -- chains do not depend on the kernel.
// The band-sum path (interpolant or sample tables; fast or safe exponentials) is chosen per wave, for the filter the wave
// is at: deterministic in the walker and the light curve alone.
// `first` (when FETCHED): the operands of the lane's first column, fetched by the caller (fetch_column, this part and vtid).
// COLD (model-specialised kernels that hold the problem in memory, `pb_mem`, and the coefficients at `c_mem`): the
// general path is the out-of-line cold_column.
template <int VARIANT, bool LDS_TAB, bool FETCHED = false, int MODEL = 0, bool COLD = false>
__device__ inline double epochs_loop(const DevProblem& pb, int part, const double* __restrict__ p,
                                     const double* __restrict__ c, const double2* tbase, const FiltDesc* fdesc,
                                     const ExpTab et, int vtid, int itab_at, const ColumnOperands& first = ColumnOperands{},
                                     bool use_first = true, const DevProblem* pb_mem = nullptr,
                                     const double* c_mem = nullptr) {
    // (`part` and the virtual wave are wave-uniform: scalar registers)
    part = __builtin_amdgcn_readfirstlane(part);
    const int c0 = part_entry(pb.part_col0, part), c1 = part_entry(pb.part_col0, part + 1);
    const int lane = vtid & 63, v64 = __builtin_amdgcn_readfirstlane(vtid & ~63);
    const bool by_itab = MODEL ? true : VARIANT != 0 && pb.use_itab;
    const bool fast_kind = columns_fast_kind<MODEL>(pb, by_itab);
    // companion-shocking models with dense columns, interpolants AND template splines staged (integer-day knots)
    const bool companion_fast = !MODEL && LDS_TAB && by_itab && pb.em_dense && pb.itab_uniform && !pb.use_sigma &&
                                pb.model >= kCompanion && pb.model <= kCompanion3 && itab_at >= 0 && pb.n_spl_lds > 0 &&
                                pb.knot_h > 0.;
    // (generic kernels only: compiled into the model-specialised ones as well it cost them their registers -- 2 to 34
    // spilled -- and photometry without shared epochs ran no faster there, 25.7 against 26.0 us)
    const bool ragged_fast = !MODEL && by_itab && !pb.em_dense && pb.itab_uniform && !pb.use_sigma &&
                             (pb.model == kShockCooling || pb.model == kShockCooling2);
    double term = 0.;
    // (a lane's columns are 256 apart: the time of the next one is requested while this one is computed -- in the generic
    // kernels; the model-specialised ones have no registers to spare for it)
    constexpr bool kAhead = MODEL == 0;
    double t_next = 0.;
    if (kAhead && c0 + v64 < c1 && !(FETCHED && use_first)) t_next = pb.em_t[min(c0 + v64 + lane, c1 - 1)];
#pragma unroll 1
    for (int cb = c0 + v64; cb < c1; cb += kBlock) {
        const bool live = cb + lane < c1;
        const int col = live ? cb + lane : c1 - 1;   // lanes beyond the part repeat its last column; their terms are dropped
        const bool fetched = FETCHED && use_first && cb < c0 + kBlock;
        const double t_in = fetched ? first.t : kAhead ? t_next : pb.em_t[col];
        if (kAhead && cb + kBlock < c1) t_next = pb.em_t[min(cb + kBlock + lane, c1 - 1)];
        // ragged columns: the first point's filter and photometry too, ahead of the state's arithmetic
        int filt0 = 0;
        double2 o0 = make_double2(0., 0.);
        if (ragged_fast) {
            filt0 = pb.em_filt[col];
            o0 = pb.em_yd[col];
        }
        double x, pr;   // the log-space pair of thermal_state_log, or (1/T, R_bb^2)
        if (by_itab) {
            thermal_state_log<MODEL>(pb, c, t_in, x, pr, et);
        } else {
            double T;
            thermal_state<MODEL>(pb, c, t_in, T, x, pr);
        }
        LCF_STAMP(0, 13);
        if (ragged_fast && __builtin_amdgcn_ballot_w64(__double2hiint(x) < 0) == 0) {
            // columns that are NOT one point of every filter (photometry whose observations have their own times: one
            // point per column): the same interpolated arithmetic with the filter read per lane
            const double lnr2k = pr * kInvLn2N;
            const int nc = pb.em_cols;
            double acc = 0.;
#pragma unroll 1
            for (int k = 0; k < pb.em_k; ++k) {
                int filt = filt0;
                double2 o = o0;
                if (k > 0) {   // (not a select between a register and a load: that puts the register in scratch memory)
                    filt = pb.em_filt[(size_t)k * nc + col];
                    o = pb.em_yd[(size_t)k * nc + col];
                }
                const double L = interp_log_band_sum(pb, itab_at, max(filt, 0), x);
                const double q = (o.x - exp_scaled<false>(fma(L, kInvLn2N, lnr2k), et)) * o.y;
                acc = filt >= 0 ? fma(q, q, acc) : acc;
            }
            term += live ? acc : 0.;
            continue;
        }
        if (companion_fast && __builtin_amdgcn_ballot_w64(__double2hiint(x) < 0) == 0) {
            // dense columns of a companion-shocking model, everything staged: groups of three filters side by side
            const int row = 8 * (int)x;
            const double s = fma(__builtin_amdgcn_fract(x), 2., -1.);
            const double lnr2k = pr * kInvLn2N, inv_s = 1. / c[4];
            const int spl_at = itab_at + pb.n_itab_lds * (int)sizeof(double), nf = pb.em_k, ncols = pb.em_cols;
            const double2* fd2 = reinterpret_cast<const double2*>(fdesc);
            const double2* yd = pb.em_yd + col;
            double acc = 0.;
            int f = 0;
#pragma unroll 1
            for (; f + 3 <= nf; f += 3)
                acc += companion_columns<3>(pb, itab_at, spl_at, fd2, p, c, f, row, s, lnr2k, t_in, inv_s, et, yd, ncols);
            if (f + 2 <= nf) {
                acc += companion_columns<2>(pb, itab_at, spl_at, fd2, p, c, f, row, s, lnr2k, t_in, inv_s, et, yd, ncols);
                f += 2;
            }
            if (f < nf) acc += companion_columns<1>(pb, itab_at, spl_at, fd2, p, c, f, row, s, lnr2k, t_in, inv_s, et, yd, ncols);
            term += live ? acc : 0.;
            continue;
        }
        if (fast_kind && __builtin_amdgcn_ballot_w64(__double2hiint(x) < 0) == 0) {
            const int row = 8 * (int)x;
            const double s = fma(__builtin_amdgcn_fract(x), 2., -1.);
            const double lnr2k = pr * kInvLn2N;
            double acc;
            if (fetched && first.have_o)
                acc = itab_at >= 0 ? fast_column_fetched<true>(pb, itab_at, row, s, lnr2k, et, first.o)
                                   : fast_column_fetched<false>(pb, itab_at, row, s, lnr2k, et, first.o);
            else
                acc = itab_at >= 0 ? fast_column<true>(pb, itab_at, col, row, s, lnr2k, et)
                                   : fast_column<false>(pb, itab_at, col, row, s, lnr2k, et);
            term += live ? acc : 0.;   // (a dead lane's sum is dropped as a whole)
            continue;
        }
        if constexpr (COLD && MODEL != 0)
            term += cold_column<VARIANT, MODEL>(pb_mem, c_mem, p, t_in, col, live, x, pr);
        else
            term += general_column<VARIANT, LDS_TAB, MODEL>(pb, p, c, tbase, fdesc, et, itab_at, t_in, col, live, x, pr);
    }
    return term;
}

// Thermal state (T, R_bb^2) of every (walker, distinct observation time): light curves observed in several filters at
// the same epochs share it, so the logarithm and exponentials of the model are paid once per epoch instead of once
// per point.  Used when the light curve has at least two points per distinct time on average.
__global__ __launch_bounds__(kBlock) void k_thermal(const DevProblem pb, int w_lo, int n_w,
                                                    const double* __restrict__ coef,
                                                    const double* __restrict__ lprior, int skip_excluded,
                                                    double2* __restrict__ therm, int log_state) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_w * pb.n_epochs) return;
    const int w = w_lo + idx / pb.n_epochs, ep = idx % pb.n_epochs;
    if (skip_excluded && lprior[w] == -INFINITY) return;
    double T, invT, pref;
    if (log_state)
        thermal_state_log(pb, coef + (size_t)w * kNCoef, pb.epoch_t[ep], invT, pref, ExpTab{pb.exp2tab});
    else
        thermal_state(pb, coef + (size_t)w * kNCoef, pb.epoch_t[ep], T, invT, pref);
    therm[(size_t)w * pb.n_epochs + ep] = make_double2(invT, pref);
}

// Stage the exp table and the staged region (band tables, descriptors, interpolants) in LDS; thread t of nt.  One flat
// copy of the engine's image of that layout, eight 16-byte loads per thread in flight before the first LDS store: a
// loop of load -> store pairs is one memory round trip per iteration, and six of them in a row made staging as long
// as the whole serial head (configs[1]: staged at 5.4 k cycles after kernel entry, now at 4.4 k; half-step 7.82 ->
// 7.51 us).  ShockCooling3: the walker's reddening goes into the staged weights (table by table).
template <int VARIANT, bool LDS_TAB>
__device__ inline void stage_tables(const DevProblem& pb, double* __restrict__ exptab, double2* __restrict__ ltab,
                                    double ebv, int t, int nt) {
    if (LDS_TAB && pb.model == kShockCooling3 && !pb.redden_slow) {
        for (int k = t; k < kExpTabSize; k += nt) exptab[k] = pb.exp2tab[k];
        double2* lfd = ltab + pb.n_lds_tab;   // filter descriptors (48 B = 3 double2 each) behind the tables
        const double2* gfd = reinterpret_cast<const double2*>(pb.f_desc);
        for (int k = t; k < kFdD2 * pb.n_filters; k += nt) lfd[k] = gfd[k];
        for (int k = t; k < pb.n_lds_tab; k += nt) {
            double2 aw = pb.tab[k];
            aw.y *= exp2(-ebv * pb.tab_ext[k]);
            ltab[k] = aw;
        }
        return;
    }
    const int n16 = LDS_TAB ? pb.stage_n16 : kLdsHead / 2;
    const double2* __restrict__ g = pb.stage_image;
    double2* __restrict__ l = reinterpret_cast<double2*>(exptab);
    const int last = n16 - 1;
#pragma unroll 1
    for (int k0 = t; k0 < n16; k0 += 8 * nt) {   // (named values, not an array: that the compiler kept in scratch memory)
        const int k1 = k0 + nt, k2 = k1 + nt, k3 = k2 + nt, k4 = k3 + nt, k5 = k4 + nt, k6 = k5 + nt, k7 = k6 + nt;
        const double2 v0 = g[min(k0, last)], v1 = g[min(k1, last)], v2 = g[min(k2, last)], v3 = g[min(k3, last)];
        const double2 v4 = g[min(k4, last)], v5 = g[min(k5, last)], v6 = g[min(k6, last)], v7 = g[min(k7, last)];
        l[k0] = v0;
        if (k1 < n16) l[k1] = v1;
        if (k2 < n16) l[k2] = v2;
        if (k3 < n16) l[k3] = v3;
        if (k4 < n16) l[k4] = v4;
        if (k5 < n16) l[k5] = v5;
        if (k6 < n16) l[k6] = v6;
        if (k7 < n16) l[k7] = v7;
    }
}

// The points of part `part` for one walker in the filter-sorted order: parameters p, coefficients c (global or LDS),
// thermal states th[pt_epoch - e_off] when THERM.  MODE 0: returns this thread's share of chi^2 (engines without
// thermal states ahead of the points only: with them the likelihood walks the epoch-major copy, epochs_loop);
// MODE 1: y_fit -> out0[row][orig];  MODE 2: T, R_bb -> out0, out1.
template <int VARIANT, int MODE, bool LDS_TAB, bool THERM, int KPRE_POINTWISE = 0>
__device__ inline double points_loop(const DevProblem& pb, int part, size_t row, const double* __restrict__ p,
                                     const double* __restrict__ c, const double2* __restrict__ th_base, int e_off,
                                     const double2* tbase, const FiltDesc* fdesc, const ExpTab et,
                                     double* __restrict__ out0, double* __restrict__ out1, int tid = threadIdx.x,
                                     int itab_at = -1) {
    static_assert(!(THERM && MODE == 0), "the likelihood of an engine with thermal states is epochs_loop's");
    double term = 0.;  // `tid`: this thread's index among the kBlock threads that walk the part
    const int p0 = part_entry(pb.part_start, part), p1 = part_entry(pb.part_start, part + 1);  // this part's points
    // chunks whose operands are fetched together, before any band sum starts (fewer when the thermal state is computed
    // per point: that code needs the registers)
    constexpr int kPre = THERM ? (KPRE_POINTWISE == 1 ? LCF_KPRE_SOLO : LCF_KPRE) : 1;
    for (int k0 = 0; k0 * kBlock < p1 - p0; k0 += kPre) {
        int idx[kPre], filt[kPre];
        double tin[kPre], yv[kPre], idy[kPre];
        double2 th[kPre];
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int i = p0 + (k0 + u) * kBlock + tid;
            idx[u] = i < p1 ? i : -1;
            if (idx[u] < 0) continue;
            // (filter and epoch share a word, and the time is fetched only by who uses it -- per-point thermal states,
            // the SN Ia template, the (T, R) output: 20 instead of 32 bytes per point from L2)
            if (THERM && MODE != 2) {
                int ep;
                if (pb.pt_fe) {
                    const unsigned int fe = (unsigned int)pb.pt_fe[i];
                    filt[u] = (int)(fe & 63u);
                    ep = (int)(fe >> 6);
                } else {
                    filt[u] = pb.pt_filt[i];
                    ep = pb.pt_epoch[i];
                }
                th[u] = th_base[ep - e_off];
                tin[u] = pb.model >= kCompanion && pb.model <= kCompanion3 ? pb.t[i] : 0.;
            } else {
                filt[u] = pb.pt_filt[i];
                tin[u] = pb.t[i];
            }
            if (MODE == 0) {
                const double2 yd = pb.pt_yd[i];   // (y, 1/dy), or (y, dy) when sigma is fitted
                yv[u] = yd.x;
                idy[u] = yd.y;
            }
        }
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int i = idx[u];
            if (i < 0) continue;
            double invT, pref, Tk = 0.;
            if (THERM && MODE != 2) {
                invT = th[u].x;   // (the log-space pair of thermal_state_log when the engine interpolates)
                pref = th[u].y;
            } else {
                thermal_state(pb, c, tin[u], Tk, invT, pref);
            }
            if (MODE == 2) {
                // R_bb = sqrt(pref) keeps the reference's NaN/0 pattern (pref = R_bb^2)
                const size_t j = row * pb.n_points + pb.pt_orig[i];
                out0[j] = Tk;
                out1[j] = sqrt(pref);
                continue;
            }
            const double yfit = point_model<VARIANT, LDS_TAB, THERM>(pb, c, p, tin[u], filt[u], tbase, fdesc, et, invT, pref,
                                                                     itab_at);
            if (MODE == 0) {  // models.py:121-135
                const double r = yv[u] - yfit;
                if (pb.use_sigma) {
                    const double dy = idy[u];
                    const double su = p[pb.n_dim - 1] * (pb.sigma_abs ? pb.sigma_unit_abs : dy);
                    const double var = fma(dy, dy, su * su);
                    term += log(kTwoPi * var) + r * r / var;
                } else {
                    const double q = r * idy[u];
                    term = fma(q, q, term);
                }
            } else {
                out0[row * pb.n_points + pb.pt_orig[i]] = yfit;
            }
        }
    }
    return term;
}

// Workgroup reduction of the chi^2 shares in a fixed order; thread 0 stores the part's partial sum.
__device__ inline double store_part_sum(double term, double* __restrict__ red, double* __restrict__ dst) {
    const int tid = threadIdx.x;
    const double ws = wave_sum(term);
    if ((tid & 63) == 0) red[tid >> 6] = ws;
    __syncthreads();
    if (tid != 0) return 0.;
    const double sum = (red[0] + red[1]) + (red[2] + red[3]);
    *dst = sum;
    return sum;  // (thread 0)
}

// MODE 0: chi^2 partial sums -> part[w][n_parts];  MODE 1: y_fit -> out0[w][orig];  MODE 2: T, R_bb -> out0, out1
// Workgroup = (walker w, part j): it walks the points of part j (a range of epochs, sorted by filter) in chunks of
// 256 with the exp table and ALL band tables staged in LDS once, and reduces once at the end.
template <int VARIANT, int MODE, bool LDS_TAB, bool THERM>
__device__ inline void points_body(const DevProblem& pb, int bid, int w_lo, int n_w, const double* __restrict__ P,
                                   const double* __restrict__ coef, const double* __restrict__ lprior,
                                   const double2* __restrict__ therm, double* __restrict__ out0,
                                   double* __restrict__ out1) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* exptab = reinterpret_cast<double*>(smem);                     // kExpTabSize doubles
    double* red = exptab + kExpTabSize;                                   // 4 doubles
    double2* ltab = reinterpret_cast<double2*>(smem + kLdsHead * sizeof(double));

    const int part = bid / n_w;
    const int w = w_lo + bid % n_w;

    if (MODE == 0 && lprior[w] == -INFINITY) return;  // prior excludes the walker: likelihood skipped (fitting.py:125)

    const double* c = coef + (size_t)w * kNCoef;   // wave-uniform -> scalar loads
    stage_tables<VARIANT, LDS_TAB>(pb, exptab, ltab, pb.model == kShockCooling3 ? c[6] : 0., threadIdx.x, kBlock);
    __syncthreads();

    const double2* tbase = LDS_TAB ? (const double2*)ltab : pb.tab;
    const FiltDesc* fdesc = LDS_TAB ? reinterpret_cast<const FiltDesc*>(ltab + pb.n_lds_tab) : pb.f_desc;
    // byte offset of the staged interpolants in the dynamic LDS (behind the tables and descriptors), or -1
    const int itab_at = (LDS_TAB && pb.n_itab_lds > 0)
                            ? (int)(kLdsHead * sizeof(double) + (pb.n_lds_tab + kFdD2 * pb.n_filters) * sizeof(double2)) : -1;
    if constexpr (MODE == 0 && THERM) {
        // (the thermal states are the lanes' own: k_thermal is not launched for the likelihood)
        const double term = epochs_loop<VARIANT, LDS_TAB>(pb, part, P + (size_t)w * pb.n_dim, c, tbase, fdesc, ExpTab{exptab},
                                                          threadIdx.x, itab_at);
        store_part_sum(term, red, out0 + (size_t)w * part_stride(pb) + part);
    } else {
        const double term = points_loop<VARIANT, MODE, LDS_TAB, THERM>(
            pb, part, (size_t)(w - w_lo), P + (size_t)w * pb.n_dim, c, THERM ? therm + (size_t)w * pb.n_epochs : nullptr, 0,
            tbase, fdesc, ExpTab{exptab}, out0, out1, threadIdx.x, itab_at);
        if (MODE == 0) store_part_sum(term, red, out0 + (size_t)w * part_stride(pb) + part);
    }
}

template <int VARIANT, int MODE, bool LDS_TAB, bool THERM>
__global__ __launch_bounds__(kBlock, LCF_WAVES) void k_points(const DevProblem pb, int w_lo, int n_w, const double* __restrict__ P,
                                                   const double* __restrict__ coef,
                                                   const double* __restrict__ lprior,
                                                   const double2* __restrict__ therm, double* __restrict__ out0,
                                                   double* __restrict__ out1) {
    points_body<VARIANT, MODE, LDS_TAB, THERM>(pb, blockIdx.x, w_lo, n_w, P, coef, lprior, therm, out0, out1);
}

// lnL (and log-posterior) per walker from the partial sums, fixed summation order.  (A separate launch on purpose:
// letting the last workgroup of a walker do this inside k_points needs an agent-scope fence per workgroup, which on
// this part writes back and invalidates the XCD's L2 -- measured 3x slower k_points.)
__global__ void k_finalize(const DevProblem pb, int n, const double* __restrict__ part,
                           const double* __restrict__ lprior, double* __restrict__ out) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    const double lp = lprior[w];
    if (lp == -INFINITY) {
        out[w] = -INFINITY;
        return;
    }
    double s = pb.use_sigma ? 0. : pb.log_norm_const;
    for (int k = 0; k < pb.n_parts; ++k) s += part[(size_t)w * part_stride(pb) + k];
    out[w] = lp - 0.5 * s;
}

// blackbody_to_filters, pointwise (models.py:1161-1162): arbitrary (filter, T, R) triples.
template <int VARIANT>
__global__ __launch_bounds__(kBlock) void k_bb_pointwise(const DevProblem pb, int m, const int* __restrict__ filt,
                                                         const int* __restrict__ tab_off,
                                                         const int* __restrict__ ctab_off,
                                                         const double* __restrict__ ctmin,
                                                         const double* __restrict__ T, const double* __restrict__ R,
                                                         double* __restrict__ out) {
    __shared__ double exptab[kExpTabSize];
    if (threadIdx.x < kExpTabSize) exptab[threadIdx.x] = pb.exp2tab[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int f = filt[i];
    const double Tk = T[i], r = R[i];
    double S = 0.;
    if (Tk > 0. && Tk < kTmax) {
        const ExpTab et{exptab};
        // the interpolated level (the engine's default where it has the interpolants): ln S_f(ln T) inside the range the
        // filter's interpolant is proved for -- a per-triple choice, a triple's result depends on nothing else
        bool done = false;
        if (VARIANT != 0 && pb.use_itab) {
            const double x = (log(Tk) - pb.itab_u0) * pb.itab_inv_h;
            if (x >= (double)pb.f_desc[f].r_min && x < (double)pb.itab_m) {
                S = exp_scaled<false>(interp_log_band_sum(pb, -1, f, x) * kInvLn2N, et);
                done = true;
            }
        }
        if (!done) {
            const TabSel<const double2*> ts{pb.tab, tab_slice(tab_off[2 * f], tab_off[2 * f + 1]),
                                            tab_slice(ctab_off[2 * f], ctab_off[2 * f + 1]), 0, 1. / ctmin[f], 0.};
            S = band_sum_at<VARIANT>(ts, pb.use_ctab != 0, 1. / Tk, et);
        }
    }
    out[i] = r * r * S;
}

// ---- ensemble sampler ---------------------------------------------------------------------------------------------
// Half-steps are numbered globally (g = 0, 1, 2, ...).  Everything a proposal needs later for its accept/reject is
// kept per proposal slot in buffers double-buffered by the parity of g, so that ONE kernel (k_next) can, fully in
// parallel and without any inter-workgroup hand-off, (i) commit the previous half-step and (ii) draw the next
// proposals: a thread that needs the position of a walker whose previous move is not committed yet simply evaluates
// that walker's accept test itself (a pure function of immutable per-slot data).
// Per proposal slot: what its accept test needs besides the new log-posterior (one 32-byte load).
struct SlotRec {
    double zl;      // (n_dim - 1) ln z
    double lnu;     // ln u of the accept test
    double lp_old;  // log-posterior of the walker when the proposal was drawn
    double lpri;    // log-prior of the proposal
};
// Per (step, half, slot): the state-independent part of the stretch move, drawn for the whole run in advance.
struct DrawRec {
    int wid, pid;      // active walker and its partner from the complementary colour
    int wprev, pprev;  // their proposal slots in the previous half-step of the run (-1: not active there)
    double z;          // stretch factor
    double zl;         // (n_dim - 1) ln z
    double lnu;        // ln u
    int wage, page;    // half-steps since the walker / the partner last moved (1..3; 0 without slot bookkeeping)
};

struct DevSampler {
    int n_walkers, n_half, n_dim, store_chain;
    uint32_t key0, key1;
    int inline_finalize;  // 1: accept tests sum the chi^2 partials themselves; 0: they read the gathered newlp
    int n_peers;          // > 0: the rows travel through peer mailboxes (below) instead of part2 + a collective
    double a;
    double* X;          // [n_walkers][n_dim]  committed positions
    double* LP;         // [n_walkers]         committed log-posteriors
    double* Q[2];       // [n_half][n_dim]     proposals
    SlotRec* rec[2];    // [n_half]
    double* newlp[2];   // [n_half]            log-posterior of the proposal (finalize kernel / all-gather)
    double* part2[2];   // [n_half][n_parts + 1] per slot: chi^2 partial sums, then the log-prior; half-step parity 0 / 1
    double* chain;      // [n_steps][n_walkers][n_dim]
    double* chain_lp;   // [n_steps][n_walkers]
    long long* nacc;    // [n_walkers]
    int* err;
    // Peer mailboxes (multi-GPU without a collective): every rank owns a mailbox [4 generations][n_half][row] of
    // 16-byte entries; a rank that has evaluated a proposal writes the entry of each of the row's numbers straight into
    // EVERY rank's mailbox (peer memory mapped through IPC; over xGMI on a node), and whoever needs the row polls its
    // own copy.  mbox = this rank's, peer_mbox[r] = rank r's as mapped here (own included).
    unsigned long long* mbox;
    unsigned long long* peer_mbox[kMaxPeers];
    // Row boards (multi-GPU, one workgroup per proposal: lcf_sampler_run_rows).  Every rank owns a board
    // [kRing versions][n_walkers][n_dim + 2] of 16-byte entries in uncached memory -- a walker's position, its
    // log-posterior and its acceptance count after each of its moves, tagged with the half-step -- followed by one
    // progress word per rank, an abort word and four words that say what an aborted launch was waiting for.  The rank
    // that moves a walker posts the row on EVERY rank's board; nobody else computes anything about that walker.
    unsigned long long* board;
    unsigned long long* peer_board[kMaxPeers];
    int n_board_ranks, board_rank;
    int ring, pad_ring;   // versions a board keeps (a power of two): kRing between ranks, kRunRing for one-launch runs
    // One-launch runs write the snapshot the host reads after a run -- [error word | X | LP | n_accepted] in pinned host
    // memory -- themselves, with the state, in their last step; a workgroup that meets a NaN or gives up a wait says so in
    // a word of its own behind it (snap_flags[blockIdx.x & (kSnapFlags - 1)] = 1 / snap_flags[kSnapFlags + ...] = 2;
    // plain stores, cleared with the state only: errors stay until set_state).  Null: the snapshot kernel does it.
    unsigned long long* snap_out;
    unsigned int* snap_flags;
    // ... and write the state into a second set of buffers (the host then swaps the two sets): a launch that gives up
    // leaves the state it started from untouched, and the host runs the same steps again, a launch per half-step.
    double* X_out;
    double* LP_out;
    long long* nacc_out;
    // Bound of every wait for another rank (mailbox entries, board rows, progress words), in ticks of the 100 MHz wall
    // clock: peer_wait_ticks().  A rank's stream holds only a few ms of launches, so a host that stalls longer than
    // this on ONE rank ends the run on ALL of them -- the default is therefore seconds, not the 0.5 s of round 2.
    unsigned long long wait_ticks;
    // ... and of the wait of a resident launch (k_solo_run) for the REST OF ITSELF: LCF_RESIDENT_WAIT_S, default 0.05 s.
    unsigned long long resident_ticks;
};

// LCF_RESIDENT_WAIT_S (seconds, default 0.05): how long a workgroup of a resident launch waits for a row before it asks
// whether the launch's other workgroups have started at all (board_take)
unsigned long long resident_wait_ticks() {
    double sec = 0.05;
    if (const char* env = std::getenv("LCF_RESIDENT_WAIT_S")) sec = std::atof(env);
    if (!(sec > 0.)) sec = 0.05;
    return (unsigned long long)(std::min(sec, 600.) * 1e8);
}

// LCF_PEER_WAIT_S (seconds, default 5; the tests of the bounded waits set 0.5)
unsigned long long peer_wait_ticks() {
    double sec = 5.;
    if (const char* env = std::getenv("LCF_PEER_WAIT_S")) sec = std::atof(env);
    if (!(sec > 0.)) sec = 5.;
    return (unsigned long long)(std::min(sec, 600.) * 1e8);
}

// One float64 as two 8-byte granules {32 data bits, 32-bit generation tag}: an 8-byte store is the largest that
// arrives whole, so a reader that sees the tag of the generation it waits for in both granules has the value -- no
// flag, no fence, no ordering between stores needed (the "LL" protocol of the collective libraries).  Mailbox
// memory is uncached (fine-grained); stores and polls are system-scope so that they bypass the XCD's L2.
__device__ inline void mbox_post(const DevSampler& sm, long long g, int slot, int col, int stride, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), tag = (unsigned long long)(uint32_t)g << 32;
    const size_t at = 2 * ((((size_t)(g & 3) * sm.n_half) + slot) * stride + col);
#pragma unroll
    for (int r = 0; r < kMaxPeers; ++r)
        if (r < sm.n_peers) {
            unsigned long long* p = sm.peer_mbox[r] + at;
            __hip_atomic_store(p, (b & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(p + 1, (b >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
}

// The value of generation g, once it has arrived; every lane's wait is bounded (DevSampler::wait_ticks of the 100 MHz wall clock): a
// peer that never delivers ends the run with an error instead of hanging the device.
__device__ inline double mbox_take(const DevSampler& sm, long long g, int slot, int col, int stride) {
    const unsigned long long tag = (unsigned long long)(uint32_t)g;
    const unsigned long long* p = sm.mbox + 2 * ((((size_t)(g & 3) * sm.n_half) + slot) * stride + col);
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((a >> 32) == tag && (b >> 32) == tag)
            return __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
        if (wall_clock64() - t0 > sm.wait_ticks) {
            atomicOr(sm.err, 2);
            return qnan();
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// Random red/blue colouring of each step (emcee's randomize_split): one workgroup per step ranks the walkers by a
// 50-bit Philox key (ties impossible: the walker id fills the low 14 bits) with a bitonic sort in LDS.
// perm[step][0 .. n/2) is colour 0.  Deterministic in (seed, step): every rank of a multi-GPU run derives the same
// split without communicating.
// `slot_of` (or null): the slot table of k_slots, written here as well -- the step's two rows, and by the block's first
// workgroup the row in front of the block (`front`, or all -1) -- so that the records need no launch in between.
__device__ __forceinline__ void make_perm_body(int n_walkers, int n_pad, uint32_t key0, uint32_t key1,
                                               long long first_step, int* __restrict__ perm, int n_half,
                                               int* __restrict__ slot_of, const int* __restrict__ front) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    const long long step = first_step + blockIdx.x;
    for (int w = threadIdx.x; w < n_pad; w += blockDim.x) {
        unsigned long long k = ~0ull;
        if (w < n_walkers) {
            uint32_t r[4];
            philox4x32((uint32_t)w, (uint32_t)step, 2u, 7u, key0, key1, r);
            const unsigned long long h = ((unsigned long long)r[0] << 32) | r[1];
            k = (h & ~0x3fffull) | (unsigned long long)w;
        }
        keys[w] = k;
    }
    __syncthreads();
    // Pairs [64 m, 64 m + 64) -- one wave's share of a stage (blockDim.x is a multiple of 64) -- touch keys
    // [128 m, 128 m + 128) only while the stride is at most 64: such stages follow each other without a workgroup
    // barrier (a wave's LDS operations execute in order); 6 of the 55 stages of 1024 keys need one on either side.
    for (int size = 2; size <= n_pad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < (n_pad >> 1); i += blockDim.x) {
                const int lo = 2 * i - (i & (stride - 1));  // index with bit `stride` clear
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            const int next = stride > 1 ? stride >> 1 : size;   // the stride of the stage that follows
            if (stride > 64 || next > 64)
                __syncthreads();
            else
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < n_walkers; w += blockDim.x) {
        const int wid = (int)(keys[w] & 0x3fffull);
        perm[(size_t)blockIdx.x * n_walkers + w] = wid;
        if (slot_of) {   // (position w of the permutation: colour 0 = the first n_half entries -- as k_slots)
            const int half = w < n_half ? 0 : 1, slot = w < n_half ? w : w - n_half;
            slot_of[((size_t)blockIdx.x * 2 + 1 + half) * n_walkers + wid] = slot;
            slot_of[((size_t)blockIdx.x * 2 + 1 + (1 - half)) * n_walkers + wid] = -1;
            if (blockIdx.x == 0) slot_of[w] = front ? front[w] : -1;
        }
    }
}

__global__ __launch_bounds__(1024) void k_make_perm(int n_walkers, int n_pad, uint32_t key0, uint32_t key1,
                                                    long long first_step, int* __restrict__ perm, int n_half,
                                                    int* __restrict__ slot_of, const int* __restrict__ front) {
    make_perm_body(n_walkers, n_pad, key0, key1, first_step, perm, n_half, slot_of, front);
}

// Population mode: the same for MANY samplers in one launch (blockIdx.y = sampler; equal walker counts and blocks).  What
// differs from sampler to sampler -- the key of its RNG, its buffers -- comes from an array in device memory.
struct GenItem {
    uint32_t key0, key1;
    int* perm[2];
    int* slot[2];
    DrawRec* draws[2];
};
// `front_row` >= 0: the half-step in front of the block is row `front_row` of the OTHER buffer's slot table (-1: none)
__global__ __launch_bounds__(1024) void k_make_perm_multi(const GenItem* __restrict__ items, int n_walkers, int n_pad,
                                                          long long first_step, int buf, int n_half, long long front_row) {
    const GenItem it = items[blockIdx.y];
    make_perm_body(n_walkers, n_pad, it.key0, it.key1, first_step, it.perm[buf], n_half, it.slot[buf],
                   front_row >= 0 ? it.slot[buf ^ 1] + (size_t)front_row * n_walkers : nullptr);
}

// Slot of every walker in each half-step of a block of steps (-1 where it is not active).  Rows of `slot_of`
// ([1 + 2 n_steps][n_walkers]): row 0 = the half-step in front of the block (copied from the previous block, or all
// -1 at the start of a run), row 1 + 2 k + half = half-step (k, half) of the block.
__global__ void k_slots(int n_walkers, int n_half, const int* __restrict__ perm, long long n_steps,
                        int* __restrict__ slot_of, const int* __restrict__ front) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n_walkers) slot_of[idx] = front ? front[idx] : -1;   // row 0 (nobody in this launch reads it)
    if (idx >= n_steps * n_walkers) return;
    const long long row = idx / n_walkers;
    const int pos = (int)(idx % n_walkers);  // position in the permutation: colour 0 = first n_half entries
    const int wid = perm ? perm[idx] : pos;
    const int half = pos < n_half ? 0 : 1, slot = pos < n_half ? pos : pos - n_half;
    slot_of[((size_t)row * 2 + 1 + half) * n_walkers + wid] = slot;
    slot_of[((size_t)row * 2 + 1 + (1 - half)) * n_walkers + wid] = -1;
}

// The state-independent half of every stretch move of a block of steps, one thread per (step, half, slot).
// n_half = ceil(n_walkers / 2) slots per half-step: colour 0 (the first n_half entries of the step's permutation)
// moves in half 0 against the n_walkers - n_half walkers of colour 1, then colour 1 against colour 0 -- the larger
// colour first, as emcee's red-blue split does for an odd ensemble; the slot an odd ensemble leaves empty in half 1
// gets wid = -1.  `slot_of` null: no slot bookkeeping (the one-workgroup-per-proposal half-step does not need it).
__device__ __forceinline__ void draws_body(const DevSampler& sm, const int* __restrict__ perm, const int* __restrict__ slot_of,
                                           long long first_step, long long n_steps, DrawRec* __restrict__ draws) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_steps * 2 * sm.n_half) return;
    const int i = (int)(idx % sm.n_half), half = (int)((idx / sm.n_half) & 1);
    const long long row = idx / (2 * sm.n_half);
    const int* pr = perm ? perm + (size_t)row * sm.n_walkers : nullptr;
    const int n_act = half == 0 ? sm.n_half : sm.n_walkers - sm.n_half, n_other = sm.n_walkers - n_act;
    DrawRec d;
    if (i >= n_act) {
        d.wid = d.pid = d.wprev = d.pprev = -1;
        d.z = 1.;
        d.zl = d.lnu = 0.;
        d.wage = d.page = 0;
        draws[idx] = d;
        return;
    }
    const int my_slot = half == 0 ? i : sm.n_half + i;  // colour 0 = first n_half entries of the permutation
    const int wid = pr ? pr[my_slot] : my_slot;
    uint32_t r[4], s2[4];
    philox4x32((uint32_t)wid, (uint32_t)(first_step + row), (uint32_t)half, 0u, sm.key0, sm.key1, r);
    philox4x32((uint32_t)wid, (uint32_t)(first_step + row), (uint32_t)half, 1u, sm.key0, sm.key1, s2);
    const double zr = (sm.a - 1.) * u01(r[0], r[1]) + 1.;
    const double z = zr * zr / sm.a;
    int j = (int)(u01(r[2], r[3]) * (double)n_other);
    j = min(j, n_other - 1);
    const int other_slot = half == 0 ? sm.n_half + j : j;
    d.wid = wid;
    d.pid = pr ? pr[other_slot] : other_slot;
    const int* before = slot_of ? slot_of + (size_t)(row * 2 + half) * sm.n_walkers : nullptr;  // the half-step in front
    d.wprev = before ? before[d.wid] : -1;
    d.pprev = before ? before[d.pid] : -1;
    d.z = z;
    d.zl = (double)(sm.n_dim - 1) * log(z);
    d.lnu = log(u01(s2[0], s2[1]));
    // Every walker moves once per step, in one of its two half-steps: a walker that was not active in the half-step in
    // front (half-step G - 1) moved in the one before it, or -- the walker of a step's SECOND half-step only -- three
    // half-steps ago (first half of the previous step).  The sharded one-workgroup-per-proposal run waits for exactly
    // that version of each row.  (Before the first step of a run every age points in front of the run: its start state.)
    d.wage = d.page = 0;
    if (slot_of) {
        if (half == 0) {
            d.wage = before[d.wid] >= 0 ? 1 : 2;
            d.page = before[d.pid] >= 0 ? 1 : 2;
        } else {
            const int* two_back = slot_of + (size_t)(row * 2) * sm.n_walkers;  // second half of the previous step
            d.wage = two_back[d.wid] >= 0 ? 2 : 3;
            d.page = 1;
        }
    }
    draws[idx] = d;
}

__global__ void k_draws(DevSampler sm, const int* __restrict__ perm, const int* __restrict__ slot_of,
                        long long first_step, long long n_steps, DrawRec* __restrict__ draws) {
    draws_body(sm, perm, slot_of, first_step, n_steps, draws);
}
// (population mode, blockIdx.y = sampler: `sm` = the samplers' common shape, the key from the item)
__global__ void k_draws_multi(const GenItem* __restrict__ items, DevSampler sm, int buf, long long first_step,
                              long long n_steps) {
    const GenItem it = items[blockIdx.y];
    sm.key0 = it.key0;
    sm.key1 = it.key1;
    draws_body(sm, it.perm[buf], it.slot[buf], first_step, n_steps, it.draws[buf]);
}

// The serial part of a half-step for slot i, executed by ONE wave (lane = 0..63): accept tests of the previous
// half-step (recomputed from the per-slot records, no hand-off between workgroups), the new proposal, and -- for a
// slot this rank evaluates (`mine`) -- its coefficients and log-prior, left in sc[0..kNCoef] (LDS, lane 0 writes).
// `primary`: this wave is the one that commits slot i of the previous half-step and publishes the proposal record
// (exactly one wave per slot and launch is primary; the others only need the proposal for themselves).
template <int ND>
__device__ inline void step_serial(const DevProblem& pb, const DevSampler& sm, int i, bool primary, int lane,
                                   int have_prev, long long prev_row, int have_next,
                                   const DrawRec* __restrict__ draws, const DrawRec* __restrict__ prev_draws,
                                   long long g, bool mine, double* __restrict__ sc, double* __restrict__ sq,
                                   double* __restrict__ coef, double* __restrict__ lprior) {
    const int prev_wid = (have_prev && primary) ? prev_draws[i].wid : 0;
    PriorDev my_prior{0, 0, 0., 0., 0., 1.};
    if (lane < pb.n_dim && pb.has_priors && have_next) my_prior = pb.priors[lane];
    const int pp = (int)((g - 1) & 1), cp = (int)(g & 1);
    constexpr int kD = ND > 0 ? ND : kMaxDim;     // array extent
    const int nd = ND > 0 ? ND : sm.n_dim;         // trip count (constant when ND > 0)
    DrawRec dr{0, 0, -1, -1, 1., 0., 0., 0, 0};
    if (have_next) dr = draws[i];
    const bool active = have_next && dr.wid >= 0;  // (an odd ensemble leaves the last slot of its second half-step empty)
    // --- roles: which accept test (if any) this lane evaluates ---
    int rw = -1, rslot = -1;
    if (lane == 0 && have_prev && primary && prev_wid >= 0) {
        rslot = i;
        rw = prev_wid;  // walker of slot i in the previous half-step (from its draw record)
    } else if ((lane == 1 || lane == 2) && active) {
        rw = lane == 1 ? dr.wid : dr.pid;
        rslot = have_prev ? (lane == 1 ? dr.wprev : dr.pprev) : -1;
    }
    double row[kD], qrow[kD], lp_cur = 0., nlp = 0.;
    bool ok = false;
#pragma unroll
    for (int d = 0; d < kD; ++d) row[d] = qrow[d] = 0.;
    if (rw >= 0) {
        // everything the accept test and both outcomes need, in one wave of loads
        const double* xs = sm.X + (size_t)rw * nd;
#pragma unroll
        for (int d = 0; d < kD; ++d)
            if (d < nd) row[d] = xs[d];
        lp_cur = sm.LP[rw];
        if (rslot >= 0) {
            const SlotRec rc = sm.rec[pp][rslot];
            const double* qs = sm.Q[pp] + (size_t)rslot * nd;
#pragma unroll
            for (int d = 0; d < kD; ++d)
                if (d < nd) qrow[d] = qs[d];
            if (sm.n_peers > 0) {
                // the slot's row from this rank's mailbox, as its owner posted it: the log-prior first (always posted;
                // the partial sums only where the prior allows the proposal)
                const int stride = part_stride(pb);
                const double lpri = mbox_take(sm, g - 1, rslot, pb.n_parts, stride);
                double sum = pb.use_sigma ? 0. : pb.log_norm_const;
                if (lpri != -INFINITY)
                    for (int k = 0; k < pb.n_parts; ++k) sum += mbox_take(sm, g - 1, rslot, k, stride);
                nlp = lpri == -INFINITY ? -INFINITY : lpri - 0.5 * sum;
            } else if (sm.inline_finalize) {
                // the slot's row: partial chi^2 sums, then the log-prior (of this rank's evaluation, or gathered)
                const double* prow = sm.part2[pp] + (size_t)rslot * part_stride(pb);
                double sum = pb.use_sigma ? 0. : pb.log_norm_const;
                for (int k = 0; k < pb.n_parts; ++k) sum += prow[k];
                const double lpri = prow[pb.n_parts];
                nlp = lpri == -INFINITY ? -INFINITY : lpri - 0.5 * sum;
            } else {
                nlp = sm.newlp[pp][rslot];
            }
            ok = (rc.zl + nlp - rc.lp_old) > rc.lnu;  // emcee: (ndim-1) ln z + lp_new - lp_old > ln u
            lp_cur = ok ? nlp : rc.lp_old;
            if (ok) {
#pragma unroll
                for (int d = 0; d < kD; ++d) row[d] = qrow[d];
            }
        }
    }
    if (lane == 0 && rslot >= 0) {  // commit of the previous half-step's slot i
        if (nlp != nlp) atomicOr(sm.err, 1);
        if (ok) {
#pragma unroll
            for (int d = 0; d < kD; ++d)
                if (d < nd) sm.X[(size_t)rw * nd + d] = row[d];
            sm.LP[rw] = nlp;
            atomicAdd((unsigned long long*)&sm.nacc[rw], 1ull);
        }
        if (sm.store_chain) {
            double* crow = sm.chain + ((size_t)prev_row * sm.n_walkers + rw) * nd;
#pragma unroll
            for (int d = 0; d < kD; ++d)
                if (d < nd) crow[d] = row[d];
            sm.chain_lp[(size_t)prev_row * sm.n_walkers + rw] = lp_cur;
        }
    }
    if (have_next && !active) {  // empty slot: nothing to propose; its workgroups skip the likelihood
        if (lane == 0) {
            if (sc) sc[kNCoef] = -INFINITY;
            if (primary && lprior) lprior[i] = -INFINITY;
        }
        return;
    }
    if (active) {
        double q[kMaxDim], lq[kMaxDim];
        double arg = 1.;
#pragma unroll
        for (int d = 0; d < kMaxDim; ++d) q[d] = lq[d] = 0.;
#pragma unroll
        for (int d = 0; d < kD; ++d) {
            const double xi = __shfl(row[d], 1, 64), cj = __shfl(row[d], 2, 64);
            q[d] = d < nd ? cj - (cj - xi) * dr.z : 0.;
            if (lane == d && d < pb.n_par) arg = q[d];
        }
        const double lp_i = __shfl(lp_cur, 1, 64);
        if (!mine) {
            // another rank evaluates this proposal: only what later accept tests need is published here
            // (its log-prior reaches this rank inside the gathered log-posterior)
            if (lane == 0 && primary) {
#pragma unroll
                for (int d = 0; d < kD; ++d)
                    if (d < nd) sm.Q[cp][(size_t)i * nd + d] = q[d];
                sm.rec[cp][i] = SlotRec{dr.zl, dr.lnu, lp_i, 0.};
            }
            return;
        }
        const double lg = flog(arg);  // one logarithm per lane, all at once
#pragma unroll
        for (int d = 0; d < kD; ++d) lq[d] = __shfl(lg, d, 64);
        double c[kNCoef];
        walker_coefficients(pb, q, lq, c, pb.use_itab != 0 && pb.variant != 0);
        // log-prior: lane d evaluates parameter d with its descriptor fetched at kernel entry, then an ordered sum
        double lpr = 0.;
        if (pb.has_priors) {
            double qv = 0.;
#pragma unroll
            for (int d = 0; d < kD; ++d)
                if (lane == d) qv = q[d];
            const double mine = lane < pb.n_dim ? prior_term(my_prior, qv) : 0.;  // one evaluation per lane
#pragma unroll
            for (int d = 0; d < kD; ++d)
                if (d < nd) lpr += __shfl(mine, d, 64);
        }
        if (lane == 0) {
            for (int k = 0; k < kNCoef; ++k) sc[k] = c[k];
            sc[kNCoef] = lpr;
            if (sq) {  // the proposal itself, for a workgroup that goes on to evaluate it
#pragma unroll
                for (int d = 0; d < kD; ++d)
                    if (d < nd) sq[d] = q[d];
            }
            if (primary) {  // publish the per-slot records
#pragma unroll
                for (int d = 0; d < kD; ++d)
                    if (d < nd) sm.Q[cp][(size_t)i * nd + d] = q[d];
                sm.rec[cp][i] = SlotRec{dr.zl, dr.lnu, lp_i, lpr};
                sm.part2[cp][(size_t)i * part_stride(pb) + pb.n_parts] = lpr;  // last column of the slot's row
                if (sm.n_peers > 0) mbox_post(sm, g, i, pb.n_parts, part_stride(pb), lpr);
                if (coef)
                    for (int k = 0; k < kNCoef; ++k) coef[(size_t)i * kNCoef + k] = c[k];
                if (lprior) lprior[i] = lpr;
            }
        }
    }
}

// One 64-thread workgroup per proposal slot i: it commits half-step g - 1 for slot i and draws slot i of half-step g
// COOPERATIVELY -- the latency chain of a single thread (three accept tests with dependent loads, then the logarithms of
// the proposal) is spread over lanes that run the same code on different data:
//   lanes 0..2: accept test of {previous slot i, own walker, partner walker}, with every load that does not depend
//   on the outcome issued up front (both candidate rows included);  lanes 0..n_par-1: ln of the proposal's parameters.
// Every lane ends with the same proposal (bitwise); lane 0 publishes it, with the coefficients and the log-prior of the
// slots in [lo, hi) (this rank's shard) for the likelihood launch behind it (k_points: thermal states are its own).
// ND > 0: the walker dimension is a compile-time constant (loops over parameters unroll with exact trip counts and
// the per-parameter arrays live in registers); ND == 0: any dimension up to kMaxDim.
template <int ND>
__device__ inline void step_body(const DevProblem& pb, const DevSampler& sm, int i, int have_prev, long long prev_row,
                                 int have_next, const DrawRec* __restrict__ draws, const DrawRec* __restrict__ prev_draws,
                                 long long g, int lo, int hi, double* __restrict__ coef, double* __restrict__ lprior) {
    __shared__ double sc[kNCoef + 1];
    const bool mine = i >= lo && i < hi;       // this rank evaluates the likelihood of slot i
    if (threadIdx.x < 64)
        step_serial<ND>(pb, sm, i, true, threadIdx.x, have_prev, prev_row, have_next, draws, prev_draws, g, mine, sc, nullptr,
                        coef, lprior);
}

template <int ND>
__global__ __launch_bounds__(64) void k_step(const DevProblem pb, const DevSampler sm, int have_prev, long long prev_row,
                                             int have_next, const DrawRec* __restrict__ draws,
                                             const DrawRec* __restrict__ prev_draws, long long g, int lo, int hi,
                                             double* __restrict__ coef, double* __restrict__ lprior) {
    step_body<ND>(pb, sm, blockIdx.x, have_prev, prev_row, have_next, draws, prev_draws, g, lo, hi, coef, lprior);
}

// ---- single-GPU fit: the whole half-step in ONE launch ---------------------------------------------------------------
// Workgroup = (proposal slot i, part j).  Wave 0 runs the serial part (every part's workgroup redundantly: a few
// hundred cycles, and no workgroup then waits for another; part 0 is the one that commits and publishes) while waves
// 1-3 stage the tables; then all threads compute the thermal states of the part's own epochs straight into LDS and walk
// the part's points.  Against k_step + k_points this saves a launch, the round trip of the thermal states and the
// coefficients through memory, and the second staging.  The partial sums are double-buffered by half-step parity: this
// launch reads the previous half-step's for its accept tests and writes its own.  Same arithmetic in the same order
// as the two-kernel path: chains are bitwise identical.
constexpr int kFusedScratch = kNCoef + 2 + kMaxDim + (kMaxDim & 1);  // doubles: coefficients + log-prior, proposal

// A value every lane holds identically (read from LDS), moved to scalar registers: the per-walker coefficients are
// used by every band sum, and as vector registers they would cost the kernel its occupancy.
__device__ inline double uniform_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffLL));
    const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int ND, int VARIANT, bool THERM>
__global__ __launch_bounds__(kBlock, LCF_WAVES) void k_fused(const DevProblem pb, const DevSampler sm, int have_prev,
                                                  long long prev_row, const DrawRec* __restrict__ draws,
                                                  const DrawRec* __restrict__ prev_draws, long long g, int lo, int hi,
                                                  double* __restrict__ lprior) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* exptab = reinterpret_cast<double*>(smem);
    double* red = exptab + kExpTabSize;
    double2* ltab = reinterpret_cast<double2*>(smem + kLdsHead * sizeof(double));
    const FiltDesc* fdesc = reinterpret_cast<const FiltDesc*>(ltab + pb.n_lds_tab);
    const int itab_at = pb.n_itab_lds > 0
                            ? (int)(kLdsHead * sizeof(double) + (pb.n_lds_tab + kFdD2 * pb.n_filters) * sizeof(double2)) : -1;
    double* sc = reinterpret_cast<double*>(ltab + pb.stage_d2);
    double* sq = sc + kNCoef + 2;
    const int tid = threadIdx.x;
    // Multi-GPU: this rank evaluates the slots [lo, hi) only.  Every other slot (commit + light proposal, no
    // likelihood) takes one WAVE of the workgroups launched behind the shard's own.
    const int n_own = hi - lo;
    if ((int)blockIdx.x >= n_own * pb.n_parts) {
        const int k = ((int)blockIdx.x - n_own * pb.n_parts) * (kBlock / 64) + (tid >> 6);
        const int slot = k < lo ? k : k + n_own;
        if (slot < sm.n_half)
            step_serial<ND>(pb, sm, slot, true, tid & 63, have_prev, prev_row, 1, draws, prev_draws, g, false, nullptr,
                            nullptr, nullptr, nullptr);
        return;
    }
    const int part = blockIdx.x / n_own, i = lo + blockIdx.x % n_own;
    const bool reddened = pb.model == kShockCooling3;
    if (tid < 64)
        step_serial<ND>(pb, sm, i, part == 0, tid, have_prev, prev_row, 1, draws, prev_draws, g, true, sc, sq, nullptr,
                        lprior);
    else if (!reddened)
        stage_tables<VARIANT, true>(pb, exptab, ltab, 0., tid - 64, kBlock - 64);
    __syncthreads();
    if (sc[kNCoef] == -INFINITY) return;  // prior excludes the proposal: likelihood skipped (fitting.py:125)
    double cs[kNCoef];
#pragma unroll
    for (int k = 0; k < kNCoef; ++k) cs[k] = uniform_f64(sc[k]);
    if (reddened) stage_tables<VARIANT, true>(pb, exptab, ltab, cs[6], tid, kBlock);
    if (reddened) __syncthreads();
    double term;
    if constexpr (THERM)
        term = epochs_loop<VARIANT, true>(pb, part, sq, cs, ltab, fdesc, ExpTab{exptab}, tid, itab_at);
    else
        term = points_loop<VARIANT, 0, true, false>(pb, part, 0, sq, cs, nullptr, 0, ltab, fdesc, ExpTab{exptab}, nullptr,
                                                    nullptr, tid, itab_at);
    const double psum = store_part_sum(term, red, sm.part2[g & 1] + (size_t)i * part_stride(pb) + part);
    if (sm.n_peers > 0 && tid == 0) mbox_post(sm, g, i, part, part_stride(pb), psum);  // straight into every rank's mailbox
}

// ---- row boards: tagged words in device memory ---------------------------------------------------------------------
// One float64 = ONE 16-byte entry of two 8-byte granules {32 data bits, 32-bit tag}, tag = half-step after which the row
// holds + 1 (the LL protocol of the mailboxes above, the entry posted with one 16-byte store and polled with one 16-byte
// load: each granule carries its own tag, so an entry that arrives in two halves is still never mistaken for complete).
// A row = the walker's n_dim + 2 numbers in consecutive entries, padded to whole 128-byte lines (board_row_entries):
// the lanes of ONE store instruction post a row, into one line (n_dim <= 6) of each board it goes to.
// Between ranks (k_solo<BOARD> per half-step, k_solo_run<..., RANKS> per block of half-steps): a ring of kRing versions.
// Safe because (a) a reader asks for exactly the version the draw record names (DrawRec::wage / page) and waits,
// bounded, until both granules carry its tag; (b) progress words bound how far ranks drift apart: with a launch per
// half-step no rank starts half-step G before every rank has finished G - 2; with a launch per block of up to kRunSpan
// half-steps no rank starts a launch before every rank has STARTED the launch before the previous one (a rank's
// progress word = the first half-step of the launch it has reached, posted by that launch itself: stream order proves
// that everything in front of it, its row collection included, is complete).  Everything a rank still reads is then at
// most 3 kRunSpan + 2 versions behind anything another rank writes: kRing = 256 versions are never overrun.
constexpr int kRing = 256;
constexpr int kSnapFlags = 1024;
// One-launch runs of ONE GPU (k_solo_run, k_pop_run): a launch covers at most kRunSpanSolo half-steps, reads versions >= G - 3 and
// writes G + 1: kRunRing versions are never overrun however far the workgroups of a launch drift apart.
// (Between ranks a launch covers at most kRunSpan half-steps -- the ring of the inter-rank boards bounds it, below; on one
// GPU up to kRunSpanSolo: a launch's start-up -- staging, first fetch, every workgroup arriving, ~18 us at configs[1] --
// is then paid once per 256 half-steps instead of once per 64: 5.41 -> 5.31 us per half-step over a 1000-step run.)
#ifndef LCF_RUN_SPAN_SOLO
#define LCF_RUN_SPAN_SOLO 256
#endif
constexpr int kRunSpanSolo = LCF_RUN_SPAN_SOLO;
constexpr int kRunRing = 2 * kRunSpanSolo;
constexpr int kRunSpan = 64;
static_assert((kRunRing & (kRunRing - 1)) == 0 && kRunRing >= kRunSpanSolo + 8, "the ring of a one-launch run covers a launch");
constexpr int kRunStoreChain = 1, kRunFlip = 2;   // k_solo_run's run_flags: the run stores its chain; its start state is in X_out / LP_out / nacc_out
// 32-bit words behind the rows: progress per rank, abort, 4 x diagnosis, arrivals (workgroups of resident launches that
// have started, counted up from launch to launch: a launch knows the count that says "all of mine are there")
constexpr int kBoardTail = kMaxPeers + 1 + 4 + 1 + 4;
constexpr int kBoardClear = 10;                 // the last words of the tail that set_state clears: abort ... arrivals, and
                                                // the four words of the entry a given-up wait last saw (diagnosis)
static_assert(kRing >= 3 * kRunSpan + 8, "the ring of the inter-rank boards must cover three launches");

__host__ __device__ inline int board_row_entries(int n_dim) { return (n_dim + 2 + 7) & ~7; }   // 16-byte entries per row
__host__ __device__ inline size_t board_rows_bytes(int ring, int n_walkers, int n_dim) {
    return (size_t)ring * n_walkers * board_row_entries(n_dim) * 16;
}
__device__ inline unsigned long long* board_entry(unsigned long long* board, const DevSampler& sm, unsigned int tag, int wid,
                                                  int col) {
    return board + 2 * ((((size_t)(tag & (unsigned int)(sm.ring - 1)) * sm.n_walkers) + wid) * board_row_entries(sm.n_dim) + col);
}
__device__ inline unsigned int* board_progress(unsigned long long* board, const DevSampler& sm) {
    return reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(board) +
                                           board_rows_bytes(sm.ring, sm.n_walkers, sm.n_dim));
}
// 16-byte accesses past every cache (sc0 sc1 = system scope: they meet in memory, whichever XCD or GPU the other side
// is on).  The store is not waited for; the load is (a poll has nothing else to do).  Written as instructions because
// the language has no 16-byte atomic: the hardware moves an aligned 16-byte lane access as one piece, and the tags make
// even 8-byte pieces safe.
typedef unsigned int lcf_u32x4 __attribute__((ext_vector_type(4)));
// AGENT: the board of a one-launch run lives in this GPU's ordinary memory and is shared by its own workgroups only.
// Its 16-byte accesses are nevertheless issued at system scope (sc0 sc1), like those of the inter-rank boards: with sc1
// alone (device scope) the same run takes TWICE as long (0.70 against 0.37 ms per launch of 64 half-steps at configs[1]
// -- polls served from a cache for a while before they see memory); only the 4-byte words of the tail keep the scope.
template <bool AGENT>
__device__ __forceinline__ void store16_past_caches(unsigned long long* p, unsigned long long lo, unsigned long long hi) {
    const lcf_u32x4 v = {(unsigned int)lo, (unsigned int)(lo >> 32), (unsigned int)hi, (unsigned int)(hi >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
template <bool AGENT>
__device__ __forceinline__ void load16_past_caches(const unsigned long long* p, unsigned long long& lo, unsigned long long& hi) {
    lcf_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    lo = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
    hi = (unsigned long long)v.z | ((unsigned long long)v.w << 32);
}
template <bool AGENT = false>
__device__ inline void board_post(unsigned long long* board, const DevSampler& sm, unsigned int tag, int wid, int col, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), t = (unsigned long long)tag << 32;
    store16_past_caches<AGENT>(board_entry(board, sm, tag, wid, col), (b & 0xffffffffull) | t, (b >> 32) | t);
}
template <bool AGENT = false>
__device__ inline bool board_aborted(const DevSampler& sm) {
    const unsigned int* flag = board_progress(sm.board, sm) + kMaxPeers;
    return (AGENT ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                  : __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0u;
}
__device__ inline unsigned int* board_arrivals(const DevSampler& sm) { return board_progress(sm.board, sm) + kMaxPeers + 5; }
// (what: 1 = a row, a = tag, b = walker, c = column; 2 = the progress words, a = half-step, b = rank that is behind;
// 3 = a row again, but the launch's workgroups were not all resident after DevSampler::resident_ticks: c = how many were)
__device__ inline void board_abort(const DevSampler& sm, unsigned int what, unsigned int a, unsigned int b, unsigned int c) {
    unsigned int* flag = board_progress(sm.board, sm) + kMaxPeers;
    if (atomicCAS_system(flag, 0u, 1u) == 0u) {   // the first one to give up says what it was waiting for
        __hip_atomic_store(flag + 1, what, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(flag + 2, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(flag + 3, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(flag + 4, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    atomicOr(sm.err, 2);
    if (sm.snap_flags) sm.snap_flags[kSnapFlags + (blockIdx.x & (kSnapFlags - 1))] = 2u;
}
// The number with tag `tag` from this rank's board, once it is there (bounded wait: DevSampler::wait_ticks, then the launch is aborted
// and the run ends with an error; NaN after an abort).
// `arrive_goal` != 0 (resident launches): the value of the arrivals word from which all workgroups of this launch have
// started.  Rows of a launch whose workgroups are all resident arrive within microseconds; a wait that has lasted
// DevSampler::resident_ticks (50 ms) looks at the word ONCE: all there -> keep waiting (a stalled peer rank, a busy
// device); not all there -> the launch cannot make progress (another resident kernel holds the CUs): give up now.
// (Measured and dropped, round 4: FOUR loads of the entry kept in flight a quarter of a round trip apart, each looked at
// when it returns -- finer sampling than one look per round trip, but whoever finds its row must still wait for the other
// three loads before their registers may be used again: 5.71 against 5.31 us per half-step at configs[1].  And the upper
// bound of a head computed AHEAD of the partner's commit -- a second wave computes it from stale rows while the first one
// polls, wrong chain, timing only: 5.13 against 5.31 us; with the two candidate heads, the second post and the second
// poll a real version needs, nothing would be left of the 0.19.)
template <bool AGENT = false>
__device__ inline double board_take(const DevSampler& sm, unsigned int tag, int wid, int col, unsigned int arrive_goal = 0u) {
    const unsigned long long* p = board_entry(sm.board, sm, tag, wid, col);
    const unsigned long long t0 = wall_clock64();
    bool resident = arrive_goal == 0u;
    for (int spin = 0;; ++spin) {
        unsigned long long a, b;
        load16_past_caches<AGENT>(p, a, b);
        if ((unsigned int)(a >> 32) == tag && (unsigned int)(b >> 32) == tag)
            return __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
        if ((spin & 15) == 15) {
            if (board_aborted<AGENT>(sm)) return qnan();
            if (!resident && wall_clock64() - t0 > sm.resident_ticks) {
                const unsigned int there = __hip_atomic_load(board_arrivals(sm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(there - arrive_goal) < 0) {
                    board_abort(sm, 3u, tag, (unsigned int)wid, there);
                    return qnan();
                }
                resident = true;
            }
            if (wall_clock64() - t0 > sm.wait_ticks) {
                const bool was_first = !board_aborted<AGENT>(sm);
                board_abort(sm, 1u, tag, (unsigned int)wid, (unsigned int)col);
                if (was_first) {   // (what the entry holds instead: an older tag = never posted)
                    unsigned int* seen = board_arrivals(sm) + 1;
                    seen[0] = (unsigned int)a;
                    seen[1] = (unsigned int)(a >> 32);
                    seen[2] = (unsigned int)b;
                    seen[3] = (unsigned int)(b >> 32);
                }
                return qnan();
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// Version of a row a half-step G asks for: the walker's last move was `age` half-steps ago; rows nobody has moved in
// this run carry the run's start tag.
__device__ inline unsigned int board_tag(long long G, int age, long long g_run0) {
    const long long t = G - age + 1;
    return (unsigned int)(t < g_run0 ? g_run0 : t);
}

// Start of a run: this rank's complete state as version `tag` (= the run's first half-step: "the state in front of it")
// on its OWN board; the abort word is cleared.  (Progress words are absolute half-step numbers and are never reset: a
// faster rank may already have posted into this board.)
__global__ void k_board_init(const DevSampler sm, unsigned int tag) {
    const int cols = sm.n_dim + 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < kBoardClear) __hip_atomic_store(board_progress(sm.board, sm) + kMaxPeers + idx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (idx >= (long long)sm.n_walkers * cols) return;
    const int wid = (int)(idx / cols), col = (int)(idx % cols);
    const double v = col < sm.n_dim ? sm.X[(size_t)wid * sm.n_dim + col] : col == sm.n_dim ? sm.LP[wid] : (double)sm.nacc[wid];
    board_post(sm.board, sm, tag, wid, col, v);
}

// The rows of half-step G, as every rank posted them, into this rank's state (end of a run: `n_hs` = 2, the last step)
// and / or into the chain (`to_chain`, every half-step of a run that stores it).  A wave takes as many slots as whole rows
// fit its 64 lanes (8 rows of 8 entries for up to six parameters), lane = (row, column): behind a resident launch of an
// 8-GPU run there are 64 x 4096 rows to collect, and a wave per row left 57 of 64 lanes idle on a kernel that is nothing
// but round trips to uncached memory (33 us per 131 k rows; 0.25 ns per row).
__global__ void k_board_collect(const DevSampler sm, long long G, long long row, const DrawRec* __restrict__ draws, int n_hs,
                                int to_state, int to_chain) {
    const int nd = sm.n_dim, re = board_row_entries(nd), rpw = 64 / re;   // rows per wave (kMaxDim + 2 <= 24 entries: >= 2)
    const int lane = threadIdx.x & 63, sub = lane / re, col = lane - sub * re;
    const long long wave = (long long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const long long item = wave * rpw + sub;
    if (sub >= rpw || item >= (long long)n_hs * sm.n_half) return;
    const int h = (int)(item / sm.n_half);
    const DrawRec dr = draws[item];
    if (dr.wid < 0 || col > nd + 1) return;
    if (board_aborted(sm)) return;
    const double v = board_take(sm, (unsigned int)(G + h + 1), dr.wid, col);
    if (to_state) {
        if (col < nd) sm.X[(size_t)dr.wid * nd + col] = v;
        else if (col == nd) sm.LP[dr.wid] = v;
        else sm.nacc[dr.wid] = (long long)v;
    }
    if (to_chain && sm.store_chain) {
        if (col < nd) sm.chain[((size_t)(row + h / 2) * sm.n_walkers + dr.wid) * nd + col] = v;
        else if (col == nd) sm.chain_lp[(size_t)(row + h / 2) * sm.n_walkers + dr.wid] = v;
    }
}

// The serial head of one proposal, executed by ONE wave (lane = 0..63): rows of the walker and of its partner ->
// proposal -> logarithms (one parameter per lane) -> coefficients, log-prior; lane 0 leaves them in LDS: sc[0 .. kNCoef)
// the coefficients, sc[kNCoef] the log-prior, sq the proposal, sx the walker's position and sx[kMaxDim] its
// log-posterior.
// BOARD: the rows come from this rank's row board instead of X / LP -- lanes 0 .. nd-1 poll the partner's position,
// lanes 16 .. 16+nd+1 the walker's position, log-posterior and acceptance count (left in sx[kMaxDim + 1]), each for the
// version the draw record names -- and are handed round by shuffles; everything after that is the same arithmetic.
// What the head reads from memory, requested first (head_fetch) and used later (proposal_head): the kernels put their
// other early loads between the two, so that nothing of theirs queues in front of these.
template <int ND>
struct HeadRows {
    static constexpr int kD = ND > 0 ? ND : kMaxDim;
    double x[kD], cj[kD];   // the walker's position, the partner's
    double lp_i, got;       // the walker's log-posterior; BOARD: this lane's word of the rows
    PriorDev prior;         // lane d: the prior of parameter d
};

template <int ND, int BOARD = 0>
__device__ __forceinline__ void head_fetch(const DevProblem& pb, const DevSampler& sm, const DrawRec& dr, int lane,
                                           HeadRows<ND>& h, long long G = 0, long long g_run0 = 0, unsigned int arrive_goal = 0u) {
    constexpr int kD = ND > 0 ? ND : kMaxDim;
    const int nd = ND > 0 ? ND : sm.n_dim;
    const double* xs = sm.X + (size_t)dr.wid * nd;
    const double* cs_ = sm.X + (size_t)dr.pid * nd;
    h.got = 0.;
    if (BOARD) {
        const bool own = lane >= 16;
        const int col = own ? lane - 16 : lane;
        if (own ? col <= nd + 1 : col < nd) {
#ifdef LCF_EXPERIMENT_NOPOLL   // (timing experiment, wrong chain: the state in memory instead of the rows the half-step needs:
            // one load past the caches, as a poll that finds its row at once, and no waiting for anybody)
            {
                const int w_ = own ? dr.wid : dr.pid;
                const double* src_ = col < nd ? sm.X + (size_t)w_ * nd + col : col == nd ? sm.LP + w_ : sm.LP + w_;
                h.got = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(src_),
                                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
            }
#else
            h.got = board_take<BOARD == 2>(sm, board_tag(G, own ? dr.wage : dr.page, g_run0), own ? dr.wid : dr.pid, col, arrive_goal);
#endif
        }
    }
    h.lp_i = BOARD ? lane_value(h.got, 16 + nd) : sm.LP[dr.wid];
#pragma unroll
    for (int d = 0; d < kD; ++d) {
        h.x[d] = d < nd ? (BOARD ? lane_value(h.got, 16 + d) : xs[d]) : 0.;
        h.cj[d] = d < nd ? (BOARD ? lane_value(h.got, d) : cs_[d]) : 0.;
    }
    // (measured: requesting the prior in FRONT of the poll instead changes nothing -- 5.45 against 5.43 us; its way from L2
    // is hidden behind the logarithms either way)
    h.prior = PriorDev{0, 0, 0., 0., 0., 1.};
    if (lane < pb.n_dim && pb.has_priors) h.prior = pb.priors[lane];
}

template <int ND, int BOARD = 0, int MODEL = 0>
__device__ __forceinline__ void proposal_head(const DevProblem& pb, const DevSampler& sm, const DrawRec& dr, int lane,
                                              double* __restrict__ sc, double* __restrict__ sq, double* __restrict__ sx,
                                              const HeadRows<ND>& h) {
    constexpr int kD = ND > 0 ? ND : kMaxDim;
    const int nd = ND > 0 ? ND : sm.n_dim;
    const PriorDev my_prior = h.prior;
    double x[kD], q[kMaxDim], lq[kMaxDim];
#pragma unroll
    for (int d = 0; d < kMaxDim; ++d) q[d] = lq[d] = 0.;
    const double got = h.got;
    const double lp_i = h.lp_i;
    double arg = 1.;
#pragma unroll
    for (int d = 0; d < kD; ++d) {
        x[d] = h.x[d];
        const double cj = h.cj[d];
        q[d] = d < nd ? cj - (cj - x[d]) * dr.z : 0.;   // emcee: c_j - (c_j - x_i) z
        if (lane == d && d < pb.n_par) arg = q[d];
    }
#ifdef LCF_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    LCF_STAMP(0, 2);
    const double lg = flog(arg);
#pragma unroll
    for (int d = 0; d < kD; ++d) lq[d] = lane_value(lg, d);
    LCF_STAMP(0, 3);
    double c[kNCoef];
    walker_coefficients<MODEL>(pb, q, lq, c, MODEL ? true : pb.use_itab != 0);
    LCF_STAMP(0, 4);
    double lpr = 0.;
    if (pb.has_priors) {
        double qv = 0.;
#pragma unroll
        for (int d = 0; d < kD; ++d)
            if (lane == d) qv = q[d];
        const double mine = lane < pb.n_dim ? prior_term(my_prior, qv) : 0.;
#pragma unroll
        for (int d = 0; d < kD; ++d)
            if (d < nd) lpr += lane_value(mine, d);   // the same ordered sum as walker_log_prior
    }
    const double count = BOARD ? lane_value(got, 16 + nd + 1) : 0.;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < kNCoef; ++k) sc[k] = c[k];
        sc[kNCoef] = lpr;
#pragma unroll
        for (int d = 0; d < kD; ++d)
            if (d < nd) {
                sq[d] = q[d];
                sx[d] = x[d];
            }
        sx[kMaxDim] = lp_i;
        if (BOARD) sx[kMaxDim + 1] = count;   // the walker's acceptance count so far
    }
    LCF_STAMP(0, 5);
}

// ---- single-GPU fit, one workgroup per PROPOSAL: a half-step that owns its accept test ------------------------------
// k_fused gives every (proposal, part) its own workgroup, so no workgroup knows the proposal's likelihood: the accept
// test is re-derived by whoever needs the walker in the next launch, from per-slot records (proposal, draw, partial
// sums) that every launch publishes and the next one loads.  Here ONE workgroup of NPARTS x 256 threads evaluates all
// parts of its proposal -- threads [256 j, 256 j + 256) walk part j exactly as a k_fused workgroup would, same chunks,
// same reduction tree, so the chain is bitwise the one k_fused and the two-kernel path produce -- and thread 0 then
// accepts or rejects and commits the walker itself.  What that removes from the serial head of every half-step:
// the second dependent round trip (slot records, proposals and partial sums of three slots), the three redundant
// accept tests, the per-part repetition of the whole serial section, and every global store except the commit.
// Within a launch X[wid] is read and written by the walker's own workgroup only (partners come from the
// complementary colour, which this half-step does not move), so there is nothing to synchronise.
constexpr int kSoloScratch = kNCoef + 2 + 2 * (kMaxDim + (kMaxDim & 1));  // doubles: coefficients, log-prior, q, x

// BOARD (multi-GPU, lcf_sampler_run_rows): the launch covers this rank's slots [slot_lo, slot_lo + gridDim.x) of
// half-step G; rows come from this rank's row board and the commit posts the walker's new row on EVERY rank's board.
// No rank computes anything about another rank's proposals, and nothing but these rows travels.
// NPARTS: 2 = engines with up to two parts, one per 256 threads; 4 = three or four parts, the two groups of 256
// threads take parts j, j + 2 one after the other.  512 threads either way, so that two workgroups share a CU and one's
// serial head overlaps the other's points (1024-thread workgroups for four parts, one per CU: companion fit 1.26e7
// walker-steps/s; this way 1.41e7).
// MODEL > 0: the kernel is compiled for that one model and for what the benchmark shapes have in common -- dense columns,
// interpolants proved from their first interval, no fitted sigma -- so that the other models' arithmetic is not in its
// instruction stream (a launch streams its code from L2 into cold instruction caches: 89 KiB of kernel, ~28 KiB of
// them executed, cost the generic kernel a third of its time).
// NPARTS = 8: three or four parts again, but FOUR groups of 256 threads, one part each at the same time (1024 threads):
// for launches of at most one workgroup per CU -- a rank's share of a strongly scaled ensemble, configs[2] on 8 GPUs:
// 256 proposals -- where a second workgroup to overlap with does not exist and the parts of one proposal can.
// BOARD = 3: the same between ranks (k_solo_run<..., RANKS>): rows from this rank's board, the commit posts on every rank's;
// state and chain are collected from the board behind the launch (k_board_collect), nothing else is written here.
// BOARD = 2: one half-step of a ONE-LAUNCH run (k_solo_run below): rows from / to this GPU's own board, the chain written
// here; `first`: the workgroup's first half-step of the launch (tables staged, first columns fetched: both stay).
// Returns true when the run is aborted (uniform over the workgroup).
template <int ND, int VARIANT, bool THERM, int NPARTS, int BOARD, int MODEL>
__device__ __forceinline__ bool solo_half_step(const DevProblem& pb, const DevProblem* __restrict__ pbp, const DevSampler& sm, long long row,
                                               const DrawRec* __restrict__ draws, const DrawRec* draws_next, long long G,
                                               long long g_run0, int i, unsigned char* smem, ColumnOperands& first_col,
                                               bool first, bool write_state, const int tid, const int run_flags = 0,
                                               const unsigned int arrive_goal = 0u) {
    double* exptab = reinterpret_cast<double*>(smem);
    double* red = exptab + kExpTabSize;                                     // 4 wave sums per part (32 reserved)
    double2* ltab = reinterpret_cast<double2*>(smem + kLdsHead * sizeof(double));
    const FiltDesc* fdesc = reinterpret_cast<const FiltDesc*>(ltab + pb.n_lds_tab);
    const int itab_at = pb.n_itab_lds > 0
                            ? (int)(kLdsHead * sizeof(double) + (pb.n_lds_tab + kFdD2 * pb.n_filters) * sizeof(double2)) : -1;
    double* sc = reinterpret_cast<double*>(ltab + pb.stage_d2);  // coefficients, then log-prior
    double* sq = sc + kNCoef + 2;                                           // the proposal
    double* sx = sq + kMaxDim + (kMaxDim & 1);                              // the walker's current position, lp, draw
    int* sctl = reinterpret_cast<int*>(sc + kSoloScratch + 2);              // BOARD: [0] = 1: the launch is aborted
    constexpr int kGroups = NPARTS == 8 ? 4 : 2;     // groups of 256 threads, each walks one part at a time
    constexpr int kThreads = kBlock * kGroups;
    constexpr int kD = ND > 0 ? ND : kMaxDim;
    const int nd = ND > 0 ? ND : sm.n_dim;
    const bool reddened = !MODEL && pb.model == kShockCooling3;
    LCF_STAMP(0, 0);
#ifdef LCF_STAMPS
    if (tid == 0 && blockIdx.x < 1024) g_wall[((G & 1) * 1024 + blockIdx.x) * 2] = wall_clock64();
#endif
    const DrawRec dr = draws[i];     // wave-uniform
    if (BOARD == 1 && blockIdx.x == 0 && tid < sm.n_board_ranks)
        // This launch runs, so every launch in front of it in the stream has finished: tell every rank that this rank
        // is through with all half-steps before G (stream order does the counting; no atomics).
        __hip_atomic_store(board_progress(sm.peer_board[tid], sm) + sm.board_rank, (unsigned int)G, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    if (dr.wid < 0) return false;    // an odd ensemble's smaller colour leaves its last slot empty
    LCF_STAMP(0, 1);
    // the operands of this thread's first column (of the first part its half of the workgroup walks): requested now,
    // needed behind the barrier
    // (the head's own loads go first; the other waves' streams -- tables, photometry: 80 KiB per workgroup through the
    // same path -- start once they are out: rows + proposal 3.8 k cycles with everything requested at once, 2.1 k alone)
    // (model-specialised kernels only: the generic ones need the registers for the point-by-point path)
    constexpr bool kFetch = MODEL != 0;
    if (tid < 64) {
        // Resident workgroups drift apart, so this head shares its SIMD with column waves of the CU's other workgroup:
        // the one wave that everybody behind it waits for goes first (5.75 against 6.03 us per half-step; with a launch
        // per half-step both workgroups are in their heads at once and the priority only starves the staging waves:
        // 8.06 against 7.71 us)
        if (BOARD >= 2) __builtin_amdgcn_s_setprio(3);
        HeadRows<ND> rows;
        head_fetch<ND, BOARD>(pb, sm, dr, tid, rows, G, g_run0, arrive_goal);
        if (kFetch && first) fetch_column<VARIANT, MODEL>(pb, tid / kBlock, tid % kBlock, first_col);
        proposal_head<ND, BOARD, MODEL>(pb, sm, dr, tid, sc, sq, sx, rows);
        if (BOARD >= 2) __builtin_amdgcn_s_setprio(0);
    } else {
        if (LCF_HEAD_START > 0) __builtin_amdgcn_s_sleep(LCF_HEAD_START);
        if (kFetch && first) fetch_column<VARIANT, MODEL>(pb, tid / kBlock, tid % kBlock, first_col);
        // Touch the draw record this block index needs in the NEXT launch: block -> XCD placement repeats from launch
        // to launch, so the record is then in this XCD's L2 instead of HBM when the next serial head starts with it
        // (a hint only: nothing depends on the value or on the placement).
        if (BOARD >= 2) {
            // Resident launches: the record is read by a SCALAR load at the top of the next half-step, so it is the scalar
            // cache of this CU that should hold it by then -- two scalar loads (the record may straddle a line), waited for
            // here, by a wave that has nothing else to do in the shadow of the head (5.51 -> 5.42 us per half-step against
            // the vector touch below, which only brings the record into the XCD's L2).
            if (tid >= 64 && tid < 128 && draws_next != nullptr) {
                const int* nxt = reinterpret_cast<const int*>(draws_next + i);
                int a_, b_;
                asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x2c\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(a_), "=&s"(b_) : "s"(nxt) : "memory");
            }
        } else if (tid == 64 && draws_next != nullptr) {
            const volatile int* nxt = reinterpret_cast<const volatile int*>(draws_next + i);
            (void)nxt[0];
            (void)nxt[sizeof(DrawRec) / sizeof(int) - 1];
        }
        if (!reddened && first) stage_tables<VARIANT, true>(pb, exptab, ltab, 0., tid - 64, kThreads - 64);
        LCF_STAMP(1, 11);
        if (BOARD >= 2 && tid == 64) sctl[0] = board_aborted<BOARD == 2>(sm) ? 1 : 0;   // (in the shadow of the head)
        if (BOARD == 1 && tid < 128) {
            // In the shadow of the head: has every rank finished half-step G - 2 (lane = rank)?  has this rank given up?
            const int lane = tid - 64;
            const unsigned int* progress = board_progress(sm.board, sm);
            const unsigned long long t0 = wall_clock64();
            bool abort = false;
            for (;;) {
                bool ok = true;
                // rank `lane` has posted P when all its half-steps before P were finished: G - 2 is finished from P = G - 1
                if (G - 2 >= g_run0 && lane < sm.n_board_ranks)   // (earlier half-steps ended with an earlier run)
                    ok = (int)(__hip_atomic_load(progress + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) -
                               (unsigned int)(G - 1)) >= 0;
                abort = board_aborted(sm);
                if (__all(ok) || abort) break;
                if (wall_clock64() - t0 > sm.wait_ticks) {
                    board_abort(sm, 2u, (unsigned int)G, (unsigned int)__builtin_ctzll(~__ballot(ok)), 0u);
                    abort = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0) sctl[0] = abort ? 1 : 0;
        }
    }
    __syncthreads();
    LCF_STAMP(0, 6);
    if (BOARD && sctl[0] != 0) return true;   // (uniform: everybody reads the same word)
    const double lpr = sc[kNCoef];
    double term = 0.;
    const bool excluded = lpr == -INFINITY;  // prior excludes the proposal: likelihood skipped (fitting.py:125)
    // The benchmark shape in a MODEL-specialised kernel: every lane has ONE column, fetched ahead of the head
    // (lean_column); anything else -- more columns than lanes, another filter count, interpolants outside LDS, a wave with
    // a state outside the interpolants -- goes through epochs_loop / the cold path, same numbers.
    bool lean = false;
    if (MODEL != 0 && NPARTS <= 2 && !excluded) {
        const int part = __builtin_amdgcn_readfirstlane(tid / kBlock), ltid = tid % kBlock;
        const int c0 = part_entry(pb.part_col0, part), c1 = part_entry(pb.part_col0, part + 1);
        lean = first_col.have_o && itab_at >= 0 && c1 - c0 <= kBlock && part < pb.n_parts;   // (wave-uniform)
        if (lean) {
            LCF_STAMP(0, 7);
            double acc = 0.;
            if (c0 + (ltid & ~63) < c1) {   // (a virtual wave without columns adds nothing)
                const bool live = c0 + ltid < c1;
#ifdef LCF_EXPERIMENT_NOCOLUMNS   // (timing experiment, wrong chain: no likelihood arithmetic)
                if (false)
#else
                if (!lean_column<MODEL>(pb, sc, first_col, ExpTab{exptab}, itab_at, acc))
#endif
                    acc = cold_column_with_state<VARIANT, MODEL>(pbp, sc, sq, first_col.t, min(c0 + ltid, c1 - 1), live);
                term = live ? acc : 0.;     // (lanes beyond the part repeated its last column)
            }
            LCF_STAMP(0, 8);
            LCF_STAMP(1, 12);
            const double ws = wave_sum(term);
            if ((tid & 63) == 0) red[tid >> 6] = ws;
        }
    }
    if (!excluded && !lean) {
        double cs[kNCoef];
#pragma unroll
        for (int k = 0; k < kNCoef; ++k) cs[k] = uniform_f64(sc[k]);
        if (reddened) stage_tables<VARIANT, true>(pb, exptab, ltab, cs[6], tid, kThreads);
        // threads [256 j, 256 j + 256) are the virtual threads of part j (NPARTS = 4: then of part j + 2, ...): each lane
        // computes the thermal state of its column and walks the column's points (epochs_loop)
        const int part = tid / kBlock, ltid = tid % kBlock;
        if (reddened) __syncthreads();
        LCF_STAMP(0, 7);
        if (NPARTS > 2) {
#pragma unroll 1
            for (int pp = part; pp < pb.n_parts; pp += kGroups) {
                term = epochs_loop<VARIANT, true, kFetch, MODEL, true>(pb, pp, sq, cs, ltab, fdesc, ExpTab{exptab}, ltid, itab_at,
                                                                       first_col, pp == part, pbp, sc);
                const double ws = wave_sum(term);
                if ((tid & 63) == 0) red[4 * pp + (ltid >> 6)] = ws;
            }
        } else {
#ifdef LCF_TWICE   // diagnostic: the column phase twice through the SAME code (second pass: warm instruction cache)
            {
                int reps_;
                asm volatile("s_mov_b32 %0, 2" : "=s"(reps_));
#pragma unroll 1
                for (int rep = 0; rep < reps_; ++rep) {
                    if (part < pb.n_parts)
                        term = epochs_loop<VARIANT, true, kFetch, MODEL, true>(pb, part, sq, cs, ltab, fdesc, ExpTab{exptab}, ltid,
                                                                               itab_at, first_col, true, pbp, sc);
                    if (rep == 0) LCF_STAMP(0, 7);
                }
            }
#else
            if (part < pb.n_parts)
                term = epochs_loop<VARIANT, true, kFetch, MODEL, true>(pb, part, sq, cs, ltab, fdesc, ExpTab{exptab}, ltid, itab_at,
                                                                       first_col, true, pbp, sc);
#endif
            LCF_STAMP(0, 8);
            LCF_STAMP(1, 12);
            const double ws = wave_sum(term);
            if ((tid & 63) == 0) red[tid >> 6] = ws;
        }
    }
    __syncthreads();
    LCF_STAMP(0, 9);
    if (BOARD) {
        // ---- accept / reject by wave 0; lanes 0 .. nd+1 post the row (position, log-posterior, acceptance count) on every
        // rank's board; this rank's X / LP / counts follow for its own walkers (the others' come from the board when
        // the run ends, and the chain is written from the board)
        if (tid >= 64) return false;
        if (BOARD >= 2) __builtin_amdgcn_s_setprio(3);   // (the row's readers wait for this)
        double nlp = -INFINITY;
        if (!excluded) {
            double sum = pb.use_sigma ? 0. : pb.log_norm_const;
            for (int k = 0; k < pb.n_parts; ++k) sum += (red[4 * k] + red[4 * k + 1]) + (red[4 * k + 2] + red[4 * k + 3]);
            nlp = lpr - 0.5 * sum;
        }
        const double lp_i = sx[kMaxDim];
        const bool ok = (dr.zl + nlp - lp_i) > dr.lnu;
        const double count = sx[kMaxDim + 1] + (ok ? 1. : 0.);
        if (tid <= nd + 1) {
            const double v = tid < nd ? (ok ? sq[tid] : sx[tid]) : tid == nd ? (ok ? nlp : lp_i) : count;
            if (BOARD == 2) {
                board_post<true>(sm.board, sm, (unsigned int)(G + 1), dr.wid, tid, v);
            } else {
#pragma unroll
                for (int r = 0; r < kMaxPeers; ++r)
                    if (r < sm.n_board_ranks) board_post(sm.peer_board[r], sm, (unsigned int)(G + 1), dr.wid, tid, v);
            }
            // (a one-launch run writes the state in its last step only: every walker moves exactly once there, while two
            // moves of a walker in one launch come from different workgroups, and whose store reaches memory last is open)
            if (BOARD == 1 || (BOARD == 2 && write_state)) {
                // (one-launch runs: into the set of state buffers the run did NOT start from -- kRunFlip says which)
                const bool to_out = BOARD == 2 && !(run_flags & kRunFlip);
                double* X = to_out ? sm.X_out : sm.X;
                double* LP = to_out ? sm.LP_out : sm.LP;
                long long* nacc = to_out ? sm.nacc_out : sm.nacc;
                if (tid < nd)
                    X[(size_t)dr.wid * nd + tid] = v;
                else if (tid == nd)
                    LP[dr.wid] = v;
                else
                    nacc[dr.wid] = (long long)v;
                if (BOARD == 2 && sm.snap_out) {   // ... and the host's copy of it
                    const size_t nw = sm.n_walkers;
                    if (tid < nd) sm.snap_out[1 + (size_t)dr.wid * nd + tid] = (unsigned long long)__double_as_longlong(v);
                    else if (tid == nd) sm.snap_out[1 + nw * nd + dr.wid] = (unsigned long long)__double_as_longlong(v);
                    else sm.snap_out[1 + nw * nd + nw + dr.wid] = (unsigned long long)(long long)v;
                }
            }
            if (BOARD == 2 && (run_flags & kRunStoreChain)) {   // (one GPU: every walker's row is decided here)
                if (tid < nd) sm.chain[((size_t)row * sm.n_walkers + dr.wid) * nd + tid] = v;
                else if (tid == nd) sm.chain_lp[(size_t)row * sm.n_walkers + dr.wid] = v;
            }
        }
        if (tid == 0 && nlp != nlp) {
            atomicOr(sm.err, 1);
            if (BOARD == 2 && sm.snap_flags) sm.snap_flags[blockIdx.x & (kSnapFlags - 1)] = 1u;
        }
        if (BOARD >= 2) __builtin_amdgcn_s_setprio(0);
        LCF_STAMP(0, 10);
        return false;
    }
    if (tid != 0) return false;
    // ---- accept / reject and commit (models.py:121-135 -> fitting.py:121-128 -> emcee's stretch move)
    double nlp = -INFINITY;
    if (!excluded) {
        double sum = pb.use_sigma ? 0. : pb.log_norm_const;   // fixed order: parts, each (w0 + w1) + (w2 + w3)
        for (int k = 0; k < pb.n_parts; ++k) sum += (red[4 * k] + red[4 * k + 1]) + (red[4 * k + 2] + red[4 * k + 3]);
        nlp = lpr - 0.5 * sum;
    }
    const double lp_i = sx[kMaxDim];
    const bool ok = (dr.zl + nlp - lp_i) > dr.lnu;   // emcee: (ndim - 1) ln z + lp_new - lp_old > ln u
    if (nlp != nlp) atomicOr(sm.err, 1);
    if (ok) {
#pragma unroll
        for (int d = 0; d < kD; ++d)
            if (d < nd) sm.X[(size_t)dr.wid * nd + d] = sq[d];
        sm.LP[dr.wid] = nlp;
        atomicAdd(reinterpret_cast<unsigned long long*>(&sm.nacc[dr.wid]), 1ull);  // (no return value: nothing waits)
    }
    if (sm.store_chain) {
        double* crow = sm.chain + ((size_t)row * sm.n_walkers + dr.wid) * nd;
#pragma unroll
        for (int d = 0; d < kD; ++d)
            if (d < nd) crow[d] = ok ? sq[d] : sx[d];
        sm.chain_lp[(size_t)row * sm.n_walkers + dr.wid] = ok ? nlp : lp_i;
    }
    LCF_STAMP(0, 10);
#ifdef LCF_STAMPS
    if (blockIdx.x < 1024) g_wall[((G & 1) * 1024 + blockIdx.x) * 2 + 1] = wall_clock64();
#endif
    return false;
}

template <int ND, int VARIANT, bool THERM, int NPARTS, bool BOARD = false, int MODEL = 0>
__global__ __launch_bounds__(kBlock * (NPARTS == 8 ? 4 : 2), LCF_WAVES)
void k_solo(const DevProblem* __restrict__ pbp, const DevSampler sm, long long row, const DrawRec* __restrict__ draws,
            const DrawRec* draws_next, long long G, long long g_run0, int slot_lo) {
    extern __shared__ __align__(16) unsigned char smem[];
    ColumnOperands first_col;
    // The problem is read through a constant-address-space pointer: scalar loads where a field is used, instead of
    // 700 bytes of kernel arguments preloaded into (and spilled from) scalar registers.
    typedef const DevProblem __attribute__((address_space(4)))* ProblemPtr;
    const DevProblem& pb = *(const DevProblem*)(ProblemPtr)pbp;
    solo_half_step<ND, VARIANT, THERM, NPARTS, BOARD ? 1 : 0, MODEL>(pb, pbp, sm, row, draws, draws_next, G, g_run0,
                                                                     blockIdx.x + (BOARD ? slot_lo : 0), smem, first_col, true,
                                                                     true, threadIdx.x);
}

// ---- a whole run (or a block of it) in ONE launch -------------------------------------------------------------------
// The workgroups stay: workgroup b takes slot b (b + gridDim.x, ...) of half-step G0, then of G0 + 1, ... -- no kernel
// boundary between half-steps, tables staged and first columns fetched once per launch, and nothing makes two
// workgroups of a CU run in step, so one's serial head overlaps the other's columns.  What a half-step needs from an
// earlier one -- the rows of its walker and of the partner -- comes from a board of tagged rows in this GPU's memory,
// exactly as between the ranks of a row-board run (BOARD = 2): a head polls for the version its draw record names.
// Every wait is for a row of an EARLIER half-step, so the launch makes progress as long as all its workgroups are
// resident at once: the host launches no more than the device holds.  The ring keeps more versions than a launch has
// half-steps (DevSampler::ring), so no workgroup can overrun a version another one still waits for.
// RANKS: the launch of ONE RANK of a row-board run.  The rank's workgroups take its slots [slot_lo, slot_lo + n_slots) of
// every half-step of the launch; rows come from the rank's own board, every commit goes to the boards of all ranks
// (BOARD = 3).  What bounds the drift between ranks is a word per rank and launch instead of one per half-step: the
// launch's first workgroup posts "this rank has reached half-step G0" on every board -- stream order proves that all the
// rank's earlier launches and their row collections are complete -- and no workgroup starts before every rank has
// reached `need_progress` (the first half-step of the launch before the previous one; 0: nothing to wait for).  With
// at most kRunSpan half-steps per launch, anything a rank still reads is then less than kRing versions behind
// anything another rank writes.
template <int ND, int VARIANT, bool THERM, int NPARTS, int MODEL, bool RANKS = false>
__global__ __launch_bounds__(kBlock * (NPARTS == 8 ? 4 : 2), LCF_WAVES)
void k_solo_run(const DevProblem* __restrict__ pbp, const DevSampler* __restrict__ smp, long long rel0,
                const DrawRec* __restrict__ draws0, long long g_run0, int n_hs, long long state_from, int n_wg, int run_flags,
                unsigned int arrive_goal, int slot_lo, int n_slots, long long need_progress) {
    // (n_wg = gridDim.x, except in the test of a launch whose workgroups are not all there: LCF_RUN_TEST_MISSING)
    extern __shared__ __align__(16) unsigned char smem[];
    // (a lane's first column: fetched by the launch's first half-step and kept -- 26 registers, a few of them spilled;
    // fetching it again every half-step costs 0.5 us, in front of the head's own loads)
    ColumnOperands first_col;
    // Problem AND sampler are read through constant-address-space pointers: scalar loads where a field is used.  As
    // kernel arguments the sampler's 60 words stayed in scalar registers across the loop over the half-steps -- and
    // most of them in spill lanes.  What varies from run to run travels in `run_flags`, so that the copy in device
    // memory is written once per sampler (lcf_sampler::run_image).
    typedef const DevProblem __attribute__((address_space(4)))* ProblemPtr;
    typedef const DevSampler __attribute__((address_space(4)))* SamplerPtr;
    const DevProblem& pb = *(const DevProblem*)(ProblemPtr)pbp;
    const DevSampler& sm = *(const DevSampler*)(SamplerPtr)smp;
    constexpr int kBoard = RANKS ? 3 : 2;
    bool first = true;
    // "this workgroup has started": what a workgroup that waits unusually long for a row looks at (board_take)
    if (threadIdx.x == 0) __hip_atomic_fetch_add(board_arrivals(sm), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (RANKS) {
        __shared__ int s_abort;
        if (blockIdx.x == 0 && threadIdx.x < sm.n_board_ranks)
            __hip_atomic_store(board_progress(sm.peer_board[threadIdx.x], sm) + sm.board_rank, (unsigned int)(g_run0 + rel0),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x < 64) {
            bool abort = false;
            if (need_progress > 0) {
                const int lane = threadIdx.x;
                const unsigned int* progress = board_progress(sm.board, sm);
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    bool ok = true;
                    if (lane < sm.n_board_ranks)
                        ok = (int)(__hip_atomic_load(progress + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) -
                                   (unsigned int)need_progress) >= 0;
                    abort = board_aborted(sm);
                    if (__all(ok) || abort) break;
                    if (wall_clock64() - t0 > sm.wait_ticks) {
                        board_abort(sm, 2u, (unsigned int)(g_run0 + rel0), (unsigned int)__builtin_ctzll(~__ballot(ok)),
                                    (unsigned int)need_progress);
                        abort = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            if (threadIdx.x == 0) s_abort = abort ? 1 : 0;
        }
        __syncthreads();
        if (s_abort != 0) return;
    }
    if (!RANKS && rel0 == 0) {
        // The run's first launch: the state in front of it is version g_run0 of every row (what a draw record's age
        // points at until the walker has moved in this run).  Whoever needs a row polls for it: no barrier behind this.
        // (RANKS: every rank holds the whole state and has posted it on its OWN board: k_board_init, in front of the run.)
        const int nd = ND > 0 ? ND : sm.n_dim;
        const int col = threadIdx.x & 31;   // (n_dim + 2 <= kMaxDim + 2 columns)
        const bool flip = (run_flags & kRunFlip) != 0;
        const double* X = flip ? sm.X_out : sm.X;
        const double* LP = flip ? sm.LP_out : sm.LP;
        const long long* nacc = flip ? sm.nacc_out : sm.nacc;
        for (int w = blockIdx.x * (int)(blockDim.x / 32) + (int)(threadIdx.x / 32); w < sm.n_walkers;
             w += n_wg * (int)(blockDim.x / 32))
            if (col <= nd + 1) {
                const double v = col < nd ? X[(size_t)w * nd + col] : col == nd ? LP[w] : (double)nacc[w];
                board_post<true>(sm.board, sm, (unsigned int)g_run0, w, col, v);
            }
    }
    const int slot_end = slot_lo + n_slots;
    // (Measured and dropped: the draw record of the NEXT half-step requested one half-step ahead -- by a wave beside the head
    // into LDS, or as a scalar load carried in registers across the half-step -- instead of the touch that only brings
    // it into this XCD's L2: 5.80 / 5.94 against 5.51 us per half-step.  The record's two round trips are hidden
    // already; what the extra live registers cost is not.)
#pragma unroll 1
    for (int h = 0; h < n_hs; ++h) {
        const DrawRec* draws = draws0 + (size_t)h * sm.n_half;
#pragma unroll 1
        for (int i = slot_lo + blockIdx.x; i < slot_end; i += n_wg) {
            if (draws[i].wid < 0) continue;   // (uniform) an odd ensemble's smaller colour leaves its last slot empty
            // the record this workgroup needs next: its next slot of this half-step, else its first of the next one
            const bool more = i + n_wg < slot_end;
            const DrawRec* hint = more ? draws + n_wg : h + 1 < n_hs ? draws + sm.n_half + (slot_lo + (int)blockIdx.x - i) : nullptr;
            const int tid = threadIdx.x;
            if (solo_half_step<ND, VARIANT, THERM, NPARTS, kBoard, MODEL>(pb, pbp, sm, (rel0 + h) >> 1, draws, hint,
                                                                          g_run0 + rel0 + h, g_run0, i, smem, first_col, first,
                                                                          rel0 + h >= state_from, tid, run_flags, arrive_goal))
                return;
            first = false;
        }
    }
}

// The images of DevProblem / DevSampler that kernels read through constant-address-space pointers are WRITTEN BY A
// KERNEL on the stream of the launches that read them: ordinary producer -> consumer order between two kernels of one
// queue.  (A host copy into an image whose address an earlier one had -- a sampler destroyed, the next created -- left
// some workgroups of the next launch with the old struct out of an XCD's L2: rows posted for another board layout, a
// launch that waited for them for ever.  Seen once the inter-rank boards grew and allocations started to recycle
// addresses; tools/debug/rows_mismatch.py.)
template <class T>
__global__ void k_put_image(T* __restrict__ dst, const T src) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = src;
}

// State of the sampler as 8-byte words into (mapped, pinned) host memory: [error flag | X | LP | n_accepted].
__global__ void k_snapshot(const DevSampler sm, unsigned long long* __restrict__ out) {
    const long long nx = (long long)sm.n_walkers * sm.n_dim, nw = sm.n_walkers;
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w == 0) out[0] = (unsigned long long)(unsigned int)*sm.err;
    else if (w <= nx) out[w] = reinterpret_cast<const unsigned long long*>(sm.X)[w - 1];
    else if (w <= nx + nw) out[w] = reinterpret_cast<const unsigned long long*>(sm.LP)[w - 1 - nx];
    else if (w <= nx + 2 * nw) out[w] = (unsigned long long)sm.nacc[w - 1 - nx - nw];
}

// ---- population mode: one launch covers the same half-step of MANY independent transients (blockIdx.y) --------------
struct MultiItem {
    DevProblem pb;
    DevSampler sm;
    const DrawRec* draws[2];        // the sampler's two block buffers (block b in buffer b & 1)
    long long blk_first, blk_steps; // steps in block 0 / in every later block
    double* coef;
    double* lprior;
    // resident population launches (k_pop_run): the transient's board is sm.board; `flip`: its start state is in
    // sm.X_out / LP_out / nacc_out; `arrive0`: its board's count of started workgroups before this run; `itab_extra`:
    // doubles of pb.itab the launch stages in LDS itself, behind the image (the engine's image has them only for long
    // light curves -- a launch per half-step cannot afford 24 KiB more per workgroup, a launch per 32 steps can)
    long long g_run0;
    int flip, itab_extra;
    unsigned int arrive0, pad;
};

// Draw records of a transient's half-step `rel` of the run (the host keeps that block resident).
__device__ inline const DrawRec* item_rows(const MultiItem& it, long long rel) {
    const long long k = rel / 2;
    const long long b = k < it.blk_first ? 0 : 1 + (k - it.blk_first) / it.blk_steps;
    const long long k0 = b == 0 ? 0 : it.blk_first + (b - 1) * it.blk_steps;
    return it.draws[b & 1] + (size_t)(rel - 2 * k0) * it.sm.n_half;
}

template <int ND>
__global__ __launch_bounds__(64) void k_step_multi(const MultiItem* __restrict__ items, int have_prev, long long prev_row,
                                                   int have_next, long long rel, long long g) {
    const MultiItem& it = items[blockIdx.y];
    const int nh = it.sm.n_half;
    if ((int)blockIdx.x >= nh) return;
    step_body<ND>(it.pb, it.sm, blockIdx.x, have_prev, prev_row, have_next,
                  have_next ? item_rows(it, rel) : nullptr, have_prev ? item_rows(it, rel - 1) : nullptr,
                  g, 0, nh, it.coef, it.lprior);
}

template <int VARIANT, bool LDS_TAB, bool THERM>
__global__ __launch_bounds__(kBlock) void k_points_multi(const MultiItem* __restrict__ items, int parity) {
    const MultiItem& it = items[blockIdx.y];
    const int nh = it.sm.n_half;
    if ((int)blockIdx.x >= nh * it.pb.n_parts) return;
    points_body<VARIANT, 0, LDS_TAB, THERM>(it.pb, blockIdx.x, 0, nh, it.sm.Q[parity], it.coef, it.lprior, nullptr,
                                            it.sm.part2[parity], nullptr);
}

// ---- population mode, ONE launch per half-step (k_pop): a workgroup takes FOUR proposals of one transient -----------
// The two launches above spend most of a half-step in k_step_multi's latency chain and in global round trips (proposal,
// coefficients, thermal states and partial sums all travel through memory).  Here one workgroup of four waves owns four
// consecutive proposal slots of a transient: it stages the transient's tables once, every wave runs the serial head of
// ONE of the proposals (four latency chains side by side instead of one after the other), all threads then compute the
// thermal states of the four proposals together and walk the points proposal by proposal -- thread t takes the points
// t, t + 256, ... of a part exactly as a k_points workgroup does, same reduction tree, so a transient's chain is bitwise
// the one the two-launch path produces -- and lane 0 of wave w accepts or rejects and commits proposal w.  Nothing
// between launches but the committed state: no slot records, no coefficients, no thermal states in memory.  The
// transient's DevProblem is read through a constant-address-space pointer: scalar loads, as kernel arguments are.
constexpr int kPopScratch = kSoloScratch + 6;   // doubles per proposal: k_solo's + (ln z term, ln u, walker id), padded
constexpr int kPopMaxParts = 4;
#ifndef LCF_POP_RUN_GROUP
#define LCF_POP_RUN_GROUP 4
#endif
constexpr int kPopRunGroup = LCF_POP_RUN_GROUP;   // proposals (= waves) per workgroup of the resident form (k_pop_run)

#ifndef LCF_POP_WAVES
#define LCF_POP_WAVES LCF_WAVES
#endif
// GROUP = proposals (= waves) per workgroup: 4, or 8 with two proposals' points walked side by side (threads
// [0, 256) and [256, 512), each half exactly as a k_points workgroup).
template <int ND, int VARIANT, int GROUP, int MODEL = 0>
__global__ __launch_bounds__(64 * GROUP, LCF_POP_WAVES) void k_pop(const MultiItem* __restrict__ items, long long rel) {
    extern __shared__ __align__(16) unsigned char smem[];
    typedef const MultiItem __attribute__((address_space(4)))* ItemPtr;
    const ItemPtr itp = (ItemPtr)(items + blockIdx.y);
    const MultiItem& it = *(const MultiItem*)itp;
    const DevProblem& pb = it.pb;
    const DevSampler& sm = it.sm;
    constexpr int kThreads = 64 * GROUP;
    const int nh = sm.n_half, slot0 = blockIdx.x * GROUP;
    if (slot0 >= nh) return;
    double* exptab = reinterpret_cast<double*>(smem);
    double2* ltab = reinterpret_cast<double2*>(smem + kLdsHead * sizeof(double));
    const FiltDesc* fdesc = reinterpret_cast<const FiltDesc*>(ltab + pb.n_lds_tab);
    const int itab_at = pb.n_itab_lds > 0
                            ? (int)(kLdsHead * sizeof(double) + (pb.n_lds_tab + kFdD2 * pb.n_filters) * sizeof(double2)) : -1;
    double* scratch = reinterpret_cast<double*>(ltab + pb.stage_d2);       // [GROUP][kPopScratch]
    double* red = scratch + GROUP * kPopScratch;                             // [GROUP][4 * kPopMaxParts] wave sums
    constexpr int kD = ND > 0 ? ND : kMaxDim;
    const int nd = ND > 0 ? ND : sm.n_dim;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        // ---- the serial heads, one per wave, side by side
        double* sc = scratch + wave * kPopScratch;
        double* sq = sc + kNCoef + 2;
        double* sx = sq + kMaxDim + (kMaxDim & 1);
        const int slot = slot0 + wave;
        DrawRec dr{-1, -1, -1, -1, 1., 0., 0., 0, 0};
        if (slot < nh) dr = item_rows(it, rel)[slot];     // wave-uniform
        const bool active = dr.wid >= 0;  // (an odd ensemble's smaller colour leaves its last slot empty)
        // (the rows are requested first, the tables staged while they are on their way: a launch per half-step pays the
        // staging every time, and in front of the heads it was a round trip of its own)
        HeadRows<ND> rows;
        if (active) head_fetch<ND>(pb, sm, dr, lane, rows);
        stage_tables<VARIANT, true>(pb, exptab, ltab, 0., tid, kThreads);
        if (active) proposal_head<ND, false, MODEL>(pb, sm, dr, lane, sc, sq, sx, rows);
        if (lane == 0) {
            if (!active) sc[kNCoef] = -INFINITY;
            sx[kMaxDim + 1] = dr.zl;
            sx[kMaxDim + 2] = dr.lnu;
            reinterpret_cast<int*>(sx + kMaxDim + 3)[0] = dr.wid;
        }
    }
    __syncthreads();
    // ---- thermal states + points: the unit of work is a VIRTUAL WAVE (proposal w, part, v) -- the 64 columns
    // part_col0 + 64 v + lane (+ 256 m) of epochs_loop, exactly what wave v of a k_points workgroup for (w, part) walks, so a
    // transient's chain is bitwise the one every other kernel produces.  The physical waves take the units round-robin,
    // proposal fastest: with up to GROUP non-empty units per (part, v) every wave stays with one proposal.
    {
        const int n_units = GROUP * pb.n_parts * 4;
#pragma unroll 1
        for (int u = wave; u < n_units; u += GROUP) {
            const int w = u % GROUP, pv = u / GROUP, part = pv >> 2, v = pv & 3;
            const double* sc = scratch + w * kPopScratch;
            double ws = 0.;
            const bool empty = part_entry(pb.part_col0, part) + 64 * v >= part_entry(pb.part_col0, part + 1);
            if (!empty && sc[kNCoef] != -INFINITY) {   // (wave-uniform)
                double cs[kNCoef];
#pragma unroll
                for (int k = 0; k < kNCoef; ++k) cs[k] = uniform_f64(sc[k]);
                const double term = epochs_loop<VARIANT, true, false, MODEL>(pb, part, sc + kNCoef + 2, cs, ltab, fdesc, ExpTab{exptab},
                                                               64 * v + lane, itab_at);
                ws = wave_sum(term);
            }
            if (lane == 0) red[(w * kPopMaxParts + part) * 4 + v] = ws;
        }
    }
    // (no barrier here: u = wave (mod GROUP), so a wave has walked the units of ITS proposal only -- the sums it reads
    // below are the ones its own lane 0 wrote, and a wave's LDS operations execute in order)
    if (lane != 0) return;
    // ---- accept / reject and commit (models.py:121-135 -> fitting.py:121-128 -> emcee's stretch move), one lane per proposal
    const double* sc = scratch + wave * kPopScratch;
    const double* sq = sc + kNCoef + 2;
    const double* sx = sq + kMaxDim + (kMaxDim & 1);
    const int wid = reinterpret_cast<const int*>(sx + kMaxDim + 3)[0];
    if (wid < 0) return;
    const double lpr = sc[kNCoef];
    double nlp = -INFINITY;
    if (lpr != -INFINITY) {
        double sum = pb.use_sigma ? 0. : pb.log_norm_const;   // fixed order: parts, each (w0 + w1) + (w2 + w3)
        const double* r = red + wave * kPopMaxParts * 4;
        for (int k = 0; k < pb.n_parts; ++k) sum += (r[4 * k] + r[4 * k + 1]) + (r[4 * k + 2] + r[4 * k + 3]);
        nlp = lpr - 0.5 * sum;
    }
    const double lp_i = sx[kMaxDim];
    const bool ok = (sx[kMaxDim + 1] + nlp - lp_i) > sx[kMaxDim + 2];   // emcee: (ndim - 1) ln z + lp_new - lp_old > ln u
    if (nlp != nlp) atomicOr(sm.err, 1);
    if (ok) {
#pragma unroll
        for (int d = 0; d < kD; ++d)
            if (d < nd) sm.X[(size_t)wid * nd + d] = sq[d];
        sm.LP[wid] = nlp;
        atomicAdd(reinterpret_cast<unsigned long long*>(&sm.nacc[wid]), 1ull);  // (no return value: nothing waits)
    }
    if (sm.store_chain) {
        const long long row = rel / 2;
        double* crow = sm.chain + ((size_t)row * sm.n_walkers + wid) * nd;
#pragma unroll
        for (int d = 0; d < kD; ++d)
            if (d < nd) crow[d] = ok ? sq[d] : sx[d];
        sm.chain_lp[(size_t)row * sm.n_walkers + wid] = ok ? nlp : lp_i;
    }
}

// What one lane of a wave wrote to LDS, read by the wave's other lanes WITHOUT a workgroup barrier in between: the hardware
// executes a wave's LDS operations in order, but the compiler sees one thread -- it may move a load of sc[k] in front of
// `if (lane == 0) sc[k] = ...` for the lanes that do not store.  A fence at wavefront scope costs no instruction and keeps
// the order of the program.
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- population mode with RESIDENT workgroups ------------------------------------------------------------------------
// k_pop's half-step in a loop over up to kRunSpanSolo half-steps, as k_solo_run is k_solo's: gridDim.y = transient,
// gridDim.x = the workgroups that stay for it (what the device holds, shared evenly); workgroup b takes the groups
// b, b + gridDim.x, ... of GROUP proposals of every half-step.  What a half-step needs from an earlier one comes from the
// transient's own board of tagged rows (the sampler's k_solo_run board): a wave's head polls the rows its draw record
// names, its commit posts the walker's new row.  No kernel boundary, the tables staged ONCE per launch -- and with them
// the interpolants, which a launch per half-step reads from L2 point by point because staging 24 KiB more per workgroup
// and half-step costs more than it saves.  A wave walks the units of ITS proposal only (u = wave mod GROUP), so beyond the
// staging there is no barrier at all.  Same arithmetic, same reduction tree: the chains are bitwise k_pop's.
constexpr int pop_run_waves(int group) { return group == 12 ? 6 : group == 10 ? 5 : 4; }   // waves per SIMD the build aims at
template <int ND, int VARIANT, int GROUP, int MODEL = 0>
__global__ __launch_bounds__(64 * GROUP, pop_run_waves(GROUP))
void k_pop_run(const MultiItem* __restrict__ items, long long rel0, int n_hs, long long state_from, int store_chain, int launch_no,
               int n_wg) {
    // (n_wg = gridDim.x, except in the test of a launch whose workgroups are not all there: LCF_RUN_TEST_MISSING)
    extern __shared__ __align__(16) unsigned char smem[];
    typedef const MultiItem __attribute__((address_space(4)))* ItemPtr;
    const ItemPtr itp = (ItemPtr)(items + blockIdx.y);
    const MultiItem& it = *(const MultiItem*)itp;
    const DevProblem& pb = it.pb;
    const DevSampler& sm = it.sm;
    constexpr int kThreads = 64 * GROUP;
    const int nh = sm.n_half;
    double* exptab = reinterpret_cast<double*>(smem);
    double2* ltab = reinterpret_cast<double2*>(smem + kLdsHead * sizeof(double));
    const FiltDesc* fdesc = reinterpret_cast<const FiltDesc*>(ltab + pb.n_lds_tab);
    const int itab_at = pb.n_itab_lds > 0
                            ? (int)(kLdsHead * sizeof(double) + (pb.n_lds_tab + kFdD2 * pb.n_filters) * sizeof(double2)) : -1;
    double* scratch = reinterpret_cast<double*>(ltab + pb.stage_d2);       // [GROUP][kPopScratch]
    double* red = scratch + GROUP * kPopScratch;                             // [GROUP][4 * kPopMaxParts] wave sums
    constexpr int kD = ND > 0 ? ND : kMaxDim;
    const int nd = ND > 0 ? ND : sm.n_dim;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long g_run0 = it.g_run0;
    const unsigned int arrive_goal = it.arrive0 + (unsigned int)(launch_no + 1) * (unsigned int)n_wg;
    const bool flip = it.flip != 0;
    if (tid == 0) __hip_atomic_fetch_add(board_arrivals(sm), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (rel0 == 0) {   // the transient's start state as version g_run0 of every row (whoever needs one polls for it)
        const int col = tid & 31;
        const double* X = flip ? sm.X_out : sm.X;
        const double* LP = flip ? sm.LP_out : sm.LP;
        const long long* nacc = flip ? sm.nacc_out : sm.nacc;
        for (int w = blockIdx.x * (kThreads / 32) + tid / 32; w < sm.n_walkers; w += n_wg * (kThreads / 32))
            if (col <= nd + 1) {
                const double v = col < nd ? X[(size_t)w * nd + col] : col == nd ? LP[w] : (double)nacc[w];
                board_post<true>(sm.board, sm, (unsigned int)g_run0, w, col, v);
            }
    }
    stage_tables<VARIANT, true>(pb, exptab, ltab, 0., tid, kThreads);
    if (it.itab_extra > 0) {   // the interpolants behind the image (DevProblem::n_itab_lds of this copy says they are there)
        double2* dst = reinterpret_cast<double2*>(smem + itab_at);
        const double2* src = reinterpret_cast<const double2*>(pb.itab);
        for (int k = tid; k < it.itab_extra / 2; k += kThreads) dst[k] = src[k];
    }
    __syncthreads();
    double* sc = scratch + wave * kPopScratch;
    double* sq = sc + kNCoef + 2;
    double* sx = sq + kMaxDim + (kMaxDim & 1);
    const int n_groups = (nh + GROUP - 1) / GROUP;
#pragma unroll 1
    for (int h = 0; h < n_hs; ++h) {
        const long long rel = rel0 + h, G = g_run0 + rel;
        const DrawRec* draws = item_rows(it, rel);
#pragma unroll 1
        for (int grp = blockIdx.x; grp < n_groups; grp += n_wg) {
            const int slot = grp * GROUP + wave;
            DrawRec dr{-1, -1, -1, -1, 1., 0., 0., 0, 0};
            if (slot < nh) dr = draws[slot];     // wave-uniform
            if (dr.wid < 0) continue;            // (an odd ensemble's smaller colour leaves its last slot empty)
            {
                // (the head is a latency chain that the proposal's points wait for; the waves it shares the SIMD with are
                // mostly in their points: 22.73 -> 21.97 us per half-step at configs[4])
                __builtin_amdgcn_s_setprio(2);
                HeadRows<ND> rows;
                head_fetch<ND, 2>(pb, sm, dr, lane, rows, G, g_run0, arrive_goal);
                proposal_head<ND, 2, MODEL>(pb, sm, dr, lane, sc, sq, sx, rows);
                __builtin_amdgcn_s_setprio(0);
                // (Measured and dropped: the rows of the wave's NEXT proposal requested here and looked at when its head
                // starts -- 23.6 against 21.9 us per half-step: loads return in order, so the first loads of the points
                // wait for that round trip past the caches, and the request costs registers.)
            }
            wave_lds_order();   // (lane 0 wrote coefficients, log-prior, proposal and row; every lane reads them)
            // ---- the units of this wave's proposal: (part, v) = the 64 columns part_col0 + 64 v + lane, as in k_pop
            const double lpr = sc[kNCoef];
            if (lpr != -INFINITY) {
                double cs[kNCoef];
#pragma unroll
                for (int k = 0; k < kNCoef; ++k) cs[k] = uniform_f64(sc[k]);
#pragma unroll 1
                for (int pv = 0; pv < pb.n_parts * 4; ++pv) {
                    const int part = pv >> 2, v = pv & 3;
                    double ws = 0.;
                    const bool empty = part_entry(pb.part_col0, part) + 64 * v >= part_entry(pb.part_col0, part + 1);
                    if (!empty) {
                        const double term = epochs_loop<VARIANT, true, false, MODEL>(pb, part, sq, cs, ltab, fdesc, ExpTab{exptab},
                                                                                     64 * v + lane, itab_at);
                        ws = wave_sum(term);
                    }
                    if (lane == 0) red[(wave * kPopMaxParts + part) * 4 + v] = ws;
                }
            }
            // ---- accept / reject; lanes 0 .. nd+1 post the walker's row (position, log-posterior, acceptance count)
            wave_lds_order();   // (lane 0 wrote the sums)
            __builtin_amdgcn_s_setprio(3);   // (the row's readers wait for this: 23.05 -> 22.70 us per half-step at configs[4])
            double nlp = -INFINITY;
            if (lpr != -INFINITY) {
                double sum = pb.use_sigma ? 0. : pb.log_norm_const;   // fixed order: parts, each (w0 + w1) + (w2 + w3)
                const double* r = red + wave * kPopMaxParts * 4;
                for (int k = 0; k < pb.n_parts; ++k) sum += (r[4 * k] + r[4 * k + 1]) + (r[4 * k + 2] + r[4 * k + 3]);
                nlp = lpr - 0.5 * sum;
            }
            const double lp_i = sx[kMaxDim];
            const bool ok = (dr.zl + nlp - lp_i) > dr.lnu;   // emcee: (ndim - 1) ln z + lp_new - lp_old > ln u
            const double count = sx[kMaxDim + 1] + (ok ? 1. : 0.);
            if (lane <= nd + 1) {
                double qv = 0., xv = 0.;
#pragma unroll
                for (int d = 0; d < kD; ++d)
                    if (lane == d && d < nd) {
                        qv = sq[d];
                        xv = sx[d];
                    }
                const double v = lane < nd ? (ok ? qv : xv) : lane == nd ? (ok ? nlp : lp_i) : count;
                board_post<true>(sm.board, sm, (unsigned int)(G + 1), dr.wid, lane, v);
                if (rel >= state_from) {   // the run's last step: the state, into the set of buffers the run did not start from
                    double* X = flip ? sm.X : sm.X_out;
                    double* LP = flip ? sm.LP : sm.LP_out;
                    long long* nacc = flip ? sm.nacc : sm.nacc_out;
                    if (lane < nd) X[(size_t)dr.wid * nd + lane] = v;
                    else if (lane == nd) LP[dr.wid] = v;
                    else nacc[dr.wid] = (long long)v;
                }
                if (store_chain) {
                    const long long row = rel / 2;
                    if (lane < nd) sm.chain[((size_t)row * sm.n_walkers + dr.wid) * nd + lane] = v;
                    else if (lane == nd) sm.chain_lp[(size_t)row * sm.n_walkers + dr.wid] = v;
                }
            }
            if (lane == 0 && nlp != nlp) atomicOr(sm.err, 1);
            __builtin_amdgcn_s_setprio(0);
            wave_lds_order();   // (the next proposal's head overwrites what the lanes have just read)
        }
    }
}

}  // namespace

// =================================================================================================================
// host side
// =================================================================================================================
namespace lcf {
thread_local std::string g_err;
lcf_status fail(lcf_status st, const std::string& msg) {
    g_err = msg;
    return st;
}
}  // namespace lcf

namespace {

}  // namespace

struct lcf_engine {
    int device = 0;
    DevProblem dp{};
    std::vector<void*> owned;
    hipStream_t stream = nullptr;
    int64_t samples_per_eval = 0;
    size_t lds_bytes = 0;
    int* d_tab_off = nullptr;   // per filter: (offset, count) of the full table in the device table
    int* d_ctab_off = nullptr;  // per filter: (offset, count) of the compressed table
    double* d_ctmin = nullptr;
    bool have_ctab = false, have_itab = false;
    int n_cus = 256;             // compute units of the device (launch shapes depend on it)
    DevProblem* d_dp = nullptr;  // `dp` in device memory, for the kernels that read it through a pointer
    lcf_status sync_dp();        // after every change of `dp`
    // workspace for n walkers
    int64_t cap = 0;
    double *wP = nullptr, *wcoef = nullptr, *wlprior = nullptr, *wpart = nullptr, *wout = nullptr;
    double2* wtherm = nullptr;
    // scratch for evaluate-type calls
    size_t big_bytes = 0;
    double* wbig = nullptr;

    ~lcf_engine() {
        hipSetDevice(device);
        for (void* p : owned) hipFree(p);
        free_ws();
        if (wbig) hipFree(wbig);
        if (stream) hipStreamDestroy(stream);
    }
    void free_ws() {
        for (double** p : {&wP, &wcoef, &wlprior, &wpart, &wout}) {
            if (*p) hipFree(*p);
            *p = nullptr;
        }
        if (wtherm) hipFree(wtherm);
        wtherm = nullptr;
        cap = 0;
    }
    lcf_status reserve(int64_t n) {
        if (n <= cap) return LCF_OK;
        LCF_HIP(hipStreamSynchronize(stream));
        free_ws();
        const int64_t c = std::max<int64_t>(n, 64);
        LCF_HIP(hipMalloc((void**)&wP, c * dp.n_dim * sizeof(double)));
        LCF_HIP(hipMalloc((void**)&wcoef, c * kNCoef * sizeof(double)));
        LCF_HIP(hipMalloc((void**)&wlprior, c * sizeof(double)));
        LCF_HIP(hipMalloc((void**)&wpart, c * (dp.n_parts + 1) * sizeof(double)));
        LCF_HIP(hipMalloc((void**)&wout, c * sizeof(double)));
        if (dp.use_therm) LCF_HIP(hipMalloc((void**)&wtherm, c * dp.n_epochs * sizeof(double2)));
        cap = c;
        return LCF_OK;
    }
    lcf_status reserve_big(size_t bytes) {
        if (bytes <= big_bytes) return LCF_OK;
        LCF_HIP(hipStreamSynchronize(stream));
        if (wbig) hipFree(wbig);
        wbig = nullptr;
        big_bytes = 0;
        LCF_HIP(hipMalloc((void**)&wbig, bytes));
        big_bytes = bytes;
        return LCF_OK;
    }
};

// `dp` into its image in device memory: by a kernel on the engine's stream (see k_put_image), complete on return.
lcf_status lcf_engine::sync_dp() {
    if (!d_dp) {
        LCF_HIP(hipMalloc((void**)&d_dp, sizeof(DevProblem)));
        owned.push_back(d_dp);
    }
    LCF_HIP(hipDeviceSynchronize());   // (nothing in flight reads the old image)
    hipLaunchKernelGGL(k_put_image<DevProblem>, dim3(1), dim3(64), 0, stream, d_dp, dp);
    LCF_HIP(hipGetLastError());
    LCF_HIP(hipStreamSynchronize(stream));
    return LCF_OK;
}

namespace {

// More than the default 64 KiB of dynamic LDS must be granted per kernel function -- and per DEVICE: the attribute
// belongs to the current device's copy of the function, so it is set before every such launch (cheap next to one; a
// per-process flag would leave the second device of a process without it).
template <class K>
inline void allow_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <int VARIANT, int MODE>
void launch_points_v(const DevProblem& pb, dim3 grid, size_t lds, hipStream_t st, int w_lo, int n, const double* dP,
                     const double* coef, const double* lprior, const double2* therm, double* out0, double* out1) {
    const dim3 block(kBlock);
#define LCF_GO(L, T) do { allow_lds(k_points<VARIANT, MODE, L, T>, lds);                                               \
                          hipLaunchKernelGGL((k_points<VARIANT, MODE, L, T>), grid, block, lds, st, pb, w_lo, n, dP, coef, \
                                             lprior, therm, out0, out1); } while (0)
    if (pb.tab_in_lds) {
        if (pb.use_therm) LCF_GO(true, true); else LCF_GO(true, false);
    } else {
        if (pb.use_therm) LCF_GO(false, true); else LCF_GO(false, false);
    }
#undef LCF_GO
}

// Launches the per-point kernel for walkers [w_lo, w_lo + n) (absolute indices into P / coef / lprior / therm).
template <int MODE>
void launch_points(const lcf_engine* e, int w_lo, int n, const double* dP, const double* coef, const double* lprior,
                   double2* therm, double* out0, double* out1, hipStream_t st, bool do_thermal = true) {
    const DevProblem& pb = e->dp;
    if (pb.use_therm && do_thermal && MODE == 1) {   // (the likelihood's lanes compute their own: epochs_loop)
        const long long total = (long long)n * pb.n_epochs;
        hipLaunchKernelGGL(k_thermal, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, pb, w_lo, n,
                           coef, lprior, MODE == 0 ? 1 : 0, therm, (MODE != 2 && pb.variant != 0 && pb.use_itab) ? 1 : 0);
    }
    const dim3 grid((unsigned)((size_t)n * pb.n_parts));
    if (pb.variant == 0)
        launch_points_v<0, MODE>(pb, grid, e->lds_bytes, st, w_lo, n, dP, coef, lprior, therm, out0, out1);
    else
        launch_points_v<1, MODE>(pb, grid, e->lds_bytes, st, w_lo, n, dP, coef, lprior, therm, out0, out1);
}

// log-likelihood / log-posterior of n walkers, device pointers, enqueue only.
lcf_status logprob_dev(lcf_engine* e, int64_t n, const double* dP, double* dout, hipStream_t st, int with_prior) {
    if (n == 0) return LCF_OK;
    const int bs = 128;
    hipLaunchKernelGGL(k_prepare, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, st, e->dp, (int)n, dP, e->wcoef,
                       e->wlprior, with_prior);
    launch_points<0>(e, 0, (int)n, dP, e->wcoef, e->wlprior, e->wtherm, e->wpart, nullptr, st);
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, st, e->dp, (int)n, e->wpart,
                       e->wlprior, dout);
    LCF_HIP(hipGetLastError());
    return LCF_OK;
}

lcf_status logprob_host(lcf_engine* e, int64_t n, const double* P, double* out, int with_prior) {
    if (!e || n < 0 || (n > 0 && (!P || !out))) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return LCF_OK;
    if (n > (1 << 22)) return fail(LCF_ERR_INVALID_ARGUMENT, "at most 2^22 walkers per call");
    LCF_HIP(hipSetDevice(e->device));
    if (lcf_status st = e->reserve(n)) return st;
    LCF_HIP(hipMemcpyAsync(e->wP, P, n * e->dp.n_dim * sizeof(double), hipMemcpyHostToDevice, e->stream));
    if (lcf_status st = logprob_dev(e, n, e->wP, e->wout, e->stream, with_prior)) return st;
    LCF_HIP(hipMemcpyAsync(out, e->wout, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    LCF_HIP(hipStreamSynchronize(e->stream));
    return LCF_OK;
}

}  // namespace

extern "C" {

int32_t lcf_abi_version(void) { return LCF_ABI_VERSION; }

const char* lcf_last_error(void) { return lcf::g_err.c_str(); }

int32_t lcf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

lcf_status lcf_engine_create(const lcf_problem* pr, int32_t device, lcf_engine** out) {
    if (!pr || !out) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (pr->abi_version != LCF_ABI_VERSION) return fail(LCF_ERR_INVALID_ARGUMENT, "abi_version mismatch");
    switch (pr->model) {
        case LCF_MODEL_SHOCK_COOLING: case LCF_MODEL_SHOCK_COOLING2: case LCF_MODEL_SHOCK_COOLING3:
        case LCF_MODEL_SHOCK_COOLING4:
        case LCF_MODEL_COMPANION_SHOCKING: case LCF_MODEL_COMPANION_SHOCKING2: case LCF_MODEL_COMPANION_SHOCKING3:
        case LCF_MODEL_BLACKBODY:
            break;
        default:
            return fail(LCF_ERR_UNSUPPORTED, "unknown or unsupported model id");
    }
    static const int kNPar[9] = {0, 5, 4, 7, 5, 8, 7, 7, 2};
    if (pr->n_par != kNPar[pr->model]) return fail(LCF_ERR_INVALID_ARGUMENT, "n_par does not match the model");
    const int n_dim = pr->n_par + (pr->use_sigma ? 1 : 0);
    if (n_dim > kMaxDim) return fail(LCF_ERR_INVALID_ARGUMENT, "too many parameters");
    if (pr->n_points < 0 || pr->n_points > (1 << 26) || pr->n_filters <= 0)
        return fail(LCF_ERR_INVALID_ARGUMENT, "bad n_points / n_filters");
    if (pr->n_points > 0 && (!pr->t || !pr->y || !pr->dy || !pr->filt_idx))
        return fail(LCF_ERR_INVALID_ARGUMENT, "null photometry");
    if (!pr->tab_off || !pr->tab_a || !pr->tab_w) return fail(LCF_ERR_INVALID_ARGUMENT, "null band tables");
    const bool reddened = pr->model == LCF_MODEL_SHOCK_COOLING3;
    if (reddened && !pr->tab_ext) return fail(LCF_ERR_INVALID_ARGUMENT, "ShockCooling3 needs tab_ext");
    if (pr->sigma_type != LCF_SIGMA_RELATIVE && pr->sigma_type != LCF_SIGMA_ABSOLUTE)
        return fail(LCF_ERR_INVALID_ARGUMENT, "sigma_type must be relative or absolute");
    const int N = (int)pr->n_points, NF = pr->n_filters;
    if (pr->tab_off[0] != 0) return fail(LCF_ERR_INVALID_ARGUMENT, "tab_off[0] must be 0");
    for (int f = 0; f < NF; ++f)
        if (pr->tab_off[f + 1] < pr->tab_off[f]) return fail(LCF_ERR_INVALID_ARGUMENT, "tab_off must be non-decreasing");
    for (int i = 0; i < N; ++i)
        if (pr->filt_idx[i] < 0 || pr->filt_idx[i] >= NF) return fail(LCF_ERR_INVALID_ARGUMENT, "filt_idx out of range");
    const bool companion = pr->model >= LCF_MODEL_COMPANION_SHOCKING && pr->model <= LCF_MODEL_COMPANION_SHOCKING3;
    if (companion) {
        if (!pr->filt_kasen_par || !pr->filt_sifto_par || !pr->filt_dt_par || !pr->spline_knots || !pr->spline_coef ||
            pr->n_knots < 2)
            return fail(LCF_ERR_INVALID_ARGUMENT, "companion-shocking model needs spline and factor tables");
        for (int f = 0; f < NF; ++f)
            for (const int32_t* a : {pr->filt_kasen_par, pr->filt_sifto_par, pr->filt_dt_par})
                if (a[f] < -1 || a[f] >= n_dim) return fail(LCF_ERR_INVALID_ARGUMENT, "factor parameter index out of range");
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(LCF_ERR_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(LCF_ERR_INVALID_ARGUMENT, "device index out of range");
    LCF_HIP(hipSetDevice(device));

    auto* e = new lcf_engine();
    e->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->n_cus = cus;
    }
    lcf_status st = LCF_OK;
    auto bail = [&](lcf_status s) {
        delete e;
        return s;
    };
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(LCF_ERR_HIP, "hipStreamCreate failed"));

    // ---- order the points: n_parts contiguous ranges of observation epochs ("parts", one workgroup per walker
    // each), sorted by filter inside a part (waves then share a band table).  A part owning whole epochs means its
    // workgroup needs the thermal states of those epochs only. ----
    // distinct observation times (exact equality): the thermal state depends on (walker, time) only
    const bool all_finite_t = std::all_of(pr->t, pr->t + N, [](double v) { return std::isfinite(v); });
    // (with a non-finite time there is no strict weak order to sort by: every point is then its own epoch, in the
    // caller's order, and nothing below compares times)
    std::vector<double> epochs(pr->t, pr->t + N);
    if (all_finite_t) {
        std::sort(epochs.begin(), epochs.end());
        epochs.erase(std::unique(epochs.begin(), epochs.end()), epochs.end());
    }
    const int n_chunks_all = std::max(1, (N + kBlock - 1) / kBlock);
    std::vector<int> order(N);
    std::iota(order.begin(), order.end(), 0);
    auto epoch_of = [&](int i) {
        return all_finite_t ? (int)(std::lower_bound(epochs.begin(), epochs.end(), pr->t[i]) - epochs.begin()) : i;
    };
    std::vector<int> ep_of(N);
    for (int i = 0; i < N; ++i) ep_of[i] = epoch_of(i);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ep_of[a] < ep_of[b]; });
    // Thermal states per (walker, observation time) ahead of the points.  With >= 2 points per distinct time that is
    // sharing; without any (real multi-band photometry: every observation has its own time) it still is what puts a
    // light curve on the fast path -- log-space states, interpolated band sums, k_solo.  LCF_SHARED_EPOCHS_ONLY=1
    // restores the round-1 rule, sharing or nothing (read at every engine creation: the tests of the
    // state-inside-the-point-loop kernels set it).
    const bool shared_only = std::getenv("LCF_SHARED_EPOCHS_ONLY") != nullptr;
    const bool sharing = 2 * (long long)epochs.size() <= N;
    const bool epochs_ahead = all_finite_t && N > 0 && (sharing || !shared_only);
    // ---- the epoch-major copy the likelihood walks (DevProblem::em_*): COLUMNS = an observation time with up to em_k of
    // its points in filter order; an epoch with more points than that takes several columns (em_k = the most points of
    // an epoch, but at most twice the mean, so that ragged photometry does not pad the arrays beyond ~3 N entries) ----
    int em_k = 1, em_dense = 0;
    std::vector<int> col_epoch, col_first;   // per column: its epoch, its first point in em_order
    std::vector<int> em_order;               // points by (epoch, filter, caller's index)
    std::vector<int> ep_col0;                // first column of every epoch (+ one past the last)
    if (epochs_ahead) {
        em_order = order;
        std::stable_sort(em_order.begin(), em_order.end(), [&](int a, int b) {
            return ep_of[a] != ep_of[b] ? ep_of[a] < ep_of[b] : pr->filt_idx[a] < pr->filt_idx[b];
        });
        const int n_ep = (int)epochs.size();
        std::vector<int> cnt(n_ep, 0);
        for (int i = 0; i < N; ++i) ++cnt[ep_of[i]];
        const int kmax = *std::max_element(cnt.begin(), cnt.end());
        em_k = std::min(kmax, std::max(2, (int)((2LL * N + n_ep - 1) / n_ep)));
        ep_col0.assign(n_ep + 1, 0);
        for (int e = 0, at = 0; e < n_ep; ++e) {
            ep_col0[e] = (int)col_epoch.size();
            for (int k = 0; k < cnt[e]; k += em_k) {
                col_epoch.push_back(e);
                col_first.push_back(at + k);
            }
            at += cnt[e];
        }
        ep_col0[n_ep] = (int)col_epoch.size();
        em_dense = em_k == NF && (long long)n_ep * NF == N;
        for (int i = 0; em_dense && i < N; ++i)   // every epoch holds filters 0 .. NF-1, one point each
            if (pr->filt_idx[em_order[i]] != i % NF) em_dense = 0;
    }
    const int n_cols = (int)col_epoch.size();
    // workgroups per walker ("parts"): every one repeats the prologue (table staging; in the fused sampler kernel also
    // the serial part of the half-step), so few of them.  Epoch-major engines: a part is what 256 lanes walk, one
    // column each and round -- one part per 256 columns; else two parts up to 16 chunks of points, then one per 8
    // (measured on the 1024-walker, 3000-point fit in round 1: 2 parts 52 us/step, 4 parts 61, 8 parts 83)
    int n_parts = std::min(n_chunks_all, std::min(kMaxParts, std::max(2, (n_chunks_all + 7) / 8)));
    if (epochs_ahead) {
        // ... or, beyond kMaxParts x 256 columns, one part per 256 r columns with the smallest r that fits: a part's
        // 256 lanes then make r full rounds (3000 single-point columns: 6 parts of 500, not 8 of 375 whose second round
        // is half empty)
        int rounds = 1;
        while ((n_cols + kBlock * rounds - 1) / (kBlock * rounds) > kMaxParts) ++rounds;
        n_parts = std::max(n_cols > 64 ? 2 : 1, (n_cols + kBlock * rounds - 1) / (kBlock * rounds));
    }
    if (const char* env = std::getenv("LCF_PARTS"))
        n_parts = std::max(1, std::min(std::min(epochs_ahead ? n_cols : n_chunks_all, kMaxParts), std::atoi(env)));
    std::vector<int> part_start(kMaxParts + 1, N), part_col0(kMaxParts + 1, n_cols);
    part_start[0] = 0;
    part_col0[0] = 0;
    {
        int made = 0;
        for (int j = 1; j < n_parts; ++j) {
            int b;  // boundary j: the first epoch change at or after the equal split (of the columns / of the points)
            if (epochs_ahead) {
                int cb = (int)((long long)n_cols * j / n_parts);
                while (cb < n_cols && cb > 0 && col_epoch[cb] == col_epoch[cb - 1]) ++cb;
                b = cb < n_cols ? col_first[cb] : N;   // (`order` and em_order agree on where an epoch starts)
                if (b > part_start[made] && b < N) part_col0[made + 1] = cb;
            } else {
                b = (int)((long long)N * j / n_parts);
                while (b < N && b > 0 && ep_of[order[b]] == ep_of[order[b - 1]]) ++b;
            }
            if (b > part_start[made] && b < N) part_start[++made] = b;
        }
        n_parts = made + 1;
        for (int j = n_parts; j <= kMaxParts; ++j) {
            part_start[j] = N;
            part_col0[j] = n_cols;
        }
    }
    for (int j = 0; j < n_parts; ++j)
        std::stable_sort(order.begin() + part_start[j], order.begin() + part_start[j + 1],
                         [&](int a, int b) { return pr->filt_idx[a] < pr->filt_idx[b]; });
    std::vector<double> ht(N), hy(N), hdy(N);
    std::vector<int> hfilt(N), horig(N);
    double lognorm = 0.;
    int64_t samples = 0;
    // device table: per filter [full | compressed], each padded to a multiple of four samples with zero weights
    // (exactly 0 contribution)
    // (a reddened model reweights the samples per walker: the compressed tables do not apply)
    const bool have_ctab = pr->ctab_off && pr->ctab_a && pr->ctab_w && pr->ctab_tmin && !reddened;
    const bool have_htab = have_ctab && pr->htab_off && pr->htab_a && pr->htab_w && pr->htab_tmin;
    const bool have_itab = !reddened && pr->itab_coef && pr->itab_tmin && pr->itab_m > 0 && pr->itab_h > 0.;
    if (have_itab) {
        if (pr->itab_m > 4096 || !(pr->itab_u0 > 0.) || !std::isfinite(pr->itab_h))
            return bail(fail(LCF_ERR_INVALID_ARGUMENT, "interpolants: 0 < itab_m <= 4096, itab_u0 > 0 (T >= 1 kK), finite itab_h"));
        for (int f = 0; f < NF; ++f) {
            if (!(pr->itab_tmin[f] > 0.)) return bail(fail(LCF_ERR_INVALID_ARGUMENT, "itab_tmin must be > 0 (+inf: none)"));
            if (std::isfinite(pr->itab_tmin[f]))
                for (int k = 0; k < pr->itab_m * 8; ++k)
                    if (!std::isfinite(pr->itab_coef[(size_t)f * pr->itab_m * 8 + k]))
                        return bail(fail(LCF_ERR_INVALID_ARGUMENT, "non-finite interpolant coefficient"));
        }
    }
    if (have_htab) {
        if (pr->htab_off[0] != 0) return bail(fail(LCF_ERR_INVALID_ARGUMENT, "htab_off[0] must be 0"));
        for (int f = 0; f < NF; ++f)
            if (pr->htab_off[f + 1] < pr->htab_off[f] || pr->htab_off[f + 1] - pr->htab_off[f] > 252)
                return bail(fail(LCF_ERR_INVALID_ARGUMENT, "htab_off must be non-decreasing, <= 252 samples a filter"));
    }
    if (have_ctab) {
        if (pr->ctab_off[0] != 0) return bail(fail(LCF_ERR_INVALID_ARGUMENT, "ctab_off[0] must be 0"));
        for (int f = 0; f < NF; ++f)
            if (pr->ctab_off[f + 1] < pr->ctab_off[f])
                return bail(fail(LCF_ERR_INVALID_ARGUMENT, "ctab_off must be non-decreasing"));
    }
    std::vector<int> pfull(2 * NF, 0), pcomp(2 * NF, 0), phot(2 * NF, 0);  // (offset, count) pairs
    std::vector<double> ptmin(NF, INFINITY), ptmin2(NF, INFINITY);
    std::vector<double2> htab;
    std::vector<double> hext;  // reddened models: 0.4 log2(10) A_k / E(B-V), aligned with htab
    auto append = [&](const double* a, const double* w, int k0, int k1, int* slot) -> bool {
        slot[0] = (int)htab.size();
        for (int k = k0; k < k1; ++k) {
            if (!(a[k] > 0.) || !std::isfinite(a[k]) || !std::isfinite(w[k])) return false;
            htab.push_back(make_double2(a[k], w[k]));
            if (reddened) {
                if (!std::isfinite(pr->tab_ext[k])) return false;
                hext.push_back(0.4 * 3.321928094887362 * pr->tab_ext[k]);
            }
        }
        const double apad = k1 > k0 ? a[k1 - 1] : 1.;
        while ((htab.size() - slot[0]) % 4) {
            htab.push_back(make_double2(apad, 0.));
            if (reddened) hext.push_back(0.);
        }
        slot[1] = (int)htab.size() - slot[0];
        return true;
    };
    // device table = [compressed levels of every filter | full tables of every filter]: when everything does not fit
    // in LDS the (short) compressed part still does, and only points colder than its validity read global memory
    for (int f = 0; f < NF; ++f) {
        bool ok = true;
        if (ok && have_ctab && pr->ctab_off[f + 1] > pr->ctab_off[f]) {
            ok = append(pr->ctab_a, pr->ctab_w, pr->ctab_off[f], pr->ctab_off[f + 1], &pcomp[2 * f]);
            ptmin[f] = pr->ctab_tmin[f];
            if (!(ptmin[f] >= 0.)) ok = false;
        }
        if (ok && have_htab && pr->htab_off[f + 1] > pr->htab_off[f]) {
            ok = append(pr->htab_a, pr->htab_w, pr->htab_off[f], pr->htab_off[f + 1], &phot[2 * f]);
            ptmin2[f] = pr->htab_tmin[f];
            // the hot level must not claim a wider range than the cool one (selection tests it first)
            if (!(ptmin2[f] >= 0.) || (pcomp[2 * f + 1] > 0 && ptmin2[f] < ptmin[f])) ok = false;
        }
        if (!ok)
            return bail(fail(LCF_ERR_INVALID_ARGUMENT,
                             "band tables need finite a_k > 0, finite W_k, t_min >= 0 (hot level: t_min >= the cool one's)"));
    }
    const int n_compressed = (int)htab.size();
    for (int f = 0; f < NF; ++f)
        if (!append(pr->tab_a, pr->tab_w, pr->tab_off[f], pr->tab_off[f + 1], &pfull[2 * f]))
            return bail(fail(LCF_ERR_INVALID_ARGUMENT, "band tables need finite a_k > 0 and finite W_k"));
    for (int i = 0; i < N; ++i) {
        const int o = order[i], f = pr->filt_idx[o];
        ht[i] = pr->t[o];
        hy[i] = pr->y[o];
        hdy[i] = pr->dy[o];
        hfilt[i] = f;
        horig[i] = o;
        samples += pr->tab_off[f + 1] - pr->tab_off[f];
    }
    for (int i = 0; i < N; ++i) lognorm += std::log(2. * M_PI * pr->dy[i] * pr->dy[i]);  // caller order, like np.sum
    std::vector<double> sorted_dy(pr->dy, pr->dy + N);
    double med = 0.;
    if (N > 0) {  // np.median
        std::sort(sorted_dy.begin(), sorted_dy.end());
        med = (N & 1) ? sorted_dy[N / 2] : 0.5 * (sorted_dy[N / 2 - 1] + sorted_dy[N / 2]);
    }
    std::vector<int> hepoch(N);
    for (int i = 0; i < N; ++i) hepoch[i] = ep_of[horig[i]];
    int n_chunks = 0, cpb = 1;  // chunks of kBlock points: total over the parts, and the most in one part
    for (int j = 0; j < n_parts; ++j) {
        const int c = (part_start[j + 1] - part_start[j] + kBlock - 1) / kBlock;
        n_chunks += c;
        cpb = std::max(cpb, c);
    }
    std::vector<int> htaboff(pfull);
    if (htab.empty()) htab.push_back(make_double2(1., 0.));
    std::vector<double> hinvdy(N);
    for (int i = 0; i < N; ++i) hinvdy[i] = 1. / hdy[i];
    std::vector<FiltDesc> hfd(NF);  // per filter: where its tables are and from which temperature each is valid
    for (int f = 0; f < NF; ++f)
        hfd[f] = FiltDesc{pfull[2 * f], pfull[2 * f + 1], pcomp[2 * f], pcomp[2 * f + 1], phot[2 * f], phot[2 * f + 1],
                          // the interpolant holds from max(its proved t_min, the table's first interval)
                          (have_itab && std::isfinite(pr->itab_tmin[f]))
                              ? std::nextafter((float)std::max((std::log(pr->itab_tmin[f]) - pr->itab_u0) / pr->itab_h, 0.),
                                               INFINITY)
                              : INFINITY,
                          have_itab ? f * pr->itab_m * 8 : 0,
                          pcomp[2 * f + 1] > 0 ? 1. / ptmin[f] : 0.,  // t_min = 0 -> inf: always valid
                          phot[2 * f + 1] > 0 ? 1. / ptmin2[f] : 0.,
                          companion ? pr->filt_kasen_par[f] : -1, companion ? pr->filt_sifto_par[f] : -1,
                          companion ? pr->filt_dt_par[f] : -1, 0};
    std::vector<double> hexp(kExpTabSize);
    for (int j = 0; j < kExpTabSize; ++j) hexp[j] = std::exp2(j / (double)kExpTabSize);

    DevProblem& dp = e->dp;
    dp.model = pr->model;
    dp.n_points = N;
    dp.n_chunks = n_chunks;
    dp.n_parts = n_parts;
    dp.cpb = cpb;
    for (int j = 0; j <= kMaxParts; ++j) dp.part_start[j] = part_start[j];
    for (int j = 0; j <= kMaxParts; ++j)  // epochs were sorted, so a part's epoch range starts at its first point's
        dp.part_ep0[j] = (j < n_parts && part_start[j] < N && all_finite_t) ? ep_of[order[part_start[j]]]
                                                                              : (int)epochs.size();
    if (all_finite_t && N > 0) {
        // `order` is filter-sorted inside a part: the part's first epoch is the minimum over its points
        for (int j = 0; j < n_parts; ++j) {
            int lo = (int)epochs.size();
            for (int i = part_start[j]; i < part_start[j + 1]; ++i) lo = std::min(lo, hepoch[i]);
            dp.part_ep0[j] = lo;
        }
    }
    dp.n_tab = (int)htab.size();
    dp.n_filters = NF;
    dp.n_dim = n_dim;
    dp.n_par = pr->n_par;
    dp.use_sigma = pr->use_sigma ? 1 : 0;
    dp.sigma_abs = pr->sigma_type == LCF_SIGMA_ABSOLUTE;
    dp.n_knots = companion ? pr->n_knots : 0;
    dp.has_priors = pr->priors ? 1 : 0;
    // LDS holds the first n_lds_tab samples of the table array: all of it, or the compressed levels only
    int lds_tab_max = kLdsTabMax;
    if (const char* env = std::getenv("LCF_LDS_TAB_MAX")) lds_tab_max = std::max(0, std::min(9500, std::atoi(env)));
    dp.n_lds_tab = NF > kLdsFiltMax ? 0
                   : (int)htab.size() <= lds_tab_max ? (int)htab.size()
                   : (n_compressed > 0 && n_compressed <= lds_tab_max) ? n_compressed : 0;
    // reddened weights are made while staging the FULL tables; when those do not fit nothing is staged and the
    // reddening is applied on the fly (slow path)
    dp.redden_slow = reddened && dp.n_lds_tab != (int)htab.size();
    if (dp.redden_slow) dp.n_lds_tab = 0;
    dp.tab_in_lds = dp.n_lds_tab > 0;
    dp.n_epochs = (int)epochs.size();
    dp.use_therm = epochs_ahead;
    dp.em_k = em_k;
    dp.em_cols = n_cols;
    dp.em_dense = em_dense;
    for (int j = 0; j <= kMaxParts; ++j) dp.part_col0[j] = part_col0[j];
    dp.variant = 1;
    dp.use_ctab = have_ctab ? 1 : 0;
    e->have_ctab = have_ctab;
    // third level: interpolants of ln S(ln T), staged in LDS behind the descriptors when they take <= 40 KiB
    // (six to ten filters), else read from global memory (L2)
    // (engines whose points share no epochs keep the thermal state inside the point loop, in linear space)
    const bool have_itab_here = have_itab && epochs_ahead;
    e->have_itab = have_itab_here;
    dp.use_itab = have_itab_here ? 1 : 0;
    dp.itab_m = have_itab ? pr->itab_m : 0;
    dp.itab_u0 = have_itab ? pr->itab_u0 : 0.;
    dp.itab_inv_h = have_itab ? 1. / pr->itab_h : 0.;
    dp.itab_umax = have_itab ? pr->itab_u0 + pr->itab_h * pr->itab_m : 0.;
    const size_t n_itab = have_itab_here ? (size_t)NF * pr->itab_m * 8 : 0;
    // ... and only where a workgroup walks enough points to pay for staging them (a part of >= 1024 points; the
    // population launches of 600-point transients read the 64 bytes a point needs from L2 instead: staging 24 KiB per
    // workgroup cost them 20 %)
    dp.n_itab_lds = (dp.tab_in_lds && n_itab * sizeof(double) <= 40 * 1024 && N >= 1024 * n_parts) ? (int)n_itab : 0;
    dp.itab_uniform = have_itab ? 1 : 0;
    for (int f = 0; have_itab && f < NF; ++f)
        if (!(pr->itab_tmin[f] <= std::exp(pr->itab_u0) * (1. + 1e-12))) dp.itab_uniform = 0;
    // ... and a companion-shocking model's template splines behind those, when their knots are exactly k0 + i h (the
    // device then needs no knot) and everything still leaves room for two workgroups per CU (80 KiB each): 26 KiB for
    // the eight SiFTO filters -- a point's template costs two LDS reads instead of a chain of dependent loads from L2
    dp.n_spl_lds = 0;
    if (companion && dp.tab_in_lds) {
        const double h1 = pr->spline_knots[1] - pr->spline_knots[0];
        bool exact = h1 > 0.;
        for (int k = 0; exact && k < pr->n_knots; ++k) exact = pr->spline_knots[k] == pr->spline_knots[0] + k * h1;
        const size_t n_spl = (size_t)NF * (pr->n_knots - 1) * 4;
        const size_t before = kLdsHead * sizeof(double) + ((size_t)dp.n_lds_tab + kFdD2 * NF) * sizeof(double2) +
                              (size_t)dp.n_itab_lds * sizeof(double);
        if (exact && before + n_spl * sizeof(double) + 2048 <= 80 * 1024) dp.n_spl_lds = (int)n_spl;
    }
    dp.stage_d2 = dp.n_lds_tab + kFdD2 * NF + dp.n_itab_lds / 2 + dp.n_spl_lds / 2;
    std::memcpy(dp.consts, pr->consts, sizeof(dp.consts));
    if (pr->model == LCF_MODEL_SHOCK_COOLING || pr->model == LCF_MODEL_SHOCK_COOLING3)
        dp.consts[11] = pr->consts[1] > 0. ? std::log(pr->consts[1] / 19.5) : 0.;  // hoisted out of the half-step
    if (pr->model == LCF_MODEL_SHOCK_COOLING) {  // logarithms of the two amplitudes' constant factors (log-space state)
        dp.consts[9] = std::log(pr->consts[6] * pr->consts[7] / kKB);
        dp.consts[10] = std::log(kC3sq * pr->consts[5] * pr->consts[0]);
    }
    dp.log_norm_const = lognorm;
    dp.sigma_unit_abs = med;
    e->samples_per_eval = samples * (pr->model == LCF_MODEL_SHOCK_COOLING4 ? 2 : 1);
    e->lds_bytes = kLdsHead * sizeof(double) + (dp.tab_in_lds ? (size_t)dp.stage_d2 * sizeof(double2) : 0);

    double *dt, *dy_, *ddy, *dkn = nullptr, *dspl = nullptr;
    int *dfilt, *dorig;
    double2* dtab;
    PriorDev* dpri = nullptr;
    // (one block, one copy: the photometry in its layouts, the tables and the staging image; sized for the light curve)
    UploadArena arena(e->owned, (size_t)256 * 1024 + (size_t)N * 256);
#define UP(h, d) if ((st = upload(h, &d, arena)) != LCF_OK) return bail(st)
    UP(ht, dt); UP(hy, dy_); UP(hdy, ddy); UP(hfilt, dfilt); UP(horig, dorig);
    UP(htab, dtab); UP(htaboff, e->d_tab_off);
    if (reddened) {
        double* dext;
        UP(hext, dext);
        dp.tab_ext = dext;
    }
    if (have_itab) {
        std::vector<double> hit(pr->itab_coef, pr->itab_coef + (size_t)NF * pr->itab_m * 8);
        for (double& v : hit)
            if (!std::isfinite(v)) v = 0.;  // (filters without an interpolant: never read, u_min = +inf)
        double* dit;
        UP(hit, dit);
        dp.itab = dit;
    }
    int* depoch;
    FiltDesc* dfd;
    double *dexp, *depocht, *dinvdy;
    UP(hexp, dexp); UP(hepoch, depoch); UP(epochs, depocht);
    {   // the image of the kernels' staged LDS (see kLdsHead / stage_tables)
        std::vector<double2> himg(kLdsHead / 2, make_double2(0., 0.));
        std::memcpy(himg.data(), hexp.data(), kExpTabSize * sizeof(double));
        if (dp.tab_in_lds) {
            himg.insert(himg.end(), htab.begin(), htab.begin() + dp.n_lds_tab);
            const double2* fd2 = reinterpret_cast<const double2*>(hfd.data());
            himg.insert(himg.end(), fd2, fd2 + kFdD2 * NF);
            for (int k = 0; k < dp.n_itab_lds; k += 2) {
                const double a = pr->itab_coef[k], b = pr->itab_coef[k + 1];
                himg.push_back(make_double2(std::isfinite(a) ? a : 0., std::isfinite(b) ? b : 0.));
            }
            for (int k = 0; k < dp.n_spl_lds; k += 2) himg.push_back(make_double2(pr->spline_coef[k], pr->spline_coef[k + 1]));
        }
        double2* dimg;
        UP(himg, dimg);
        dp.stage_image = dimg;
        dp.stage_n16 = (int)himg.size();
    }
    {   // observation and the factor its residual is scaled with, side by side
        std::vector<double2> hyd(N);
        for (int i = 0; i < N; ++i) hyd[i] = make_double2(hy[i], pr->use_sigma ? hdy[i] : hinvdy[i]);
        double2* dyd;
        UP(hyd, dyd);
        dp.pt_yd = dyd;
    }
    if (epochs_ahead) {   // the epoch-major copy: [em_k][n_cols], a column's points in filter order
        std::vector<double> hct(n_cols);
        std::vector<double2> hcyd((size_t)em_k * n_cols, make_double2(0., 0.));
        std::vector<int> hcf((size_t)em_k * n_cols, -1);
        for (int cidx = 0; cidx < n_cols; ++cidx) {
            const int ep = col_epoch[cidx];
            hct[cidx] = epochs[ep];
            for (int k = 0; k < em_k; ++k) {
                const int at = col_first[cidx] + k;
                if (at >= N || ep_of[em_order[at]] != ep) break;
                const int o = em_order[at];
                hcyd[(size_t)k * n_cols + cidx] = make_double2(pr->y[o], pr->use_sigma ? pr->dy[o] : 1. / pr->dy[o]);
                hcf[(size_t)k * n_cols + cidx] = pr->filt_idx[o];
            }
        }
        double* dct;
        double2* dcyd;
        int* dcf;
        UP(hct, dct); UP(hcyd, dcyd); UP(hcf, dcf);
        dp.em_t = dct;
        dp.em_yd = dcyd;
        dp.em_filt = dcf;
    }
    if (NF <= 64) {  // filter and epoch of a point in one word (epochs < 2^26 = the most points an engine takes)
        std::vector<int> hfe(N);
        for (int i = 0; i < N; ++i) hfe[i] = (int)((unsigned int)hfilt[i] | ((unsigned int)hepoch[i] << 6));
        int* dfe;
        UP(hfe, dfe);
        dp.pt_fe = dfe;
    }
    UP(hfd, dfd); UP(hinvdy, dinvdy); UP(pcomp, e->d_ctab_off); UP(ptmin, e->d_ctmin);
    dp.f_desc = dfd;
    dp.inv_dy = dinvdy;
    dp.pt_epoch = depoch;
    dp.epoch_t = depocht;
    dp.exp2tab = dexp;
    if (companion) {
        std::vector<double> hkn(pr->spline_knots, pr->spline_knots + pr->n_knots),
            hsp(pr->spline_coef, pr->spline_coef + (size_t)NF * (pr->n_knots - 1) * 4);
        for (int k = 1; k < pr->n_knots; ++k)
            if (!(hkn[k] > hkn[k - 1])) return bail(fail(LCF_ERR_INVALID_ARGUMENT, "spline knots must ascend"));
        const double h = (hkn.back() - hkn.front()) / (pr->n_knots - 1);
        bool uniform = true;
        for (int k = 0; k < pr->n_knots; ++k)
            uniform = uniform && std::fabs(hkn[k] - (hkn.front() + k * h)) <= 1e-9 * h;
        dp.knot_inv_h = uniform ? 1. / h : 0.;
        // knots that ARE k0 + i h in float64 (SiFTO: integer days): interval by arithmetic, no knot read on the device
        const double h1 = hkn[1] - hkn[0];
        bool exact = pr->n_knots >= 2 && h1 > 0.;
        for (int k = 0; exact && k < pr->n_knots; ++k) exact = hkn[k] == hkn[0] + k * h1;
        dp.knot0 = hkn.front();
        dp.knot_last = hkn.back();
        dp.knot_h = exact ? h1 : 0.;
        if (exact) dp.knot_inv_h = 1. / h1;
        UP(hkn, dkn); UP(hsp, dspl);
    }
    if (pr->priors) {
        std::vector<PriorDev> hp(n_dim);
        for (int i = 0; i < n_dim; ++i) {
            if (pr->priors[i].kind < 0 || pr->priors[i].kind > 2) return bail(fail(LCF_ERR_INVALID_ARGUMENT, "bad prior kind"));
            hp[i] = PriorDev{pr->priors[i].kind, 0, pr->priors[i].p_min, pr->priors[i].p_max, pr->priors[i].mean,
                             pr->priors[i].stddev};
        }
        UP(hp, dpri);
    }
#undef UP
    dp.t = dt; dp.y = dy_; dp.dy = ddy; dp.pt_filt = dfilt; dp.pt_orig = dorig;
    dp.tab = dtab;
    dp.knots = dkn; dp.spl = dspl; dp.priors = dpri;
    if ((st = arena.flush()) != LCF_OK) return bail(st);
    if ((st = e->sync_dp()) != LCF_OK) return bail(st);
    *out = e;
    return LCF_OK;
}

void lcf_engine_destroy(lcf_engine* e) { delete e; }
int32_t lcf_engine_ndim(const lcf_engine* e) { return e ? e->dp.n_dim : 0; }
int64_t lcf_engine_npoints(const lcf_engine* e) { return e ? e->dp.n_points : 0; }
int64_t lcf_engine_samples_per_eval(const lcf_engine* e) { return e ? e->samples_per_eval : 0; }

lcf_status lcf_engine_set_variant(lcf_engine* e, int32_t variant) {
    if (!e || variant < 0 || variant > 3) return fail(LCF_ERR_INVALID_ARGUMENT, "variant must be 0, 1, 2 or 3");
    if (variant >= 2 && !e->have_ctab) return fail(LCF_ERR_INVALID_ARGUMENT, "no compressed tables were given");
    if (variant == 3 && !e->have_itab) return fail(LCF_ERR_INVALID_ARGUMENT, "no interpolants were given");
    e->dp.variant = variant == 0 ? 0 : 1;
    e->dp.use_ctab = variant >= 2 ? 1 : 0;
    e->dp.use_itab = variant == 3 ? 1 : 0;
    LCF_HIP(hipSetDevice(e->device));
    LCF_HIP(hipStreamSynchronize(e->stream));   // (launches in flight read the old copy)
    return e->sync_dp();
}

lcf_status lcf_log_likelihood(lcf_engine* e, int64_t n, const double* P, double* out) {
    return logprob_host(e, n, P, out, 0);
}
lcf_status lcf_log_posterior(lcf_engine* e, int64_t n, const double* P, double* out) {
    return logprob_host(e, n, P, out, 1);
}
lcf_status lcf_log_likelihood_dev(lcf_engine* e, int64_t n, const double* dP, double* dout, void* stream) {
    if (!e || n < 0 || (n > 0 && (!dP || !dout))) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    LCF_HIP(hipSetDevice(e->device));
    if (lcf_status st = e->reserve(n)) return st;
    return logprob_dev(e, n, dP, dout, stream ? (hipStream_t)stream : e->stream, 0);
}
lcf_status lcf_log_posterior_dev(lcf_engine* e, int64_t n, const double* dP, double* dout, void* stream) {
    if (!e || n < 0 || (n > 0 && (!dP || !dout))) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    LCF_HIP(hipSetDevice(e->device));
    if (lcf_status st = e->reserve(n)) return st;
    return logprob_dev(e, n, dP, dout, stream ? (hipStream_t)stream : e->stream, 1);
}

static lcf_status evaluate_impl(lcf_engine* e, int64_t n, const double* P, double* o0, double* o1, int mode) {
    if (!e || n < 0 || (n > 0 && (!P || !o0 || (mode == 2 && !o1)))) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0 || e->dp.n_points == 0) return LCF_OK;
    LCF_HIP(hipSetDevice(e->device));
    if (lcf_status st = e->reserve(n)) return st;
    const size_t N = e->dp.n_points;
    // bound the device scratch to ~256 MiB per pass
    const int64_t per = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)((256u << 20) / (N * sizeof(double) * (mode == 2 ? 2 : 1)))));
    if (lcf_status st = e->reserve_big(per * N * sizeof(double) * (mode == 2 ? 2 : 1))) return st;
    LCF_HIP(hipMemcpyAsync(e->wP, P, n * e->dp.n_dim * sizeof(double), hipMemcpyHostToDevice, e->stream));
    const int bs = 128;
    hipLaunchKernelGGL(k_prepare, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, e->stream, e->dp, (int)n, e->wP,
                       e->wcoef, e->wlprior, 0);
    for (int64_t lo = 0; lo < n; lo += per) {
        const int64_t m = std::min(per, n - lo);
        double* b0 = e->wbig;
        double* b1 = e->wbig + per * N;
        if (mode == 1)
            launch_points<1>(e, (int)lo, (int)m, e->wP, e->wcoef, e->wlprior, e->wtherm, b0, nullptr, e->stream);
        else
            launch_points<2>(e, (int)lo, (int)m, e->wP, e->wcoef, e->wlprior, e->wtherm, b0, b1, e->stream);
        LCF_HIP(hipGetLastError());
        LCF_HIP(hipMemcpyAsync(o0 + lo * N, b0, m * N * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        if (mode == 2) LCF_HIP(hipMemcpyAsync(o1 + lo * N, b1, m * N * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        LCF_HIP(hipStreamSynchronize(e->stream));
    }
    return LCF_OK;
}

lcf_status lcf_model_evaluate(lcf_engine* e, int64_t n, const double* P, double* y_fit) {
    return evaluate_impl(e, n, P, y_fit, nullptr, 1);
}
lcf_status lcf_temperature_radius(lcf_engine* e, int64_t n, const double* P, double* T_K, double* R_bb) {
    return evaluate_impl(e, n, P, T_K, R_bb, 2);
}

lcf_status lcf_blackbody_to_filters(lcf_engine* e, int64_t m, const int32_t* filt_idx, const double* T, const double* R,
                                    double* out) {
    if (!e || m < 0 || (m > 0 && (!filt_idx || !T || !R || !out))) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (m == 0) return LCF_OK;
    for (int64_t i = 0; i < m; ++i)
        if (filt_idx[i] < 0 || filt_idx[i] >= e->dp.n_filters) return fail(LCF_ERR_INVALID_ARGUMENT, "filt_idx out of range");
    LCF_HIP(hipSetDevice(e->device));
    const size_t bytes = m * (3 * sizeof(double) + sizeof(int));
    if (lcf_status st = e->reserve_big(bytes + 64)) return st;
    double* dT = e->wbig;
    double* dR = dT + m;
    double* dO = dR + m;
    int* dF = reinterpret_cast<int*>(dO + m);
    LCF_HIP(hipMemcpyAsync(dT, T, m * sizeof(double), hipMemcpyHostToDevice, e->stream));
    LCF_HIP(hipMemcpyAsync(dR, R, m * sizeof(double), hipMemcpyHostToDevice, e->stream));
    LCF_HIP(hipMemcpyAsync(dF, filt_idx, m * sizeof(int), hipMemcpyHostToDevice, e->stream));
    const dim3 grid((unsigned)((m + kBlock - 1) / kBlock));
    if (e->dp.variant == 0)
        hipLaunchKernelGGL(k_bb_pointwise<0>, grid, dim3(kBlock), 0, e->stream, e->dp, (int)m, dF, e->d_tab_off, e->d_ctab_off,
                           e->d_ctmin, dT, dR, dO);
    else
        hipLaunchKernelGGL(k_bb_pointwise<1>, grid, dim3(kBlock), 0, e->stream, e->dp, (int)m, dF, e->d_tab_off, e->d_ctab_off,
                           e->d_ctmin, dT, dR, dO);
    LCF_HIP(hipGetLastError());
    LCF_HIP(hipMemcpyAsync(out, dO, m * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    LCF_HIP(hipStreamSynchronize(e->stream));
    return LCF_OK;
}

}  // extern "C"

extern "C" lcf_status lcf_profile_loglike_kernel(lcf_engine* e, int64_t n, const double* P, int32_t reps,
                                                 double* avg_ms) {
    if (!e || n <= 0 || !P || reps <= 0 || !avg_ms) return fail(LCF_ERR_INVALID_ARGUMENT, "bad argument");
    LCF_HIP(hipSetDevice(e->device));
    if (lcf_status st = e->reserve(n)) return st;
    LCF_HIP(hipMemcpyAsync(e->wP, P, n * e->dp.n_dim * sizeof(double), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_prepare, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, e->stream, e->dp, (int)n, e->wP,
                       e->wcoef, e->wlprior, 0);
    launch_points<0>(e, 0, (int)n, e->wP, e->wcoef, e->wlprior, e->wtherm, e->wpart, nullptr, e->stream);  // warm-up (+ thermal states)
    hipEvent_t a, b;
    LCF_HIP(hipEventCreate(&a));
    LCF_HIP(hipEventCreate(&b));
    LCF_HIP(hipEventRecord(a, e->stream));
    for (int r = 0; r < reps; ++r)
        launch_points<0>(e, 0, (int)n, e->wP, e->wcoef, e->wlprior, e->wtherm, e->wpart, nullptr, e->stream, false);
    LCF_HIP(hipEventRecord(b, e->stream));
    LCF_HIP(hipStreamSynchronize(e->stream));
    float ms = 0.f;
    LCF_HIP(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a);
    hipEventDestroy(b);
    LCF_HIP(hipGetLastError());
    *avg_ms = (double)ms / reps;
    return LCF_OK;
}

namespace {
// ---- memory that kernels POLL (row boards, mailboxes) is never handed back to the driver ---------------------------------
// A board that was freed (hipFree) and whose address range the driver then gave to the next sampler's board left single
// workgroups of the next launches reading the OLD contents of those addresses for as long as they polled -- rows that
// every other workgroup (and the host) could see never arrived for them, 5 s waits, once also a stale row with a valid
// tag (a wrong chain).  Reproduced deterministically by tools/debug/rows_mismatch.py once the inter-rank boards were
// megabytes (freed uncached memory recycled into the next board); gone when such memory is not freed.  So: polled
// memory goes back to a list of this process and is taken from there by the next sampler that needs the same size;
// whoever takes it clears it (stale tags of an earlier life would be valid tags of the next) before anything reads it.
struct PolledBlock { int dev; bool uncached; size_t bytes; void* p; };
std::mutex g_polled_mutex;
std::vector<PolledBlock> g_polled;

void* polled_take(int dev, bool uncached, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_polled_mutex);
    for (size_t k = 0; k < g_polled.size(); ++k)
        if (g_polled[k].dev == dev && g_polled[k].uncached == uncached && g_polled[k].bytes == bytes) {
            void* p = g_polled[k].p;
            g_polled.erase(g_polled.begin() + (long)k);
            return p;
        }
    return nullptr;
}
void polled_give(int dev, bool uncached, size_t bytes, void* p) {
    std::lock_guard<std::mutex> lock(g_polled_mutex);
    g_polled.push_back(PolledBlock{dev, uncached, bytes, p});
}
// `bytes` of polled memory on the current device, cleared (complete on return).
lcf_status polled_alloc(int dev, bool uncached, size_t bytes, void** out) {
    void* p = polled_take(dev, uncached, bytes);
    if (!p) {
        if (uncached)   // fine-grained device memory: peers' stores over the fabric and this rank's polls meet in memory
            LCF_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached));
        else
            LCF_HIP(hipMalloc(&p, bytes));
    }
    *out = p;
    LCF_HIP(hipMemset(p, 0, bytes));        // tag 0: no version / generation (half-steps are numbered from 2)
    LCF_HIP(hipDeviceSynchronize());
    return LCF_OK;
}

}  // namespace

// =================================================================================================================
// sampler
// =================================================================================================================
struct lcf_sampler {
    lcf_engine* e = nullptr;
    int device = 0;           // e->device, kept for the destructor
    DevSampler ds{};
    std::vector<void*> owned;
    double *coef = nullptr, *lprior = nullptr;
    // The state-independent draws of a run are produced in BLOCKS of steps, two buffers (block b lives in buffer b & 1):
    // device memory does not grow with the run, the first half-step starts after a short first block, and nothing on
    // the host waits for the generation.  The generation kernels go on the SAME stream as the half-steps, between two
    // of them: block b + 1 right behind the first launch of block b (the last reader of the buffer it overwrites), so
    // stream order is all the synchronisation there is.  (Measured at 1024 walkers x 2000 steps: a low- or
    // normal-priority side stream with events cost 3-4 % of the whole run however rarely it was used; the inline
    // kernels cost 40 us per 256 steps.)
    int64_t blk_first = 0, blk_steps = 0;          // steps in block 0 and in every later block
    int64_t blk_cap = 0;                           // steps a buffer holds
    int* d_perm[2] = {nullptr, nullptr};           // [blk_cap][n_walkers]
    DrawRec* d_draws[2] = {nullptr, nullptr};      // [blk_cap][2][n_half]
    int* d_slot[2] = {nullptr, nullptr};           // [1 + 2 blk_cap][n_walkers] (row 0: the half-step in front)
    int* d_perm_host = nullptr;                    // LCF_SPLIT_HOST: the caller's permutations of the whole run
    int64_t perm_host_rows = 0;
    int split_mode = LCF_SPLIT_IDENTITY;
    bool need_slots = true;                        // draw records carry the slots of the previous half-step
    int64_t blk_generated = -1;                    // last block whose generation is enqueued
    int64_t blk_current = -1;                      // block the half-steps are in
    int64_t run_first = 0, run_steps = 0;
    int64_t spec_first = -1;   // >= 0: buffer 0 holds the first block of a run starting at this step (speculated)
    int spec_mode = 0;
    bool spec_slots = false;
    int64_t chain_cap = 0;
    bool has_state = false;
    long long g_next = 2;     // global half-step counter (never reused: see lcf_sampler_begin)
    long long g_run0 = 2;     // first half-step of the current run
    bool pending = false;     // the last proposed half-step is not committed yet
    bool foreign_stream = false;  // half-steps of the current run were enqueued on a caller's stream
    int half_step_kernel = LCF_HALF_STEP_AUTO;
    int last_kernel = -1;     // what the last run's half-steps were (lcf_sampler_last_run_kernel)
    bool last_rows = false;   // ... and whether it was a row-board run (between ranks)
    long long last_launches = 0;   // launches of that kernel in the last run (lcf_sampler_last_run_launches)
    unsigned long long* mailbox = nullptr;   // this rank's peer mailbox (uncached device memory), see DevSampler
    size_t mailbox_cap = 0;
    void* board_mem = nullptr;               // this rank's row board (uncached device memory), see DevSampler
    std::vector<void*> board_opened;         // peers' boards mapped through IPC
    // One-launch runs write their final state into the other of two sets of state buffers (DevSampler::X_out ...):
    // ds.X / LP / nacc name the set that holds the state behind everything enqueued so far.
    double* alt_X = nullptr;
    double* alt_LP = nullptr;
    long long* alt_nacc = nullptr;
    bool run_off = false;                    // a one-launch run of this sampler gave up once: launches per half-step from then on
    int replay_split = 0, replay_store = 0;  // the last one-launch run, should it have to be repeated
    int64_t replay_first = 0, replay_steps = -1;
    void* run_board_mem = nullptr;           // the board of one-launch runs (k_solo_run): kRunRing versions, this GPU only
    DevSampler* d_run_image = nullptr;       // the sampler as k_solo_run reads it (run_image)
    DevSampler run_image_host{};
    bool run_image_valid = false;
    bool run_flip = false;                   // ds.X / LP / nacc name the SECOND set of state buffers
    unsigned int run_arrivals = 0;           // workgroups of resident launches enqueued so far (the board's arrivals word)
    DevSampler* d_rows_image = nullptr;      // the sampler as k_solo_run<..., RANKS> reads it (rows_image)
    DevSampler rows_image_host{};
    bool rows_image_valid = false;
    unsigned int rows_arrivals = 0;          // the same count for the resident launches of row-board runs (cleared per run)
    int run_capacity = -1;                   // workgroups of k_solo_run the device holds at once (-1: not asked yet)
    int run_capacity_wide = -1;              // ... of its 1024-thread form (NPARTS = 8)
    size_t run_board_bytes() const {
        return board_rows_bytes(kRunRing, ds.n_walkers, ds.n_dim) + (size_t)kBoardTail * sizeof(unsigned int);
    }
    size_t board_bytes() const {
        return board_rows_bytes(kRing, ds.n_walkers, ds.n_dim) + (size_t)kBoardTail * sizeof(unsigned int);
    }
    std::vector<void*> opened;               // peers' mailboxes mapped through IPC
    int peer_ranks = 0, peer_rank = 0;
    // Snapshot of (error flag, positions, log-posteriors, acceptance counts) in pinned host memory, copied behind the
    // last launch of a run: the calls that read them back after the run wait for nothing more.
    unsigned char* snap = nullptr;
    bool snap_enqueued = false, snap_valid = false;
    size_t snap_x() const { return 8; }
    size_t snap_lp() const { return snap_x() + (size_t)ds.n_walkers * ds.n_dim * sizeof(double); }
    size_t snap_acc() const { return snap_lp() + (size_t)ds.n_walkers * sizeof(double); }
    size_t snap_bytes() const { return snap_acc() + (size_t)ds.n_walkers * sizeof(long long); }
    size_t snap_alloc() const { return snap_bytes() + 2 * kSnapFlags * sizeof(unsigned int); }   // + the workgroups' error words
    unsigned int* snap_flags() const { return reinterpret_cast<unsigned int*>(snap + snap_bytes()); }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_snap = nullptr;   // behind the snapshot kernel: what a caller of a finished run waits for
    double last_ms = 0.;

    ~lcf_sampler() {
        // (the device is remembered here: a garbage collector may destroy the engine first, and nothing below needs it)
        hipSetDevice(device);
        for (void* p : owned) hipFree(p);
        if (ds.chain) hipFree(ds.chain);
        if (ds.chain_lp) hipFree(ds.chain_lp);
        free_blocks();
        for (void* p : opened) hipIpcCloseMemHandle(p);
        if (mailbox) polled_give(device, true, mailbox_cap, mailbox);
        for (void* p : board_opened) hipIpcCloseMemHandle(p);
        if (board_mem) polled_give(device, true, board_bytes(), board_mem);
        if (run_board_mem) polled_give(device, false, run_board_bytes(), run_board_mem);
        if (snap) hipHostFree(snap);
        if (d_perm_host) hipFree(d_perm_host);
        if (ev0) hipEventDestroy(ev0);
        if (ev1) hipEventDestroy(ev1);
        if (ev_snap) hipEventDestroy(ev_snap);
    }
    void free_blocks() {
        for (int b = 0; b < 2; ++b) {
            if (d_perm[b]) hipFree(d_perm[b]);
            if (d_draws[b]) hipFree(d_draws[b]);
            if (d_slot[b]) hipFree(d_slot[b]);
            d_perm[b] = nullptr;
            d_draws[b] = nullptr;
            d_slot[b] = nullptr;
        }
        blk_cap = 0;
    }
    // block of the run's step k (relative), and the block's first step / length
    int64_t block_of_step(int64_t k) const { return k < blk_first ? 0 : 1 + (k - blk_first) / blk_steps; }
    int64_t block_start(int64_t b) const { return b == 0 ? 0 : blk_first + (b - 1) * blk_steps; }
    int64_t block_len(int64_t b) const {
        return std::min(run_steps, block_start(b) + (b == 0 ? blk_first : blk_steps)) - block_start(b);
    }
    // draw records of the run's half-step `rel` (its block must be resident)
    const DrawRec* rows(long long rel) const {
        const int64_t b = block_of_step(rel / 2);
        return d_draws[b & 1] + (size_t)(rel - 2 * block_start(b)) * ds.n_half;
    }
};

namespace {

// Enqueue, on stream `gs` (the one the half-steps run on: behind the last reader of the buffer), the generation of
// `len` steps of draw records starting at absolute step `step0` into buffer `buf`.  `front`: where the slots of the
// half-step in front of the block come from (null: nothing in front, the start of a run).
lcf_status generate_steps(lcf_sampler* s, int buf, int64_t step0, int64_t len, int split_mode, const int* host_perm,
                          bool need_slots, const int* front, hipStream_t gs) {
    const DevSampler& ds = s->ds;
    const int* perm = nullptr;
    if (split_mode == LCF_SPLIT_RANDOM) {
        int n_pad = 2;
        while (n_pad < ds.n_walkers) n_pad <<= 1;
        const int threads = std::min(1024, std::max(64, n_pad / 2));
        hipLaunchKernelGGL(k_make_perm, dim3((unsigned)len), dim3(threads), (size_t)n_pad * 8, gs, ds.n_walkers, n_pad,
                           ds.key0, ds.key1, (long long)step0, s->d_perm[buf], ds.n_half,
                           need_slots ? s->d_slot[buf] : nullptr, front);
        perm = s->d_perm[buf];
    } else if (split_mode == LCF_SPLIT_HOST) {
        perm = host_perm;
    }
    const long long total = (long long)len * ds.n_walkers;
    int* slots = nullptr;
    if (need_slots) {
        slots = s->d_slot[buf];
        if (split_mode != LCF_SPLIT_RANDOM)   // (a random split's table comes with its permutations, from k_make_perm)
            hipLaunchKernelGGL(k_slots, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, gs, ds.n_walkers, ds.n_half, perm,
                               (long long)len, slots, front);
    }
    const long long recs = (long long)len * 2 * ds.n_half;
    hipLaunchKernelGGL(k_draws, dim3((unsigned)((recs + 255) / 256)), dim3(256), 0, gs, ds, perm, slots,
                       (long long)step0, (long long)len, s->d_draws[buf]);
    LCF_HIP(hipGetLastError());
    return LCF_OK;
}

// Block b of the current run (block b lives in buffer b & 1).
lcf_status generate_block(lcf_sampler* s, int64_t b, hipStream_t consumer) {
    const int buf = (int)(b & 1);
    const int64_t k0 = s->block_start(b);
    const int* host_perm = s->split_mode == LCF_SPLIT_HOST ? s->d_perm_host + (size_t)k0 * s->ds.n_walkers : nullptr;
    // the half-step in front of a later block: the last row of the previous block (the other buffer)
    const int* front = (b > 0 && s->need_slots)
                           ? s->d_slot[buf ^ 1] + (size_t)2 * s->block_len(b - 1) * s->ds.n_walkers : nullptr;
    if (lcf_status r = generate_steps(s, buf, s->run_first + k0, s->block_len(b), s->split_mode, host_perm, s->need_slots,
                                      front, consumer))
        return r;
    s->blk_generated = b;
    return LCF_OK;
}

// A run usually continues where the last one stopped (burn-in -> sampling; run_mcmc(None, ...) in a loop).  Behind the
// last launch of a run, generate the first block of such a continuation, so that its first half-step finds its draw
// records ready: sampler_begin adopts them when the new run matches (first step, split mode, slot bookkeeping).
// (Measured and dropped: the same BESIDE a one-block run instead of behind it -- into the other buffer, on a stream of the
// sampler's own, of the lowest priority, enqueued before or after the run's launch -- so that the caller's
// synchronisation does not wait for it.  The 20 us it takes behind the run disappear, but the resident launch beside it
// takes 16-36 us longer -- its workgroups arrive later: 14.4-15.0 against 14.2 us per step of a 20-step run.)
lcf_status speculate_continuation(lcf_sampler* s, hipStream_t st) {
    s->spec_first = -1;
    if (s->pending || s->split_mode == LCF_SPLIT_HOST || s->run_steps == 0) return LCF_OK;
    const int64_t first = s->run_first + s->run_steps;
    if (lcf_status r = generate_steps(s, 0, first, s->blk_first, s->split_mode, nullptr, s->need_slots, nullptr, st))
        return r;
    s->spec_first = first;
    s->spec_mode = s->split_mode;
    s->spec_slots = s->need_slots;
    return LCF_OK;
}

// Before launching the run's half-step `rel` on stream `st`: its block of draw records must be generated (it is,
// unless the caller jumped ahead).
lcf_status enter_half_step(lcf_sampler* s, long long rel, hipStream_t st) {
    const int64_t b = s->block_of_step(rel / 2);
    if (b == s->blk_current) return LCF_OK;
    while (s->blk_generated < b)
        if (lcf_status r = generate_block(s, s->blk_generated + 1, st)) return r;
    s->blk_current = b;
    return LCF_OK;
}

// After that launch (the last reader of the block left behind, through the previous half-step's records): generate
// the next block into the buffer that is now free.
lcf_status leave_half_step(lcf_sampler* s, hipStream_t st) {
    const int64_t last = s->block_of_step(s->run_steps - 1);
    if (s->blk_generated == s->blk_current && s->blk_current < last)
        return generate_block(s, s->blk_current + 1, st);
    return LCF_OK;
}

template <class T>
lcf_status dalloc(T** p, size_t n, std::vector<void*>& owned) {
    LCF_HIP(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
    owned.push_back(*p);
    return LCF_OK;
}

// Commit half-step g_next - 1 (if pending) and draw half-step g_next (if have_next); coefficients and log-priors of the
// slots [lo, hi) for the likelihood launch behind it.
lcf_status launch_next(lcf_sampler* s, bool have_next, int lo, int hi, hipStream_t st) {
    lcf_engine* e = s->e;
    const DevSampler& ds = s->ds;
    const long long g = s->g_next;
    const int have_prev = s->pending ? 1 : 0;
    if (!have_prev && !have_next) return LCF_OK;
    const long long prev_row = have_prev ? (g - 1 - s->g_run0) / 2 : 0;
    const long long rel = g - s->g_run0;
    if (st != e->stream) s->foreign_stream = true;
    if (have_next)
        if (lcf_status r = enter_half_step(s, rel, st)) return r;
    const DrawRec* draws = have_next ? s->rows(rel) : nullptr;
    const DrawRec* prev_draws = have_prev ? s->rows(rel - 1) : nullptr;
#define LCF_STEP(ND) hipLaunchKernelGGL(k_step<ND>, dim3((unsigned)ds.n_half), dim3(64), 0, st, \
                                        e->dp, ds, have_prev, prev_row, have_next ? 1 : 0, draws, prev_draws, g, lo, hi, \
                                        s->coef, s->lprior)
    switch (ds.n_dim) {  // the fit dimensions of the supported models (+ sigma) get dedicated instantiations
#ifndef LCF_DEV_BUILD
        case 2: LCF_STEP(2); break;
        case 3: LCF_STEP(3); break;
        case 4: LCF_STEP(4); break;
        case 6: LCF_STEP(6); break;
        case 7: LCF_STEP(7); break;
        case 9: LCF_STEP(9); break;
#endif
        case 5: LCF_STEP(5); break;
        case 8: LCF_STEP(8); break;
        default: LCF_STEP(0); break;
    }
#undef LCF_STEP
    LCF_HIP(hipGetLastError());
    if (have_next)
        if (lcf_status r = leave_half_step(s, st)) return r;
    s->pending = have_next;
    if (have_next) s->g_next = g + 1;
    return LCF_OK;
}

// Likelihood of the proposals [lo, hi) of the half-step drawn last.
lcf_status launch_eval(lcf_sampler* s, int lo, int hi, bool finalize, hipStream_t st) {
    lcf_engine* e = s->e;
    if (hi <= lo) return LCF_OK;
    double* part = s->ds.part2[(s->g_next - 1) & 1];
    launch_points<0>(e, lo, hi - lo, s->ds.Q[(s->g_next - 1) & 1], s->coef, s->lprior, nullptr, part, nullptr, st);
    if (finalize) {
        const int bs = 128;
        hipLaunchKernelGGL(k_finalize, dim3((hi - lo + bs - 1) / bs), dim3(bs), 0, st, e->dp, hi - lo,
                           part + (size_t)lo * (e->dp.n_parts + 1), s->lprior + lo, s->ds.newlp[(s->g_next - 1) & 1] + lo);
    }
    LCF_HIP(hipGetLastError());
    return LCF_OK;
}

constexpr size_t kLdsPerCU = 160 * 1024;

size_t fused_lds_bytes(const lcf_engine* e) {
    return e->lds_bytes + kFusedScratch * sizeof(double);
}

bool fused_eligible(const lcf_sampler* s) {
    static const bool disabled = std::getenv("LCF_NO_FUSED") != nullptr;
    const lcf_engine* e = s->e;
    return !disabled && s->half_step_kernel != LCF_HALF_STEP_PHASES && e->dp.tab_in_lds &&
           fused_lds_bytes(e) <= 80 * 1024;  // (two workgroups per CU; measured with 160 KiB allowed: no gain)
}

// One launch for a whole half-step of a single-GPU run: commit half-step g_next - 1 (if pending), draw half-step
// g_next and evaluate its likelihood.
lcf_status launch_fused(lcf_sampler* s, int lo, int hi, hipStream_t st) {
    lcf_engine* e = s->e;
    const DevSampler& ds = s->ds;
    const long long g = s->g_next;
    const int have_prev = s->pending ? 1 : 0;
    const long long prev_row = have_prev ? (g - 1 - s->g_run0) / 2 : 0;
    const long long rel = g - s->g_run0;
    if (st != e->stream) s->foreign_stream = true;
    if (lcf_status r = enter_half_step(s, rel, st)) return r;
    const DrawRec* draws = s->rows(rel);
    const DrawRec* prev_draws = have_prev ? s->rows(rel - 1) : nullptr;
    const int foreign = ds.n_half - (hi - lo);
    const dim3 grid((unsigned)((size_t)(hi - lo) * e->dp.n_parts + (foreign + kBlock / 64 - 1) / (kBlock / 64)));
    const size_t lds = fused_lds_bytes(e);
    double* lprior = ds.inline_finalize ? nullptr : s->lprior;  // the finalize launch of a sharded run reads it
#define LCF_FUSED3(ND, V, T) do { allow_lds(k_fused<ND, V, T>, lds);                                                     \
                                  hipLaunchKernelGGL((k_fused<ND, V, T>), grid, dim3(kBlock), lds, st, e->dp, ds,       \
                                                     have_prev, prev_row, draws, prev_draws, g, lo, hi, lprior); } while (0)
#define LCF_FUSED(ND)                                                                             \
    do {                                                                                          \
        if (e->dp.variant == 0) { if (e->dp.use_therm) LCF_FUSED3(ND, 0, true); else LCF_FUSED3(ND, 0, false); } \
        else { if (e->dp.use_therm) LCF_FUSED3(ND, 1, true); else LCF_FUSED3(ND, 1, false); }      \
    } while (0)
    switch (ds.n_dim) {
#ifndef LCF_DEV_BUILD
        case 4: LCF_FUSED(4); break;
        case 6: LCF_FUSED(6); break;
        case 7: LCF_FUSED(7); break;
        case 9: LCF_FUSED(9); break;
#endif
        case 5: LCF_FUSED(5); break;
        case 8: LCF_FUSED(8); break;
        default: LCF_FUSED(0); break;
    }
#undef LCF_FUSED
#undef LCF_FUSED3
    LCF_HIP(hipGetLastError());
    if (lcf_status r = leave_half_step(s, st)) return r;
    s->pending = true;
    s->g_next = g + 1;
    return LCF_OK;
}

// ---- one workgroup per proposal (k_solo): single-GPU runs whose parts fit one workgroup -----------------------------
size_t solo_lds_bytes(const lcf_engine* e) {
    return kLdsHead * sizeof(double) + (size_t)e->dp.stage_d2 * sizeof(double2) +
           (kSoloScratch + 4) * sizeof(double);
}

bool solo_eligible(const lcf_sampler* s) {
    static const bool disabled = std::getenv("LCF_NO_SOLO") != nullptr;
    const lcf_engine* e = s->e;
    // (the libm band sum, variant 0, exists to mirror the reference instruction for instruction: it keeps k_fused)
    // (and light curves without shared epochs -- thermal state per point, inside the point loop -- keep k_fused too: there
    // the serial head's registers on top of the point loop's do not fit 128 without spilling)
    return !disabled && (s->half_step_kernel == LCF_HALF_STEP_AUTO || s->half_step_kernel == LCF_HALF_STEP_SOLO) &&
           e->dp.variant != 0 && e->dp.use_therm && e->dp.tab_in_lds && e->dp.n_parts <= kMaxParts &&
           solo_lds_bytes(e) <= kLdsPerCU;
}

// The model-specialised kernels (k_solo / k_pop <..., MODEL>) take engines of the shape the benchmarks have: a power-law
// model (ShockCooling, ShockCooling2) without a fitted sigma, dense columns, interpolated band sums proved from their
// first interval.  0: the generic kernel.
int specialised_model(const DevProblem& dp) {
    static const bool disabled = std::getenv("LCF_NO_SPECIALISED") != nullptr;
    if (disabled || !dp.use_therm || !dp.use_itab || dp.variant == 0 || !dp.em_dense || !dp.itab_uniform || dp.use_sigma)
        return 0;
    if (dp.model == kShockCooling && dp.n_dim == 5) return kShockCooling;
    if (dp.model == kShockCooling2 && dp.n_dim == 4) return kShockCooling2;
    return 0;
}

// One half-step of a single-GPU run: ONE launch, one workgroup per proposal, accept test and commit included.
// `board`: this rank's slots [lo, hi) only, rows from / to the row boards (lcf_sampler_run_rows).
lcf_status launch_solo(lcf_sampler* s, long long rel, hipStream_t st, bool board = false, int lo = 0, int hi = 0) {
    lcf_engine* e = s->e;
    const DevSampler& ds = s->ds;
    if (lcf_status r = enter_half_step(s, rel, st)) return r;
    const DrawRec* draws = s->rows(rel);
    const bool next_here = rel + 1 < 2 * s->run_steps && s->block_of_step((rel + 1) / 2) == s->blk_current;
    const DrawRec* draws_next = next_here ? draws + ds.n_half : nullptr;
    const size_t lds = solo_lds_bytes(e);
    const long long row = rel / 2, G = s->g_run0 + rel, g_run0 = s->g_run0;
    const dim3 grid((unsigned)(board ? hi - lo : ds.n_half));
    if (board && hi <= lo) return leave_half_step(s, st);
    const int spec = specialised_model(e->dp);
#define LCF_SOLO6(ND, V, T, NP, B, M)                                                                                 \
    do {                                                                                                              \
        if (lds > 64 * 1024) /* allow more than the default 64 KiB of dynamic LDS (per function and device) */       \
            LCF_HIP(hipFuncSetAttribute((const void*)k_solo<ND, V, T, NP, B, M>,                                      \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCU));                 \
        hipLaunchKernelGGL((k_solo<ND, V, T, NP, B, M>), grid, dim3(kBlock * (NP == 8 ? 4 : 2)), lds, st, e->d_dp, ds, row, \
                           draws, draws_next, G, g_run0, lo);                                                         \
    } while (0)
#define LCF_SOLO5(ND, V, T, NP, B)                                                                                    \
    do {                                                                                                              \
        if (NP != 8 && ND == 5 && spec == kShockCooling) LCF_SOLO6(5, V, T, (NP == 8 ? 4 : NP), B, kShockCooling);    \
        else if (NP != 8 && ND == 4 && spec == kShockCooling2) LCF_SOLO6(4, V, T, (NP == 8 ? 4 : NP), B, kShockCooling2); \
        else LCF_SOLO6(ND, V, T, NP, B, 0);                                                                           \
    } while (0)
#define LCF_SOLO4(ND, V, T, NP)                                                                                       \
    do {                                                                                                              \
        if (board) LCF_SOLO5(ND, V, T, NP, true); else LCF_SOLO5(ND, V, T, NP, false);                                \
    } while (0)
    // workgroups of 512 threads: one part per 256 threads (up to two parts) or two (three or four)
    // ... or four (1024 threads) where the launch has at most one workgroup per CU
    static const bool no_wide = std::getenv("LCF_NO_WIDE_SOLO") != nullptr;
    const bool wide = !no_wide && e->dp.n_parts > 2 && (int)grid.x <= e->n_cus;
#define LCF_SOLO3(ND, V, T)                                                                                           \
    do {                                                                                                              \
        if (e->dp.n_parts <= 2) LCF_SOLO4(ND, V, T, 2); else if (wide) LCF_SOLO4(ND, V, T, 8); else LCF_SOLO4(ND, V, T, 4); \
    } while (0)
#define LCF_SOLO(ND) LCF_SOLO3(ND, 1, true)
    switch (ds.n_dim) {
#ifndef LCF_DEV_BUILD
        case 4: LCF_SOLO(4); break;
        case 6: LCF_SOLO(6); break;
        case 7: LCF_SOLO(7); break;
        case 9: LCF_SOLO(9); break;
#endif
        case 5: LCF_SOLO(5); break;
        case 8: LCF_SOLO(8); break;
        default: LCF_SOLO(0); break;
    }
#undef LCF_SOLO
#undef LCF_SOLO3
#undef LCF_SOLO4
#undef LCF_SOLO5
#undef LCF_SOLO6
    LCF_HIP(hipGetLastError());
    return leave_half_step(s, st);
}

// ---- a block of half-steps in ONE launch (k_solo_run) ---------------------------------------------------------------
// Runs of one GPU whose half-steps k_solo can execute.  The launch's workgroups wait for each other's rows, so they must
// all be resident at once: the grid is what the device holds (occupancy x CUs; each workgroup then takes several slots
// of a half-step), and a process keeps ONE such launch in flight per device (`g_run_busy`): a second sampler's run
// enqueued meanwhile on another stream takes a launch per half-step.
// ... and whose proposals have a workgroup of their own (n_half <= kRunSlots: 2 workgroups of 512 threads per CU of an
// MI355X) or, for light curves of at most two parts, share one with up to three others.  With several proposals per
// workgroup and half-step the fixed share of each workgroup competes with the hardware's dispatch of one workgroup per
// proposal, where a proposal that the prior excludes costs nothing: measured per half-step, k_solo_run against k_solo,
// configs[1]'s light curve with 512 / 1024 / 2048 / 4096 walkers 4.9 / 6.0 / 11.8 / 22.7 against 6.1 / 7.8 / 13.5 /
// 23.6 us; configs[2] (four parts, 2048 proposals) 80.7 against 76.9 us -- the boundary it saves is 2 % of that launch.
// Light curves of more than two parts gain little (3000 observations at times of their own, 8 parts, 1024 walkers: 21.8
// against 23.0 us) and lose below a few hundred proposals (100 walkers: 15.2 against 13.0 us -- the rows' way through the
// board costs more than the boundary of so small a launch; configs[2]'s light curve with 512 walkers: 14.1 against the 12.8 us
// of the 1024-thread workgroups that k_solo uses for launches of at most one workgroup per CU): they take it above one
// proposal per CU, up to 512.
constexpr int kRunSlots = 512;
// Launches of at most one workgroup per CU of light curves with more than two parts (a rank's share of a strongly scaled
// ensemble: configs[2] on 8 GPUs, 256 proposals): 1024-thread workgroups, all four parts of a proposal side by side, as
// k_solo uses for such launches -- resident, for the eight-parameter models (the instantiation that exists).
// (LCF_WIDE_RUNS=0: launches that k_solo's 1024-thread workgroups would take keep a launch per half-step)
bool wide_runs() {
    const char* env = std::getenv("LCF_WIDE_RUNS");
    return !(env && env[0] == '0');
}
bool run_wide(const lcf_sampler* s, int proposals) {
    static const bool no_wide = std::getenv("LCF_NO_WIDE_SOLO") != nullptr;
    return !no_wide && s->e->dp.n_parts > 2 && proposals <= s->e->n_cus && s->ds.n_dim == 8;
}
bool run_eligible(const lcf_sampler* s) {
    static const bool disabled = std::getenv("LCF_NO_RUN_KERNEL") != nullptr;
    static const bool any_size = std::getenv("LCF_RUN_ANY_SIZE") != nullptr;   // (tests: several slots per workgroup)
    return !disabled && !s->run_off && s->half_step_kernel == LCF_HALF_STEP_AUTO && solo_eligible(s) && s->ds.n_peers == 0 &&
           ((s->e->dp.n_parts <= 2 ? s->ds.n_half <= 4 * kRunSlots
                                   : (s->ds.n_half > s->e->n_cus || (wide_runs() && run_wide(s, s->ds.n_half))) && s->ds.n_half <= kRunSlots) ||
            any_size);
}

struct RunBusy { hipEvent_t ev = nullptr; hipStream_t stream = nullptr; bool used = false; bool enqueuing = false; };
RunBusy g_run_busy[64];
std::mutex g_run_mutex;

// May a one-launch run go on stream `st` of device `dev` now?  Yes unless another stream's is being enqueued (between its
// claim and its release: the device counts as busy from the claim on, not only once the release has recorded the event)
// or still in flight.
bool run_claim(int dev, hipStream_t st) {
    if (dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(g_run_mutex);
    RunBusy& b = g_run_busy[dev];
    if (b.stream != st) {
        if (b.enqueuing) return false;
        if (b.used) {
            // (the event may have been recorded last on a stream that no longer exists -- its engine destroyed: whatever
            // the query then says, nothing is in flight there; and what it says must not stay behind as the "last error")
            const hipError_t q = hipEventQuery(b.ev);
            (void)hipGetLastError();
            if (q == hipErrorNotReady) return false;
        }
    }
    if (!b.ev && hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess) return false;
    b.stream = st;
    b.enqueuing = true;
    return true;
}
void run_release(int dev, hipStream_t st) {   // behind the last launch of the run (or on the way out of a failed enqueue)
    std::lock_guard<std::mutex> lock(g_run_mutex);
    RunBusy& b = g_run_busy[dev];
    b.used = hipEventRecord(b.ev, st) == hipSuccess;
    b.enqueuing = false;
}
struct RunClaim {   // releases on every path out of the enqueue
    int dev;
    hipStream_t st;
    bool held;
    ~RunClaim() { if (held) run_release(dev, st); }
};

template <class K>
lcf_status run_capacity(lcf_sampler* s, K kernel, int threads, size_t lds) {
    int& cap = threads > kBlock * 2 ? s->run_capacity_wide : s->run_capacity;
    if (cap >= 0) return LCF_OK;
    if (lds > 64 * 1024)   // (what the launch asks for, not the CU's whole LDS: the kernel may hold static words of its own)
        LCF_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    LCF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds));
    // (the compute units this stream may use: a CU mask on the stream -- or on the process -- leaves fewer than the device has)
    int cus = s->e->n_cus;
    uint32_t mask[16] = {0};
    if (hipExtStreamGetCUMask(s->e->stream, 16, mask) == hipSuccess) {
        int bits = 0;
        for (uint32_t m : mask) bits += __builtin_popcount(m);
        if (bits > 0 && bits < cus) cus = bits;
    }
    (void)hipGetLastError();
    cap = per_cu * cus;
    if (const char* env = std::getenv("LCF_RUN_GRID")) cap = std::min(cap, std::atoi(env));  // (tests)
    return LCF_OK;
}

// The sampler as k_solo_run reads it, in device memory (a constant-address-space pointer in the kernel): written when a
// sampler's first one-launch run is enqueued and again only when something in it has changed (a longer chain).  The
// two sets of state buffers keep their places in it -- X / LP / nacc = the set the sampler was created with -- and the
// launch's flags say which of them holds the run's start state.
DevSampler run_struct(const lcf_sampler* s) {
    DevSampler rs = s->ds;
    const bool flip = s->run_flip;
    rs.X = flip ? s->alt_X : s->ds.X;
    rs.LP = flip ? s->alt_LP : s->ds.LP;
    rs.nacc = flip ? s->alt_nacc : s->ds.nacc;
    rs.X_out = flip ? s->ds.X : s->alt_X;
    rs.LP_out = flip ? s->ds.LP : s->alt_LP;
    rs.nacc_out = flip ? s->ds.nacc : s->alt_nacc;
    rs.store_chain = rs.inline_finalize = rs.n_peers = 0;   // (per run: in the flags, or not read by this kernel)
    rs.board = static_cast<unsigned long long*>(s->run_board_mem);
    rs.peer_board[0] = rs.board;
    rs.n_board_ranks = 1;
    rs.board_rank = 0;
    rs.ring = kRunRing;
    rs.snap_out = reinterpret_cast<unsigned long long*>(s->snap);
    rs.snap_flags = s->snap_flags();
    return rs;
}

// The buffers a sampler's resident launches need beside its own: the board of tagged rows and the second set of state.
lcf_status run_buffers(lcf_sampler* s) {
    if (s->run_board_mem) return LCF_OK;
    if (lcf_status r = polled_alloc(s->e->device, false, s->run_board_bytes(), &s->run_board_mem)) return r;
    const size_t nw = s->ds.n_walkers;
    if (lcf_status r = dalloc(&s->alt_X, nw * s->ds.n_dim, s->owned)) return r;
    if (lcf_status r = dalloc(&s->alt_LP, nw, s->owned)) return r;
    return dalloc(&s->alt_nacc, nw, s->owned);
}

lcf_status run_image(lcf_sampler* s, hipStream_t st, const DevSampler** out, int* flags) {
    const DevSampler rs = run_struct(s);
    const bool flip = s->run_flip;
    if (!s->d_run_image) {
        LCF_HIP(hipMalloc((void**)&s->d_run_image, sizeof(DevSampler)));
        s->owned.push_back(s->d_run_image);
        s->run_image_valid = false;
    }
    if (!s->run_image_valid || std::memcmp(&rs, &s->run_image_host, sizeof rs) != 0) {
        // (by a kernel, in stream order behind the launches that read the old image: k_put_image)
        hipLaunchKernelGGL(k_put_image<DevSampler>, dim3(1), dim3(64), 0, st, s->d_run_image, rs);
        LCF_HIP(hipGetLastError());
        std::memcpy(&s->run_image_host, &rs, sizeof rs);
        s->run_image_valid = true;
    }
    *out = s->d_run_image;
    *flags = (s->ds.store_chain ? kRunStoreChain : 0) | (flip ? kRunFlip : 0);
    return LCF_OK;
}

// The same for a rank of a row-board run (k_solo_run<..., RANKS>): the rank's own board and its peers', no second set of
// state buffers (state and chain are collected from the board), no snapshot words.
lcf_status rows_image(lcf_sampler* s, hipStream_t st, const DevSampler** out) {
    DevSampler rs = s->ds;
    rs.store_chain = rs.inline_finalize = rs.n_peers = 0;
    rs.X_out = nullptr;
    rs.LP_out = nullptr;
    rs.nacc_out = nullptr;
    rs.snap_out = nullptr;
    rs.snap_flags = nullptr;
    if (!s->d_rows_image) {
        LCF_HIP(hipMalloc((void**)&s->d_rows_image, sizeof(DevSampler)));
        s->owned.push_back(s->d_rows_image);
        s->rows_image_valid = false;
    }
    if (!s->rows_image_valid || std::memcmp(&rs, &s->rows_image_host, sizeof rs) != 0) {
        hipLaunchKernelGGL(k_put_image<DevSampler>, dim3(1), dim3(64), 0, st, s->d_rows_image, rs);
        LCF_HIP(hipGetLastError());
        std::memcpy(&s->rows_image_host, &rs, sizeof rs);
        s->rows_image_valid = true;
    }
    *out = s->d_rows_image;
    return LCF_OK;
}

// Half-steps [rel, rel + n_hs) of the run, all inside the current block of draw records.
// `ranks`: this rank's slots [lo, hi) only, rows between the ranks' boards (lcf_sampler_run_rows); `need_progress`: see
// k_solo_run<..., RANKS>.
// `dry`: nothing is launched -- the kernel the launch would take is resolved, its LDS attribute set and its capacity on this
// device asked (what a process pays ONCE per kernel: lcf_sampler_board_connect does it ahead of the first run, so that no
// rank's first launch makes such calls while another rank's resident workgroups already wait for it).
lcf_status launch_run(lcf_sampler* s, long long rel, int n_hs, hipStream_t st, bool ranks = false, int lo = 0, int hi = 0,
                      long long need_progress = 0, bool dry = false) {
    lcf_engine* e = s->e;
    const DevSampler* rs = nullptr;
    int run_flags = 0;
    if (dry) {
        if (!ranks) hi = s->ds.n_half;
    } else if (ranks) {
        if (lcf_status r = rows_image(s, st, &rs)) return r;
    } else {
        if (lcf_status r = run_image(s, st, &rs, &run_flags)) return r;
        lo = 0;
        hi = s->ds.n_half;
    }
    const DrawRec* draws = dry ? nullptr : s->rows(rel);
    const size_t lds = solo_lds_bytes(e);
    const long long g_run0 = s->g_run0;
    const long long state_from = 2 * (s->run_steps - 1);   // X / LP / counts: written by the run's last step
    // (test of the recovery from a launch whose workgroups are not all resident: the last one is not launched at all)
    const bool test_missing = std::getenv("LCF_RUN_TEST_MISSING") != nullptr;
    const int spec = specialised_model(e->dp);
    unsigned int& arrivals = ranks ? s->rows_arrivals : s->run_arrivals;
#define LCF_RUN6(ND, NP, M, R)                                                                                        \
    do {                                                                                                              \
        constexpr int kThr = kBlock * (NP == 8 ? 4 : 2);                                                              \
        if (lcf_status r = run_capacity(s, k_solo_run<ND, 1, true, NP, M, R>, kThr, lds)) return r;                   \
        const int cap = NP == 8 ? s->run_capacity_wide : s->run_capacity;                                             \
        if (cap < 1) return fail(LCF_ERR_UNSUPPORTED, "k_solo_run does not fit the device");                          \
        if (dry) break;                                                                                               \
        const int n_wg = std::min(hi - lo, cap);                                                                      \
        const dim3 grid((unsigned)(test_missing && n_wg > 1 ? n_wg - 1 : n_wg));                                      \
        arrivals += (unsigned int)n_wg;   /* (0 = "no check": skipped when the count wraps onto it) */                 \
        if (arrivals == 0u) arrivals = 1u;                                                                            \
        hipLaunchKernelGGL((k_solo_run<ND, 1, true, NP, M, R>), grid, dim3(kThr), lds, st, e->d_dp, rs, rel, draws,    \
                           g_run0, n_hs, state_from, n_wg, run_flags, arrivals, lo, hi - lo, need_progress);          \
    } while (0)
#define LCF_RUN5(ND, NP, M) LCF_RUN6(ND, NP, M, false)
#define LCF_RUN4(ND, NP)                                                                                              \
    do {                                                                                                              \
        if (ND == 5 && spec == kShockCooling) LCF_RUN5(5, NP, kShockCooling);                                         \
        else if (ND == 4 && spec == kShockCooling2) LCF_RUN5(4, NP, kShockCooling2);                                  \
        else LCF_RUN5(ND, NP, 0);                                                                                     \
    } while (0)
#define LCF_RUN(ND) do { if (e->dp.n_parts <= 2) LCF_RUN4(ND, 2); else LCF_RUN4(ND, 4); } while (0)
    if (ranks) {
        // (between ranks: the benchmark shapes' own kernels, the companion fit's dimension, and the generic kernel with the
        // dimension at run time for everything else -- every instantiation is a minute of compile time)
        const bool two = e->dp.n_parts <= 2;
        const bool wide = run_wide(s, hi - lo);   // (1024-thread workgroups, the four parts of a proposal side by side)
        if (s->ds.n_dim == 5 && spec == kShockCooling && two) LCF_RUN6(5, 2, kShockCooling, true);
#ifndef LCF_DEV_BUILD
        else if (s->ds.n_dim == 4 && spec == kShockCooling2 && two) LCF_RUN6(4, 2, kShockCooling2, true);
        else if (s->ds.n_dim == 8 && two) LCF_RUN6(8, 2, 0, true);
        else if (s->ds.n_dim == 8 && wide) LCF_RUN6(8, 8, 0, true);
        else if (s->ds.n_dim == 8) LCF_RUN6(8, 4, 0, true);
#endif
        else if (two) LCF_RUN6(0, 2, 0, true);
        else LCF_RUN6(0, 4, 0, true);
        LCF_HIP(hipGetLastError());
        return LCF_OK;
    }
    if (s->ds.n_dim == 8 && run_wide(s, hi - lo)) {
        LCF_RUN6(8, 8, 0, false);
        LCF_HIP(hipGetLastError());
        return LCF_OK;
    }
    switch (s->ds.n_dim) {
#ifndef LCF_DEV_BUILD
        case 4: LCF_RUN(4); break;
        case 6: LCF_RUN(6); break;
        case 7: LCF_RUN(7); break;
        case 9: LCF_RUN(9); break;
#endif
        case 5: LCF_RUN(5); break;
        case 8: LCF_RUN(8); break;
        default: LCF_RUN(0); break;
    }
#undef LCF_RUN
#undef LCF_RUN4
#undef LCF_RUN5
#undef LCF_RUN6
    LCF_HIP(hipGetLastError());
    return LCF_OK;
}

// Log-posteriors of the shard's proposals from their partial sums (sharded runs: the all-gather sends these).
lcf_status launch_finalize(lcf_sampler* s, int lo, int hi, hipStream_t st) {
    if (hi <= lo) return LCF_OK;
    lcf_engine* e = s->e;
    const int par = (int)((s->g_next - 1) & 1), bs = 128;
    hipLaunchKernelGGL(k_finalize, dim3((hi - lo + bs - 1) / bs), dim3(bs), 0, st, e->dp, hi - lo,
                       s->ds.part2[par] + (size_t)lo * (e->dp.n_parts + 1), s->lprior + lo, s->ds.newlp[par] + lo);
    LCF_HIP(hipGetLastError());
    return LCF_OK;
}

// One half-step of a sharded run on this rank, up to the log-posteriors of its shard [lo, hi).
lcf_status launch_half_step_sharded(lcf_sampler* s, int lo, int hi, hipStream_t st) {
    if (fused_eligible(s) && hi > lo) {
        if (lcf_status r = launch_fused(s, lo, hi, st)) return r;
        return launch_finalize(s, lo, hi, st);
    }
    if (lcf_status r = launch_next(s, true, lo, hi, st)) return r;
    return launch_eval(s, lo, hi, true, st);
}

// The same without the finalize launch: the shard's rows (partial sums + log-prior) are the result.
lcf_status launch_half_step_rows(lcf_sampler* s, int lo, int hi, hipStream_t st) {
    if (fused_eligible(s) && hi > lo) return launch_fused(s, lo, hi, st);
    if (lcf_status r = launch_next(s, true, lo, hi, st)) return r;
    return launch_eval(s, lo, hi, false, st);
}

lcf_status flush_pending(lcf_sampler* s, hipStream_t st) {
    if (!s->pending) return LCF_OK;
    return launch_next(s, false, 0, 0, st);
}

// Commit what is pending and copy the snapshot behind it, all on the engine's stream; enqueue only.
lcf_status enqueue_snapshot(lcf_sampler* s) {
    lcf_engine* e = s->e;
    hipStream_t st = e->stream;
    if (s->foreign_stream) {  // half-steps were driven on a caller's stream: order this stream behind them
        LCF_HIP(hipDeviceSynchronize());
        s->foreign_stream = false;
    }
    if (lcf_status r = flush_pending(s, st)) return r;
    const DevSampler& ds = s->ds;
    // one small kernel writes the snapshot straight into the pinned host buffer (four separate copies cost 4x the
    // fixed price of a device-to-host transfer)
    const long long words = (long long)(s->snap_bytes() / 8);
    hipLaunchKernelGGL(k_snapshot, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, ds,
                       reinterpret_cast<unsigned long long*>(s->snap));
    LCF_HIP(hipGetLastError());
    LCF_HIP(hipEventRecord(s->ev_snap, st));
    s->snap_enqueued = true;
    s->snap_valid = false;
    return LCF_OK;
}

// The snapshot of the sampler's present state, complete in host memory on return.
lcf_status settle(lcf_sampler* s) {
    LCF_HIP(hipSetDevice(s->e->device));
    if (s->snap_valid && !s->pending && !s->foreign_stream) return LCF_OK;
    if (!s->snap_enqueued || s->pending || s->foreign_stream)
        if (lcf_status r = enqueue_snapshot(s)) return r;
    // Wait for the snapshot, not for the stream: what a run enqueues behind it (the draw records of a possible
    // continuation, 30-40 us of kernels) is nobody's business here -- everything later on the stream is ordered behind
    // it anyway.  A short run ends within a millisecond of this call: poll for that long before handing the wait to
    // the driver.
    {
        const auto t0 = std::chrono::steady_clock::now();
        while (hipEventQuery(s->ev_snap) == hipErrorNotReady &&
               std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(2)) {
        }
    }
    LCF_HIP(hipEventSynchronize(s->ev_snap));
    s->snap_enqueued = false;
    s->snap_valid = true;
    return LCF_OK;
}

// Whatever changes the state on the device makes the host's copy stale.
void invalidate_snapshot(lcf_sampler* s) { s->snap_enqueued = s->snap_valid = false; }

// Device memory for the chain of a run of n_steps steps (kept until a longer run needs more).
lcf_status reserve_chain(lcf_sampler* s, int64_t n_steps) {
    DevSampler& ds = s->ds;
    if (n_steps <= s->chain_cap) return LCF_OK;
    LCF_HIP(hipStreamSynchronize(s->e->stream));
    if (ds.chain) hipFree(ds.chain);
    if (ds.chain_lp) hipFree(ds.chain_lp);
    ds.chain = nullptr;
    ds.chain_lp = nullptr;
    s->chain_cap = 0;
    LCF_HIP(hipMalloc((void**)&ds.chain, (size_t)n_steps * ds.n_walkers * ds.n_dim * sizeof(double)));
    LCF_HIP(hipMalloc((void**)&ds.chain_lp, (size_t)n_steps * ds.n_walkers * sizeof(double)));
    s->chain_cap = n_steps;
    return LCF_OK;
}

// Start a run of n_steps steps: settle what the previous run left pending, size the chain and the draw blocks, and
// enqueue the generation of the first block.  Nothing here waits for the device unless a buffer has to grow.
// `need_slots`: the draw records carry each walker's slot in the previous half-step (every path except k_solo).
// `gen`: the stream the first block of draw records is generated on (default: the engine's own -- where a single
// sampler's half-steps follow; a population's half-steps all run on ONE stream, and so do its samplers' records).
// `defer`: the first block is NOT generated here (a population generates the blocks of all its samplers in one launch).
lcf_status sampler_begin(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode, const int32_t* perm,
                         int32_t store_chain, bool need_slots, hipStream_t gen = nullptr, bool defer = false) {
    if (!s || n_steps < 0 || first_step < 0) return fail(LCF_ERR_INVALID_ARGUMENT, "bad argument");
    if (split_mode < LCF_SPLIT_IDENTITY || split_mode > LCF_SPLIT_HOST)
        return fail(LCF_ERR_INVALID_ARGUMENT, "bad split_mode");
    if (split_mode == LCF_SPLIT_HOST && !perm && n_steps > 0)
        return fail(LCF_ERR_INVALID_ARGUMENT, "LCF_SPLIT_HOST needs perm");
    if (split_mode == LCF_SPLIT_RANDOM && s->ds.n_walkers > 16384)
        return fail(LCF_ERR_UNSUPPORTED, "device-generated splits support at most 16384 walkers; pass perm");
    if (split_mode != LCF_SPLIT_HOST) perm = nullptr;
    if (!s->has_state) return fail(LCF_ERR_STATE, "lcf_sampler_set_state must be called first");
    lcf_engine* e = s->e;
    LCF_HIP(hipSetDevice(e->device));
    if (s->foreign_stream) {  // the previous run was driven on a caller's stream: order everything behind it
        LCF_HIP(hipDeviceSynchronize());
        s->foreign_stream = false;
    }
    if (lcf_status st = flush_pending(s, e->stream)) return st;  // with the previous run's chain and draw records
    DevSampler& ds = s->ds;
    // leave a gap in the half-step numbering: no stale (last_g == g - 1) match across runs or set_state calls
    s->g_next += 2;
    s->g_run0 = s->g_next;
    ds.store_chain = store_chain ? 1 : 0;
    if (store_chain)
        if (lcf_status st = reserve_chain(s, n_steps)) return st;
    if (perm && n_steps > 0) {
        // validate: every row must be a permutation of 0..n_walkers-1 (out-of-range ids would fault the GPU)
        std::vector<char> seen(ds.n_walkers);
        for (int64_t r = 0; r < n_steps; ++r) {
            std::fill(seen.begin(), seen.end(), 0);
            const int32_t* row = perm + (size_t)r * ds.n_walkers;
            for (int i = 0; i < ds.n_walkers; ++i) {
                if (row[i] < 0 || row[i] >= ds.n_walkers || seen[row[i]])
                    return fail(LCF_ERR_INVALID_ARGUMENT, "perm rows must be permutations of the walker ids");
                seen[row[i]] = 1;
            }
        }
        LCF_HIP(hipStreamSynchronize(e->stream));
        if (n_steps > s->perm_host_rows) {
            if (s->d_perm_host) hipFree(s->d_perm_host);
            s->d_perm_host = nullptr;
            s->perm_host_rows = 0;
            LCF_HIP(hipMalloc((void**)&s->d_perm_host, (size_t)n_steps * ds.n_walkers * sizeof(int)));
            s->perm_host_rows = n_steps;
        }
        LCF_HIP(hipMemcpy(s->d_perm_host, perm, (size_t)n_steps * ds.n_walkers * sizeof(int), hipMemcpyHostToDevice));
    }
    invalidate_snapshot(s);
    s->run_first = first_step;
    s->run_steps = n_steps;
    s->split_mode = split_mode;
    s->need_slots = need_slots;
    s->blk_generated = s->blk_current = -1;
    if (n_steps == 0) return LCF_OK;
    // Block geometry: about 2^18 draw records (12 MiB) per buffer however long the run; a short first block, so that
    // the first half-step waits for a few steps' worth of records only.
    int64_t cap = std::max<int64_t>(4, std::min<int64_t>(256, (int64_t)(1 << 18) / ds.n_walkers));
    if (const char* env = std::getenv("LCF_DRAW_BLOCK")) cap = std::max<int64_t>(1, std::atoll(env));  // (tests: tiny blocks)
    bool grown = false;
    if (cap > s->blk_cap) {
        grown = true;
        LCF_HIP(hipStreamSynchronize(e->stream));
        s->free_blocks();
        for (int b = 0; b < 2; ++b) {
            LCF_HIP(hipMalloc((void**)&s->d_perm[b], (size_t)cap * ds.n_walkers * sizeof(int)));
            LCF_HIP(hipMalloc((void**)&s->d_draws[b], (size_t)cap * 2 * ds.n_half * sizeof(DrawRec)));
            LCF_HIP(hipMalloc((void**)&s->d_slot[b], (size_t)(1 + 2 * cap) * ds.n_walkers * sizeof(int)));
        }
        s->blk_cap = cap;
        int n_pad = 2;
        while (n_pad < ds.n_walkers) n_pad <<= 1;
        if ((size_t)n_pad * 8 > 65536)
            LCF_HIP(hipFuncSetAttribute((const void*)k_make_perm, hipFuncAttributeMaxDynamicSharedMemorySize, n_pad * 8));
    }
    s->blk_steps = s->blk_cap;
    s->blk_first = std::min<int64_t>(s->blk_cap, LCF_FIRST_BLOCK);
    if (s->spec_first == first_step && s->spec_mode == split_mode && s->spec_slots == need_slots && !grown && !defer &&
        (gen == nullptr || gen == e->stream)) {   // (a block speculated on the engine's stream is not ordered with another)
        s->spec_first = -1;  // the previous run left this run's first block behind (speculate_continuation)
        s->blk_generated = 0;
        return LCF_OK;
    }
    s->spec_first = -1;
    if (defer) return LCF_OK;
    return generate_block(s, 0, gen ? gen : e->stream);
}

// k_pop_run<ND, 1, G, M> for the population's shape: `L` null = its LDS attribute set and the workgroups a CU holds asked;
// else the launch.
struct PopRunLaunch {
    dim3 grid;
    hipStream_t st;
    const MultiItem* items;
    long long rel, state_from;
    int n_hs, store_chain, launch_no, n_wg;
};
template <int ND, int G, int M>
hipError_t pop_run_do(const PopRunLaunch* L, size_t lds, int* per_cu) {
    const auto kernel = k_pop_run<ND, 1, G, M>;
    if (!L) {
        if (lds > 64 * 1024)
            if (hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kernel, 64 * G, lds);
    }
    hipLaunchKernelGGL(kernel, L->grid, dim3(64 * G), lds, L->st, L->items, L->rel, L->n_hs, L->state_from, L->store_chain,
                       L->launch_no, L->n_wg);
    return hipGetLastError();
}
template <int G>
hipError_t pop_run_shape(int same_dim, int spec, const PopRunLaunch* L, size_t lds, int* per_cu) {
    if (same_dim == 5 && spec == kShockCooling) return pop_run_do<5, G, kShockCooling>(L, lds, per_cu);
#ifndef LCF_DEV_BUILD
    if (same_dim == 4 && spec == kShockCooling2) return pop_run_do<4, G, kShockCooling2>(L, lds, per_cu);
#endif
    return pop_run_do<0, G, 0>(L, lds, per_cu);
}
hipError_t pop_run_kernel(int group, int same_dim, int spec, const PopRunLaunch* L, size_t lds, int* per_cu) {
    switch (group) {
#ifdef LCF_POP_GROUPS_ALL
        case 4: return pop_run_shape<4>(same_dim, spec, L, lds, per_cu);
        case 10: return pop_run_shape<10>(same_dim, spec, L, lds, per_cu);
        case 12: return pop_run_shape<12>(same_dim, spec, L, lds, per_cu);
#endif
        default: return pop_run_shape<kPopRunGroup>(same_dim, spec, L, lds, per_cu);
    }
}

}  // namespace

extern "C" {

lcf_status lcf_sampler_create(lcf_engine* e, int32_t n_walkers, uint64_t seed, double a, lcf_sampler** out) {
    if (!e || !out) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (n_walkers < 2) return fail(LCF_ERR_INVALID_ARGUMENT, "n_walkers must be >= 2");
    if (!(a > 1.)) return fail(LCF_ERR_INVALID_ARGUMENT, "stretch scale a must be > 1");
    LCF_HIP(hipSetDevice(e->device));
    auto* s = new lcf_sampler();
    s->e = e;
    s->device = e->device;
    DevSampler& ds = s->ds;
    ds.n_walkers = n_walkers;
    ds.n_half = (n_walkers + 1) / 2;  // slots per half-step: the larger colour of an odd ensemble
    ds.n_dim = e->dp.n_dim;
    ds.key0 = (uint32_t)(seed & 0xffffffffu);
    ds.key1 = (uint32_t)(seed >> 32);
    ds.a = a;
    ds.wait_ticks = peer_wait_ticks();
    ds.resident_ticks = resident_wait_ticks();
    ds.ring = kRing;
    const size_t nw = n_walkers, nh = ds.n_half, nd = ds.n_dim;
    lcf_status st;
#define AL(p, n) if ((st = dalloc(&p, n, s->owned)) != LCF_OK) { delete s; return st; }
    AL(ds.X, nw * nd); AL(ds.LP, nw); AL(ds.nacc, nw); AL(ds.err, 1);
    for (int b = 0; b < 2; ++b) {
        AL(ds.Q[b], nh * nd); AL(ds.rec[b], nh); AL(ds.newlp[b], nh);
    }
    AL(s->coef, nh * kNCoef); AL(s->lprior, nh);
    for (int b = 0; b < 2; ++b) AL(ds.part2[b], nh * (e->dp.n_parts + 1));
#undef AL
    LCF_HIP(hipMemset(ds.nacc, 0, nw * sizeof(long long)));
    LCF_HIP(hipMemset(ds.err, 0, sizeof(int)));
    LCF_HIP(hipEventCreate(&s->ev0));
    LCF_HIP(hipEventCreate(&s->ev1));
    LCF_HIP(hipEventCreateWithFlags(&s->ev_snap, hipEventDisableTiming));
    LCF_HIP(hipHostMalloc((void**)&s->snap, s->snap_alloc(), hipHostMallocDefault));
    std::memset(s->snap, 0, s->snap_alloc());
    *out = s;
    return LCF_OK;
}

void lcf_sampler_destroy(lcf_sampler* s) { delete s; }

lcf_status lcf_sampler_reserve_chain(lcf_sampler* s, int64_t n_steps) {
    if (!s || n_steps < 0) return fail(LCF_ERR_INVALID_ARGUMENT, "bad argument");
    LCF_HIP(hipSetDevice(s->e->device));
    if (s->ds.store_chain && s->run_steps > 0 && n_steps > s->chain_cap)
        return fail(LCF_ERR_STATE, "the stored chain of the last run must be read (lcf_sampler_get_chain) before its buffer grows");
    return reserve_chain(s, n_steps);
}

lcf_status lcf_sampler_set_state(lcf_sampler* s, const double* coords) {
    if (!s || !coords) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    lcf_engine* e = s->e;
    LCF_HIP(hipSetDevice(e->device));
    LCF_HIP(hipDeviceSynchronize());
    s->pending = false;  // an uncommitted move of the old state is dropped with it
    s->foreign_stream = false;
    invalidate_snapshot(s);
    const DevSampler& ds = s->ds;
    if (lcf_status st = e->reserve(ds.n_walkers)) return st;
    LCF_HIP(hipMemcpyAsync(ds.X, coords, (size_t)ds.n_walkers * ds.n_dim * sizeof(double), hipMemcpyHostToDevice, e->stream));
    if (lcf_status st = logprob_dev(e, ds.n_walkers, ds.X, ds.LP, e->stream, 1)) return st;
    LCF_HIP(hipMemsetAsync(ds.nacc, 0, (size_t)ds.n_walkers * sizeof(long long), e->stream));
    LCF_HIP(hipMemsetAsync(ds.err, 0, sizeof(int), e->stream));
    std::memset(s->snap_flags(), 0, 2 * kSnapFlags * sizeof(unsigned int));   // (after the device synchronisation above)
    if (s->run_board_mem)   // (the abort word and its diagnosis behind the rows of the one-launch runs' board)
    {
        LCF_HIP(hipMemsetAsync(static_cast<unsigned char*>(s->run_board_mem) + s->run_board_bytes() - kBoardClear * sizeof(unsigned int), 0,
                               kBoardClear * sizeof(unsigned int), e->stream));
        s->run_arrivals = 0;
    }
    LCF_HIP(hipStreamSynchronize(e->stream));
    s->has_state = true;
    return LCF_OK;
}

lcf_status lcf_sampler_get_state(lcf_sampler* s, double* coords, double* log_prob) {
    if (!s) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (lcf_status st = settle(s)) return st;
    if (coords) std::memcpy(coords, s->snap + s->snap_x(), s->snap_lp() - s->snap_x());
    if (log_prob) std::memcpy(log_prob, s->snap + s->snap_lp(), s->snap_acc() - s->snap_lp());
    return LCF_OK;
}

lcf_status lcf_sampler_begin(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                             const int32_t* perm, int32_t store_chain) {
    if (lcf_status st = sampler_begin(s, first_step, n_steps, split_mode, perm, store_chain, true)) return st;
    // the half-steps of the phase API may be enqueued on a caller's stream, which is not ordered with the engine's own:
    // the first block of draw records (generated on the engine's stream) must be complete before this returns
    LCF_HIP(hipStreamSynchronize(s->e->stream));
    return LCF_OK;
}

// ---- phase-by-phase API (multi-GPU): propose -> evaluate(shard) -> [all-gather newlp] -> accept -------------------
lcf_status lcf_sampler_propose(lcf_sampler* s, int64_t step, int32_t half, void* stream) {
    if (!s || half < 0 || half > 1 || step < s->run_first || step >= s->run_first + s->run_steps)
        return fail(LCF_ERR_INVALID_ARGUMENT, "bad step/half");
    const long long g = s->g_run0 + 2 * (step - s->run_first) + half;
    if (g != s->g_next) return fail(LCF_ERR_STATE, "half-steps must be proposed in order, each exactly once");
    s->ds.inline_finalize = 0;  // accept tests read the gathered newlp
    // the shard is not known yet: every slot gets its coefficients (lcf_sampler_half_step knows it and is cheaper)
    return launch_next(s, true, 0, s->ds.n_half, stream ? (hipStream_t)stream : s->e->stream);
}
// propose + evaluate in one call: the shard is known, so the thermal states of [lo, hi) are computed by the same
// launch that draws the proposals (3 launches per half-step and rank: k_step, k_points, k_finalize).
lcf_status lcf_sampler_half_step(lcf_sampler* s, int64_t step, int32_t half, int32_t lo, int32_t hi, void* stream) {
    if (!s || half < 0 || half > 1 || step < s->run_first || step >= s->run_first + s->run_steps)
        return fail(LCF_ERR_INVALID_ARGUMENT, "bad step/half");
    if (lo < 0 || hi < lo || hi > s->ds.n_half) return fail(LCF_ERR_INVALID_ARGUMENT, "bad shard range");
    const long long g = s->g_run0 + 2 * (step - s->run_first) + half;
    if (g != s->g_next) return fail(LCF_ERR_STATE, "half-steps must be proposed in order, each exactly once");
    hipStream_t st = stream ? (hipStream_t)stream : s->e->stream;
    s->ds.inline_finalize = 0;
    return launch_half_step_sharded(s, lo, hi, st);
}

lcf_status lcf_sampler_evaluate(lcf_sampler* s, int32_t lo, int32_t hi, void* stream) {
    if (!s || lo < 0 || hi < lo || hi > s->ds.n_half) return fail(LCF_ERR_INVALID_ARGUMENT, "bad shard range");
    if (!s->pending) return fail(LCF_ERR_STATE, "lcf_sampler_propose must precede lcf_sampler_evaluate");
    return launch_eval(s, lo, hi, true, stream ? (hipStream_t)stream : s->e->stream);
}
lcf_status lcf_sampler_accept(lcf_sampler* s, int64_t step, int32_t half, void* stream) {
    if (!s || half < 0 || half > 1 || step < s->run_first || step >= s->run_first + s->run_steps)
        return fail(LCF_ERR_INVALID_ARGUMENT, "bad step/half");
    // The accept/reject of a half-step is applied by the kernel that draws the next one (it needs the gathered
    // newlp, which is complete once this call is reached); only the run's last half-step is committed here.
    if (step == s->run_first + s->run_steps - 1 && half == 1)
        return flush_pending(s, stream ? (hipStream_t)stream : s->e->stream);
    return LCF_OK;
}
void* lcf_sampler_newlp_ptr(lcf_sampler* s) { return s ? s->ds.newlp[(s->g_next - 1) & 1] : nullptr; }
lcf_status lcf_sampler_set_half_step_kernel(lcf_sampler* s, int32_t choice, int32_t* used) {
    if (!s || choice < LCF_HALF_STEP_AUTO || choice > LCF_HALF_STEP_SOLO)
        return fail(LCF_ERR_INVALID_ARGUMENT, "bad half-step kernel choice");
    s->half_step_kernel = choice;
    if (used) *used = run_eligible(s) ? 3 : solo_eligible(s) ? 2 : fused_eligible(s) ? 1 : 0;
    return LCF_OK;
}

int32_t lcf_sampler_last_run_kernel(const lcf_sampler* s) { return s ? s->last_kernel : -1; }

int64_t lcf_sampler_last_run_launches(const lcf_sampler* s) { return s ? s->last_launches : 0; }

int32_t lcf_sampler_one_launch(const lcf_sampler* s) { return s && (solo_eligible(s) || fused_eligible(s)) ? 1 : 0; }

lcf_status lcf_sampler_half_step_rows(lcf_sampler* s, int64_t step, int32_t half, int32_t lo, int32_t hi,
                                      void* stream) {
    if (!s || half < 0 || half > 1 || step < s->run_first || step >= s->run_first + s->run_steps)
        return fail(LCF_ERR_INVALID_ARGUMENT, "bad step/half");
    if (lo < 0 || hi < lo || hi > s->ds.n_half) return fail(LCF_ERR_INVALID_ARGUMENT, "bad shard range");
    const long long g = s->g_run0 + 2 * (step - s->run_first) + half;
    if (g != s->g_next) return fail(LCF_ERR_STATE, "half-steps must be proposed in order, each exactly once");
    hipStream_t st = stream ? (hipStream_t)stream : s->e->stream;
    s->ds.inline_finalize = 1;  // accept tests add up the gathered rows
    return launch_half_step_rows(s, lo, hi, st);
}

void* lcf_sampler_rows_ptr(lcf_sampler* s, int32_t* row_doubles) {
    if (!s) return nullptr;
    if (row_doubles) *row_doubles = s->e->dp.n_parts + 1;
    return s->ds.part2[(s->g_next - 1) & 1];
}

lcf_status lcf_sampler_check(lcf_sampler* s) {
    if (!s) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (lcf_status st = settle(s)) return st;
    int err = 0;
    std::memcpy(&err, s->snap, sizeof(int));
    {   // (what the workgroups of one-launch runs reported themselves)
        const unsigned int* flags = s->snap_flags();
        unsigned int any = 0;
        for (int k = 0; k < 2 * kSnapFlags; ++k) any |= flags[k];
        err |= (int)any;
    }
    if (err & 2) {
        // (an aborted multi-rank run leaves the ranks with different states -- a rank has committed its own walkers of
        // the half-step the others gave up on: the ensemble must be set again, on every rank, before the next run)
        const double sec = (double)s->ds.wait_ticks / 1e8;
        unsigned int w[kBoardClear] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bool run = s->last_kernel == LCF_KERNEL_RUN && s->run_board_mem && !s->last_rows;
        if (run)
            hipMemcpy(w, reinterpret_cast<unsigned char*>(s->run_board_mem) + s->run_board_bytes() - kBoardClear * sizeof(unsigned int),
                      sizeof w, hipMemcpyDeviceToHost);
        else if (s->board_mem)
            hipMemcpy(w, reinterpret_cast<unsigned char*>(s->board_mem) + s->board_bytes() - kBoardClear * sizeof(unsigned int), sizeof w,
                      hipMemcpyDeviceToHost);
        if (run && s->replay_steps >= 0 && s->replay_split != LCF_SPLIT_HOST) {
            // The launch's workgroups were not all resident (somebody else's resident kernel on this GPU): it gave up
            // within the bound of its waits and has written no state -- that goes into the other set of buffers, in the
            // last step.  Take the state it started from, drop what it reported, and run the same steps again with a
            // launch per half-step (as every later run of this sampler).
            LCF_HIP(hipStreamSynchronize(s->e->stream));
            std::swap(s->ds.X, s->alt_X);
            std::swap(s->ds.LP, s->alt_LP);
            std::swap(s->ds.nacc, s->alt_nacc);
            s->run_flip = !s->run_flip;
            int sticky = 0;
            std::memcpy(&sticky, s->snap, sizeof(int));
            sticky &= 1;                                   // (a NaN of an earlier run stays reported)
            LCF_HIP(hipMemcpy(s->ds.err, &sticky, sizeof(int), hipMemcpyHostToDevice));
            std::memcpy(s->snap, &sticky, sizeof(int));
            std::memset(s->snap_flags(), 0, 2 * kSnapFlags * sizeof(unsigned int));
            LCF_HIP(hipMemset(static_cast<unsigned char*>(s->run_board_mem) + s->run_board_bytes() - kBoardClear * sizeof(unsigned int), 0,
                              kBoardClear * sizeof(unsigned int)));
            s->run_arrivals = 0;
            static bool told = false;
            if (!told)
                std::fprintf(stderr, "liblcf_hip: a one-launch run waited %.2f s for version %u of walker %u: its workgroups were "
                             "not all resident (another resident kernel on this GPU?); the steps are repeated with a launch per "
                             "half-step, as are this sampler's later runs (LCF_NO_RUN_KERNEL=1 avoids the wait)\n",
                             w[1] == 3 ? (double)s->ds.resident_ticks / 1e8 : sec, w[2], w[3]);
            if (!told && std::getenv("LCF_TRACE_RUN"))
                std::fprintf(stderr, "liblcf_hip: (what %u, column %u, workgroups started %u; the entry held {%08x tag %u | %08x tag %u}; host: "
                             "%d walkers, board %08x)\n", w[1], w[4], w[5], w[6], w[7], w[8], w[9], s->ds.n_walkers,
                             (unsigned int)(unsigned long long)s->run_board_mem);
            told = true;
            s->run_off = true;
            s->spec_first = -1;
            invalidate_snapshot(s);
            const int64_t n = s->replay_steps;
            s->replay_steps = -1;
            if (lcf_status st = lcf_sampler_run_async(s, s->replay_first, n, s->replay_split, nullptr, s->replay_store)) return st;
            return lcf_sampler_check(s);
        }
        if (run) {
            char msg[260];
            std::snprintf(msg, sizeof msg, "one-launch run: version %u of walker %u (column %u) was not posted within %.1f s: "
                          "the launch's workgroups were not all resident (another process's persistent kernel on this "
                          "GPU?); set the state again and run with LCF_NO_RUN_KERNEL=1", w[2], w[3], w[4], sec);
            return fail(LCF_ERR_STATE, msg);
        }
        if (w[0]) {
            char msg[300];
            if (w[1] == 1)
                std::snprintf(msg, sizeof msg, "row-board run: version %u of walker %u (column %u) was not posted within "
                              "%.1f s: a rank is missing or behind (set_state is required on all ranks before the next run)",
                              w[2], w[3], w[4], sec);
            else if (w[1] == 3)
                std::snprintf(msg, sizeof msg, "row-board run: version %u of walker %u did not arrive within %.2f s and only "
                              "%u workgroups of this rank's resident launch had started: another resident kernel holds this "
                              "GPU (set_state is required on all ranks before the next run)", w[2], w[3],
                              (double)s->ds.resident_ticks / 1e8, w[4]);
            else if (s->last_kernel == LCF_KERNEL_RUN)
                std::snprintf(msg, sizeof msg, "row-board run: the launch from half-step %u waited %.1f s for rank %u to "
                              "reach half-step %u (set_state is required on all ranks before the next run)", w[2], sec, w[3], w[4]);
            else
                std::snprintf(msg, sizeof msg, "row-board run: half-step %u waited %.1f s for rank %u to finish half-step "
                              "%u (set_state is required on all ranks before the next run)", w[2], sec, w[3], w[2] - 2);
            return fail(LCF_ERR_STATE, msg);
        }
        char msg[200];
        std::snprintf(msg, sizeof msg, "a peer's rows did not arrive within %.1f s (peer-mailbox run; set_state is required "
                      "on all ranks before the next run)", sec);
        return fail(LCF_ERR_STATE, msg);
    }
    if (err) return fail(LCF_ERR_NAN_LOGPROB, "Probability function returned NaN");
    return LCF_OK;
}

// ---- RCCL, bound at run time (no link-time dependency; the library PyTorch ships is reused when torch is loaded) ----
namespace {
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, lcf_comm_id, int) = nullptr;  // ncclUniqueId is a 128-byte struct by value
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

lcf_status rccl_load(const char* path) {
    if (g_rccl.handle) return LCF_OK;
    void* h = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(LCF_ERR_UNSUPPORTED, std::string("cannot load RCCL: ") + dlerror());
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(h, "ncclCommUserRank");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy)
        return fail(LCF_ERR_UNSUPPORTED, "RCCL library lacks the expected symbols");
    g_rccl = r;
    return LCF_OK;
}

lcf_status rccl_check(int rc, const char* what) {
    if (rc == 0) return LCF_OK;
    return fail(LCF_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}
}  // namespace

struct lcf_comm {
    void* comm = nullptr;
    int n_ranks = 1, rank = 0;
};

lcf_status lcf_comm_probe(const char* rccl_path) { return rccl_load(rccl_path); }

lcf_status lcf_comm_unique_id(const char* rccl_path, lcf_comm_id* out) {
    if (!out) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (lcf_status st = rccl_load(rccl_path)) return st;
    return rccl_check(g_rccl.GetUniqueId(out), "ncclGetUniqueId");
}

lcf_status lcf_comm_create(const char* rccl_path, const lcf_comm_id* id, int32_t n_ranks, int32_t rank, int32_t device,
                           lcf_comm** out) {
    if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(LCF_ERR_INVALID_ARGUMENT, "bad argument");
    *out = nullptr;
    if (lcf_status st = rccl_load(rccl_path)) return st;
    LCF_HIP(hipSetDevice(device));
    auto* c = new lcf_comm();
    c->n_ranks = n_ranks;
    c->rank = rank;
    if (lcf_status st = rccl_check(g_rccl.CommInitRank(&c->comm, n_ranks, *id, rank), "ncclCommInitRank")) {
        delete c;
        return st;
    }
    *out = c;
    return LCF_OK;
}

lcf_status lcf_comm_count(const lcf_comm* c, int32_t* n_ranks, int32_t* rank) {
    if (!c || !c->comm) return fail(LCF_ERR_INVALID_ARGUMENT, "null communicator");
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(LCF_ERR_UNSUPPORTED, "RCCL lacks ncclCommCount");
    int n = 0, r = 0;
    if (lcf_status st = rccl_check(g_rccl.CommCount(c->comm, &n), "ncclCommCount")) return st;
    if (lcf_status st = rccl_check(g_rccl.CommUserRank(c->comm, &r), "ncclCommUserRank")) return st;
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = r;
    return LCF_OK;
}

lcf_status lcf_comm_time_allgather(lcf_comm* c, lcf_sampler* s, int32_t reps, double* avg_ms) {
    if (!c || !s || reps <= 0 || !avg_ms) return fail(LCF_ERR_INVALID_ARGUMENT, "bad argument");
    const int nh = s->ds.n_half;
    if (nh % c->n_ranks) return fail(LCF_ERR_INVALID_ARGUMENT, "the slots of a half-step must divide evenly over the ranks");
    LCF_HIP(hipSetDevice(s->e->device));
    hipStream_t st = s->e->stream;
    const size_t stride = (size_t)s->e->dp.n_parts + 1, width = (size_t)(nh / c->n_ranks);
    double* scratch = nullptr;  // not the sampler's rows: a measurement must not disturb a chain
    LCF_HIP(hipMalloc((void**)&scratch, (size_t)nh * stride * sizeof(double)));
    LCF_HIP(hipMemsetAsync(scratch, 0, (size_t)nh * stride * sizeof(double), st));
    lcf_status rc = LCF_OK;
    for (int k = -2; k < reps && rc == LCF_OK; ++k) {  // two warm-up rounds, then the timed ones
        if (k == 0 && hipEventRecord(s->ev0, st) != hipSuccess) rc = fail(LCF_ERR_HIP, "hipEventRecord");
        if (rc == LCF_OK)
            rc = rccl_check(g_rccl.AllGather(scratch + c->rank * width * stride, scratch, width * stride, /*ncclDouble*/ 8,
                                             c->comm, st), "ncclAllGather");
    }
    if (rc == LCF_OK && hipEventRecord(s->ev1, st) != hipSuccess) rc = fail(LCF_ERR_HIP, "hipEventRecord");
    const hipError_t serr = hipStreamSynchronize(st);
    float ms = 0.f;
    if (rc == LCF_OK && serr == hipSuccess && hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) *avg_ms = ms / reps;
    hipFree(scratch);
    if (rc != LCF_OK) return rc;
    LCF_HIP(serr);
    return LCF_OK;
}

void lcf_comm_destroy(lcf_comm* c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    delete c;
}

// Whole sharded run enqueued natively: per half-step  k_fused (replicated commit + proposals; thermal states and
// likelihood of the shard) -> in-place ncclAllGather of the shard's rows of partial sums.  No host
// round-trip and no Python between half-steps; every rank must call it with the same arguments.
lcf_status lcf_sampler_run_sharded(lcf_sampler* s, lcf_comm* c, int64_t first_step, int64_t n_steps,
                                   int32_t split_mode, const int32_t* perm, int32_t store_chain) {
    if (!s || !c) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    const int nh = s->ds.n_half;
    if (nh % c->n_ranks)
        return fail(LCF_ERR_INVALID_ARGUMENT, "the slots of a half-step must divide evenly over the ranks");
    if (lcf_status st = lcf_sampler_begin(s, first_step, n_steps, split_mode, perm, store_chain)) return st;
    hipStream_t st = s->e->stream;
    const int width = nh / c->n_ranks, lo = c->rank * width, hi = lo + width;
    // What travels is each slot's ROW (partial chi^2 sums + log-prior, n_parts + 1 doubles): the accept tests of the
    // next launch add them up themselves, exactly as on one GPU, so no finalize launch sits in front of the collective.
    s->ds.inline_finalize = 1;
    const size_t stride = (size_t)s->e->dp.n_parts + 1;
    LCF_HIP(hipEventRecord(s->ev0, st));
    for (int64_t k = 0; k < 2 * n_steps; ++k) {
        if (lcf_status r = launch_half_step_rows(s, lo, hi, st)) return r;
        double* buf = s->ds.part2[(s->g_next - 1) & 1];
        if (lcf_status r = rccl_check(g_rccl.AllGather(buf + lo * stride, buf, (size_t)width * stride, /*ncclDouble*/ 8,
                                                       c->comm, st),
                                      "ncclAllGather"))
            return r;
    }
    if (lcf_status r = flush_pending(s, st)) return r;
    LCF_HIP(hipEventRecord(s->ev1, st));
    if (lcf_status r = enqueue_snapshot(s)) return r;
    return lcf_sampler_wait(s);
}

// ---- peer mailboxes: a sharded run without a collective ---------------------------------------------------------------
namespace {
size_t mailbox_bytes(const lcf_sampler* s) {
    return (size_t)4 * s->ds.n_half * (s->e->dp.n_parts + 1) * 2 * sizeof(unsigned long long);
}
lcf_status mailbox_alloc(lcf_sampler* s) {
    if (s->mailbox) return LCF_OK;
    LCF_HIP(hipSetDevice(s->e->device));
    // uncached (fine-grained) device memory: peers' stores over the fabric and this rank's polls meet in memory
    s->mailbox_cap = mailbox_bytes(s);
    if (lcf_status st = polled_alloc(s->e->device, true, s->mailbox_cap, (void**)&s->mailbox)) return st;
    return LCF_OK;
}
}  // namespace

lcf_status lcf_sampler_mailbox_export(lcf_sampler* s, lcf_ipc_handle* out, void** local_ptr) {
    if (!s) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (lcf_status st = mailbox_alloc(s)) return st;
    if (out) {
        static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(lcf_ipc_handle), "IPC handle size");
        hipIpcMemHandle_t h;
        LCF_HIP(hipIpcGetMemHandle(&h, s->mailbox));
        std::memset(out, 0, sizeof(*out));
        std::memcpy(out, &h, sizeof(h));
    }
    if (local_ptr) *local_ptr = s->mailbox;
    return LCF_OK;
}

lcf_status lcf_sampler_mailbox_connect(lcf_sampler* s, int32_t n_ranks, int32_t rank, const lcf_ipc_handle* handles,
                                       void* const* local_ptrs) {
    if (!s || n_ranks < 1 || n_ranks > kMaxPeers || rank < 0 || rank >= n_ranks || (!handles && !local_ptrs))
        return fail(LCF_ERR_INVALID_ARGUMENT, "bad argument (at most 8 ranks)");
    if (s->ds.n_half % n_ranks)
        return fail(LCF_ERR_INVALID_ARGUMENT, "the slots of a half-step must divide evenly over the ranks");
    if (!fused_eligible(s)) return fail(LCF_ERR_UNSUPPORTED, "peer-mailbox runs need the one-launch half-step (k_fused)");
    if (lcf_status st = mailbox_alloc(s)) return st;
    LCF_HIP(hipSetDevice(s->e->device));
    for (int r = 0; r < n_ranks; ++r) {
        void* p = nullptr;
        if (r == rank) {
            p = s->mailbox;
        } else if (local_ptrs) {  // ranks emulated inside one process: plain device pointers
            p = local_ptrs[r];
        } else {
            hipIpcMemHandle_t h;
            std::memcpy(&h, &handles[r], sizeof(h));
            LCF_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
            s->opened.push_back(p);
        }
        if (!p) return fail(LCF_ERR_INVALID_ARGUMENT, "null peer mailbox");
        s->ds.peer_mbox[r] = static_cast<unsigned long long*>(p);
    }
    s->ds.mbox = s->mailbox;
    s->peer_ranks = n_ranks;
    s->peer_rank = rank;
    return LCF_OK;
}

// The run of lcf_sampler_run_sharded without its collective: every launch posts the rows of this rank's shard into
// all ranks' mailboxes and polls its own for the rows it needs.  Collective in effect: every rank must call it with the
// same arguments, after ALL ranks have returned from the previous run (the caller's barrier).
lcf_status lcf_sampler_run_peers_async(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                                       const int32_t* perm, int32_t store_chain) {
    if (!s) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (s->peer_ranks < 1) return fail(LCF_ERR_STATE, "lcf_sampler_mailbox_connect must be called first");
    if (!fused_eligible(s)) return fail(LCF_ERR_UNSUPPORTED, "peer-mailbox runs need the one-launch half-step (k_fused)");
    if (lcf_status st = sampler_begin(s, first_step, n_steps, split_mode, perm, store_chain, true)) return st;
    hipStream_t st = s->e->stream;
    const int width = s->ds.n_half / s->peer_ranks, lo = s->peer_rank * width, hi = lo + width;
    s->ds.inline_finalize = 1;
    s->ds.n_peers = s->peer_ranks;
    LCF_HIP(hipEventRecord(s->ev0, st));
    lcf_status rc = LCF_OK;
    for (int64_t k = 0; k < 2 * n_steps && rc == LCF_OK; ++k) rc = launch_fused(s, lo, hi, st);
    if (rc == LCF_OK) rc = flush_pending(s, st);
    s->ds.n_peers = 0;  // (kernel arguments are captured at launch: later runs of other kinds are unaffected)
    if (rc != LCF_OK) return rc;
    LCF_HIP(hipEventRecord(s->ev1, st));
    return enqueue_snapshot(s);
}

lcf_status lcf_sampler_run_peers(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                                 const int32_t* perm, int32_t store_chain) {
    if (lcf_status st = lcf_sampler_run_peers_async(s, first_step, n_steps, split_mode, perm, store_chain)) return st;
    return lcf_sampler_wait(s);
}

// ---- multi-GPU without replicated bookkeeping: rows over the boards (lcf_sampler_run_rows) ---------------------------
}  // extern "C"

namespace {
// Row-board runs with RESIDENT workgroups (k_solo_run<..., RANKS>): where a single-GPU run of the rank's share would
// take resident workgroups (run_eligible), and for the same reasons.  LCF_ROWS_PER_HALF_STEP=1: never.
// (Light curves of more than two parts: up to TWO slots per workgroup -- a rank of a 2-GPU run of configs[2] has 1024
// proposals: 36.1 us per half-step resident against 39.8 with a launch per half-step, whose row collection comes per
// half-step as well; on one GPU the two forms are level at that size, 36.6 / 36.5, and the hardware's dispatch of a
// workgroup per proposal wins beyond it.)
bool rows_resident_eligible(const lcf_sampler* s, int width) {
    const bool disabled = std::getenv("LCF_NO_RUN_KERNEL") != nullptr || std::getenv("LCF_ROWS_PER_HALF_STEP") != nullptr;
    const bool any_size = std::getenv("LCF_RUN_ANY_SIZE") != nullptr;
    return !disabled && s->half_step_kernel == LCF_HALF_STEP_AUTO && solo_eligible(s) &&
           ((s->e->dp.n_parts <= 2 ? width <= 4 * kRunSlots
                                   : (width > s->e->n_cus || (wide_runs() && run_wide(s, width))) && width <= 2 * kRunSlots) || any_size);
}

lcf_status board_alloc(lcf_sampler* s) {
    if (s->board_mem) return LCF_OK;
    LCF_HIP(hipSetDevice(s->e->device));
    // uncached (fine-grained) device memory: peers' stores over the fabric and this rank's polls meet in memory
    if (lcf_status st = polled_alloc(s->e->device, true, s->board_bytes(), &s->board_mem)) return st;
    s->ds.board = static_cast<unsigned long long*>(s->board_mem);
    return LCF_OK;
}
}  // namespace

extern "C" {

lcf_status lcf_sampler_board_export(lcf_sampler* s, lcf_ipc_handle* out, void** local_ptr) {
    if (!s) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (lcf_status st = board_alloc(s)) return st;
    if (out) {
        hipIpcMemHandle_t h;
        LCF_HIP(hipIpcGetMemHandle(&h, s->board_mem));
        std::memset(out, 0, sizeof(*out));
        std::memcpy(out, &h, sizeof(h));
    }
    if (local_ptr) *local_ptr = s->board_mem;
    return LCF_OK;
}

lcf_status lcf_sampler_board_connect(lcf_sampler* s, int32_t n_ranks, int32_t rank, const lcf_ipc_handle* handles,
                                     void* const* local_ptrs) {
    if (!s || n_ranks < 1 || n_ranks > kMaxPeers || rank < 0 || rank >= n_ranks || (!handles && !local_ptrs))
        return fail(LCF_ERR_INVALID_ARGUMENT, "bad argument (at most 8 ranks)");
    if (s->ds.n_half % n_ranks)
        return fail(LCF_ERR_INVALID_ARGUMENT, "the slots of a half-step must divide evenly over the ranks");
    if (!solo_eligible(s))
        return fail(LCF_ERR_UNSUPPORTED, "row-board runs need the one-workgroup-per-proposal half-step (k_solo)");
    if (lcf_status st = board_alloc(s)) return st;
    LCF_HIP(hipSetDevice(s->e->device));
    for (int r = 0; r < n_ranks; ++r) {
        void* p = nullptr;
        if (r == rank) {
            p = s->board_mem;
        } else if (local_ptrs) {  // ranks emulated inside one process: plain device pointers
            p = local_ptrs[r];
        } else {
            hipIpcMemHandle_t h;
            std::memcpy(&h, &handles[r], sizeof(h));
            LCF_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
            s->board_opened.push_back(p);
        }
        if (!p) return fail(LCF_ERR_INVALID_ARGUMENT, "null peer board");
        s->ds.peer_board[r] = static_cast<unsigned long long*>(p);
    }
    s->ds.n_board_ranks = n_ranks;
    s->ds.board_rank = rank;
    // (the kernel a resident run of this rank will take, resolved now: see launch_run's `dry`; and the image of the
    // sampler it reads, allocated now: an allocation made by one rank of a process while another rank's resident
    // workgroups already wait for it can hold its launch back for seconds)
    if (!s->d_rows_image) {
        LCF_HIP(hipMalloc((void**)&s->d_rows_image, sizeof(DevSampler)));
        s->owned.push_back(s->d_rows_image);
        s->rows_image_valid = false;
    }
    const int width = s->ds.n_half / n_ranks;
    if (rows_resident_eligible(s, width))
        if (lcf_status st = launch_run(s, 0, 0, s->e->stream, true, rank * width, rank * width + width, 0, true)) return st;
    return LCF_OK;
}

// The sharded run in which nothing is replicated: rank r evaluates, accepts and commits the proposals
// [r w, (r + 1) w) of every half-step with k_solo and posts each walker's new row (position, log-posterior, acceptance
// count) on every rank's board; the serial head of a later half-step polls its own board for the rows it needs.  No
// collective, no launch between half-steps, no work about other ranks' walkers.  When the run ends every rank takes
// the last step's rows of ALL walkers from its board, so state, acceptance counts and (if stored) the chain are
// complete on every rank.  Collective in effect: every rank calls it with the same arguments, after ALL ranks have
// returned from the previous run (the caller's barrier); the ranks' samplers must have seen the same sequence of runs.
lcf_status lcf_sampler_run_rows_async(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                                      const int32_t* perm, int32_t store_chain) {
    if (!s) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (s->ds.n_board_ranks < 1) return fail(LCF_ERR_STATE, "lcf_sampler_board_connect must be called first");
    if (!solo_eligible(s))
        return fail(LCF_ERR_UNSUPPORTED, "row-board runs need the one-workgroup-per-proposal half-step (k_solo)");
    static const bool trace = std::getenv("LCF_TRACE_RUN") != nullptr;   // (diagnostic: host time of the enqueue's parts)
    const auto t_in = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (trace)
            std::fprintf(stderr, "lcf_sampler_run_rows (rank %d): %s at %.1f us\n", s->ds.board_rank, what,
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_in).count());
    };
    if (lcf_status st = sampler_begin(s, first_step, n_steps, split_mode, perm, store_chain, true)) return st;
    mark("draw records enqueued");
    hipStream_t st = s->e->stream;
    DevSampler& ds = s->ds;
    const int width = ds.n_half / ds.n_board_ranks, lo = ds.board_rank * width, hi = lo + width;
    const bool resident = rows_resident_eligible(s, width) && n_steps > 0;
    s->last_rows = true;
    s->last_kernel = resident ? LCF_KERNEL_RUN : LCF_KERNEL_SOLO;
    s->last_launches = resident ? 0 : 2 * n_steps;
    LCF_HIP(hipEventRecord(s->ev0, st));
    const long long cells = (long long)ds.n_walkers * (ds.n_dim + 2);
    hipLaunchKernelGGL(k_board_init, dim3((unsigned)((std::max<long long>(cells, kBoardClear) + 255) / 256)), dim3(256), 0, st, ds,
                       (unsigned int)s->g_run0);
    LCF_HIP(hipGetLastError());
    s->rows_arrivals = 0;   // (k_board_init has cleared the word)
    // (k_board_collect: workgroups of four waves, each wave 64 / board_row_entries rows)
    const int collect_rpw = 64 / board_row_entries(ds.n_dim);
    auto collect_grid = [&](long long rows) { return dim3((unsigned)((rows + 4 * collect_rpw - 1) / (4 * collect_rpw))); };
    if (resident) {
        // The rank's workgroups stay for a block of up to kRunSpan half-steps (k_solo_run<..., RANKS>); behind every launch
        // the rows of its half-steps -- as every rank posted them -- go from the board into the chain, behind the last one
        // the rows of the last step into X / LP / counts.  A launch may start once every rank has reached the launch before
        // the previous one (need_progress).
        long long starts[2] = {-1, -1};   // first half-steps (relative) of the two launches in front
        for (long long rel = 0; rel < 2 * n_steps;) {
            ++s->last_launches;
            if (lcf_status r = enter_half_step(s, rel, st)) return r;
            const int64_t b = s->blk_current;
            const long long end = 2 * (s->block_start(b) + s->block_len(b));
            const int n = (int)std::min<long long>(kRunSpan, end - rel);
            const long long need = starts[0] >= 0 ? s->g_run0 + starts[0] : 0;
            if (lcf_status r = launch_run(s, rel, n, st, true, lo, hi, need)) return r;
            mark("resident launch enqueued");
            if (store_chain) {
                hipLaunchKernelGGL(k_board_collect, collect_grid((long long)n * ds.n_half), dim3(256), 0, st, ds,
                                   (long long)(s->g_run0 + rel), (long long)(rel / 2), s->rows(rel), n, 0, 1);
                LCF_HIP(hipGetLastError());
            }
            if (lcf_status r = leave_half_step(s, st)) return r;
            starts[0] = starts[1];
            starts[1] = rel;
            rel += n;
        }
    } else {
        for (int64_t k = 0; k < 2 * n_steps; ++k) {
            if (lcf_status r = launch_solo(s, k, st, true, lo, hi)) return r;
            if (store_chain && (k & 1)) {  // the rows of this step's two half-steps, from every rank, into the chain
                hipLaunchKernelGGL(k_board_collect, collect_grid(2ll * ds.n_half), dim3(256), 0, st, ds,
                                   (long long)(s->g_run0 + k - 1), (long long)(k / 2), s->rows(k - 1), 2, 0, 1);
                LCF_HIP(hipGetLastError());
            }
        }
    }
    if (n_steps > 0) {  // the last step's rows of all walkers into X / LP / counts
        const long long k = 2 * (n_steps - 1);
        hipLaunchKernelGGL(k_board_collect, collect_grid(2ll * ds.n_half), dim3(256), 0, st, ds, (long long)(s->g_run0 + k),
                           (long long)(k / 2), s->rows(k), 2, 1, 0);
        LCF_HIP(hipGetLastError());
    }
    s->g_next += 2 * n_steps;
    LCF_HIP(hipEventRecord(s->ev1, st));
    const lcf_status rc = enqueue_snapshot(s);
    mark("all enqueued");
    return rc;
}

lcf_status lcf_sampler_run_rows(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                                const int32_t* perm, int32_t store_chain) {
    if (lcf_status st = lcf_sampler_run_rows_async(s, first_step, n_steps, split_mode, perm, store_chain)) return st;
    return lcf_sampler_wait(s);
}

// Enqueue a whole run on the engine's stream and return: several samplers (one engine each = one transient of a
// population) then execute concurrently on the device.  lcf_sampler_wait() completes it.
lcf_status lcf_sampler_run_async(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                                 const int32_t* perm, int32_t store_chain) {
    if (!s) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    const bool one_launch = run_eligible(s) && n_steps > 0 && run_claim(s->e->device, s->e->stream);
    RunClaim claim{s->e->device, s->e->stream, one_launch};
    s->last_rows = false;
    if (lcf_status st = sampler_begin(s, first_step, n_steps, split_mode, perm, store_chain,
                                      !solo_eligible(s) || run_eligible(s))) return st;
    hipStream_t st = s->e->stream;
    s->ds.inline_finalize = 1;  // single GPU: no separate finalize / accept launches
    LCF_HIP(hipEventRecord(s->ev0, st));
    if (one_launch) {   // the workgroups stay for a block of half-steps and hand each other rows (k_solo_run)
        if (lcf_status r = run_buffers(s)) return r;
        s->replay_first = first_step;
        s->replay_steps = n_steps;
        s->replay_split = split_mode;
        s->replay_store = store_chain;
        s->last_kernel = LCF_KERNEL_RUN;
        s->last_launches = 0;
        for (long long rel = 0; rel < 2 * n_steps;) {   // (the first launch posts the start state on the board itself)
            ++s->last_launches;
            if (lcf_status r = enter_half_step(s, rel, st)) return r;
            const int64_t b = s->blk_current;
            const long long end = 2 * (s->block_start(b) + s->block_len(b));
            const int n = (int)std::min<long long>(kRunSpanSolo, end - rel);
            if (lcf_status r = launch_run(s, rel, n, st)) return r;
            if (lcf_status r = leave_half_step(s, st)) return r;
            rel += n;
        }
        s->g_next += 2 * n_steps;
        std::swap(s->ds.X, s->alt_X);          // the state behind this run is in the other set now
        std::swap(s->ds.LP, s->alt_LP);
        std::swap(s->ds.nacc, s->alt_nacc);
        s->run_flip = !s->run_flip;
        LCF_HIP(hipEventRecord(s->ev1, st));
        claim.held = false;
        run_release(s->e->device, st);
        // (the last step wrote the snapshot with the state: no snapshot kernel; the caller waits for this event)
        LCF_HIP(hipEventRecord(s->ev_snap, st));
        s->snap_enqueued = true;
        s->snap_valid = false;
        return speculate_continuation(s, st);
    }
    // per half-step: ONE launch (k_fused) when everything a workgroup needs fits in LDS, else
    // [commit previous + draw + thermal states] -> [per-point likelihood]; one trailing commit
    const bool fused = fused_eligible(s);
    s->last_kernel = solo_eligible(s) ? LCF_KERNEL_SOLO : fused ? LCF_KERNEL_FUSED : LCF_KERNEL_PHASES;
    s->last_launches = 2 * n_steps;
    if (solo_eligible(s)) {  // one workgroup per proposal, nothing pending between launches
        for (int64_t k = 0; k < 2 * n_steps; ++k)
            if (lcf_status r = launch_solo(s, k, st)) return r;
        s->g_next += 2 * n_steps;
        LCF_HIP(hipEventRecord(s->ev1, st));
        if (lcf_status r = enqueue_snapshot(s)) return r;
        return speculate_continuation(s, st);
    }
    for (int64_t k = 0; k < 2 * n_steps; ++k) {
        if (fused) {
            if (lcf_status r = launch_fused(s, 0, s->ds.n_half, st)) return r;
            continue;
        }
        if (lcf_status r = launch_next(s, true, 0, s->ds.n_half, st)) return r;
        if (lcf_status r = launch_eval(s, 0, s->ds.n_half, false, st)) return r;
    }
    if (lcf_status r = flush_pending(s, st)) return r;
    LCF_HIP(hipEventRecord(s->ev1, st));
    if (lcf_status r = enqueue_snapshot(s)) return r;
    return speculate_continuation(s, st);
}

lcf_status lcf_sampler_wait(lcf_sampler* s) {
    if (!s) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (lcf_status st = lcf_sampler_check(s)) return st;  // (waits for the run and its snapshot)
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->last_ms = ms;
    return LCF_OK;
}

lcf_status lcf_sampler_run(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                           const int32_t* perm, int32_t store_chain) {
    static const bool trace = std::getenv("LCF_TRACE_RUN") != nullptr;   // (diagnostic: host time of the two halves)
    const auto t0 = std::chrono::steady_clock::now();
    if (lcf_status st = lcf_sampler_run_async(s, first_step, n_steps, split_mode, perm, store_chain)) return st;
    const auto t1 = std::chrono::steady_clock::now();
    const lcf_status r = lcf_sampler_wait(s);
    if (trace) {
        const auto t2 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "lcf_sampler_run: %lld steps enqueued in %.1f us, waited %.1f us, device %.1f us\n",
                     (long long)n_steps, std::chrono::duration<double, std::micro>(t1 - t0).count(),
                     std::chrono::duration<double, std::micro>(t2 - t1).count(), 1e3 * s->last_ms);
    }
    return r;
}

// Population mode: the same n_steps for `n` samplers (one transient each, same walker count) with ONE k_step and ONE
// k_points launch per half-step covering all of them (blockIdx.y = transient).
// (`resident`: the transients' workgroups may stay for blocks of half-steps, k_pop_run; false = a launch per half-step,
// what a run falls back to -- for the rest of the process -- after a resident launch whose workgroups were not all there)
static bool g_pop_run_off = false;
static lcf_status population_run(lcf_sampler** ss, int32_t n, int64_t first_step, int64_t n_steps, int32_t split_mode,
                                 int32_t store_chain, double* elapsed_ms, bool resident) {
    if (!ss || n <= 0) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (n > 65535) return fail(LCF_ERR_INVALID_ARGUMENT, "at most 65535 transients per call");
    if (split_mode == LCF_SPLIT_HOST) return fail(LCF_ERR_UNSUPPORTED, "population runs use identity or random splits");
    const lcf_sampler* s0 = ss[0];
    if (!s0) return fail(LCF_ERR_INVALID_ARGUMENT, "null sampler");
    for (int t = 0; t < n; ++t) {
        const lcf_sampler* s = ss[t];
        if (!s) return fail(LCF_ERR_INVALID_ARGUMENT, "null sampler");
        const DevProblem &a = s->e->dp, &b = s0->e->dp;
        if (s->e->device != s0->e->device || s->ds.n_walkers != s0->ds.n_walkers || a.variant != b.variant ||
            a.use_therm != b.use_therm || a.tab_in_lds != b.tab_in_lds)
            return fail(LCF_ERR_UNSUPPORTED, "transients of one batched run must agree on device, walker count, "
                                             "band-sum variant, thermal sharing and table placement");
    }
    // Everything of this run -- every transient's draw records, the half-steps, the snapshots -- goes on ONE stream (the
    // first transient's): stream order is all the synchronisation there is.  What the transients' own streams still hold
    // (set_state, an earlier run of their own) is waited for once, here.
    LCF_HIP(hipSetDevice(s0->e->device));
    LCF_HIP(hipDeviceSynchronize());
    hipStream_t pop_stream = s0->e->stream;
    long long g = 0;
    for (int t = 0; t < n; ++t) {
        if (lcf_status st = sampler_begin(ss[t], first_step, n_steps, split_mode, nullptr, store_chain, true, pop_stream,
                                          /*defer: see pop_generate*/ true)) return st;
        g = std::max(g, ss[t]->g_next);
    }
    // The draw records of ALL transients come from one launch of each generation kernel per block of steps (random
    // splits of samplers with the same block geometry -- what a population has; else sampler by sampler as before).
    bool batched_gen = split_mode == LCF_SPLIT_RANDOM;
    for (int t = 1; t < n; ++t)
        batched_gen = batched_gen && ss[t]->blk_first == s0->blk_first && ss[t]->blk_steps == s0->blk_steps;
    GenItem* dgen = nullptr;
    if (batched_gen) {
        std::vector<GenItem> gen(n);
        for (int t = 0; t < n; ++t)
            gen[t] = GenItem{ss[t]->ds.key0, ss[t]->ds.key1, {ss[t]->d_perm[0], ss[t]->d_perm[1]},
                             {ss[t]->d_slot[0], ss[t]->d_slot[1]}, {ss[t]->d_draws[0], ss[t]->d_draws[1]}};
        LCF_HIP(hipMalloc((void**)&dgen, (size_t)n * sizeof(GenItem)));
        if (hipMemcpy(dgen, gen.data(), (size_t)n * sizeof(GenItem), hipMemcpyHostToDevice) != hipSuccess) {
            hipFree(dgen);
            return fail(LCF_ERR_HIP, "hipMemcpy of the generation items");
        }
    }
    // block b of every sampler (enqueued on the population's stream)
    auto pop_generate = [&](int64_t b) -> lcf_status {
        if (!batched_gen) {
            for (int t = 0; t < n; ++t)
                if (lcf_status r = generate_block(ss[t], b, pop_stream)) return r;
            return LCF_OK;
        }
        const DevSampler& d0 = s0->ds;
        const int buf = (int)(b & 1);
        const int64_t k0 = s0->block_start(b), len = s0->block_len(b);
        int n_pad = 2;
        while (n_pad < d0.n_walkers) n_pad <<= 1;
        const int threads = std::min(1024, std::max(64, n_pad / 2));
        const long long front_row = b > 0 ? 2 * (long long)s0->block_len(b - 1) : -1;
        if ((size_t)n_pad * 8 > 65536)
            LCF_HIP(hipFuncSetAttribute((const void*)k_make_perm_multi, hipFuncAttributeMaxDynamicSharedMemorySize, n_pad * 8));
        hipLaunchKernelGGL(k_make_perm_multi, dim3((unsigned)len, (unsigned)n), dim3(threads), (size_t)n_pad * 8, pop_stream, dgen,
                           d0.n_walkers, n_pad, (long long)(s0->run_first + k0), buf, d0.n_half, front_row);
        const long long recs = (long long)len * 2 * d0.n_half;
        hipLaunchKernelGGL(k_draws_multi, dim3((unsigned)((recs + 255) / 256), (unsigned)n), dim3(256), 0, pop_stream, dgen, d0,
                           buf, (long long)(s0->run_first + k0), (long long)len);
        LCF_HIP(hipGetLastError());
        for (int t = 0; t < n; ++t) ss[t]->blk_generated = b;
        return LCF_OK;
    };
    auto pop_enter = [&](long long rel) -> lcf_status {      // (enter_half_step for all samplers)
        const int64_t b = s0->block_of_step(rel / 2);
        if (b == s0->blk_current) return LCF_OK;
        while (s0->blk_generated < b)
            if (lcf_status r = pop_generate(s0->blk_generated + 1)) return r;
        for (int t = 0; t < n; ++t) ss[t]->blk_current = b;
        return LCF_OK;
    };
    auto pop_leave = [&]() -> lcf_status {                    // (leave_half_step for all samplers)
        const int64_t last = s0->block_of_step(s0->run_steps - 1);
        if (s0->blk_generated == s0->blk_current && s0->blk_current < last) return pop_generate(s0->blk_current + 1);
        return LCF_OK;
    };
    if (n_steps > 0)
        if (lcf_status r = pop_generate(0)) {
            if (dgen) hipFree(dgen);
            return r;
        }
    std::vector<MultiItem> items(n);
    size_t lds = 0;
    int max_parts = 1;
    const bool thermal = s0->e->dp.use_therm != 0;
    int same_dim = s0->ds.n_dim;  // compile-time walker dimension when every transient has the same
    for (int t = 0; t < n; ++t)
        if (ss[t]->ds.n_dim != same_dim) same_dim = 0;
    for (int t = 0; t < n; ++t) {
        lcf_sampler* s = ss[t];
        s->g_next = s->g_run0 = g;  // lock-step half-step numbering across the population
        s->ds.inline_finalize = 1;
        items[t] = MultiItem{s->e->dp, s->ds, {s->d_draws[0], s->d_draws[1]}, (long long)s->blk_first,
                             (long long)s->blk_steps, s->coef, s->lprior};
        // With many transients in one launch a single workgroup per proposal already fills the chip, and it stages
        // the tables and reduces once for all of the proposal's chunks: fewer parts than the engine's default.
        DevProblem& ip = items[t].pb;
        const long long slots = (long long)n * s->ds.n_half;
        const int want = (int)std::max<long long>(1, (kTargetGroups + slots - 1) / slots);
        const int parts = std::min(ip.n_parts, want);
        const int mg = (ip.n_parts + parts - 1) / parts;  // engine parts merged into one workgroup's share
        const int np = (ip.n_parts + mg - 1) / mg;
        const DevProblem& ep = s->e->dp;
        for (int J = 0; J <= kMaxParts; ++J) {
            ip.part_start[J] = J < np ? ep.part_start[J * mg] : ep.n_points;
            ip.part_col0[J] = J < np ? ep.part_col0[J * mg] : ep.em_cols;
        }
        ip.n_parts = np;
        lds = std::max(lds, s->e->lds_bytes);
        max_parts = std::max(max_parts, items[t].pb.n_parts);
    }
    // One launch per half-step (k_pop: a workgroup per four proposals, accept test included) where every transient has
    // shared epochs, staged tables, the fast band sum, at most kPopMaxParts merged parts and tables that do not depend
    // on the proposal (ShockCooling3's reddened tables do); else the two launches below.
    const DevProblem& p0 = s0->e->dp;
    bool one_launch = std::getenv("LCF_NO_POP") == nullptr && thermal && p0.variant != 0 && p0.tab_in_lds &&
                      max_parts <= kPopMaxParts;
    size_t pop_lds = 0;
    // Proposals per workgroup: 4.  (Measured at 32 x 512 walkers x 600 points: 49.0 us per half-step; staging the
    // interpolants in LDS as well costs occupancy, 61.7 us; workgroups of 8 proposals that can afford it, 50.4 us;
    // five waves per SIMD at the price of 8 spilled registers, 47.0 us.)
    constexpr int pop_group = 4;
    for (int t = 0; t < n; ++t) one_launch = one_launch && ss[t]->e->dp.model != kShockCooling3;
    for (int t = 0; t < n && one_launch; ++t) {
        const DevProblem& ip = items[t].pb;
        pop_lds = std::max(pop_lds, kLdsHead * sizeof(double) + (size_t)ip.stage_d2 * sizeof(double2) +
                                        (size_t)pop_group * (kPopScratch + 4 * kPopMaxParts) * sizeof(double));
    }
    one_launch = one_launch && pop_lds <= kLdsPerCU;
    int pop_spec = specialised_model(items[0].pb);   // every transient of the population of that shape, or the generic kernel
    for (int t = 1; t < n; ++t)
        if (specialised_model(items[t].pb) != pop_spec) pop_spec = 0;
    LCF_HIP(hipSetDevice(s0->e->device));
    // Resident form (k_pop_run): the workgroups stay for a block of half-steps and hand each other rows through the
    // transients' boards.  The launch stages the interpolants in LDS as well, where the engine's image leaves them out.
    const bool no_pop_run = std::getenv("LCF_NO_POP_RUN") != nullptr || std::getenv("LCF_NO_RUN_KERNEL") != nullptr;
    const bool pop_itab_lds = !(std::getenv("LCF_POP_ITAB_LDS") && std::atoi(std::getenv("LCF_POP_ITAB_LDS")) == 0);
    resident = resident && one_launch && !no_pop_run && !g_pop_run_off && n_steps > 0;
    if (resident) {
        // (a board of tagged rows and a second set of state buffers per transient -- 64 KB per walker: a population of
        // thousands of transients stays with a launch per half-step rather than take more than half of the free memory)
        size_t need = 0, free_b = 0, total_b = 0;
        for (int t = 0; t < n; ++t)
            if (!ss[t]->run_board_mem) need += ss[t]->run_board_bytes() + (size_t)ss[t]->ds.n_walkers * (ss[t]->ds.n_dim + 2) * 8;
        if (need > 0 && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need > free_b / 2)) resident = false;
        (void)hipGetLastError();
    }
    resident = resident && run_claim(s0->e->device, pop_stream);
    RunClaim claim{s0->e->device, pop_stream, resident};
    size_t run_lds = 0;
    int run_group = kPopRunGroup;   // proposals (= waves) per workgroup of the resident form
#ifdef LCF_POP_GROUPS_ALL
    if (const char* env = std::getenv("LCF_POP_GROUP")) run_group = std::atoi(env);   // (experiments: 4, 8, 10, 12)
#endif
    if (resident) {
        for (int t = 0; t < n; ++t) {
            lcf_sampler* s = ss[t];
            if (lcf_status r = run_buffers(s)) {
                hipFree(dgen);
                return r;
            }
            MultiItem& it = items[t];
            it.sm = run_struct(s);
            it.g_run0 = g;
            it.flip = s->run_flip ? 1 : 0;
            it.itab_extra = 0;
            it.arrive0 = s->run_arrivals;
            it.pad = 0u;
            DevProblem& ip = it.pb;
            const size_t n_itab = ip.use_itab ? (size_t)ip.n_filters * ip.itab_m * 8 : 0;
            if (pop_itab_lds && n_itab > 0 && ip.n_itab_lds == 0 && ip.n_spl_lds == 0 && n_itab * sizeof(double) <= 40 * 1024) {
                it.itab_extra = (int)n_itab;
                ip.n_itab_lds = (int)n_itab;
                ip.stage_d2 += (int)(n_itab / 2);
            }
            run_lds = std::max(run_lds, kLdsHead * sizeof(double) + (size_t)ip.stage_d2 * sizeof(double2) +
                                            (size_t)run_group * (kPopScratch + 4 * kPopMaxParts) * sizeof(double));
        }
    }
    MultiItem* ditems = nullptr;
    LCF_HIP(hipMalloc((void**)&ditems, (size_t)n * sizeof(MultiItem)));
    hipStream_t st = s0->e->stream;
    hipError_t err = hipMemcpyAsync(ditems, items.data(), (size_t)n * sizeof(MultiItem), hipMemcpyHostToDevice, st);
    const int nh = s0->ds.n_half;
    hipEvent_t ev0 = s0->ev0, ev1 = s0->ev1;
    if (err == hipSuccess) err = hipEventRecord(ev0, st);
    const dim3 gs((unsigned)nh, (unsigned)n), gp((unsigned)(nh * max_parts), (unsigned)n);
    const dim3 bs(64), bp(kBlock);
    int run_launches = 0, run_grid = 0;
    if (resident) {
        // gridDim.x workgroups per transient, all of them on the device at once: what the device holds, shared evenly --
        // and no more than give every workgroup the same number of groups of proposals per half-step
        const int n_groups = (nh + run_group - 1) / run_group;
        int per_cu = 0;
        err = pop_run_kernel(run_group, same_dim, pop_spec, nullptr, run_lds, &per_cu);
        int cap = per_cu * s0->e->n_cus;
        if (const char* env = std::getenv("LCF_RUN_GRID")) cap = std::min(cap, std::atoi(env));  // (tests)
        const int chunk = std::min<int>(n, std::max(cap, 1));          // transients per launch
        const int room = std::max(1, cap / chunk);
        const int per_wg = (n_groups + std::min(room, n_groups) - 1) / std::min(room, n_groups);
        run_grid = (n_groups + per_wg - 1) / per_wg;
        if (cap < 1 && err == hipSuccess) err = hipErrorInvalidConfiguration;
        const bool test_missing = std::getenv("LCF_RUN_TEST_MISSING") != nullptr;
        const long long state_from = 2 * (long long)(n_steps - 1);
        for (long long rel = 0; rel < 2 * n_steps && err == hipSuccess;) {
            if (pop_enter(rel) != LCF_OK) err = hipErrorUnknown;
            if (err != hipSuccess) break;
            const int64_t b = s0->blk_current;
            const long long end = 2 * (s0->block_start(b) + s0->block_len(b));
            const int n_hs = (int)std::min<long long>(kRunSpanSolo, end - rel);
            for (int c0 = 0; c0 < n && err == hipSuccess; c0 += chunk) {
                const int nc = std::min(chunk, n - c0);
                const PopRunLaunch L{dim3((unsigned)(test_missing && run_grid > 1 ? run_grid - 1 : run_grid), (unsigned)nc), st,
                                     ditems + c0, rel, state_from, n_hs, store_chain ? 1 : 0, run_launches, run_grid};
                err = pop_run_kernel(run_group, same_dim, pop_spec, &L, run_lds, nullptr);
            }
            ++run_launches;
            for (int t = 0; t < n; ++t)
                if (st != ss[t]->e->stream) ss[t]->foreign_stream = true;
            if (err == hipSuccess && pop_leave() != LCF_OK) err = hipErrorUnknown;
            rel += n_hs;
        }
        for (int t = 0; t < n; ++t) {   // the state behind this run is in the other set of buffers now
            lcf_sampler* s = ss[t];
            s->run_arrivals += (unsigned int)(run_launches * run_grid);
            std::swap(s->ds.X, s->alt_X);
            std::swap(s->ds.LP, s->alt_LP);
            std::swap(s->ds.nacc, s->alt_nacc);
            s->run_flip = !s->run_flip;
        }
        claim.held = false;
        run_release(s0->e->device, st);
    } else if (one_launch) {
        const dim3 gq((unsigned)((nh + pop_group - 1) / pop_group), (unsigned)n), bq(64 * pop_group);
        for (int64_t k = 0; k < 2 * n_steps && err == hipSuccess; ++k) {
            if (pop_enter(k) != LCF_OK) err = hipErrorUnknown;
            if (err != hipSuccess) break;
#define LCF_POP3(ND, G, M) do { allow_lds(k_pop<ND, 1, G, M>, pop_lds);                                             \
                                hipLaunchKernelGGL((k_pop<ND, 1, G, M>), gq, bq, pop_lds, st, ditems, (long long)k); } while (0)
#define LCF_POP2(ND, G) do { if (ND == 5 && pop_spec == kShockCooling) LCF_POP3(5, G, kShockCooling);               \
                             else if (ND == 4 && pop_spec == kShockCooling2) LCF_POP3(4, G, kShockCooling2);        \
                             else LCF_POP3(ND, G, 0); } while (0)
#define LCF_POP(ND) LCF_POP2(ND, pop_group)
            switch (same_dim) {
#ifndef LCF_DEV_BUILD
                case 4: LCF_POP(4); break;
                case 6: LCF_POP(6); break;
#endif
                case 5: LCF_POP(5); break;
                case 8: LCF_POP(8); break;
                default: LCF_POP(0); break;
            }
#undef LCF_POP2
#undef LCF_POP3
#undef LCF_POP
            for (int t = 0; t < n; ++t)
                if (st != ss[t]->e->stream) ss[t]->foreign_stream = true;
            if (err == hipSuccess && pop_leave() != LCF_OK) err = hipErrorUnknown;
            if (err == hipSuccess) err = hipGetLastError();
        }
    }
    for (int64_t k = 0; !one_launch && !resident && k <= 2 * n_steps && err == hipSuccess; ++k) {
        const bool have_next = k < 2 * n_steps, have_prev = k > 0;
        if (!have_next && !have_prev) break;
        if (have_next && pop_enter(k) != LCF_OK) err = hipErrorUnknown;   // (every transient's block of draw records)
        if (err != hipSuccess) break;
#define LCF_STEPM(ND) hipLaunchKernelGGL(k_step_multi<ND>, gs, bs, 0, st, ditems, have_prev ? 1 : 0,                      \
                                         (long long)((k - 1) / 2), have_next ? 1 : 0, (long long)k, (long long)(g + k))
        switch (same_dim) {
#ifndef LCF_DEV_BUILD
            case 4: LCF_STEPM(4); break;
            case 6: LCF_STEPM(6); break;
#endif
            case 5: LCF_STEPM(5); break;
            case 8: LCF_STEPM(8); break;
            default: LCF_STEPM(0); break;
        }
#undef LCF_STEPM
        if (have_next) {
            for (int t = 0; t < n; ++t)
                if (st != ss[t]->e->stream) ss[t]->foreign_stream = true;
            if (err == hipSuccess && pop_leave() != LCF_OK) err = hipErrorUnknown;
        }
        if (!have_next) break;
        const int parity = (int)((g + k) & 1);
#define LCF_PM(V, L, T) do { allow_lds(k_points_multi<V, L, T>, lds);                                       \
                             hipLaunchKernelGGL((k_points_multi<V, L, T>), gp, bp, lds, st, ditems, parity); } while (0)
        if (p0.variant == 0) {
            if (p0.tab_in_lds) { if (thermal) LCF_PM(0, true, true); else LCF_PM(0, true, false); }
            else { if (thermal) LCF_PM(0, false, true); else LCF_PM(0, false, false); }
        } else {
            if (p0.tab_in_lds) { if (thermal) LCF_PM(1, true, true); else LCF_PM(1, true, false); }
            else { if (thermal) LCF_PM(1, false, true); else LCF_PM(1, false, false); }
        }
#undef LCF_PM
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipEventRecord(ev1, st);
    // every transient's snapshot (error word, state, counts) behind the run, on the same stream: ONE wait below serves
    // the 32 state / count / check calls that follow (each of them used to synchronise and launch on its own)
    for (int t = 0; t < n && err == hipSuccess; ++t) {
        const long long words = (long long)(ss[t]->snap_bytes() / 8);
        hipLaunchKernelGGL(k_snapshot, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, ss[t]->ds,
                           reinterpret_cast<unsigned long long*>(ss[t]->snap));
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    hipFree(ditems);
    if (dgen) hipFree(dgen);
    LCF_HIP(err);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess && elapsed_ms) *elapsed_ms = ms;
    for (int t = 0; t < n; ++t) {
        ss[t]->g_next = g + 2 * n_steps;
        ss[t]->pending = false;
        ss[t]->foreign_stream = false;      // (the stream has been waited for)
        ss[t]->snap_enqueued = false;
        ss[t]->snap_valid = true;
        ss[t]->last_ms = ms;
        ss[t]->last_rows = false;
        ss[t]->last_kernel = resident ? LCF_KERNEL_POPULATION_RUN : one_launch ? LCF_KERNEL_POPULATION : LCF_KERNEL_POPULATION_PHASES;
        ss[t]->last_launches = resident ? run_launches : 2 * n_steps * (one_launch ? 1 : 2);
    }
    if (resident) {
        // A resident launch whose workgroups were not all on the device (somebody else's resident kernel holds CUs) has
        // given up within the bound of its waits.  No state has been written -- that goes into the other set of buffers,
        // in the last step: take the states the run started from, drop what it reported, and run the same steps with a
        // launch per half-step, as every later population run of this process.
        bool gave_up = false;
        for (int t = 0; t < n; ++t) {
            int e = 0;
            std::memcpy(&e, ss[t]->snap, sizeof(int));
            const unsigned int* flags = ss[t]->snap_flags();
            for (int k = 0; k < 2 * kSnapFlags; ++k) e |= (int)flags[k];
            gave_up = gave_up || (e & 2);
        }
        if (gave_up) {
            for (int t = 0; t < n; ++t) {
                lcf_sampler* s = ss[t];
                std::swap(s->ds.X, s->alt_X);
                std::swap(s->ds.LP, s->alt_LP);
                std::swap(s->ds.nacc, s->alt_nacc);
                s->run_flip = !s->run_flip;
                int sticky = 0;
                std::memcpy(&sticky, s->snap, sizeof(int));
                sticky &= 1;                                   // (a NaN of an earlier run stays reported)
                LCF_HIP(hipMemcpy(s->ds.err, &sticky, sizeof(int), hipMemcpyHostToDevice));
                std::memcpy(s->snap, &sticky, sizeof(int));
                std::memset(s->snap_flags(), 0, 2 * kSnapFlags * sizeof(unsigned int));
                LCF_HIP(hipMemset(static_cast<unsigned char*>(s->run_board_mem) + s->run_board_bytes() - kBoardClear * sizeof(unsigned int),
                                  0, kBoardClear * sizeof(unsigned int)));
                s->run_arrivals = 0;
                invalidate_snapshot(s);
            }
            std::fprintf(stderr, "liblcf_hip: a resident population launch gave up waiting for a row: its workgroups were not all "
                         "resident (another resident kernel on this GPU?); the steps are repeated with a launch per half-step, as are "
                         "this process's later population runs (LCF_NO_POP_RUN=1 avoids the wait)\n");
            g_pop_run_off = true;
            return population_run(ss, n, first_step, n_steps, split_mode, store_chain, elapsed_ms, false);
        }
    }
    for (int t = 0; t < n; ++t)
        if (lcf_status r = lcf_sampler_check(ss[t])) return r;
    return LCF_OK;
}

lcf_status lcf_population_run(lcf_sampler** ss, int32_t n, int64_t first_step, int64_t n_steps, int32_t split_mode,
                              int32_t store_chain, double* elapsed_ms) {
    return population_run(ss, n, first_step, n_steps, split_mode, store_chain, elapsed_ms, true);
}

lcf_status lcf_sampler_get_chain(lcf_sampler* s, double* chain, double* log_prob) {
    if (!s) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (!s->ds.store_chain || s->run_steps == 0) return fail(LCF_ERR_STATE, "no stored chain");
    if (lcf_status st = settle(s)) return st;  // (the trailing commit writes the last chain row)
    const DevSampler& ds = s->ds;
    if (chain)
        LCF_HIP(hipMemcpy(chain, ds.chain, (size_t)s->run_steps * ds.n_walkers * ds.n_dim * sizeof(double), hipMemcpyDeviceToHost));
    if (log_prob)
        LCF_HIP(hipMemcpy(log_prob, ds.chain_lp, (size_t)s->run_steps * ds.n_walkers * sizeof(double), hipMemcpyDeviceToHost));
    return LCF_OK;
}

lcf_status lcf_sampler_get_naccepted(lcf_sampler* s, int64_t* n_accepted) {
    if (!s || !n_accepted) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (lcf_status st = settle(s)) return st;
    std::memcpy(n_accepted, s->snap + s->snap_acc(), s->snap_bytes() - s->snap_acc());
    return LCF_OK;
}

lcf_status lcf_sampler_get_snapshot(lcf_sampler* s, double* coords, double* log_prob, int64_t* n_accepted) {
    if (!s) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (lcf_status st = settle(s)) return st;
    if (coords) std::memcpy(coords, s->snap + s->snap_x(), s->snap_lp() - s->snap_x());
    if (log_prob) std::memcpy(log_prob, s->snap + s->snap_lp(), s->snap_acc() - s->snap_lp());
    if (n_accepted) std::memcpy(n_accepted, s->snap + s->snap_acc(), s->snap_bytes() - s->snap_acc());
    return LCF_OK;
}

double lcf_sampler_last_run_ms(const lcf_sampler* s) { return s ? s->last_ms : 0.; }

}  // extern "C"

// (diagnostic, not in lcf.h) a copy of the board of the sampler's one-launch runs: rows, then the tail words
extern "C" long long lcf_debug_read_run_board(lcf_sampler* s, void* out, long long max_bytes) {
    if (!s || !s->run_board_mem) return -1;
    const long long n = std::min<long long>(max_bytes, (long long)s->run_board_bytes());
    if (hipMemcpy(out, s->run_board_mem, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    return n;
}
#ifdef LCF_PROGRESS
extern "C" int lcf_debug_read_progress(unsigned int* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_progress), sizeof(unsigned int) * 64 * 16);
}
#endif
#ifdef LCF_STAMPS
// (sums and counts, 64 x 16 each; `reset` != 0: cleared afterwards, with the "previous stamp" of every workgroup)
extern "C" int lcf_debug_read_stamp_sums(unsigned long long* acc, unsigned long long* cnt, int reset) {
    int rc = (int)hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_acc), sizeof(unsigned long long) * 64 * 16);
    rc |= (int)hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_cnt), sizeof(unsigned long long) * 64 * 16);
    if (reset) {
        static unsigned long long zeros[64 * 16];
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_acc), zeros, sizeof zeros);
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cnt), zeros, sizeof zeros);
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_last), zeros, sizeof(unsigned long long) * 64);
    }
    return rc;
}
extern "C" int lcf_debug_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 16);
}
extern "C" int lcf_debug_read_wall(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wall), sizeof(unsigned long long) * 2 * 1024 * 2);
}
#endif
