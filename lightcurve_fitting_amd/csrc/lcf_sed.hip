// Per-epoch blackbody SED likelihood on gfx950: the band-integrated Planck primitive with (T, R[, sigma]) as direct
// parameters, batched over (epoch x candidate).  Replaces the inner log_posterior of the reference's
// bolometric.spectrum_mcmc (bolometric.py:154-164): for every epoch, [f.synthesize(planck_fast, T, R) for f in the
// epoch's filters] followed by the Gaussian log-likelihood.
//
// Work decomposition: lane = one candidate (T, R) of one epoch: loops over the epoch's observations.  Three modes:
//   precision 2: float64 through the INTERPOLANTS of ln S_f(ln T) (the light-curve engine's default level): one
//                logarithm per candidate, and per observation four 16-byte LDS reads, a degree-7 Horner and one table
//                exponential instead of the filter's 11-89 Planck samples.  Workgroups of 256 walk (epoch, tile) items
//                with a grid stride, so the 2 + 4 n_filters KiB they stage are paid once per workgroup, not per tile.
//                A candidate outside the range the interpolants are proved for (T < 2 kK, > 256 kK) is appended to a
//                list and finished by k_sed_rest over the sample tables: a few cold lanes do not stall their waves.
//   precision 0: float64 over the sample tables (same band sum as the light-curve engine's levels 1 / 2)
//   precision 1: float32 over the sample tables (v_exp_f32 / v_rcp_f32, accumulate in f32) -- BASELINE configs[3] as
//                specified; precision 2 is both faster and exact to 1e-12, so it is what the bench line reports
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "lcf.h"
#include "lcf_device.h"
#include "lcf_host.h"

using namespace lcf;

namespace {

constexpr int kSedBlock = 128;
constexpr int kSedLdsMax = 3500;  // samples staged in LDS (56 KiB as double2)

struct SedDev {
    int n_filters, n_tab, tab_in_lds, itab_m;
    const double2* tab;    // (a, W) float64: per filter [full | Gauss-compressed], each padded to quads
    const float2* tab32;   // same in float32
    const int4* desc;      // [n_filters] (full offset, full count, compressed offset, compressed count)
    const double* inv_tmin; // [n_filters] 1 / t_min of the compressed table (0: none)
    const double* exp2tab; // 2^(j/256)
    // interpolants of ln S_f(ln T) (filters.interp_planck_table): [n_filters][itab_m][8] doubles, highest power first,
    // on itab_m equal intervals of ln T from itab_u0; itab_rmin[f] = the interval coordinate from which filter f's is proved
    const double2* itab;
    const float* itab_rmin;
    double itab_u0, itab_inv_h;
};

struct SedObs {
    long long n_epochs;
    const int* ep_off;      // [n_epochs + 1]
    const int* filt;        // [n_obs]
    const double* y;        // [n_obs]
    const double* dy;       // [n_obs]
    const double* dy_med;   // [n_epochs] median(dy) of the epoch (sigma_type 'absolute')
    const double* inv_dy;   // [n_obs] 1 / dy
    const double* lognorm;  // [n_obs] ln(2 pi dy^2)
    const float* ep_rmin;   // [n_epochs] the largest itab_rmin of the epoch's filters (+inf without interpolants)
    const double4* rec;     // [n_obs] (y, 1 / dy, ln(2 pi dy^2), dy): what one observation's term needs, one 32-byte load
};

// ln S_f(T) from filter f's interpolant at the interval coordinate x (rows of 8 doubles; LDS or global memory)
// STRIDE: 16-byte words from row to row.  In LDS the rows are 80 bytes apart (kSedRowLds): the lanes of a wave are
// candidates with temperatures of their own, so they read DIFFERENT rows, and rows 64 bytes apart put every lane's
// first 16 bytes on the same 8 of the 32 banks (a 4-way conflict on each of the row's four reads).
template <class RowPtr, int STRIDE = 4>
__device__ __forceinline__ double sed_interp(RowPtr rows, int m, int f, double x) {
    const int j = (int)x;
    const double s = fma(__builtin_amdgcn_fract(x), 2., -1.);
    const RowPtr q = rows + ((size_t)f * m + j) * STRIDE;
    const double2 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    double g = fma(q0.x, s, q0.y);
    g = fma(g, s, q1.x);
    g = fma(g, s, q1.y);
    g = fma(g, s, q2.x);
    g = fma(g, s, q2.y);
    g = fma(g, s, q3.x);
    return fma(g, s, q3.y);
}

// One observation's term of -2 lnL (bolometric.py:154-164 -> models.py:121-135)
__device__ __forceinline__ double sed_term(double y, double yfit, double dy, double inv_dy, double lognorm, bool with_sigma,
                                           double su) {
    const double r = y - yfit;
    if (with_sigma) {
        const double var = fma(dy, dy, su * su);
        return log(kTwoPi * var) + r * r / var;
    }
    const double q = r * inv_dy;
    return fma(q, q, lognorm);
}

// Threads of the interpolated kernel's workgroups: as many waves as possible behind ONE staged copy of the interpolants
// (1536 workgroups of 256 staged 46 MB for 31 MB of candidates and results; 512 of 1024 stage 15 MB and fill the CU's
// 32 wave slots with two workgroups)
constexpr int kSedWide = 1024;
#ifndef LCF_SED_ROW_LDS
#define LCF_SED_ROW_LDS 5
#endif
constexpr int kSedRowLds = LCF_SED_ROW_LDS;   // 16-byte words per staged row of 8 coefficients (4 + padding)

// precision 2, the fast part: every candidate whose temperature is inside the range of all its epoch's interpolants.
// The unit of work is a WAVE: 64 candidates of one epoch (the epoch's observations are then wave-uniform: scalar loads);
// the four waves of a workgroup walk their own items with a grid stride and share the staged interpolants.
__global__ __launch_bounds__(kSedWide) void k_sed_interp(const SedDev sd, const SedObs ob, long long n_cand, int n_par,
                                                         int sigma_abs, const double* __restrict__ cand,
                                                         double* __restrict__ out, unsigned int* __restrict__ n_rest,
                                                         long long* __restrict__ rest) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* exptab = reinterpret_cast<double*>(smem);
    double2* rows = reinterpret_cast<double2*>(smem + kExpTabSize * sizeof(double));
    for (int k = threadIdx.x; k < kExpTabSize; k += kSedWide) exptab[k] = sd.exp2tab[k];
    for (int k = threadIdx.x; k < sd.n_filters * sd.itab_m * 4; k += kSedWide) rows[(k >> 2) * kSedRowLds + (k & 3)] = sd.itab[k];
    __syncthreads();
    const ExpTab et{exptab};
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long tiles = (n_cand + 63) / 64, items = ob.n_epochs * tiles;
    const int m = sd.itab_m;
    // (the candidate of the NEXT item is requested while this one is computed: a wave has ~3 items and nothing else to
    // hide that round trip behind)
    const long long stride = (long long)gridDim.x * (kSedWide / 64);
    long long item = (long long)blockIdx.x * (kSedWide / 64) + wave;
    double Tn = 0., Rn = 0., sn = 0.;
    if (item < items) {
        const long long ep = item / tiles, ci = (item % tiles) * 64 + lane;
        const double* p = cand + ((size_t)ep * n_cand + (ci < n_cand ? ci : 0)) * n_par;
        Tn = p[0], Rn = p[1], sn = n_par > 2 ? p[2] : 0.;
    }
#pragma unroll 1
    for (; item < items; item += stride) {
        const long long ep = item / tiles, ci = (item % tiles) * 64 + lane;
        const bool have = ci < n_cand;
        const double T = Tn, R = Rn, sig = sn;
        if (item + stride < items) {
            const long long ep2 = (item + stride) / tiles, ci2 = ((item + stride) % tiles) * 64 + lane;
            const double* p = cand + ((size_t)ep2 * n_cand + (ci2 < n_cand ? ci2 : 0)) * n_par;
            Tn = p[0], Rn = p[1], sn = n_par > 2 ? p[2] : 0.;
        }
        const bool hot = T > 0. && T < kTmax;   // (else the band integrals vanish: models.py:1127 with exp -> inf)
        const double x = (flog(hot ? T : 1.) - sd.itab_u0) * sd.itab_inv_h;
        const bool inside = x >= (double)ob.ep_rmin[ep] && x < (double)m;
        // -> the sample tables, by k_sed_rest: one append per wave (a counter that thousands of lanes add to one by one
        // serialises: 13 000 candidates colder than the interpolants cost 130 us that way)
        const bool listed = have && hot && !inside;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(listed);
        if (mask != 0) {
            unsigned int base = 0;
            const int first = __builtin_ctzll(mask);
            if (lane == first) base = atomicAdd(n_rest, (unsigned int)__builtin_popcountll(mask));
            base = (unsigned int)__builtin_amdgcn_readlane((int)base, first);
            if (listed)
                rest[base + __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u))] = ep * n_cand + ci;
        }
        const double xs = (hot && inside) ? x : 0.;
        const double r2 = hot ? R * R : 0.;      // planck_fast: R ** 2 * ...
        const int o0 = ob.ep_off[ep], o1 = ob.ep_off[ep + 1];   // (wave-uniform: scalar loads below)
        const double su_abs = sigma_abs ? sig * ob.dy_med[ep] : 0.;
        double acc = 0.;
        // observations in pairs, their records and coefficient rows requested together, two Horner chains side by side
        // (an odd one out is its own partner and counted once)
        for (int o = o0; o < o1; o += 2) {
            const int ob2 = min(o + 1, o1 - 1);
            const double4 ra = ob.rec[o], rb = ob.rec[ob2];
            const double La = sed_interp<const double2*, kSedRowLds>(rows, m, ob.filt[o], xs);
            const double Lb = sed_interp<const double2*, kSedRowLds>(rows, m, ob.filt[ob2], xs);
            const double ya = r2 * exp_scaled<false>(La * kInvLn2N, et), yb = r2 * exp_scaled<false>(Lb * kInvLn2N, et);
            const double ta = sed_term(ra.x, ya, ra.w, ra.y, ra.z, n_par > 2, sigma_abs ? su_abs : sig * ra.w);
            const double tb = sed_term(rb.x, yb, rb.w, rb.y, rb.z, n_par > 2, sigma_abs ? su_abs : sig * rb.w);
            acc += ta;
            acc += o + 1 < o1 ? tb : 0.;
        }
        if (have && !listed) out[(size_t)ep * n_cand + ci] = -0.5 * acc;
    }
}

__device__ inline void sed_rest_one(const SedDev& sd, const SedObs& ob, long long n_cand, int n_par, int sigma_abs,
                                    int use_ctab, const double* __restrict__ cand, double* __restrict__ out, long long at,
                                    const ExpTab et);

// precision 2, the rest: the listed candidates, one per lane, observation by observation through the interpolant where
// that filter's holds and through the (Gauss-compressed or full) sample table where it does not.  Tables from memory.
__global__ __launch_bounds__(kSedBlock) void k_sed_rest(const SedDev sd, const SedObs ob, long long n_cand, int n_par,
                                                        int sigma_abs, int use_ctab, const double* __restrict__ cand,
                                                        double* __restrict__ out, unsigned int* n_rest,
                                                        const long long* __restrict__ rest) {
    __shared__ double exptab[kExpTabSize];
    const unsigned int n_list = *n_rest;
    // (n_rest[1] counts the workgroups that have read the list's length: the last one clears both words for the next
    // call -- a fill launch of 4 us in front of a 30 us kernel otherwise)
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(n_rest + 1, 1u) == gridDim.x - 1) {
        n_rest[0] = 0u;
        n_rest[1] = 0u;
    }
    if ((unsigned int)(blockIdx.x * kSedBlock) >= n_list) return;
    for (int k = threadIdx.x; k < kExpTabSize; k += kSedBlock) exptab[k] = sd.exp2tab[k];
    __syncthreads();
#pragma unroll 1
    for (unsigned int idx = blockIdx.x * kSedBlock + threadIdx.x; idx < n_list; idx += gridDim.x * kSedBlock)
        sed_rest_one(sd, ob, n_cand, n_par, sigma_abs, use_ctab, cand, out, rest[idx], ExpTab{exptab});
}

__device__ inline void sed_rest_one(const SedDev& sd, const SedObs& ob, long long n_cand, int n_par, int sigma_abs,
                                    int use_ctab, const double* __restrict__ cand, double* __restrict__ out, long long at,
                                    const ExpTab et) {
    const long long ep = at / n_cand;
    const double* p = cand + (size_t)at * n_par;
    const double T = p[0], R = p[1], sig = n_par > 2 ? p[2] : 0.;
    const double invT = 1. / T;   // (listed candidates are hot)
    const double x = (flog(T) - sd.itab_u0) * sd.itab_inv_h;
    double acc = 0.;
    for (int o = ob.ep_off[ep]; o < ob.ep_off[ep + 1]; ++o) {
        const int f = ob.filt[o];
        double S;
        if (x >= (double)sd.itab_rmin[f] && x < (double)sd.itab_m) {
            S = exp_scaled<false>(sed_interp(sd.itab, sd.itab_m, f, x) * kInvLn2N, et);
        } else {
            const int4 ds = sd.desc[f];
            const bool comp = use_ctab && invT <= sd.inv_tmin[f];
            S = band_sum_fast(sd.tab + (comp ? ds.z : ds.x), comp ? ds.w : ds.y, invT, et);
        }
        const double dy = ob.dy[o];
        acc += sed_term(ob.y[o], R * R * S, dy, ob.inv_dy[o], ob.lognorm[o], n_par > 2, sig * (sigma_abs ? ob.dy_med[ep] : dy));
    }
    out[at] = -0.5 * acc;
}

// float32 band sum: per sample one multiply, v_exp_f32, one subtraction, v_rcp_f32 and one FMA (the hardware
// exponential and reciprocal are good to 1 ulp; beyond x ~ 88.7 the exponential is +inf and the term 0, below -126 it is
// flushed to 0 like the float32 result would be).  Tables are padded to quads with zero weights.
template <class TabPtr>
__device__ __forceinline__ float band_sum_f32(TabPtr tb, int cnt, float s2) {
    float S0 = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f;
    for (int k = 0; k < cnt; k += 4) {
        const float2 a0 = tb[k], a1 = tb[k + 1], a2 = tb[k + 2], a3 = tb[k + 3];
        S0 = fmaf(a0.y, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a0.x * s2) - 1.f), S0);
        S1 = fmaf(a1.y, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a1.x * s2) - 1.f), S1);
        S2 = fmaf(a2.y, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a2.x * s2) - 1.f), S2);
        S3 = fmaf(a3.y, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a3.x * s2) - 1.f), S3);
    }
    return (S0 + S1) + (S2 + S3);
}

template <int PREC>
__global__ __launch_bounds__(kSedBlock) void k_sed(const SedDev sd, const SedObs ob, long long n_cand, int n_par,
                                                   int sigma_abs, int use_ctab, const double* __restrict__ cand,
                                                   double* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* exptab = reinterpret_cast<double*>(smem);
    unsigned char* tabmem = smem + kExpTabSize * sizeof(double);
    const int tiles = (int)((n_cand + kSedBlock - 1) / kSedBlock);
    const long long ep = blockIdx.x / tiles;
    const long long ci = (long long)(blockIdx.x % tiles) * kSedBlock + threadIdx.x;
    if (PREC == 0)
        for (int k = threadIdx.x; k < kExpTabSize; k += kSedBlock) exptab[k] = sd.exp2tab[k];
    if (sd.tab_in_lds) {
        if (PREC == 0) {
            double2* lt = reinterpret_cast<double2*>(tabmem);
            for (int k = threadIdx.x; k < sd.n_tab; k += kSedBlock) lt[k] = sd.tab[k];
        } else {
            float2* lt = reinterpret_cast<float2*>(tabmem);
            for (int k = threadIdx.x; k < sd.n_tab; k += kSedBlock) lt[k] = sd.tab32[k];
        }
    }
    __syncthreads();
    if (ci >= n_cand) return;
    const double* p = cand + ((size_t)ep * n_cand + ci) * n_par;
    const int o0 = ob.ep_off[ep], o1 = ob.ep_off[ep + 1];
    if (PREC == 0) {
        const double T = p[0], R = p[1];
        const double sig = n_par > 2 ? p[2] : 0.;
        const bool hot = T > 0. && T < kTmax;
        const double invT = hot ? 1. / T : 0.;
        const ExpTab et{exptab};
        double acc = 0.;
        for (int o = o0; o < o1; ++o) {
            const int f = ob.filt[o];
            const int4 ds = sd.desc[f];
            const bool comp = use_ctab && invT <= sd.inv_tmin[f];
            const int off = comp ? ds.z : ds.x, cnt = comp ? ds.w : ds.y;
            double S = 0.;
            if (hot) {
                S = sd.tab_in_lds ? band_sum_fast(reinterpret_cast<const double2*>(tabmem) + off, cnt, invT, et)
                                  : band_sum_fast(sd.tab + off, cnt, invT, et);
            }
            const double yfit = R * R * S;  // planck_fast: R ** 2 * ...   models.py:1127
            const double dy = ob.dy[o];
            const double r = ob.y[o] - yfit;
            if (n_par > 2) {
                const double su = sig * (sigma_abs ? ob.dy_med[ep] : dy);
                const double var = fma(dy, dy, su * su);
                acc += log(kTwoPi * var) + r * r / var;
            } else {
                const double q = r / dy;
                acc += log(kTwoPi * dy * dy) + q * q;
            }
        }
        out[(size_t)ep * n_cand + ci] = -0.5 * acc;
    } else {
        const float T = (float)p[0], R = (float)p[1];
        const float sig = n_par > 2 ? (float)p[2] : 0.f;
        const bool hot = T > 0.f && T < 1e15f;
        const float invT = hot ? 1.f / T : 0.f;
        const float2* lt = reinterpret_cast<const float2*>(tabmem);
        float acc = 0.f;
        for (int o = o0; o < o1; ++o) {
            const int f = ob.filt[o];
            const int4 ds = sd.desc[f];
            const bool comp = use_ctab && (double)invT <= sd.inv_tmin[f];
            const int off = comp ? ds.z : ds.x, cnt = comp ? ds.w : ds.y;
            float S = 0.f;
            if (hot) {
                const float s2 = invT * 1.4426950408889634f;  // exp(x) = 2^(x log2 e)
                // (two branches so that the staged tables are read with LDS instructions, not through a generic pointer)
                S = sd.tab_in_lds ? band_sum_f32(lt + off, cnt, s2) : band_sum_f32(sd.tab32 + off, cnt, s2);
            }
            const float yfit = R * R * S;
            const float dy = (float)ob.dy[o];
            const float q0 = ((float)ob.y[o] - yfit) / dy;  // never square raw luminosities: dy^2 ~ 1e38 overflows f32
            float lnsig = __logf(dy), scale2 = 1.f;
            if (n_par > 2) {
                const float ratio = sig * (sigma_abs ? (float)ob.dy_med[ep] / dy : 1.f);
                scale2 = fmaf(ratio, ratio, 1.f);  // sigma^2 = dy^2 (1 + ratio^2)
                lnsig += 0.5f * __logf(scale2);
            }
            acc += 1.8378770664093453f + 2.f * lnsig + q0 * q0 / scale2;
        }
        out[(size_t)ep * n_cand + ci] = (double)(-0.5f * acc);
    }
}

}  // namespace

struct lcf_sed {
    int device = 0;
    SedDev sd{};
    SedObs ob{};
    std::vector<void*> owned, obs_owned;
    bool have_ctab = false;
    hipStream_t stream = nullptr;
    double *dcand = nullptr, *dout = nullptr;
    size_t cand_cap = 0, out_cap = 0;
    long long n_obs = 0;
    bool have_itab = false;
    std::vector<float> rmin;            // per filter, host copy (the epochs' thresholds are made from it)
    bool nrest_dirty = false;           // a call ended between k_sed_interp and the completion of k_sed_rest: the words are stale
    unsigned int* d_nrest = nullptr;    // precision 2: how many candidates the fast kernel left to k_sed_rest ...
    long long* d_rest = nullptr;        // ... and which (capacity: out_cap)

    void free_obs() {
        for (void* p : obs_owned) hipFree(p);
        obs_owned.clear();
        ob = SedObs{};
    }
    ~lcf_sed() {
        hipSetDevice(device);
        for (void* p : owned) hipFree(p);
        free_obs();
        if (dcand) hipFree(dcand);
        if (dout) hipFree(dout);
        if (d_nrest) hipFree(d_nrest);
        if (d_rest) hipFree(d_rest);
        if (stream) hipStreamDestroy(stream);
    }
};

extern "C" {

lcf_status lcf_sed_create(int32_t n_filters, const int32_t* tab_off, const double* tab_a, const double* tab_w,
                          const int32_t* ctab_off, const double* ctab_a, const double* ctab_w,
                          const double* ctab_tmin, const double* itab_coef, const double* itab_tmin, int32_t itab_m,
                          double itab_u0, double itab_h, int32_t device, lcf_sed** out) {
    if (!out) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const bool have_itab = itab_coef && itab_tmin && itab_m > 0;
    if (have_itab) {
        // (unlike the light-curve engine's, these may start below 1 kK -- ln T < 0: no sign bit carries a meaning here)
        if (itab_m > 4096 || !std::isfinite(itab_u0) || !(itab_h > 0.) || !std::isfinite(itab_h))
            return fail(LCF_ERR_INVALID_ARGUMENT, "interpolants: 0 < itab_m <= 4096, finite itab_u0, finite itab_h > 0");
        for (int f = 0; f < n_filters; ++f) {
            if (!(itab_tmin[f] > 0.)) return fail(LCF_ERR_INVALID_ARGUMENT, "itab_tmin must be > 0 (+inf: none)");
            if (std::isfinite(itab_tmin[f]))
                for (int k = 0; k < itab_m * 8; ++k)
                    if (!std::isfinite(itab_coef[(size_t)f * itab_m * 8 + k]))
                        return fail(LCF_ERR_INVALID_ARGUMENT, "non-finite interpolant coefficient");
        }
    }
    if (n_filters <= 0 || !tab_off || !tab_a || !tab_w || tab_off[0] != 0)
        return fail(LCF_ERR_INVALID_ARGUMENT, "bad band tables");
    const bool have_ctab = ctab_off && ctab_a && ctab_w && ctab_tmin;
    for (int f = 0; f < n_filters; ++f) {
        if (tab_off[f + 1] < tab_off[f]) return fail(LCF_ERR_INVALID_ARGUMENT, "tab_off must be non-decreasing");
        if (have_ctab && (ctab_off[f + 1] < ctab_off[f] || ctab_off[0] != 0))
            return fail(LCF_ERR_INVALID_ARGUMENT, "ctab_off must start at 0 and be non-decreasing");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(LCF_ERR_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(LCF_ERR_INVALID_ARGUMENT, "device index out of range");
    LCF_HIP(hipSetDevice(device));
    std::vector<double2> htab;
    std::vector<float2> htab32;
    std::vector<int4> hdesc(n_filters);
    std::vector<double> hinvt(n_filters, 0.);
    auto append = [&](const double* a, const double* w, int k0, int k1, int& off, int& cnt) -> bool {
        off = (int)htab.size();
        for (int k = k0; k < k1; ++k) {
            if (!(a[k] > 0.) || !std::isfinite(a[k]) || !std::isfinite(w[k])) return false;
            htab.push_back(make_double2(a[k], w[k]));
        }
        const double apad = k1 > k0 ? a[k1 - 1] : 1.;
        while ((htab.size() - off) % 4) htab.push_back(make_double2(apad, 0.));
        cnt = (int)htab.size() - off;
        return true;
    };
    for (int f = 0; f < n_filters; ++f) {
        int4 d = make_int4(0, 0, 0, 0);
        bool ok = append(tab_a, tab_w, tab_off[f], tab_off[f + 1], d.x, d.y);
        if (ok && have_ctab && ctab_off[f + 1] > ctab_off[f]) {
            ok = append(ctab_a, ctab_w, ctab_off[f], ctab_off[f + 1], d.z, d.w) && ctab_tmin[f] >= 0.;
            hinvt[f] = 1. / ctab_tmin[f];
        }
        if (!ok) return fail(LCF_ERR_INVALID_ARGUMENT, "band tables need finite a_k > 0, finite W_k, t_min >= 0");
        hdesc[f] = d;
    }
    if (htab.empty()) htab.push_back(make_double2(1., 0.));
    for (const double2& v : htab) htab32.push_back(make_float2((float)v.x, (float)v.y));
    std::vector<double> hexp(kExpTabSize);
    for (int j = 0; j < kExpTabSize; ++j) hexp[j] = std::exp2(j / (double)kExpTabSize);
    auto* s = new lcf_sed();
    s->device = device;
    s->have_ctab = have_ctab;
    // the interpolants fit the fast kernel's LDS up to 19 filters (2 + 4 KiB each of 160); beyond that precision 2 is
    // not offered and the callers get the sample-table sums
    s->have_itab = have_itab && (size_t)n_filters * itab_m * 16 * kSedRowLds + kExpTabSize * sizeof(double) <= 80 * 1024;
    std::vector<double2> hitab;
    s->rmin.assign(n_filters, INFINITY);
    if (s->have_itab) {
        hitab.resize((size_t)n_filters * itab_m * 4);
        for (size_t k = 0; k < hitab.size(); ++k) {
            const double a = itab_coef[2 * k], b = itab_coef[2 * k + 1];
            hitab[k] = make_double2(std::isfinite(a) ? a : 0., std::isfinite(b) ? b : 0.);
        }
        for (int f = 0; f < n_filters; ++f)   // (the same threshold the light-curve engine uses: lcf_engine_create)
            if (std::isfinite(itab_tmin[f]))
                s->rmin[f] = std::nextafter((float)std::max((std::log(itab_tmin[f]) - itab_u0) / itab_h, 0.), INFINITY);
    }
    lcf_status st;
    double2 *dtab, *ditab = nullptr;
    float2* dtab32;
    int4* ddesc;
    double *dexp, *dinvt;
    float* drmin;
#define UP(h, d) if ((st = upload(h, &d, s->owned)) != LCF_OK) { delete s; return st; }
    UP(htab, dtab); UP(htab32, dtab32); UP(hdesc, ddesc); UP(hexp, dexp); UP(hinvt, dinvt); UP(s->rmin, drmin);
    if (s->have_itab) UP(hitab, ditab);
#undef UP
    s->sd = SedDev{n_filters, (int)htab.size(), (int)htab.size() <= kSedLdsMax, s->have_itab ? itab_m : 0, dtab, dtab32,
                   ddesc, dinvt, dexp, ditab, drmin, s->have_itab ? itab_u0 : 0., s->have_itab ? 1. / itab_h : 0.};
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        delete s;
        return fail(LCF_ERR_HIP, "hipStreamCreate failed");
    }
    *out = s;
    return LCF_OK;
}

void lcf_sed_destroy(lcf_sed* s) { delete s; }

lcf_status lcf_sed_set_observations(lcf_sed* s, int64_t n_epochs, const int32_t* ep_off, const int32_t* filt_idx,
                                    const double* y, const double* dy) {
    if (!s || n_epochs < 0 || !ep_off) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (ep_off[0] != 0) return fail(LCF_ERR_INVALID_ARGUMENT, "ep_off[0] must be 0");
    for (int64_t e = 0; e < n_epochs; ++e)
        if (ep_off[e + 1] < ep_off[e]) return fail(LCF_ERR_INVALID_ARGUMENT, "ep_off must be non-decreasing");
    const long long n_obs = ep_off[n_epochs];
    if (n_obs > 0 && (!filt_idx || !y || !dy)) return fail(LCF_ERR_INVALID_ARGUMENT, "null observations");
    for (long long o = 0; o < n_obs; ++o)
        if (filt_idx[o] < 0 || filt_idx[o] >= s->sd.n_filters) return fail(LCF_ERR_INVALID_ARGUMENT, "filt_idx out of range");
    LCF_HIP(hipSetDevice(s->device));
    LCF_HIP(hipStreamSynchronize(s->stream));
    s->free_obs();
    std::vector<int> hoff(ep_off, ep_off + n_epochs + 1), hf(filt_idx, filt_idx + n_obs);
    std::vector<double> hy(y, y + n_obs), hdy(dy, dy + n_obs), hmed(std::max<int64_t>(n_epochs, 1), 0.);
    for (int64_t e = 0; e < n_epochs; ++e) {  // np.median of the epoch's uncertainties (bolometric.py:147-152)
        std::vector<double> v(dy + ep_off[e], dy + ep_off[e + 1]);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        const size_t m = v.size();
        hmed[e] = (m & 1) ? v[m / 2] : 0.5 * (v[m / 2 - 1] + v[m / 2]);
    }
    std::vector<double> hinv(n_obs), hln(n_obs);
    for (long long o = 0; o < n_obs; ++o) {
        hinv[o] = 1. / dy[o];
        hln[o] = std::log(kTwoPi * dy[o] * dy[o]);
    }
    std::vector<float> hrmin(std::max<int64_t>(n_epochs, 1), 0.f);
    for (int64_t e = 0; e < n_epochs; ++e)
        for (int o = ep_off[e]; o < ep_off[e + 1]; ++o) hrmin[e] = std::max(hrmin[e], s->rmin[filt_idx[o]]);
    lcf_status st;
    int *doff, *df;
    double *dy_, *ddy, *dmed, *dinv, *dln;
    float* drm;
#define UP(h, d) if ((st = upload(h, &d, s->obs_owned)) != LCF_OK) return st
    std::vector<double4> hrec(n_obs);
    for (long long o = 0; o < n_obs; ++o) hrec[o] = make_double4(y[o], hinv[o], hln[o], dy[o]);
    double4* drec;
    UP(hoff, doff); UP(hf, df); UP(hy, dy_); UP(hdy, ddy); UP(hmed, dmed); UP(hinv, dinv); UP(hln, dln); UP(hrmin, drm);
    UP(hrec, drec);
#undef UP
    s->ob = SedObs{n_epochs, doff, df, dy_, ddy, dmed, dinv, dln, drm, drec};
    s->n_obs = n_obs;
    return LCF_OK;
}

lcf_status lcf_sed_log_likelihood(lcf_sed* s, int64_t n_cand, int32_t n_par, int32_t sigma_type, const double* cand,
                                  int32_t precision, int32_t use_compressed, double* out, double* kernel_ms) {
    if (!s || n_cand < 0 || (n_cand > 0 && (!cand || !out))) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (n_par != 2 && n_par != 3) return fail(LCF_ERR_INVALID_ARGUMENT, "n_par must be 2 (T, R) or 3 (T, R, sigma)");
    if (precision < 0 || precision > 2)
        return fail(LCF_ERR_INVALID_ARGUMENT, "precision must be 0 (f64, sample tables), 1 (f32) or 2 (f64, interpolants)");
    if (precision == 2 && !s->have_itab) precision = 0;   // (no interpolants were given: the same sums, sample by sample)
    if (sigma_type != LCF_SIGMA_RELATIVE && sigma_type != LCF_SIGMA_ABSOLUTE)
        return fail(LCF_ERR_INVALID_ARGUMENT, "sigma_type must be relative or absolute");
    const long long ne = s->ob.n_epochs;
    if (ne == 0 || n_cand == 0) return LCF_OK;
    if (!s->ob.ep_off) return fail(LCF_ERR_STATE, "lcf_sed_set_observations must be called first");
    LCF_HIP(hipSetDevice(s->device));
    const size_t nc = (size_t)ne * n_cand * n_par, no = (size_t)ne * n_cand;
    if (nc > s->cand_cap) {
        if (s->dcand) hipFree(s->dcand);
        s->dcand = nullptr;
        s->cand_cap = 0;
        LCF_HIP(hipMalloc((void**)&s->dcand, nc * sizeof(double)));
        s->cand_cap = nc;
    }
    if (no > s->out_cap) {
        if (s->dout) hipFree(s->dout);
        if (s->d_rest) hipFree(s->d_rest);
        s->dout = nullptr;
        s->d_rest = nullptr;
        s->out_cap = 0;
        LCF_HIP(hipMalloc((void**)&s->dout, no * sizeof(double)));
        LCF_HIP(hipMalloc((void**)&s->d_rest, no * sizeof(long long)));
        if (!s->d_nrest) {   // {list length, workgroups of k_sed_rest that have read it}: cleared by that kernel itself
            LCF_HIP(hipMalloc((void**)&s->d_nrest, 2 * sizeof(unsigned int)));
            LCF_HIP(hipMemset(s->d_nrest, 0, 2 * sizeof(unsigned int)));
        }
        s->out_cap = no;
    }
    LCF_HIP(hipMemcpyAsync(s->dcand, cand, nc * sizeof(double), hipMemcpyHostToDevice, s->stream));
    const int uc = (use_compressed && s->have_ctab) ? 1 : 0;
    const long long tiles = (n_cand + kSedBlock - 1) / kSedBlock;
    const dim3 grid((unsigned)(ne * tiles));
    const size_t lds = kExpTabSize * sizeof(double) +
                       (s->sd.tab_in_lds ? (size_t)s->sd.n_tab * (precision == 0 ? sizeof(double2) : sizeof(float2)) : 0);
    hipEvent_t a = nullptr, b = nullptr;
    if (kernel_ms) {
        LCF_HIP(hipEventCreate(&a));
        LCF_HIP(hipEventCreate(&b));
        LCF_HIP(hipEventRecord(a, s->stream));
    }
    if (precision == 2) {
        const size_t lds2 = kExpTabSize * sizeof(double) + (size_t)s->sd.n_filters * s->sd.itab_m * 16 * kSedRowLds;
        if (lds2 > 64 * 1024)
            LCF_HIP(hipFuncSetAttribute((const void*)k_sed_interp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        // workgroups of 256 with a grid stride over the (epoch, tile) items: as many as keep every CU busy with the
        // waves its LDS allows, and no more (each stages the interpolants once)
        const long long items = ne * ((n_cand + 63) / 64);   // one per wave
        const long long per_cu = std::max<long long>(1, std::min<long long>(2, (long long)(160 * 1024 / std::max<size_t>(lds2, 1))));
        const long long wpg = kSedWide / 64;
        const unsigned groups = (unsigned)std::max<long long>(1, std::min<long long>((items + wpg - 1) / wpg, 256 * per_cu));
        // (the list's words are cleared by k_sed_rest's last workgroup; a call that failed before that left them stale --
        // the next append would start behind the stale length, past the list's capacity)
        if (s->nrest_dirty) LCF_HIP(hipMemsetAsync(s->d_nrest, 0, 2 * sizeof(unsigned int), s->stream));
        s->nrest_dirty = true;
        hipLaunchKernelGGL(k_sed_interp, dim3(groups), dim3(kSedWide), lds2, s->stream, s->sd, s->ob, (long long)n_cand,
                           n_par, sigma_type == LCF_SIGMA_ABSOLUTE, s->dcand, s->dout, s->d_nrest, s->d_rest);
        // (a fixed, small grid with a stride over the list: its length stays on the device, and usually it is short)
        hipLaunchKernelGGL(k_sed_rest, dim3((unsigned)std::min<size_t>(512, (no + kSedBlock - 1) / kSedBlock)), dim3(kSedBlock), 0, s->stream,
                           s->sd, s->ob, (long long)n_cand, n_par, sigma_type == LCF_SIGMA_ABSOLUTE, uc, s->dcand, s->dout,
                           s->d_nrest, s->d_rest);
    } else if (precision == 0)
        hipLaunchKernelGGL(k_sed<0>, grid, dim3(kSedBlock), lds, s->stream, s->sd, s->ob, (long long)n_cand, n_par,
                           sigma_type == LCF_SIGMA_ABSOLUTE, uc, s->dcand, s->dout);
    else
        hipLaunchKernelGGL(k_sed<1>, grid, dim3(kSedBlock), lds, s->stream, s->sd, s->ob, (long long)n_cand, n_par,
                           sigma_type == LCF_SIGMA_ABSOLUTE, uc, s->dcand, s->dout);
    LCF_HIP(hipGetLastError());
    if (kernel_ms) LCF_HIP(hipEventRecord(b, s->stream));
    LCF_HIP(hipMemcpyAsync(out, s->dout, no * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    LCF_HIP(hipStreamSynchronize(s->stream));
    s->nrest_dirty = false;   // (both kernels of a precision-2 call have completed)
    if (kernel_ms) {
        float ms = 0.f;
        LCF_HIP(hipEventElapsedTime(&ms, a, b));
        *kernel_ms = ms;
        hipEventDestroy(a);
        hipEventDestroy(b);
    }
    return LCF_OK;
}

}  // extern "C"
