// Device-side model arithmetic for the gfx950 light-curve likelihood kernels.
//
// Everything here follows the *behaviour* of the reference's models.py / filters.py (cited per function) but is
// organised for a 64-lane wavefront: one lane = one (walker, data point) pair, the walker's derived coefficients are
// wave-uniform (scalar registers), the per-filter band tables sit in LDS as interleaved (a_k, W_k) pairs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcf {

constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kMaxParts = 8;     // workgroups per walker in the likelihood kernel, at most
constexpr int kFewEpochs = 128;  // up to this many epochs one wave per proposal computes the thermal states
constexpr int kTargetGroups = 4096;  // population mode: workgroups per likelihood launch worth splitting proposals for
constexpr int kNCoef = 8;        // derived per-walker coefficients
constexpr int kMaxDim = 16;      // max parameters per walker
constexpr int kLdsFiltMax = 64;   // filters whose descriptors are staged with the tables
constexpr int kLdsTabMax = 3700; // (a,W) pairs staged per workgroup (< 64 KiB with the exp table)
constexpr int kMaxPeers = 8;     // ranks of a peer-mailbox run (the GPUs of one node)

constexpr double kKB = 0.08617333262145178;   // eV / kK                 models.py:10
constexpr double kC3 = 5.38477047522316e-19;  //                          models.py:11
constexpr double kC3sq = kC3 * kC3;
constexpr double kTwoPi = 6.283185307179586;
// Above this temperature [kK] a_k/T < 1e-13 and the reference's exp(x) - 1 is pure cancellation noise (and 0 beyond
// 1e17 kK); the engine returns a zero band integral there.  Physical fits stay below 1e3 kK.
constexpr double kTmax = 1e15;
// 1 / (4 pi Mpc^2) in m^-2 (models.py:12), Mpc = 1e6 * 648000 / pi au
constexpr double kC4 = 8.357743635931361e-47;

enum Model : int {
    kShockCooling = 1,
    kShockCooling2 = 2,
    kShockCooling3 = 3,
    kShockCooling4 = 4,
    kCompanion = 5,
    kCompanion2 = 6,
    kCompanion3 = 7,
    kBlackbody = 8
};

struct PriorDev {
    int kind;
    int pad;
    double p_min, p_max, mean, stddev;
};

// Immutable per-problem device data (passed to kernels by value).
// Tables of one filter inside `tab` (offsets / counts in samples, counts padded to quads): the full table and up to
// two Gauss-compressed companions, "cool" (valid for 1/T <= inv_tmin) and the shorter "hot" (1/T <= inv_tmin2).
struct FiltDesc {
    int off, cnt, coff, ccnt, hoff, hcnt;
    float r_min;   // interval coordinate (ln T - itab_u0) / h from which the filter's interpolant is proved; +inf: none
    int ioff;      // its coefficients in the interpolant array, in doubles
    double inv_tmin, inv_tmin2;  // 0 = that level does not exist
    // companion-shocking models: which parameters are the filter's factors (models.py:909-917), -1 = none.  Here, next
    // to the rest of what a point reads about its filter: from the staged copy these are LDS reads, from three separate
    // arrays they were three dependent scalar loads per point.
    int kpar, spar, dtpar, pad;
};
constexpr int kFdD2 = 4;   // double2 entries per descriptor
static_assert(sizeof(FiltDesc) == 16 * kFdD2, "whole 16-byte entries per filter in LDS");

struct DevProblem {
    int model, n_points, n_chunks, n_filters;
    int n_dim, n_par, use_sigma, sigma_abs;
    int n_knots, has_priors, tab_in_lds, variant;
    int n_epochs, use_therm, use_ctab, n_tab;
    int n_lds_tab;    // samples of `tab` staged in LDS: all, or the compressed levels only (they come first)
    int redden_slow;  // ShockCooling3 whose tables do not fit in LDS: reddening applied per sample from global memory
    int n_parts, cpb;  // workgroups per walker ("parts"), most point chunks in one part
    // Third table level: ln S_f(T) of every filter as piecewise polynomials of degree 7 in u = ln T on itab_m equal
    // intervals from itab_u0 (packed and proved by the host, filters.interp_planck_table).  With it a model whose
    // thermal state is known in log space costs one lookup + 7 FMA + ONE exponential per data point.
    int use_itab, itab_m;
    int itab_uniform;     // every filter's interpolant is proved from the first interval on (no per-filter threshold)
    int n_itab_lds;       // doubles of `itab` staged in LDS behind the filter descriptors (all of it, or 0)
    int stage_d2;         // double2 entries of the staged region: tables + descriptors + interpolants
    double itab_u0, itab_inv_h, itab_umax;
    const double* itab;   // [n_filters][itab_m][8], highest power first
    int part_start[kMaxParts + 1];  // part j owns the points [part_start[j], part_start[j+1]): whole epochs ...
    int part_ep0[kMaxParts + 1];    // ... namely the epochs [part_ep0[j], part_ep0[j+1]) (when use_therm)
    // Epoch-major copy of the photometry (engines with use_therm; the likelihood walks it, one lane per COLUMN): a column
    // is one observation time with up to em_k of its points, ordered by filter; an epoch with more points than that
    // takes several columns.  The lane computes the column's thermal state once, in registers, and loops over the
    // column's points: em_yd / em_filt are [em_k][em_cols] (consecutive lanes read consecutive addresses).
    int part_col0[kMaxParts + 1];   // part j owns the columns [part_col0[j], part_col0[j+1])
    int em_k, em_cols;
    int em_dense;                   // every column holds exactly one point of every filter, in filter order
    const double* em_t;             // [em_cols] observation time
    const double2* em_yd;           // [em_k][em_cols] (y, 1/dy), or (y, dy) when sigma is fitted
    const int* em_filt;             // [em_k][em_cols] filter index, -1 = no point
    double consts[12];
    double log_norm_const;  // sum_i ln(2 pi dy_i^2), used when there is no sigma parameter
    double knot_inv_h;      // 1 / spacing of the spline knots when they are equally spaced (to 1e-9), else 0
    // Template splines whose knots are EXACTLY k0 + i h in float64 (the SiFTO template: integer days): the interval of an
    // argument is computed, not searched, and no knot is read.  knot_h = 0: not such a spline.
    double knot0, knot_last, knot_h;
    int n_spl_lds, pad6;    // doubles of `spl` staged in LDS behind the interpolants (all of it, or 0)
    double sigma_unit_abs;  // median(dy)
    // points, ordered by (part, filter)
    const double* t;
    const double* y;
    const double* dy;
    const double* inv_dy;     // 1 / dy
    const FiltDesc* f_desc;   // [n_filters] where each filter's tables are, and from which temperature they hold
    const int* pt_filt;  // filter index
    const int* pt_orig;  // index in the caller's order
    const int* pt_epoch; // index into epoch_t (distinct observation times)
    const int* pt_fe;    // filter | epoch << 6 in one word (engines of at most 64 filters), else null
    const double2* pt_yd;  // (y, 1/dy) per point -- (y, dy) when sigma is fitted -- for one 16-byte load
    const double* epoch_t; // [n_epochs]
    const double* exp2tab;   // 2^(j/256), j = 0..255
    const double2* stage_image;  // [exp table | 32 doubles | first n_lds_tab samples of tab | f_desc | staged itab | splines]
    int stage_n16, pad5;         // its length in 16-byte units (the exp table and its pad only when !tab_in_lds)
    const double2* tab;  // (a_k, W_k)
    const double* tab_ext;  // ShockCooling3: 0.4 log2(10) A_k / E(B-V) per table sample (0 for padding), else null
    const double* knots;
    const double* spl;  // [n_filters][n_knots-1][4]
    const PriorDev* priors;
};

// Doubles per walker / proposal slot in a buffer of partial chi^2 sums: one per part, then one for the log-prior.
__device__ __host__ inline int part_stride(const DevProblem& pb) { return pb.n_parts + 1; }

// a[j] for a small array that lives in the kernel arguments: a chain of scalar selects (indexing it dynamically
// would make the compiler copy the array to scratch memory)
__device__ inline int part_entry(const int (&a)[kMaxParts + 1], int j) {
    int v = a[0];
#pragma unroll
    for (int k = 1; k <= kMaxParts; ++k) v = j == k ? a[k] : v;
    return v;
}

__device__ inline double qnan() { return __longlong_as_double(0x7ff8000000000000LL); }

// power() of the reference: base > 0 ? base**exp : 0, also for NaN bases.  models.py:42-48
__device__ inline double pw(double base, double e) { return base > 0. ? pow(base, e) : 0.; }

// ---------------------------------------------------------------------------------------------------------------
// Band sum  S(T) = sum_k W_k / (exp(a_k / T) - 1)           filters.py:308-310 + models.py:1127-1128
// ---------------------------------------------------------------------------------------------------------------

// Variant 0: libm, shaped like the reference (one exp-minus-one and one divide per sample).
template <class TabPtr>
__device__ inline double band_sum_ref(TabPtr tab, int cnt, double invT) {
    double acc = 0.;
    for (int k = 0; k < cnt; ++k) {
        const double2 aw = tab[k];
        acc += aw.y / expm1(aw.x * invT);
    }
    return acc;
}

// Variant 1 (default).  exp(+-x) through t = +-x * 256/ln2 = n + r (n integer, |r| <= 1/2):
//   exp(+-x) = 2^(n >> 8) * 2^((n & 255)/256) * exp(r ln2/256)
// with a 256-entry table of 2^(j/256) in LDS (2 KiB) and a degree-4 polynomial (truncation 3.8e-17).  Relative
// error ~2e-16 + |x| * 2.2e-16 (the second term is the conditioning of exp: x itself carries one rounding).
// MAIN path (every sample of the wave has x < 170): E = e^x, term = W/(E - 1), the binary exponent is added with
//   one integer add; the four denominators of a quad multiply to < e^680, no overflow.  E - 1 has the same
//   cancellation behaviour as the reference's own exp(x) - 1.
// SAFE path (some x >= 170, i.e. T below ~0.3 kK): u = e^-x, term = W u/(1 - u), x clamped so that the result
//   underflows gradually and is exactly 0 from x = 746.5 on (like the reference's 1/inf), through v_ldexp_f64.
// The path is chosen per wave = 64 consecutive points of ONE walker, so a walker's result never depends on how
// walkers are batched or sharded.
struct ExpTab {
    const double* t;  // LDS: t[j] = 2^(j/256), j = 0..255
};

constexpr int kExpTabSize = 256;
// Head of every kernel's dynamic LDS: the exp table, then 32 doubles for the wave sums of a reduction; the staged
// region (band tables, filter descriptors, interpolants) follows.  DevProblem::stage_image is a copy of all of it in
// global memory, in this layout, so that staging is ONE flat copy with all its loads in flight together.
constexpr int kLdsHead = kExpTabSize + 32;   // doubles (4 wave sums for each of up to kMaxParts parts)
constexpr double kInvLn2N = 369.3299304675746;  // 256 / ln 2
constexpr double kQ1 = 0.0027076061740622863, kQ2 = 3.665565596910106e-06, kQ3 = 3.308302680541371e-09, kQ4 = 2.239395190875157e-12;

template <bool SAFE>
__device__ inline double exp_scaled(double tp, const ExpTab et) {
    if (SAFE) tp = fmax(tp, -746.5 * kInvLn2N);
    const double nf = rint(tp);
    const double r = tp - nf;  // exact
    const int n = (int)nf;
    const double tj = et.t[n & (kExpTabSize - 1)];
    double p = fma(r, kQ4, kQ3);
    p = fma(p, r, kQ2);
    p = fma(p, r, kQ1);
    p = fma(p, r, 1.0);
    const double v = p * tj;  // in [0.99, 2.01)
    const int e = n >> 8;
    if (SAFE) return ldexp(v, e);
    return __hiloint2double(__double2hiint(v) + (int)((unsigned)e << 20), __double2loint(v));
}

// n / d for positive normal d: reciprocal seed + two Newton steps (error ~1 ulp).
__device__ inline double div_pos(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}

// Natural logarithm for the per-walker / per-epoch model arithmetic of the log-space state: libm's costs ~90 mostly
// dependent FP64 instructions, this one ~32 with an error of ~1 ulp.  x = m 2^e with m in [sqrt(1/2), sqrt(2));
// ln m = 2 atanh(s), s = (m - 1)/(m + 1), |s| <= 0.1716: odd series through s^21 (truncation < 3e-17 relative);
// e ln 2 added as a hi/lo pair.  Subnormal arguments included; flog(+inf) = +inf; NaN for x <= 0 and NaN (no caller
// uses the logarithm of a non-positive number).
__device__ inline double flog(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.7071067811865476;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double s = div_pos(m - 1., m + 1.);
    const double z = s * s;
    double p = 1. / 21.;
    p = fma(p, z, 1. / 19.);
    p = fma(p, z, 1. / 17.);
    p = fma(p, z, 1. / 15.);
    p = fma(p, z, 1. / 13.);
    p = fma(p, z, 1. / 11.);
    p = fma(p, z, 1. / 9.);
    p = fma(p, z, 1. / 7.);
    p = fma(p, z, 1. / 5.);
    p = fma(p, z, 1. / 3.);
    const double lnm = fma(s * z, p + p, s + s);
    const double ef = (double)e;
    const double r = fma(ef, 0.6931471803691238, fma(ef, 1.9082149292705877e-10, lnm));  // ln 2 = hi (32 bits) + lo
    return x > 0. ? (x < INFINITY ? r : x) : __longlong_as_double(0x7ff8000000000000LL);
}

// Four samples share one division:  n1/d1 + n2/d2 = (n1 d2 + n2 d1)/(d1 d2).
// Tables are padded to a multiple of four samples with zero weights (contribute exactly 0).
template <class TabPtr>
__device__ inline double band_sum_main(TabPtr tab, int cnt, double spos, const ExpTab et) {
    double acc = 0.;
    for (int k = 0; k < cnt; k += 4) {
        const double2 s0 = tab[k], s1 = tab[k + 1], s2 = tab[k + 2], s3 = tab[k + 3];
        const double d0 = exp_scaled<false>(s0.x * spos, et) - 1., d1 = exp_scaled<false>(s1.x * spos, et) - 1.;
        const double d2 = exp_scaled<false>(s2.x * spos, et) - 1., d3 = exp_scaled<false>(s3.x * spos, et) - 1.;
        const double n01 = fma(s0.y, d1, s1.y * d0), d01 = d0 * d1;
        const double n23 = fma(s2.y, d3, s3.y * d2), d23 = d2 * d3;
        acc += div_pos(fma(n01, d23, n23 * d01), d01 * d23);
    }
    return acc;
}

template <class TabPtr>
__device__ inline double band_sum_safe(TabPtr tab, int cnt, double sneg, const ExpTab et) {
    double acc = 0.;
    for (int k = 0; k < cnt; k += 4) {
        const double2 s0 = tab[k], s1 = tab[k + 1], s2 = tab[k + 2], s3 = tab[k + 3];
        const double u0 = exp_scaled<true>(s0.x * sneg, et), u1 = exp_scaled<true>(s1.x * sneg, et);
        const double u2 = exp_scaled<true>(s2.x * sneg, et), u3 = exp_scaled<true>(s3.x * sneg, et);
        const double d0 = 1. - u0, d1 = 1. - u1, d2 = 1. - u2, d3 = 1. - u3;  // all in (0, 1]
        const double n01 = fma(s0.y * u0, d1, s1.y * u1 * d0), d01 = d0 * d1;
        const double n23 = fma(s2.y * u2, d3, s3.y * u3 * d2), d23 = d2 * d3;
        acc += div_pos(fma(n01, d23, n23 * d01), d01 * d23);
    }
    return acc;
}

template <class TabPtr>
__device__ inline double band_sum_fast(TabPtr tab, int cnt, double invT, const ExpTab et) {
    if (cnt <= 0) return 0.;
    const double amax = fmax(tab[0].x, tab[cnt - 1].x);  // tables are monotonic in a_k
    const double spos = invT * kInvLn2N;
    const bool needs_safe = !(amax * invT < 170.);
    if (__builtin_amdgcn_ballot_w64(needs_safe) == 0) return band_sum_main(tab, cnt, spos, et);
    return band_sum_safe(tab, cnt, -spos, et);
}

// ---------------------------------------------------------------------------------------------------------------
// Per-walker derived coefficients (wave-uniform).  One thread per walker computes them once; the likelihood kernel
// reads them through scalar loads.
//   c[0] = explosion time t_0
//   power-law thermal models (ShockCooling, ShockCooling2):
//     T_K(t) = c[1] t^eT,  L(t) = c[2] t^eL exp(-(g t)^alpha) with c[3] = alpha ln g (NaN: no cut-off term),
//     c[4] != 0: a negative phase yields NaN (L < 0 before the explosion)
//   ShockCooling4:  c[1] = T_col_br/k_B, c[2] = L_br, c[3] = t_br, c[4] = t_tr
//   Companion:      c[1] = T coefficient, c[2] = R^2 coefficient, c[3] = t_peak, c[4] = stretch, c[5] = shock factor
//   Blackbody:      c[1] = T, c[2] = R^2
//   Every model but ShockCooling3 also leaves the logarithms of its two amplitudes for the log-space thermal state
//   (thermal_state_log):  c[6] = ln(temperature coefficient),  c[7] = ln(coefficient of R_bb^2) -- for the
//   shock-cooling family ln(c3^2 L coefficient), so that ln R_bb^2 = c[7] + ... - 4 ln T -- or NaN where the amplitude
//   is not positive and finite (the linear-space rules of thermal_state then decide).
// ---------------------------------------------------------------------------------------------------------------
// `lq[d]` = log(p[d]) for d < n_par (may be NaN/-inf where p[d] <= 0: only used when the parameter is positive).
// `log_only`: the caller's thermal states stay in log space (thermal_state_log of a power-law model): the linear
// amplitudes c[1], c[2] -- one exponential each -- are then not needed and are left NaN.
// MODEL > 0: the model is a compile-time constant (kernels specialised for one model: the other models' code is not
// even in the instruction stream); 0: pb.model decides at run time.
template <int MODEL = 0>
__device__ inline void walker_coefficients(const DevProblem& pb, const double* __restrict__ p,
                                           const double* __restrict__ lq, double* __restrict__ c,
                                           bool log_only = false) {
    const double* k = pb.consts;
    const int model = MODEL ? MODEL : pb.model;
    for (int i = 0; i < kNCoef; ++i) c[i] = 0.;
    if (model != kShockCooling3) c[6] = c[7] = qnan();
    switch (model) {
        case kShockCooling:
        case kShockCooling3: {  // models.py:260-267; ShockCooling3 (models.py:493-495): + distance and reddening
            const double A = k[0], a = k[1], alpha = k[2], eps1 = k[3], eps2 = k[4], L0 = k[5], T0 = k[6], ratio = k[7];
            const double v = p[0], M = p[1], f = p[2], R = p[3];
            if (model == kShockCooling3) {
                c[0] = p[6];
                c[5] = kC4 / (p[4] * p[4]);  // flux = c4 * lum / dist ** 2
                c[6] = p[5];                 // E(B-V): scales the band-table weights (points_body)
            } else {
                c[0] = p[4];
            }
            if (v > 0. && M > 0. && f > 0. && R > 0. && v < 1e100 && M < 1e100 && f < 1e100 && R < 1e100) {
                // all bases positive: every power() is a plain power; share the four logarithms
                const double lv = lq[0], lM = lq[1], lf = lq[2], lR = lq[3];
                const double a1 = eps1 * (2. * lv - lf) + 0.25 * lR, a2 = -eps2 * (lv - lf) + 2. * lv + lR;
                const bool skip = log_only && model == kShockCooling;
                c[1] = skip ? qnan() : (T0 * ratio / kKB) * exp(a1);
                c[2] = skip ? qnan() : (L0 * A) * exp(a2);
                c[3] = a > 0. ? alpha * (k[11] - 0.5 * (lM - lv)) : qnan();  // k[11] = ln(a / 19.5), set at create
                c[4] = 0.;
                if (model == kShockCooling) {  // k[9] = ln(T0 ratio / k_B), k[10] = ln(c3^2 L0 A), set at create
                    c[6] = k[9] + a1;
                    c[7] = k[10] + a2;
                }
                break;
            }
            const double Lc = L0 * pw(v / f, -eps2) * v * v * R;  // L_RW = Lc * |t|^(-2 eps2)
            const double t_tr = 19.5 * sqrt(M / v);
            const double g = a / t_tr;
            c[1] = T0 * pw(v * v / f, eps1) * pow(R, 0.25) * ratio / kKB;
            c[2] = Lc * A;
            c[3] = g > 0. ? alpha * log(g) : qnan();
            c[4] = (Lc >= 0.) ? 0. : 1.;
            if (model == kShockCooling) {
                c[6] = (c[1] > 0. && c[1] < INFINITY) ? log(c[1]) : qnan();
                c[7] = (c[2] >= 0. && c[2] < INFINITY) ? log(kC3sq * c[2]) : qnan();
            }
            break;
        }
        case kShockCooling2: {  // models.py:403-406
            const double a = k[1], alpha = k[2];
            const double g = a / p[2];
            c[0] = p[3];
            c[1] = p[0];
            c[2] = p[1] * 1e42;
            c[3] = g > 0. ? alpha * log(g) : qnan();
            c[6] = (c[1] > 0. && c[1] < INFINITY) ? lq[0] : qnan();
            c[7] = (c[2] >= 0. && c[2] < INFINITY) ? log(kC3sq * c[2]) : qnan();
            break;
        }
        case kShockCooling4: {  // models.py:584-587 (quirks kept: no kappa in t_br, right-associative ** chain)
            const double v = p[0], M = p[1], f = p[2], R = p[3];
            c[0] = p[4];
            c[4] = k[6] * sqrt(M / v);
            if (v > 0. && f > 0. && R > 0. && v < 1e100 && f < 1e100 && R < 1e100) {
                const double lv = lq[0], lf = lq[2], lR = lq[3];
                const double ex = exp(-0.5447271754416722 * exp(0.03 * lf));  // 0.58 ** (f ** 0.03)
                c[1] = (k[4] / kKB) * exp(-0.32 * lR + ex * lv);
                c[2] = k[3] * exp(0.78 * lR + 2.11 * lv + 0.11 * lf);
                c[3] = k[5] * exp(1.26 * lR - 1.13 * lv - 0.13 * lf);
                break;
            }
            c[1] = k[4] * pow(R, -0.32) * pow(v, pow(0.58, pow(f, 0.03))) / kKB;
            c[2] = k[3] * pow(R, 0.78) * pow(v, 2.11) * pow(f, 0.11);
            c[3] = k[5] * pow(R, 1.26) * pow(v, -1.13) * pow(f, -0.13);
            break;
        }
        case kCompanion:
        case kCompanion2:
        case kCompanion3: {  // models.py:752-755, 1040-1044
            const double a13 = p[1];
            const double Mv = model == kCompanion3 ? 1. : p[2];
            c[0] = p[0];
            if (a13 > 0. && Mv > 0. && a13 < 1e8 && a13 > 1e-8 && Mv < 1e100) {
                const double la = lq[1], lm = model == kCompanion3 ? 0. : lq[2];
                c[1] = log_only ? qnan() : 25. * exp((36. * la + lm) * (1. / 144.));
                c[2] = log_only ? qnan() : 7.29 * exp(lm * (2. / 9.));
                c[6] = 3.2188758248682006 + (36. * la + lm) * (1. / 144.);  // ln 25
                c[7] = 1.9865035460205669 + lm * (2. / 9.);                  // ln 7.29
            } else {
                c[1] = 25. * pw(pow(a13, 36.) * Mv, 1. / 144.);
                const double rc = 2.7 * pw(Mv, 1. / 9.);
                c[2] = rc * rc;
                c[6] = (c[1] > 0. && c[1] < INFINITY) ? log(c[1]) : qnan();
                c[7] = (c[2] >= 0. && c[2] < INFINITY) ? log(c[2]) : qnan();
            }
            c[3] = p[3];
            c[4] = p[4];
            if (model == kCompanion3) {
                const double th = p[2] * 0.017453292519943295;
                c[5] = (0.5 * cos(th) + 0.5) * (0.14 * th * th - 0.4 * th + 1.);
            } else {
                c[5] = 1.;
            }
            break;
        }
        case kBlackbody:
            c[1] = p[0];
            c[2] = p[1] * p[1];
            break;
        default:
            break;
    }
}

// Convenience for one-thread-per-walker callers: logarithms computed in place.
__device__ inline void walker_coefficients(const DevProblem& pb, const double* __restrict__ p, double* __restrict__ c) {
    double lq[kMaxDim];
    for (int d = 0; d < pb.n_par; ++d) lq[d] = flog(p[d]);  // (the serial heads of the sampler take the same logarithm)
    walker_coefficients(pb, p, lq, c);
}

// One parameter's log-prior term; -inf outside the strict bounds.  models.py:1055-1098
__device__ inline double prior_term(const PriorDev& pr, double x) {
    if (!(pr.p_min < x && x < pr.p_max)) return -INFINITY;
    if (pr.kind == 1) return -log(x);
    if (pr.kind == 2) {
        const double u = (x - pr.mean) / pr.stddev;
        return -0.5 * u * u;
    }
    return 0.;
}

// log-prior of one walker; -inf outside the strict bounds.  models.py:1055-1098, fitting.py:122-126
__device__ inline double walker_log_prior(const DevProblem& pb, const double* __restrict__ p) {
    if (!pb.has_priors) return 0.;
    double lp = 0.;  // same ordered sum of the same terms as the lane-parallel form in k_step
    for (int i = 0; i < pb.n_dim; ++i) lp += prior_term(pb.priors[i], p[i]);
    return lp;
}

// Piecewise-cubic SiFTO template, 0 outside the knot range (and for NaN arguments).  models.py:717, 816-826
// `inv_h` > 0: the knots are (nearly) equally spaced, h = 1 / inv_h apart -- the interval is then computed and
// corrected against the neighbouring knots instead of searched (the SiFTO template: 103 knots one day apart).
__device__ inline double spline_eval(const double* __restrict__ knots, int nk, const double* __restrict__ coef,
                                     double x, double inv_h) {
    const double k0 = knots[0];
    if (!(x >= k0 && x <= knots[nk - 1])) return 0.;
    int lo;  // the interval of x: the largest lo <= nk - 2 with knots[lo] <= x
    if (inv_h > 0.) {
        lo = min(max((int)((x - k0) * inv_h), 0), nk - 2);
        while (lo < nk - 2 && x >= knots[lo + 1]) ++lo;
        while (lo > 0 && x < knots[lo]) --lo;
    } else {
        int hi = nk - 1;  // invariant: knots[lo] <= x <= knots[hi]
        lo = 0;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (x >= knots[mid]) lo = mid; else hi = mid;
        }
    }
    const double dx = x - knots[lo];
    const double* q = coef + 4 * lo;
    return fma(fma(fma(q[0], dx, q[1]), dx, q[2]), dx, q[3]);
}

// The same for knots that are exactly k0 + i h (DevProblem::knot_h > 0), coefficients at `coef` (the filter's, any
// address space): no knot is read, the interval comes from one multiplication.  An argument within a rounding of a knot may
// land in the neighbouring interval with dx = -1e-16 or h + 1e-16: the two cubics agree there to the third order.
template <class CoefPtr>
__device__ __forceinline__ double spline_eval_uniform(const DevProblem& pb, CoefPtr coef, double x) {
    if (!(x >= pb.knot0 && x <= pb.knot_last)) return 0.;
    const int lo = min(max((int)((x - pb.knot0) * pb.knot_inv_h), 0), pb.n_knots - 2);
    const double dx = x - fma((double)lo, pb.knot_h, pb.knot0);
    const double2 q01 = coef[2 * lo], q23 = coef[2 * lo + 1];
    return fma(fma(fma(q01.x, dx, q01.y), dx, q23.x), dx, q23.y);
}

// Temperature [kK], its reciprocal, and the factor `pref` such that  L_nu = pref * S(T)  (pref = R_bb^2), for one
// data point.  Sets T = 1/T = 0 (and pref = 0 or NaN) where the reference's power() zeroing makes the band integral
// vanish.
__device__ inline double inv_temperature(double Tk) {  // 1 / Tk for 0 < Tk < kTmax
    return 1. / Tk;
}

template <int MODEL = 0>
__device__ inline void thermal_state(const DevProblem& pb, const double* __restrict__ c, double t_in, double& T,
                                     double& invT, double& pref) {
    const double* k = pb.consts;
    const int model = MODEL ? MODEL : pb.model;
    const double t = t_in - c[0];
    T = 0.;
    invT = 0.;
    pref = 0.;
    switch (model) {
        case kShockCooling:
        case kShockCooling2:
        case kShockCooling3: {
            const double eps1 = k[3], eps2 = k[4], alpha = k[2];
            if (t > 0.) {
                const double lt = log(t);
                const double Tk = c[1] * exp((2. * eps1 - 0.5) * lt);
                const double E = (c[3] == c[3]) ? exp(fma(alpha, lt, c[3])) : 0.;
                const double L = c[2] * exp(-2. * eps2 * lt - E);
                if (!(L >= 0.)) {
                    pref = qnan();
                } else if (Tk > 0. && Tk < kTmax) {
                    const double i1 = inv_temperature(Tk), i2 = i1 * i1;
                    T = Tk;
                    invT = i1;
                    pref = kC3sq * L * i2 * i2;  // R_bb^2 = c3^2 L T^-4      models.py:268
                }  // else: T <= 0 or NaN -> power(T, -2) = 0 -> R_bb = 0
            } else if (t < 0. && c[4] != 0.) {
                pref = qnan();  // sqrt(L) with L < 0 before the explosion
            }
            break;
        }
        case kShockCooling4: {  // models.py:588-597
            const double tt = t / c[3];
            double P1 = 0., P2 = 0., P3 = 0., P4 = 0.;
            if (tt > 0.) {
                const double ltt = log(tt);
                P1 = exp(-4. / 3. * ltt);
                P2 = exp(-0.17 * ltt);
                P3 = exp(-1. / 3. * ltt);
                P4 = exp(-0.45 * ltt);
            }
            const double u = k[1] * t / c[4];
            const double E = u > 0. ? pow(u, k[2]) : 0.;
            const double L = c[2] * (P1 + k[0] * exp(-E) * P2);
            const double Tk = c[1] * fmin(0.97 * P3, P4);
            if (!(L >= 0.)) {
                pref = qnan();
            } else if (Tk > 0. && Tk < kTmax) {
                const double i1 = inv_temperature(Tk), i2 = i1 * i1;
                T = Tk;
                invT = i1;
                pref = kC3sq * L * i2 * i2;
            }
            break;
        }
        case kCompanion:
        case kCompanion2:
        case kCompanion3: {  // models.py:752-755
            if (t > 0.) {
                const double lt = log(t);
                // power(t, -74) overflows to inf below t ~ 6.8e-5 d (T = inf -> band integral 0) and underflows to 0
                // above t ~ 2.3e4 d (T = 0), exactly as in the reference's evaluation order.
                if (lt > -9.5916 && lt < 10.06) {
                    const double Tk = c[1] * exp(-74. / 144. * lt);
                    if (Tk > 0. && Tk < kTmax) {
                        T = Tk;
                        invT = inv_temperature(Tk);
                        pref = c[2] * exp(14. / 9. * lt);  // R^2 = (2.7 (Mv t^7)^(1/9))^2
                    }
                }
            }
            break;
        }
        case kBlackbody:
            if (c[1] > 0. && c[1] < kTmax) {
                T = c[1];
                invT = inv_temperature(c[1]);
                pref = c[2];
            } else if (c[2] != c[2]) {
                pref = qnan();
            }
            break;
        default:
            break;
    }
}

// The same thermal state in LOG space, for the interpolated band sums (DevProblem::itab): a pair (x, p) with
//   x >= +0:          x = (ln T_K - itab_u0) / h, the temperature as a coordinate on the interpolants' intervals (its
//                     integer part is the interval every filter's polynomial is taken from, 0 <= x < itab_m: 2 to
//                     256 kK), p = ln R_bb^2 with |p| <= 600
//   sign bit of x set: x = -1/T_K (-0.0 where the band integral vanishes), p = R_bb^2 (0, or NaN to propagate) -- the
//                      linear-space state, for the sample-table band sum.
// Power-law models never leave log space: ln T = c[6] + eT ln t and ln R_bb^2 = c[7] + eL ln t - E - 4 ln T come from
// ONE logarithm and at most one exponential per epoch.  The special cases follow thermal_state rule for rule.
// Logarithm / exponential of the per-epoch state: the short versions (flog; the exponential through the 2^(j/256) table).
// Measured at 1024 walkers x 3000 points, kernel time per half-step: libm + libm 12.58 us, flog + libm 12.34, libm +
// table 12.50, flog + table 12.20.  (With machine LICM on -- before the compiler flag of the Makefile -- it was the
// other way round: the short versions' constants were hoisted out of the epoch loop into registers and pushed the
// point loop into spills, 3.20e7 against 3.73e7 walker-steps/s.)
#ifndef LCF_TLOG
#define LCF_TLOG 1   // 1 = flog, 0 = libm
#endif
#ifndef LCF_TEXP
#define LCF_TEXP 1   // 1 = through the 2^(j/256) table, 0 = libm
#endif
__device__ inline double tlog(double x) { return LCF_TLOG ? flog(x) : log(x); }

__device__ inline void encode_linear(double invT, double pref, double& x, double& p) {
    x = -invT;
    p = pref;
}

// Log-space state of a power-law model (ShockCooling, ShockCooling2) at a POSITIVE phase t: u = ln T_K and lp = ln R_bb^2
// from one logarithm and at most one exponential; lL = ln(c3^2 L), NaN where L < 0 or NaN (and everything is NaN for
// t <= 0: the callers decide on the phase themselves).  c3, c6, c7 = the walker's coefficients c[3], c[6], c[7].
__device__ __forceinline__ void powerlaw_log_state(const DevProblem& pb, double c3, double c6, double c7, double t,
                                                   const ExpTab et, double& u, double& lp, double& lL) {
    const double* k = pb.consts;
    const double eps1 = k[3], eps2 = k[4], alpha = k[2];
    const double lt = tlog(t);
    // (the exponential through the 2^(j/256) table: same arithmetic whether `et` points to LDS or memory)
    // (argument clamped from above: beyond e^1100 the table exponential overflows to +inf through its ldexp, as libm's
    // does -- an infinite argument, M_env = 0, would otherwise be inf - inf = NaN inside it, where the reference's
    // exp(-inf) makes the luminosity 0)
    const double E = !(c3 == c3) ? 0.
                     : LCF_TEXP ? exp_scaled<true>(fmin(fma(alpha, lt, c3) * kInvLn2N, 1100. * kInvLn2N), et)
                                : exp(fma(alpha, lt, c3));
    lL = fma(-2. * eps2, lt, c7) - E;
    u = fma(2. * eps1 - 0.5, lt, c6);
    lp = fma(-4., u, lL);
}

template <int MODEL = 0>
__device__ inline void thermal_state_log(const DevProblem& pb, const double* __restrict__ c, double t_in, double& x,
                                         double& p, const ExpTab et) {
    const int model = MODEL ? MODEL : pb.model;
    const double t = t_in - c[0];
    x = -0.;
    p = 0.;
    double u = qnan(), lp = 0.;  // ln T, ln R_bb^2 when both exist
    switch (model) {
        case kShockCooling:
        case kShockCooling2: {
            if (t > 0.) {
                double lL;
                powerlaw_log_state(pb, c[3], c[6], c[7], t, et, u, lp, lL);
                if (!(lL == lL)) {
                    p = qnan();          // L < 0 or NaN
                    return;
                }
            } else {
                if (t < 0. && c[4] != 0.) p = qnan();  // sqrt(L) with L < 0 before the explosion
                return;
            }
            break;
        }
        case kCompanion:
        case kCompanion2:
        case kCompanion3: {
            if (!(t > 0.)) return;
            const double lt = tlog(t);
            if (!(lt > -9.5916 && lt < 10.06)) return;   // power(t, -74) over- / underflows: zero band integral
            u = fma(-74. / 144., lt, c[6]);
            lp = fma(14. / 9., lt, c[7]);
            if (!(lp == lp)) u = qnan();                   // (M v^7 <= 0: R = 0 -> nothing)
            break;
        }
        default: {  // ShockCooling4, Blackbody: linear-space state, then two logarithms
            double T, invT, pref;
            thermal_state<MODEL>(pb, c, t_in, T, invT, pref);
            if (T > 0. && pref > 0. && pref < INFINITY) {
                u = log(T);
                lp = log(pref);
            } else {
                encode_linear(invT, pref, x, p);
                return;
            }
            break;
        }
    }
    if (!(u == u) || !(u < 34.538776394910684)) return;  // T <= 0, NaN or >= 1e15 kK: zero band integral
    if (lp < -600.) return;                                // R_bb^2 < e^-600: nothing (the reference: ~1e-260 of a datum)
    const double r = (u - pb.itab_u0) * pb.itab_inv_h;
    if (r >= 0. && r < (double)pb.itab_m && lp <= 600.) {
        x = r;
        p = lp;
    } else if (u > -50.) {  // outside the interpolants' range: the sample tables, in linear space
        encode_linear(exp(-u), exp(lp), x, p);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11): counter-based, so a draw depends only on (seed, walker, step, half).
// ---------------------------------------------------------------------------------------------------------------
__device__ inline void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                  uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform in (0,1) from two words: 52 random bits + 1/2, exactly representable
__device__ inline double u01(uint32_t hi, uint32_t lo) {
    const uint64_t v = ((uint64_t)hi << 20) ^ ((uint64_t)lo >> 12);
    return ((double)v + 0.5) * (1. / 4503599627370496.);
}

}  // namespace lcf
