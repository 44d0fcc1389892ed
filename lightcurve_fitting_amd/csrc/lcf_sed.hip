// Per-epoch blackbody SED likelihood on gfx950: the band-integrated Planck primitive with (T, R[, sigma]) as direct
// parameters, batched over (epoch x candidate).  Replaces the inner log_posterior of the reference's
// bolometric.spectrum_mcmc (bolometric.py:154-164): for every epoch, [f.synthesize(planck_fast, T, R) for f in the
// epoch's filters] followed by the Gaussian log-likelihood.
//
// Work decomposition: workgroup = (epoch, tile of 128 candidates); lane = one candidate (T, R): loops over the epoch's
// observations and, per observation, over the filter's samples staged in LDS.  Two arithmetic modes:
//   precision 0: float64 (same band sum as the light-curve engine, parity 1e-11)
//   precision 1: float32 (v_exp_f32 / v_rcp_f32, accumulate in f32) -- BASELINE configs[3]
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
    int n_filters, n_tab, tab_in_lds, pad;
    const double2* tab;    // (a, W) float64: per filter [full | Gauss-compressed], each padded to quads
    const float2* tab32;   // same in float32
    const int4* desc;      // [n_filters] (full offset, full count, compressed offset, compressed count)
    const double* inv_tmin; // [n_filters] 1 / t_min of the compressed table (0: none)
    const double* exp2tab; // 2^(j/256)
};

struct SedObs {
    long long n_epochs;
    const int* ep_off;      // [n_epochs + 1]
    const int* filt;        // [n_obs]
    const double* y;        // [n_obs]
    const double* dy;       // [n_obs]
    const double* dy_med;   // [n_epochs] median(dy) of the epoch (sigma_type 'absolute')
};

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
        if (stream) hipStreamDestroy(stream);
    }
};

extern "C" {

lcf_status lcf_sed_create(int32_t n_filters, const int32_t* tab_off, const double* tab_a, const double* tab_w,
                          const int32_t* ctab_off, const double* ctab_a, const double* ctab_w,
                          const double* ctab_tmin, int32_t device, lcf_sed** out) {
    if (!out) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
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
    lcf_status st;
    double2* dtab;
    float2* dtab32;
    int4* ddesc;
    double *dexp, *dinvt;
#define UP(h, d) if ((st = upload(h, &d, s->owned)) != LCF_OK) { delete s; return st; }
    UP(htab, dtab); UP(htab32, dtab32); UP(hdesc, ddesc); UP(hexp, dexp); UP(hinvt, dinvt);
#undef UP
    s->sd = SedDev{n_filters, (int)htab.size(), (int)htab.size() <= kSedLdsMax, 0, dtab, dtab32, ddesc, dinvt, dexp};
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
    lcf_status st;
    int *doff, *df;
    double *dy_, *ddy, *dmed;
#define UP(h, d) if ((st = upload(h, &d, s->obs_owned)) != LCF_OK) return st
    UP(hoff, doff); UP(hf, df); UP(hy, dy_); UP(hdy, ddy); UP(hmed, dmed);
#undef UP
    s->ob = SedObs{n_epochs, doff, df, dy_, ddy, dmed};
    s->n_obs = n_obs;
    return LCF_OK;
}

lcf_status lcf_sed_log_likelihood(lcf_sed* s, int64_t n_cand, int32_t n_par, int32_t sigma_type, const double* cand,
                                  int32_t precision, int32_t use_compressed, double* out, double* kernel_ms) {
    if (!s || n_cand < 0 || (n_cand > 0 && (!cand || !out))) return fail(LCF_ERR_INVALID_ARGUMENT, "null argument");
    if (n_par != 2 && n_par != 3) return fail(LCF_ERR_INVALID_ARGUMENT, "n_par must be 2 (T, R) or 3 (T, R, sigma)");
    if (precision != 0 && precision != 1) return fail(LCF_ERR_INVALID_ARGUMENT, "precision must be 0 (f64) or 1 (f32)");
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
        s->dout = nullptr;
        s->out_cap = 0;
        LCF_HIP(hipMalloc((void**)&s->dout, no * sizeof(double)));
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
    if (precision == 0)
        hipLaunchKernelGGL(k_sed<0>, grid, dim3(kSedBlock), lds, s->stream, s->sd, s->ob, (long long)n_cand, n_par,
                           sigma_type == LCF_SIGMA_ABSOLUTE, uc, s->dcand, s->dout);
    else
        hipLaunchKernelGGL(k_sed<1>, grid, dim3(kSedBlock), lds, s->stream, s->sd, s->ob, (long long)n_cand, n_par,
                           sigma_type == LCF_SIGMA_ABSOLUTE, uc, s->dcand, s->dout);
    LCF_HIP(hipGetLastError());
    if (kernel_ms) LCF_HIP(hipEventRecord(b, s->stream));
    LCF_HIP(hipMemcpyAsync(out, s->dout, no * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    LCF_HIP(hipStreamSynchronize(s->stream));
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
