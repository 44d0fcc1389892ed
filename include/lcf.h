/*
 * lcf.h -- C ABI of the MI355X (gfx950) batched light-curve log-likelihood engine.
 *
 * This is the drop-in boundary for the emcee-driven hot path of griffin-h/lightcurve_fitting.  The reference has no
 * native code: the seam is the Python callable that `fitting.lightcurve_mcmc` hands to `emcee.EnsembleSampler`
 * (reference fitting.py:121-130).  Each entry point below names the reference interface it replaces.  All arrays are
 * caller-owned; the engine copies photometry and tables to the device at create time and never writes to caller
 * memory except the documented outputs.  No exceptions cross this boundary: every call returns an lcf_status.
 *
 * Shapes: a "walker block" P is row-major (n x n_dim) float64, exactly what emcee passes to a `vectorize=True`
 * log-probability function.  Units follow the reference (days, kK, 1000 Rsun, THz, W/Hz).
 *
 * Thread-safety: calls on one engine (or one sampler) must be serialised by the caller; distinct engines are
 * independent.  One engine is bound to one HIP device.
 */
#ifndef LCF_H
#define LCF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCF_ABI_VERSION 8

typedef enum lcf_status {
    LCF_OK = 0,
    LCF_ERR_INVALID_ARGUMENT = 1, /* NULL pointer, bad size, unknown model, inconsistent tables */
    LCF_ERR_HIP = 2,              /* a HIP runtime call failed; see lcf_last_error() */
    LCF_ERR_NO_DEVICE = 3,        /* no usable GPU: this library has no CPU fallback */
    LCF_ERR_OUT_OF_MEMORY = 4,
    LCF_ERR_UNSUPPORTED = 5,      /* e.g. an unknown model id, or a host-supplied split in a population run */
    LCF_ERR_NAN_LOGPROB = 6,      /* sampler: a log-probability evaluated to NaN (emcee raises ValueError here) */
    LCF_ERR_STATE = 7             /* call sequence error (e.g. run before set_state) */
} lcf_status;

/* Model families (reference models.py).  Values are stable ABI. */
typedef enum lcf_model {
    LCF_MODEL_SHOCK_COOLING = 1,      /* ShockCooling          models.py:301-353  p = v_s, M_env, f_rho_M, R, t_0     */
    LCF_MODEL_SHOCK_COOLING2 = 2,     /* ShockCooling2         models.py:356-411  p = T_1, L_1, t_tr, t_0             */
    LCF_MODEL_SHOCK_COOLING3 = 3,     /* ShockCooling3         models.py:433-496  p = v_s,M_env,f_rho_M,R,d_L,E(B-V),t_0 */
    LCF_MODEL_SHOCK_COOLING4 = 4,     /* ShockCooling4         models.py:507-632  p = v_s, M_env, f_rho_M, R, t_0     */
    LCF_MODEL_COMPANION_SHOCKING = 5, /* CompanionShocking     models.py:848-918  p = t_0,a,Mv7,t_max,s,r_r,r_i,r_U   */
    LCF_MODEL_COMPANION_SHOCKING2 = 6,/* CompanionShocking2    models.py:921-980  p = t_0,a,Mv7,t_max,s,dt_U,dt_i     */
    LCF_MODEL_COMPANION_SHOCKING3 = 7,/* CompanionShocking3    models.py:983-1045 p = t_0,a,theta,t_max,s,dt_U,dt_i   */
    LCF_MODEL_BLACKBODY = 8           /* direct (T, R) blackbody, bolometric.py:154-164                              */
} lcf_model;

/* Priors (reference models.py:1048-1098).  Bounds are strict: p_min < p < p_max, else log-prior = -inf. */
typedef enum lcf_prior_kind { LCF_PRIOR_UNIFORM = 0, LCF_PRIOR_LOG_UNIFORM = 1, LCF_PRIOR_GAUSSIAN = 2 } lcf_prior_kind;

typedef struct lcf_prior {
    int32_t kind; /* lcf_prior_kind */
    int32_t reserved;
    double p_min, p_max;
    double mean, stddev; /* Gaussian only */
} lcf_prior;

enum { LCF_SIGMA_RELATIVE = 0, LCF_SIGMA_ABSOLUTE = 1 };
enum { LCF_N_CONSTS = 12 };

/*
 * One fitting problem = one light curve + one model instance (with its construction-time constants baked in).
 *
 * consts[] by model:
 *   SHOCK_COOLING, SHOCK_COOLING2: A, a, alpha, epsilon_1, epsilon_2, L_0, T_0, Tph_to_Tcol  (models.py:192-226)
 *   SHOCK_COOLING4:                A, a, alpha, L_br_0, T_col_br_0, t_br_0, t_tr_0           (models.py:567-577)
 *   others:                        unused
 *   (consts[8..11] are scratch for the engine: whatever the caller puts there is overwritten)
 *
 * Band tables: filter i owns samples tab_off[i] .. tab_off[i+1]-1 of (tab_a, tab_w) with
 *   a_k = c1 nu_k (1+z)  [kK],   W_k = c2 nu'_k^3 min(1, nu_cut/nu'_k) tw_k Tnorm_k,
 * so that L_nu(filter; T, R) = R^2 sum_k W_k / (exp(a_k/T) - 1)   (filters.py:308-310 + models.py:1127-1128).
 *
 * Companion-shocking extras (NULL / 0 otherwise): per-filter parameter indices (or -1) for the factor on the shock
 * term, the factor on the SiFTO term and the SiFTO time offset (models.py:804-807, 913-916), and one piecewise-cubic
 * per filter: spline_coef[f][i][0..3] are the coefficients of (x - knot_i)^3..^0 on [knot_i, knot_i+1]; the
 * template is 0 outside [knot_0, knot_{n-1}] (models.py:717, 826).
 */
typedef struct lcf_problem {
    int32_t abi_version; /* LCF_ABI_VERSION */
    int32_t model;       /* lcf_model */
    int32_t n_par;       /* model parameters (without the optional intrinsic-scatter parameter) */
    int32_t use_sigma;   /* 1: the last of n_dim = n_par + 1 parameters is sigma (models.py:128-130) */
    int32_t sigma_type;  /* LCF_SIGMA_RELATIVE | LCF_SIGMA_ABSOLUTE (models.py:121-126) */
    int32_t n_filters;
    int64_t n_points;
    double consts[LCF_N_CONSTS];
    const double* t;          /* [n_points] observation times (lc['MJD'])                     models.py:117 */
    const double* y;          /* [n_points] observed luminosity density (lc['lum'])           models.py:118 */
    const double* dy;         /* [n_points] its uncertainty (lc['dlum'])                      models.py:119 */
    const int32_t* filt_idx;  /* [n_points] filter of each point, 0 <= . < n_filters          models.py:116 */
    const int32_t* tab_off;   /* [n_filters + 1] */
    const double* tab_a;      /* [tab_off[n_filters]] */
    const double* tab_w;      /* [tab_off[n_filters]] */
    /* Reddening, LCF_MODEL_SHOCK_COOLING3 only (NULL otherwise): A_lambda / E(B-V) of the extinction law at every
     * table sample, so that sample k is weighted by 10^(-0.4 E(B-V) tab_ext[k]) per walker
     * (filters.py:32-33, 308-310: extinction_law(freq, ebv) inside Filter.synthesize). */
    const double* tab_ext;    /* [tab_off[n_filters]] */
    /* Optional compressed companions (all NULL = none): per filter a shorter table (the Gauss quadrature of the full
     * table's own discrete measure, computed by the host packer) that reproduces the band sum to 2e-14 for every
     * temperature T >= ctab_tmin[i]; the engine switches per data point and uses the full table below it. */
    const int32_t* ctab_off;  /* [n_filters + 1] */
    const double* ctab_a;     /* [ctab_off[n_filters]] */
    const double* ctab_w;     /* [ctab_off[n_filters]] */
    const double* ctab_tmin;  /* [n_filters] kK; +inf = never use the compressed table of this filter */
    /* Optional second compressed level (all NULL = none; needs the first): a still shorter table per filter, valid
     * above a higher temperature htab_tmin[i] >= ctab_tmin[i].  Per data point the engine takes the shortest table
     * that is valid at the point's temperature: hot, else cool, else full. */
    const int32_t* htab_off;  /* [n_filters + 1] */
    const double* htab_a;     /* [htab_off[n_filters]] */
    const double* htab_w;     /* [htab_off[n_filters]] */
    const double* htab_tmin;  /* [n_filters] kK */
    /* Optional third level (NULL = none; not for LCF_MODEL_SHOCK_COOLING3): the band sum of every filter as a FUNCTION
     * of temperature -- ln S_f(T) as piecewise polynomials of degree 7 in u = ln T on itab_m equal intervals of width
     * itab_h from itab_u0 = ln(first temperature in kK), 8 monomial coefficients in s in [-1, 1] per interval, highest
     * power first -- valid for T >= itab_tmin[i] up to the last interval's end.  The host packer builds and proves them
     * (filters.interp_planck_table).  With it the engine evaluates a data point by one lookup, 7 fused multiply-adds
     * and one exponential wherever the point's temperature is inside the range; elsewhere it walks the sample tables. */
    const double* itab_coef;  /* [n_filters][itab_m][8] */
    const double* itab_tmin;  /* [n_filters] kK; +inf = no interpolant for this filter */
    int32_t itab_m;
    int32_t reserved2;
    double itab_u0, itab_h;
    const int32_t* filt_kasen_par; /* [n_filters] or NULL */
    const int32_t* filt_sifto_par; /* [n_filters] or NULL */
    const int32_t* filt_dt_par;    /* [n_filters] or NULL */
    int32_t n_knots;
    int32_t reserved;
    const double* spline_knots;    /* [n_knots] ascending */
    const double* spline_coef;     /* [n_filters][n_knots - 1][4] */
    const lcf_prior* priors;       /* [n_par + use_sigma] or NULL (then log_posterior == log_likelihood) */
} lcf_problem;

typedef struct lcf_engine lcf_engine;
typedef struct lcf_sampler lcf_sampler;

/* ---- library ------------------------------------------------------------------------------------------------ */
int32_t lcf_abi_version(void);
/* Human-readable description of the last failure on the calling thread ("" if none). */
const char* lcf_last_error(void);
/* Number of HIP devices visible (0 if none / no driver).  Never initialises a context on a device. */
int32_t lcf_device_count(void);

/* ---- engine --------------------------------------------------------------------------------------------------- */
/* Replaces the closure set-up of fitting.py:68-128 (photometry columns + model + priors captured once). */
lcf_status lcf_engine_create(const lcf_problem* problem, int32_t device, lcf_engine** out);
void lcf_engine_destroy(lcf_engine* e);
int32_t lcf_engine_ndim(const lcf_engine* e);
int64_t lcf_engine_npoints(const lcf_engine* e);
/* Planck samples of one log-likelihood evaluation over the full tables: sum over points of K_filter. */
int64_t lcf_engine_samples_per_eval(const lcf_engine* e);
/* Select the band-sum level: 0 = libm expm1 + divide over the full tables (reference-shaped), 1 = fused fast path
 * over the full tables, 2 = fused fast path over the Gauss-compressed tables where valid, 3 = the interpolants of
 * ln S_f(ln T) where a point's temperature is inside the range they are proved for, else level 2.  An engine starts
 * at the highest level its problem gave tables for (3 with itab_*, else 2 with ctab_*, else 1). */
lcf_status lcf_engine_set_variant(lcf_engine* e, int32_t variant);

/* Model.log_likelihood (models.py:93-136) for a block of n walkers.  Host pointers. out[n]. */
lcf_status lcf_log_likelihood(lcf_engine* e, int64_t n, const double* P, double* out);
/* log_posterior closure (fitting.py:121-128): -inf where a prior excludes the walker (likelihood skipped). */
lcf_status lcf_log_posterior(lcf_engine* e, int64_t n, const double* P, double* out);
/* Same with DEVICE pointers, enqueued on `stream` (a hipStream_t), no host sync.  NULL selects the engine's own
 * non-blocking stream, which is NOT ordered with the legacy default stream: pass a real stream to chain with other
 * work (the Python driver runs under a side stream for that reason). */
lcf_status lcf_log_likelihood_dev(lcf_engine* e, int64_t n, const double* dP, double* dout, void* stream);
lcf_status lcf_log_posterior_dev(lcf_engine* e, int64_t n, const double* dP, double* dout, void* stream);

/* Model.__call__ / evaluate (models.py:86-87, pointwise branch :1161-1162): y_fit[n][n_points], original point order. */
lcf_status lcf_model_evaluate(lcf_engine* e, int64_t n, const double* P, double* y_fit);
/* temperature_radius (models.py:231-269, 583-597, 727-755): T_K, R_bb as [n][n_points]. */
lcf_status lcf_temperature_radius(lcf_engine* e, int64_t n, const double* P, double* T_K, double* R_bb);
/* blackbody_to_filters pointwise (models.py:1131-1165): out[m] = L_nu(filter filt_idx[m]; T[m], R[m]). */
lcf_status lcf_blackbody_to_filters(lcf_engine* e, int64_t m, const int32_t* filt_idx, const double* T,
                                    const double* R, double* out);

/* Measurement hook for bench.py: average duration [ms] of the dominant kernel alone (the per-point likelihood
 * kernel over n walkers), reps back-to-back launches bracketed by HIP events on the engine's stream. */
lcf_status lcf_profile_loglike_kernel(lcf_engine* e, int64_t n, const double* P, int32_t reps, double* avg_ms);

/* ---- device-resident ensemble sampler (replaces emcee.EnsembleSampler for this path, fitting.py:130-145) -------- */
/* Goodman-Weare stretch move, red/blue halves, scale a.  RNG: Philox4x32-10 keyed by seed with counters
 * (walker, step, half) -- independent of how walkers are sharded over devices. */
lcf_status lcf_sampler_create(lcf_engine* e, int32_t n_walkers, uint64_t seed, double a, lcf_sampler** out);
void lcf_sampler_destroy(lcf_sampler* s);
/* coords[n_walkers][n_dim] host; evaluates the initial log-posterior on the device. */
lcf_status lcf_sampler_set_state(lcf_sampler* s, const double* coords);
lcf_status lcf_sampler_get_state(lcf_sampler* s, double* coords, double* log_prob);
/* Device memory for the chain of a later run of n_steps steps with store_chain, allocated now instead of inside that
 * run (emcee grows its backend inside run_mcmc, fitting.py:133-148; a caller that times a run reserves first). */
lcf_status lcf_sampler_reserve_chain(lcf_sampler* s, int64_t n_steps);
/* Red/blue colouring of each step.  RANDOM = emcee's randomize_split, generated on the device from (seed, step);
 * HOST = caller-provided perm[n_steps][n_walkers] int32 (the first half of each row is colour 0). */
enum { LCF_SPLIT_IDENTITY = 0, LCF_SPLIT_RANDOM = 1, LCF_SPLIT_HOST = 2 };
/* Run n_steps ensemble steps numbered from first_step (the step number is an RNG counter word). */
lcf_status lcf_sampler_run(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                           const int32_t* perm, int32_t store_chain);
/* The same split in two: enqueue the whole run on the engine's stream and return at once / wait for it and check.
 * Samplers on different engines (population mode: one transient each) run concurrently between the two calls. */
lcf_status lcf_sampler_run_async(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                                 const int32_t* perm, int32_t store_chain);
lcf_status lcf_sampler_wait(lcf_sampler* s);
/* Population mode: run n samplers (independent transients with the same walker count, on one device) in lock step:
 * ONE launch per half-step covers all of them where every transient has shared epochs, tables staged in LDS and
 * proposal-independent tables (k_pop), else one proposal launch and one likelihood launch.  split_mode: identity or
 * random.  LCF_NO_POP=1 in the environment forces the two launches (tests). */
lcf_status lcf_population_run(lcf_sampler** samplers, int32_t n, int64_t first_step, int64_t n_steps,
                              int32_t split_mode, int32_t store_chain, double* elapsed_ms);
/* Chain of the last run: chain[n_steps][n_walkers][n_dim], log_prob[n_steps][n_walkers] (either may be NULL). */
lcf_status lcf_sampler_get_chain(lcf_sampler* s, double* chain, double* log_prob);
lcf_status lcf_sampler_get_naccepted(lcf_sampler* s, int64_t* n_accepted /* [n_walkers] */);
/* lcf_sampler_get_state and lcf_sampler_get_naccepted in ONE call (any pointer may be NULL): what a driver reads after
 * every run -- emcee's State and the acceptance counts (fitting.py:133, 145) -- for one trip through the binding. */
lcf_status lcf_sampler_get_snapshot(lcf_sampler* s, double* coords, double* log_prob, int64_t* n_accepted);
/* Device time of the last lcf_sampler_run in milliseconds (HIP events on the sampler's stream). */
double lcf_sampler_last_run_ms(const lcf_sampler* s);
/* 1 if half-steps of this sampler run as one launch (everything a workgroup needs fits in LDS), 0 if as
 * proposal + likelihood launches.  Same chain either way. */
int32_t lcf_sampler_one_launch(const lcf_sampler* s);
/* Which kernels a single-GPU run (lcf_sampler_run / _run_async) may use for a half-step.  All of them produce the
 * same chain bit for bit; the choice exists for tests and measurements.
 *   AUTO:   one workgroup per proposal that also accepts / rejects, resident for a whole block of half-steps
 *           (k_solo_run: ONE launch per up to 128 steps (256 half-steps), rows handed from workgroup to workgroup through a board of
 *           tagged rows in device memory) where a proposal's parts fit one workgroup; else one workgroup per
 *           (proposal, part) and launch (k_fused); else proposal + likelihood launches
 *   SOLO:   as AUTO, but one launch per half-step (k_solo)
 *   FUSED:  never k_solo        PHASES: always proposal + likelihood launches
 * Returns in *used (optional) what a run would use now: 3 = k_solo_run, 2 = k_solo, 1 = k_fused, 0 = separate launches.
 * (LCF_NO_RUN_KERNEL=1 in the environment: AUTO never picks k_solo_run -- for several processes sharing one GPU, whose
 * resident launches could keep each other's workgroups out.) */
enum { LCF_HALF_STEP_AUTO = 0, LCF_HALF_STEP_FUSED = 1, LCF_HALF_STEP_PHASES = 2, LCF_HALF_STEP_SOLO = 3 };
lcf_status lcf_sampler_set_half_step_kernel(lcf_sampler* s, int32_t choice, int32_t* used);

/* What the half-steps of the sampler's last run were executed by (-1: no run yet): separate proposal / likelihood
 * launches, k_fused, k_solo; for lcf_population_run resident workgroups for all transients (k_pop_run), one launch per
 * half-step for all transients (k_pop: a workgroup per four proposals, accept test included) or the two launches
 * k_step_multi + k_points_multi. */
enum { LCF_KERNEL_PHASES = 0, LCF_KERNEL_FUSED = 1, LCF_KERNEL_SOLO = 2, LCF_KERNEL_POPULATION = 3,
       LCF_KERNEL_POPULATION_PHASES = 4, LCF_KERNEL_RUN = 5 /* k_solo_run: a block of half-steps per launch */,
       LCF_KERNEL_POPULATION_RUN = 6 /* k_pop_run: the same for all transients of a population */ };
int32_t lcf_sampler_last_run_kernel(const lcf_sampler* s);
/* Launches of that kernel in the last single-GPU run (lcf_sampler_run / _run_async): two per step, or -- k_solo_run --
 * one per block of up to 128 steps (between ranks: 32); lcf_population_run: launches per run of the kernel it took. */
int64_t lcf_sampler_last_run_launches(const lcf_sampler* s);

/* Multi-GPU building blocks: one half-step split into phases so that the caller can all-gather the shard's new
 * log-probabilities (RCCL) between phase 2 and phase 3.  All enqueue on `stream` without host sync.
 *   1. propose:  every rank draws the same proposals for the whole active half (replicated, cheap)
 *   2. evaluate: this rank evaluates proposals [lo, hi) of the active half -> newlp[lo:hi)
 *   3. accept:   every rank applies the same accept/reject to the whole half given the gathered newlp[0:n/2) */
lcf_status lcf_sampler_begin(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                             const int32_t* perm, int32_t store_chain);
lcf_status lcf_sampler_propose(lcf_sampler* s, int64_t step, int32_t half, void* stream);
lcf_status lcf_sampler_evaluate(lcf_sampler* s, int32_t lo, int32_t hi, void* stream);
lcf_status lcf_sampler_accept(lcf_sampler* s, int64_t step, int32_t half, void* stream);
/* Phases 1 + 2 in one call (fewer launches: the thermal states of the shard ride on the proposal kernel). */
lcf_status lcf_sampler_half_step(lcf_sampler* s, int64_t step, int32_t half, int32_t lo, int32_t hi, void* stream);
/* Device pointer to newlp[n_walkers/2] (float64) for the collective. */
void* lcf_sampler_newlp_ptr(lcf_sampler* s);
/* The same half-step without the finalize launch: what the collective then carries is each proposal's ROW --
 * its partial chi^2 sums followed by its log-prior, *row_doubles float64 -- and the accept tests of the next launch add
 * a row up themselves (the protocol of lcf_sampler_run_sharded).  Use one protocol throughout a run.
 * lcf_sampler_rows_ptr: device pointer to rows[n_walkers/2][*row_doubles] of the half-step drawn last. */
lcf_status lcf_sampler_half_step_rows(lcf_sampler* s, int64_t step, int32_t half, int32_t lo, int32_t hi, void* stream);
void* lcf_sampler_rows_ptr(lcf_sampler* s, int32_t* row_doubles);
lcf_status lcf_sampler_check(lcf_sampler* s); /* syncs; returns LCF_ERR_NAN_LOGPROB if a NaN was seen */

/* ---- native multi-GPU run: RCCL bound at run time -------------------------------------------------------------- */
/* rccl_path: the librccl.so to dlopen (NULL/"" = the default search path; pass the one PyTorch ships when torch is
 * in the process).  One rank obtains an id, every rank creates its communicator with it (collective call). */
typedef struct { char internal[128]; } lcf_comm_id;
typedef struct lcf_comm lcf_comm;
/* Local, non-collective check that RCCL can be bound (call it on every rank and agree before the collective calls). */
lcf_status lcf_comm_probe(const char* rccl_path);
lcf_status lcf_comm_unique_id(const char* rccl_path, lcf_comm_id* out);
lcf_status lcf_comm_create(const char* rccl_path, const lcf_comm_id* id, int32_t n_ranks, int32_t rank, int32_t device,
                           lcf_comm** out);
void lcf_comm_destroy(lcf_comm* c);
/* Number of ranks as the communicator itself reports it (ncclCommCount), and this rank's index (ncclCommUserRank). */
lcf_status lcf_comm_count(const lcf_comm* c, int32_t* n_ranks, int32_t* rank);
/* Measurement hook for bench.py: average time [ms] of ONE in-place all-gather as lcf_sampler_run_sharded issues it per
 * half-step for sampler s (its rows, float64), `reps` of them back to back on the engine's stream between two HIP
 * events.  Collective: every rank calls it with the same arguments. */
lcf_status lcf_comm_time_allgather(lcf_comm* c, lcf_sampler* s, int32_t reps, double* avg_ms);
/* The whole run of lcf_sampler_run, sharded: this rank evaluates proposals [rank*w, (rank+1)*w) of each half-step
 * (w = n_walkers / 2 / n_ranks) and one in-place ncclAllGather per half-step (each proposal's partial chi^2 sums and
 * log-prior, which every rank then adds up in the same order) makes the ranks agree; enqueued on the
 * engine's stream without host synchronisation between half-steps.  Collective: same arguments on every rank. */
lcf_status lcf_sampler_run_sharded(lcf_sampler* s, lcf_comm* c, int64_t first_step, int64_t n_steps,
                                   int32_t split_mode, const int32_t* perm, int32_t store_chain);

/* ---- sharded run WITHOUT a collective: peer mailboxes (behind a switch until measured on a multi-GPU node) ---------- */
/* Every rank owns a mailbox in uncached device memory; a rank that has evaluated a proposal stores the numbers of its
 * row -- each as two 8-byte {data, generation} granules -- straight into every rank's mailbox (peers' memory mapped
 * through HIP IPC: xGMI writes on a node) and the accept tests of the next launch poll their own copy.  No kernel and
 * no host call sits between two half-steps.  Set-up: every rank exports its mailbox, the handles travel by any means
 * (the Python driver uses torch.distributed), every rank connects with the full list.  `local_ptrs` (instead of
 * handles): ranks emulated inside one process pass each other's device pointers.  At most 8 ranks. */
typedef struct { char internal[64]; } lcf_ipc_handle;
lcf_status lcf_sampler_mailbox_export(lcf_sampler* s, lcf_ipc_handle* out, void** local_ptr);
lcf_status lcf_sampler_mailbox_connect(lcf_sampler* s, int32_t n_ranks, int32_t rank, const lcf_ipc_handle* handles,
                                       void* const* local_ptrs);
/* The run of lcf_sampler_run_sharded over the mailboxes.  Every rank calls it with the same arguments, and only after
 * ALL ranks have returned from their previous run (barrier of the caller).  A rank whose peers' rows do not arrive
 * within 0.5 s returns LCF_ERR_STATE instead of waiting for ever. */
lcf_status lcf_sampler_run_peers(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                                 const int32_t* perm, int32_t store_chain);
/* Enqueue only (lcf_sampler_wait completes it): lets one host thread drive several emulated ranks. */
lcf_status lcf_sampler_run_peers_async(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                                       const int32_t* perm, int32_t store_chain);

/* Row boards: the sharded run in which nothing is replicated.  Rank r evaluates, accepts and commits the proposals
 * [r w, (r + 1) w) of every half-step with the one-workgroup-per-proposal kernel (k_solo) and stores each walker's new
 * row -- position, log-posterior, acceptance count, as {32 data bits, 32-bit half-step tag} words -- straight into
 * EVERY rank's board (uncached device memory, mapped through IPC: xGMI writes on a node); the serial head of a later
 * half-step polls its own board for exactly the versions of the two rows it needs.  No collective, no launch between
 * half-steps, no work about other ranks' walkers; every wait is bounded (0.5 s, then LCF_ERR_STATE).  When the run
 * ends, state, acceptance counts and (if stored) the chain are complete on every rank.  Same chain as every other
 * driver, bit for bit.  export / connect as for the mailboxes (local_ptrs: ranks emulated inside one process).
 * lcf_sampler_run_rows is collective in effect: same arguments on every rank, called after ALL ranks have returned
 * from the previous run, on samplers that have seen the same sequence of runs. */
lcf_status lcf_sampler_board_export(lcf_sampler* s, lcf_ipc_handle* out, void** local_ptr);
lcf_status lcf_sampler_board_connect(lcf_sampler* s, int32_t n_ranks, int32_t rank, const lcf_ipc_handle* handles,
                                     void* const* local_ptrs);
lcf_status lcf_sampler_run_rows(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                                const int32_t* perm, int32_t store_chain);
lcf_status lcf_sampler_run_rows_async(lcf_sampler* s, int64_t first_step, int64_t n_steps, int32_t split_mode,
                                      const int32_t* perm, int32_t store_chain);

/* ---- per-epoch blackbody SED likelihood (bolometric.py:154-164: spectrum_mcmc's inner log_posterior) --------- */
/* For every epoch e, observations ep_off[e] .. ep_off[e+1]-1 (filter index, luminosity density y, uncertainty dy);
 * for every candidate (T [kK], R [1000 Rsun][, sigma]) of that epoch the Gaussian log-likelihood of the band-averaged
 * blackbody [f.synthesize(planck_fast, T, R) for f in filters].  precision 0 = float64 over the band tables sample by
 * sample, 1 = float32 over the band tables, 2 = float64 through the interpolants of ln S_f(ln T) where a candidate's
 * temperature is inside their proved range and over the band tables where it is not (0 when no interpolants were given). */
typedef struct lcf_sed lcf_sed;
/* ctab_*: optional Gauss-compressed companions of the band tables (all NULL = none), itab_*: optional interpolants of
 * ln S_f(ln T) (itab_coef NULL = none); both as in lcf_problem. */
lcf_status lcf_sed_create(int32_t n_filters, const int32_t* tab_off, const double* tab_a, const double* tab_w,
                          const int32_t* ctab_off, const double* ctab_a, const double* ctab_w,
                          const double* ctab_tmin, const double* itab_coef, const double* itab_tmin, int32_t itab_m,
                          double itab_u0, double itab_h, int32_t device, lcf_sed** out);
void lcf_sed_destroy(lcf_sed* s);
lcf_status lcf_sed_set_observations(lcf_sed* s, int64_t n_epochs, const int32_t* ep_off, const int32_t* filt_idx,
                                    const double* y, const double* dy);
/* cand[n_epochs][n_cand][n_par] (n_par = 2 or 3), out[n_epochs][n_cand]; kernel_ms (optional) receives the device
 * time of the kernel alone (HIP events on the engine's stream). */
lcf_status lcf_sed_log_likelihood(lcf_sed* s, int64_t n_cand, int32_t n_par, int32_t sigma_type, const double* cand,
                                  int32_t precision, int32_t use_compressed, double* out, double* kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* LCF_H */
