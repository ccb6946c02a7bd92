/*
 * chimera_hip.h -- C ABI of libchimera_hip.so: the MI355X (gfx950) implementation of CHIMERA's
 * hyper-likelihood hot path.
 *
 * The reference (CosmoStatGW/CHIMERA v2.0.0) is pure Python on JAX and has NO foreign-function boundary;
 * its "operator API" for this path is the Python surface
 *     hyperlikelihood.__call__/compute_log_hyperlike/compute_all      CHIMERA/likelihood.py:307-338
 *     selection_function.N_exp                                         CHIMERA/selection_function.py:34-48
 *     population.update                                                CHIMERA/population/pop_wrapper.py:56-64
 * This header is the boundary a maintainer would bind with ctypes underneath that unchanged surface
 * (see INTEGRATION.md).  Each entry point cites the reference code it replaces.
 *
 * Conventions: every array is C-contiguous; floating point is IEEE fp64 (the reference enables jax x64,
 * CHIMERA/utils/config.py:5); padded per-pixel arrays carry the reference's sentinel -100
 * (CHIMERA/catalog/catalog.py:174-176, CHIMERA/data.py:348-351).  Host buffers passed to *_create are
 * copied to the device and may be freed on return; outputs are written to caller-allocated host buffers.
 * Every function returns 0 on success or a negative CHM_E_* code, with a message in chm_last_error()
 * (thread-local).  NaN / -inf likelihood values are results, not errors (likelihood.py:296-297).
 * A handle is not thread-safe; calls are synchronous (internally asynchronous on one HIP stream).
 */
#ifndef CHIMERA_HIP_H
#define CHIMERA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHM_OK        0
#define CHM_E_ARG    -1   /* bad argument (Python wrapper raises ValueError)     */
#define CHM_E_HIP    -2   /* HIP runtime error (RuntimeError)                    */
#define CHM_E_NOMEM  -3   /* device allocation failed                            */
#define CHM_E_RCCL   -4   /* RCCL error                                          */

/* kind of GW kernel: likelihood.py:88-97 */
enum { CHM_MODE_1D = 0, CHM_MODE_APPROX = 1, CHM_MODE_MARG = 2, CHM_MODE_FULL = 3 };
/* kde1d kernel: CHIMERA/utils/math.py:77,83-89 */
enum { CHM_KERNEL_EPAN = 0, CHM_KERNEL_GAUSS = 1 };
/* bandwidth rule: math.py:65-75, 178-185 */
enum { CHM_BW_SCOTT = 0, CHM_BW_SILVERMAN = 1, CHM_BW_SCALAR = 2 };
/* model ids: cosmo.py:50-115, mass.py:56-149, rate.py:32-88 */
enum { CHM_COSMO_FLRW = 0, CHM_COSMO_MG_FLRW = 1 };
enum { CHM_MASS_TPL = 0, CHM_MASS_BPL = 1, CHM_MASS_PLP = 2 };
enum { CHM_RATE_PL = 0, CHM_RATE_MD = 1, CHM_RATE_TPL = 2, CHM_RATE_TMD = 3 };

/* slots of chm_params.cosmo[] */
enum { CHM_C_H0 = 0, CHM_C_OM0, CHM_C_OK0, CHM_C_OR0, CHM_C_W0, CHM_C_WA, CHM_C_XI0, CHM_C_N, CHM_NCOSMO };
/* slots of chm_params.mass[]: common m_low, m_high; then per model
 *   tpl: alpha, beta            bpl: alpha_1, alpha_2, beta, delta_m, break_fraction
 *   plp: lambda_peak, alpha, beta, delta_m, mu_g, sigma_g                                    */
enum { CHM_M_MLOW = 0, CHM_M_MHIGH = 1, CHM_M_P0 = 2, CHM_NMASS = 8 };
/* slots of chm_params.rate[]: gamma, kappa, zp, zmax */
enum { CHM_R_GAMMA = 0, CHM_R_KAPPA, CHM_R_ZP, CHM_R_ZMAX, CHM_NRATE };

/* One hyper-parameter draw = the state of population.update(**lambda) (pop_wrapper.py:56-64). */
typedef struct chm_params {
  int32_t cosmo_model, mass_model, rate_model;
  int32_t z_grid_res;            /* T_c: cosmology table length (cosmo.py:77, default 1500)          */
  int32_t mass_grid_res;         /* T_m: mass table length (mass.py:20, default 1000)               */
  int32_t scale_free;            /* pop_wrapper.py:31                                               */
  int32_t has_catalog;           /* 0: empty_catalog (catalog.py:19-43); 1: pixelated_catalog       */
  int32_t _pad;
  double  z_max;                 /* cosmo.py:18                                                     */
  double  cosmo[CHM_NCOSMO];
  double  mass[CHM_NMASS];
  double  rate[CHM_NRATE];
  double  R0, Tobs;              /* pop_wrapper.py:18-20                                            */
  double  compl_z0, compl_z1;    /* dVdz_completeness.z_range (completeness.py:35)                  */
} chm_params;

/* GW events + per-event grids + catalogue term: everything hyperlikelihood.__init__ receives
 * (likelihood.py:48-99) apart from the population model.                                            */
typedef struct chm_like_desc {
  int32_t E, S, Z, P;            /* events, samples/event, z-grid points, max_npixels (0 if not pixelated) */
  int32_t ev_begin, ev_end;      /* shard: only events [ev_begin, ev_end) are uploaded (0,E = all)   */
  const double*  dL;             /* (E,S) theta_pe_det.dL         data.py:30                          */
  const double*  m1det;          /* (E,S)                         data.py:28                          */
  const double*  m2det;          /* (E,S)                         data.py:29                          */
  const double*  pe_prior;       /* (E,S)                         data.py:35,45-47                    */
  const double*  ra;             /* (E,S) full mode only, else NULL   data.py:33                      */
  const double*  dec;            /* (E,S) full mode only, else NULL   data.py:34                      */
  const int32_t* pix_of_sample;  /* (E,S) index into the event's pixel list (position of
                                    pixels_pe_opt_nside in pixels_opt_nsides), -1 = in no pixel;
                                    NULL if not pixelated.   likelihood.py:174-179                    */
  const double*  z_grids;        /* (E,Z)                         likelihood.py:67                    */
  const double*  p_cat;          /* (E,P,Z) -100 padded, NULL if not pixelated  catalog.py:189        */
  const double*  P_compl;        /* (E,Z)   NULL if not pixelated               catalog.py:195        */
  const double*  gw_loc2d_pdf;   /* (E,P) -100 padded             data.py:42                          */
  const double*  ra_pix;         /* (E,P) -100 padded (full mode) data.py:40                          */
  const double*  dec_pix;        /* (E,P) -100 padded (full mode) data.py:41                          */
  const int32_t* neff_pixels;    /* (E,)                          catalog.py:118                      */
  int32_t mode;                  /* CHM_MODE_*                                                        */
  int32_t kernel;                /* CHM_KERNEL_* (ignored by marginalized: always epan, likelihood.py:192) */
  int32_t bw_method;             /* CHM_BW_*                                                          */
  int32_t binning;               /* likelihood.py:58                                                  */
  int32_t num_bins;              /* likelihood.py:59                                                  */
  int32_t device;                /* HIP device ordinal                                                */
  double  bw_scalar;             /* used when bw_method == CHM_BW_SCALAR                              */
  double  cut_grid;              /* NaN = None (likelihood.py:115,185)                                */
  double  pe_neff;               /* likelihood.py:61                                                  */
} chm_like_desc;

/* Detected injections: selection_function.__init__ (selection_function.py:24-32). */
typedef struct chm_sel_desc {
  int64_t I;                     /* detected injections in the arrays                                 */
  int64_t inj_begin, inj_end;    /* shard [inj_begin, inj_end) (0,I = all)                            */
  const double* dL;              /* (I,) theta_inj_det.dL      data.py:52                             */
  const double* m1det;           /* (I,)                                                             */
  const double* m2det;           /* (I,)                                                             */
  const double* p_draw;          /* (I,)                       data.py:53                             */
  double N_inj;                  /* total generated injections (all shards)                           */
  double N_eff;                  /* NaN = None (selection_function.py:43)                             */
  int32_t device;
  int32_t _pad;
} chm_sel_desc;

/* Outputs of one evaluation; any pointer may be NULL (not wanted). nb = number of draws. */
typedef struct chm_out {
  double* log_hyper;     /* (nb,)       compute_log_hyperlike           likelihood.py:307-316         */
  double* log_num;       /* (nb,)       compute_log_likenum             likelihood.py:294-301         */
  double* N_exp;         /* (nb,)       selection_function.N_exp        selection_function.py:34-48   */
  double* log_like_evs;  /* (nb,E_loc)  nan_to_num(log L_i)             likelihood.py:329-330         */
  double* numlike_evs;   /* (nb,E_loc)  L_i = compute_numlike_evs       likelihood.py:266-292         */
  double* p_gw;          /* (nb,E_loc,P,Z) p_gw3d, or (nb,E_loc,Z) p_gw1d in 1-D mode  likelihood.py:105-260 */
  double* partials;      /* (nb,3)      this shard's [sum_i log L_i, nansum dN, sum dN^2]             */
} chm_out;

/* Plug-in models ("plugin fallback").  The reference's population pieces are open classes: a user adds a mass,
 * rate or completeness model by writing a new struct and new plum overloads of p_m1m2 (mass.py:334-345), merger_rate
 * (rate.py:96-122) or p_bkg / fR (completeness.py:43-67).  Such Python functions cannot run inside a kernel; the caller
 * evaluates them on the host and hands the values over per draw.  Every pointer may be NULL (= the built-in model selected by
 * chm_params is used for that piece); arrays are for THIS shard, C-contiguous, in the caller's original sample order.        */
typedef struct chm_tab {
  const double* pm_samples;    /* (nb,E_loc,S)  p_m1m2(m1src, m2src) of every posterior sample      pop_wrapper.py:79          */
  const double* pm_inj;        /* (nb,I_loc)    p_m1m2 of every injection                           pop_wrapper.py:108         */
  const double* rate_grid;     /* (nb,E_loc,Z)  merger_rate(z) on the event grids                   pop_wrapper.py:85          */
  const double* rate_inj;      /* (nb,I_loc)    merger_rate(z_inj)                                  pop_wrapper.py:107         */
  const double* bkg_grid;      /* (nb,E_loc,Z)  completeness.p_bkg(cosmo, z) on the event grids     catalog.py:200             */
  const double* bkg_inj;       /* (nb,I_loc)    p_bkg(cosmo, z_inj, original distances)             pop_wrapper.py:106         */
  const double* fR;            /* (nb,)         completeness.fR(cosmo)                              catalog.py:199             */
  /* Plug-in COSMOLOGY (a user struct with its own plum overloads of the distance functions, cosmo.py:122-264): the path needs a
   * cosmology only through the (dL, z) table of z_from_dGW, the Jacobian |ddL/dz| (1+z)^2 and p_bkg -- all four arrays below must be
   * given together with bkg_grid / bkg_inj (and fR for a catalogue); z_grid_res must be the same for every draw of the call and
   * chm_params.cosmo is then ignored.  NULL = the built-in flrw / mg_flrw of chm_params.                                        */
  const double* z_table;       /* (nb,z_grid_res)  cosmo.z_grid_interp                              cosmo.py:43-46             */
  const double* dL_table;      /* (nb,z_grid_res)  dL_at_z(cosmo, z_grid_interp): the table of z_from_dGW   cosmo.py:260-264     */
  const double* jac_grid;      /* (nb,E_loc,Z)  ddLdz_at_z(cosmo, z_grids) (1 + z_grids)^2                  likelihood.py:272    */
  const double* jac_inj;       /* (nb,I_loc)    |ddLdz_at_z(cosmo, z_inj, original distances)| (1 + z_inj)^2   pop_wrapper.py:109  */
} chm_tab;

typedef struct chm_like chm_like;
typedef struct chm_sel  chm_sel;
typedef struct chm_comm chm_comm;

const char* chm_version(void);
int         chm_device_count(void);
const char* chm_last_error(void);

/* hyperlikelihood.__init__ (likelihood.py:48-99): upload one shard of events to desc->device. */
int chm_like_create(const chm_like_desc* desc, chm_like** out);
int chm_like_destroy(chm_like* h);

/* selection_function.__init__ (selection_function.py:24-32). */
int chm_sel_create(const chm_sel_desc* desc, chm_sel** out);
int chm_sel_destroy(chm_sel* h);

/* A second evaluation LANE on the data a handle already holds in HBM: the clone shares every array *_create uploaded (reference
 * counted: handles may be destroyed in any order) and owns its streams, per-draw tables, workspaces and captured graph.  One host
 * thread per lane, each calling chm_eval on its own lane, keeps two evaluations in flight on one GPU -- what a sampler that evaluates
 * several walkers per step (emcee's log_prob_fn over a pool, CHIMERA/utils/emcee_utils.py:281-288) can use, and what hides the per-call
 * fixed costs of the small shards of a multi-GPU run (CHIMERA/parallel.py:94-99).  With a communicator every lane needs its OWN
 * chm_comm (collectives of one communicator must be issued in one order by every rank).                                          */
int chm_like_clone(const chm_like* src, chm_like** out);
int chm_sel_clone(const chm_sel* src, chm_sel** out);

/* hyperlikelihood.compute_all / __call__ for nb draws (likelihood.py:307-338).  `like` or `sel` may be NULL:
 * with sel == NULL only the numerator outputs are produced; with like == NULL only N_exp.
 * With comm != NULL the shard partials are summed over ranks with one RCCL all-reduce of 3*nb doubles
 * (what CHIMERA/parallel.py:366-376,406-407 intended with mpi4jax.allreduce) before the combination;
 * E_total is the number of events over all shards (ignored when comm == NULL).                        */
int chm_eval(chm_like* like, chm_sel* sel, chm_comm* comm, const chm_params* params, int32_t nb,
             int64_t E_total, chm_out* out);
/* chm_eval with plug-in pieces evaluated by the caller (tab may be NULL = chm_eval).  Slow path: the tables travel host ->
 * device on every call.  Source-frame quantities for the caller's functions: z = chm_model_eval(CHM_F_Z_FROM_DGW, dL),
 * m_src = m_det / (1 + z).                                                                                               */
int chm_eval_tabulated(chm_like* like, chm_sel* sel, chm_comm* comm, const chm_params* params, int32_t nb,
                       int64_t E_total, const chm_tab* tab, chm_out* out);

/* Elementwise model functions on the device (cosmo.py:122-264, mass.py:334-341, rate.py:96-122),
 * used by the Python free functions and by compute_z_grids (pop_wrapper.py:133-208).                  */
enum { CHM_F_E = 0, CHM_F_INT_INVE, CHM_F_DCR, CHM_F_DCT, CHM_F_DL, CHM_F_DDLDZ, CHM_F_DVCDZ, CHM_F_VC,
       CHM_F_XI, CHM_F_Z_FROM_DGW, CHM_F_RATE, CHM_F_PM1M2, CHM_F_PRIMARY, CHM_F_SECONDARY, CHM_F_SMOOTHING,
       CHM_F_PM1M2_FUSED /* the reduced-operation form of PM1M2 used inside the per-sample kernels (for tests) */,
       /* generic helpers of mass.py, parameters carried by a mass struct: tpl_cdf(alpha, m_low, m) with tpl(alpha = -alpha, m_low)
        * (mass.py:247-252); gaussian(x, mu, sigma) with plp(mu_g, sigma_g) (:267-269); truncated_gaussian(x, mu, sigma, x_min,
        * x_max) with plp(mu_g, sigma_g, m_low = x_min, m_high = x_max) (:271-279) */
       CHM_F_TPL_CDF, CHM_F_GAUSSIAN, CHM_F_TRUNC_GAUSSIAN };
/* out[i] = f(a[i] [, b[i]]):  cosmology functions take a = z and optional b = original distances
 * (cosmo.py:155-221; b may be NULL); Z_FROM_DGW takes a = dGW; RATE takes a = z; PM1M2 / SECONDARY take
 * a = m1, b = m2 (SECONDARY: a = m2, b = m1); PRIMARY / SMOOTHING take a = m.                           */
int chm_model_eval(const chm_params* p, int32_t func, const double* a, const double* b, int64_t n,
                   double* out, int32_t device);
/* Per-draw tables (cosmo.py:43-46, 263; mass.py:45-52). Any pointer may be NULL.
 * zt, It, dLt: (z_grid_res,)   mgrid, cdf_m2: (mass_grid_res,)   scalars: [norm_p_m1, fR]             */
int chm_model_tables(const chm_params* p, double* zt, double* It, double* dLt, double* mgrid,
                     double* cdf_m2, double* scalars, int32_t device);

/* pixelated_catalog.precompute_p_cat (catalog.py:152-195, _sum_gaussians_ucv :212-221): per (event, pixel) the sum over the
 * pixel's galaxies of N(z | z_gal, sigma_gal) dVc/dz(z), each normalised by its trapezoid integral over the event grid,
 * weighted and divided by the summed weights; non-finite entries -> 0.  The galaxies of pixel (e,p) are entries
 * [offsets[e*P+p], offsets[e*P+p+1]) of gal_z / gal_sig / gal_w (host selection by HEALPix index and z range:
 * catalog.py:143-150).  Writes p_cat rows for every (e,p) -- rows of padded pixels are left to the caller (-100).        */
typedef struct chm_pcat_desc {
  int32_t E, P, Z, device;
  const double*  z_grids;        /* (E,Z)                                                                */
  const int64_t* offsets;        /* (E*P+1) CSR offsets into the galaxy entry arrays                     */
  const double*  gal_z;          /* (nnz,) galaxy redshifts                   catalog.py:159             */
  const double*  gal_sig;        /* (nnz,) z_err * (1 + z_gal)                catalog.py:115,160         */
  const double*  gal_w;          /* (nnz,) host weights                       catalog.py:114,162         */
  const double*  weight_grid;    /* (E,Z) or NULL.  NULL: the Gaussians are weighted by dVc/dz of `cosmo` (_sum_gaussians_ucv,
                                  * catalog.py:212-221).  Given: by these values -- p_bkg(cosmo, z_grid) of the completeness model,
                                  * evaluated by the caller (_sum_gaussians_pbkg, catalog.py:223-231; sumgauss='pbkg')            */
} chm_pcat_desc;
int chm_pcat_compute(const chm_params* cosmo, const chm_pcat_desc* desc, double* p_cat /* (E,P,Z) */);

/* 2-D sky-localisation density at the pixel centres: jax_gkde_nd((ra, dec) samples, pixel centres) of pixelize_gw_catalog
 * (CHIMERA/data.py:343-345, CHIMERA/utils/math.py:95-148; unweighted Gaussian KDE, Scott factor, covariance whitening).
 * ra, dec: (E,S); ra_pix, dec_pix: (E,P) with npix[e] valid entries per event; out: (E,P), entries >= npix[e] untouched.   */
int chm_kde2d_pixels(int32_t E, int32_t S, int32_t P, const double* ra, const double* dec, const double* ra_pix,
                     const double* dec_pix, const int32_t* npix, double* out, int32_t device);

/* Stand-alone forms of CHIMERA/utils/math.py -- the building blocks chm_eval fuses, for callers that use them directly
 * (host arrays in, host arrays out, dense sums in the reference's order of operations):
 *   chm_kde1d      kde1d(dataset, grid, weights, kernel, bw_method)            math.py:52-89   kernel 0 = 'epan', 1 = 'gauss';
 *                  bw_method 0 = 'scott' / None, 1 = 'silverman', 2 = scalar (bw_scalar);  weights NULL = None
 *   chm_binning1d  binning1d(dataset, weights, num_bins) -> centres, counts     math.py:32-46
 *   chm_gkde_nd    jax_gkde_nd / numba_gkde_nd(dataset (d,N), points (d,M), weights, bw_method), in_log=False, d <= 4
 *                                                                                math.py:95-148, 154-229
 *   chm_gkde_nd_log  the same with in_log=True: logsumexp_j(log W_j + log_norm - |.|^2 / 2), np.logaddexp from -inf in dataset order
 *                                                                                math.py:217-227 (the numba kernel; the JAX branch's in_log raises, Q11)
 *   chm_trapz      trapz(y, x, axis=-1) of `rows` rows of n points; x one row (x_per_row = 0) or one per row   math.py:10-16
 *   chm_cumtrapz   cumtrapz(y, x) of one row                                     math.py:22-26                                  */
int chm_kde1d(const double* dataset, const double* weights, int64_t N, const double* grid, int64_t G, int32_t kernel,
              int32_t bw_method, double bw_scalar, double* out, int32_t device);
int chm_binning1d(const double* dataset, const double* weights, int64_t N, int32_t num_bins, double* centers, double* counts,
                  int32_t device);
int chm_gkde_nd(const double* dataset, const double* weights, int32_t d, int64_t N, const double* points, int64_t M,
                int32_t bw_method, double bw_scalar, double* out, int32_t device);
int chm_gkde_nd_log(const double* dataset, const double* weights, int32_t d, int64_t N, const double* points, int64_t M,
                    int32_t bw_method, double bw_scalar, double* out, int32_t device);
int chm_trapz(const double* y, const double* x, int64_t rows, int32_t n, int32_t x_per_row, double* out, int32_t device);
int chm_cumtrapz(const double* y, const double* x, int32_t n, double* out, int32_t device);

/* Event/injection sharding across GPUs: one process per GPU, RCCL over xGMI.
 * Replaces the MPI layer CHIMERA/parallel.py:94-99,68-73,366-376 (dead code in v2.0.0).               */
int chm_comm_unique_id(char id[128]);                       /* rank 0; broadcast the bytes out-of-band */
int chm_comm_init_rank(const char id[128], int32_t nranks, int32_t rank, int32_t device, chm_comm** out);
int chm_comm_destroy(chm_comm* c);
/* sum a small fp64 vector over ranks in place (host pointer; staged through the device). */
int chm_comm_allreduce_sum(chm_comm* c, double* buf, int32_t n);
/* number of ranks RCCL reports for the communicator (ncclCommCount); 0 for a NULL / dead handle.  The sharded job checks it
 * against the launcher's WORLD_SIZE (the partition of CHIMERA/parallel.py:68-73,94-99 assumes every rank is present).   */
int chm_comm_nranks(chm_comm* c);
/* Several evaluation lanes per rank (chm_like_clone), each with a communicator of its own and a host thread of its own: collectives must reach
 * the device in the same order on every rank.  chm_comm_set_ticket(c, t) makes the NEXT chm_eval on c enqueue its all-reduce only after the
 * calls carrying tickets < t (on any communicator of this process) have enqueued theirs; the job numbers its steps identically on every rank
 * (step k: ticket k).  chm_comm_ticket_reset(n): the next ticket to be served.  Calls without a ticket are not sequenced.                   */
int chm_comm_set_ticket(chm_comm* c, int64_t ticket);
int chm_comm_ticket_reset(int64_t next);
/* forfeit ticket t: a step that will not reach its collective (its host thread failed before chm_eval) lets the higher tickets pass.  A ticketed
 * call whose turn does not come within 120 s fails with CHM_E_RCCL instead of hanging the lane and its RCCL peers.                              */
int chm_comm_ticket_skip(int64_t ticket);
/* [r6] The turn passes to ticket t only when EVERY ticket below t has enqueued its collective or been forfeited (a skipped ticket never lets a higher
 * one overtake a lower one that is still on its way).  chm_comm_ticket_timeout(ms): how long a ticketed call waits for its turn (default 120 000 ms);
 * on timeout the call fails with CHM_E_RCCL once and its ticket is forfeited.  chm_comm_ticket_wait / _done: the sequencer on its own (host only) --
 * wait for the turn of ticket t / pass it on -- for callers that place collectives of their own between evaluations.                              */
int chm_comm_ticket_timeout(int64_t milliseconds);
int chm_comm_ticket_wait(int64_t ticket);
int chm_comm_ticket_done(int64_t ticket);
/* hipDeviceSynchronize on `device`: the barrier bracket of a timed region (chm_eval itself returns after its stream drained). */
int chm_device_synchronize(int32_t device);
/* PCI bus id of `device` ("0000:c1:00.0", NUL-terminated; len >= 16): a sharded job (CHIMERA/parallel.py:94-99: one rank per chunk of events) records one per
 * rank, so that its report shows N ranks on N distinct GPUs.                                                                                      */
int chm_device_pci_bus_id(int32_t device, char* out, int32_t len);

/* Timing of the last chm_eval on a handle, from HIP events recorded on the handle's own stream:
 * ms[0] = whole evaluation, ms[1] = tables, ms[2] = sample stage, ms[3] = KDE+integrand kernel,
 * ms[4] = selection kernel, ms[5] = reduce/combine.  Used by bench.py for the roofline line.          */
int chm_last_timing(chm_like* like, chm_sel* sel, double ms[8]);
/* Diagnostics of the 3-D mode (kind_p_gw3d='full', likelihood.py:211-260): how many (draw, event, pixel) triples of the last chm_eval of nb
 * draws on this handle were left to the general KDE kernel by the sample-stationary one (a stretch of the event's z grid that is not
 * uniform, a grid step of more than 15/32 kernel widths, more than 1024 grid points inside the mask).  Both
 * kernels compute the same sums; the tests use the count to know which one they are looking at.                                          */
int chm_like_full_general_pixels(chm_like* like, int32_t nb, int64_t* count);

/* Evaluation options of a handle (a clone starts with its source's).  The reference steers its evaluation through constructor arguments of
 * the jitted object alone (CHIMERA/likelihood.py:48-62 are jit-static); this library likewise reads NO environment variable -- what a call
 * computes, and on which streams, follows from the handle.  A call that carries both handles uses the options of `like`.
 *   CHM_OPT_SERIAL        1: every kernel of a call on one stream (per-kernel timings without overlap)
 *   CHM_OPT_GROUPS        event groups alternating between two streams: 0 automatic (one per 250 events, at most 8, for calls of more than
 *                         8 draws), 1 one group, n <= 128
 *   CHM_OPT_FUSED         the fused event kernel (one block per (event, draw): samples, statistics, histograms, KDE, integrand; results equal
 *                         to the separate kernels' to rounding, ~1e-15 per event): 0 never (default), 1 calls of <= 8 draws, 2 every call.
 *                         [r5] A VARIANT build only (-DCHM_WITH_FUSED; chm_has_fused() tells): it is slower than the separate kernels at every
 *                         call size measured (profiles/r04/ab_fused_event_kernel.txt), so the release library refuses values > 0
 *   CHM_OPT_TIMING        0 no timing events in the streams, 1 (default) the whole evaluation only (ms[0], ms[1]), 2 per-kernel events too (ms[2..6]; ~4 us of stream time each)
 *   CHM_OPT_GRAPH_MAX_NB  calls of at most this many draws without per-event outputs are replayed from a HIP graph (default 8; 0: never)
 *   CHM_OPT_SPIN_WAIT     1 (default): calls of <= 8 draws poll their stream for completion instead of sleeping on an interrupt
 * Options >= 100 select other kernels for the same quantity, launch geometries or switch a safeguard off (CHM_OPT_DIAG_NO_DENSE_NODE gives
 * WRONG results for weights spanning many decades): they exist for same-box A/B runs and for the tests that compare code paths, and are
 * refused (CHM_E_ARG) unless the library was built with -DCHM_DIAG -- chm_diag_build() tells which build is loaded.                      */
enum {
  CHM_OPT_SERIAL = 1, CHM_OPT_GROUPS = 2, CHM_OPT_FUSED = 3, CHM_OPT_TIMING = 4, CHM_OPT_GRAPH_MAX_NB = 5, CHM_OPT_SPIN_WAIT = 6,
  CHM_OPT_DIAG_FULL_CHAIN = 100, CHM_OPT_DIAG_NO_DENSE_NODE = 101, CHM_OPT_DIAG_MARG_GENERIC = 102, CHM_OPT_DIAG_SAMPLES_GENERIC = 103,
  CHM_OPT_DIAG_SELECTION_GENERIC = 104, CHM_OPT_DIAG_NO_GRID_PREP = 105, CHM_OPT_DIAG_ZF_FULL = 106, CHM_OPT_DIAG_KDE_IPW = 107,
  CHM_OPT_DIAG_SAMP_CPB = 108, CHM_OPT_DIAG_SELF_BLOCKS = 109, CHM_OPT_DIAG_FEW_NB = 110, CHM_OPT_DIAG_NO_ZERO_COPY = 111,
  CHM_OPT_DIAG_NO_ZF_SEL = 112, CHM_OPT_DIAG_HOST_PROF = 113, CHM_OPT_DIAG_FUSED_NW = 114,
  CHM_OPT_DIAG_POISON = 115       /* bit mask of per-call workspaces filled with a finite garbage pattern before every evaluation: a result that moves read one of them before writing it */
};
int chm_like_set_option(chm_like* like, int32_t option, int64_t value);
int chm_sel_set_option(chm_sel* sel, int32_t option, int64_t value);
int chm_diag_build(void);                                   /* 1: built with -DCHM_DIAG (diagnostic options, CHM_* environment defaults) */
int chm_has_fused(void);                                    /* 1: built with -DCHM_WITH_FUSED (the fused event kernel; CHM_OPT_FUSED > 0 accepted) */

#ifdef __cplusplus
}
#endif
#endif
