// chimera_hip.hip -- C ABI (include/chimera_hip.h) over the HIP kernels in chm_kernels.h.  gfx950 only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/chimera_hip.h"
#include "chm_kernels.h"
// [r5] the fused per-(event, draw) kernel of round 4 (chm_fused.h: parity-green, 5.8x less fabric traffic, 40 % slower -- two waves per SIMD under 78.75 KB
// of LDS per block, profiles/r04/ab_fused_event_kernel.txt) is NOT part of the release library: -DCHM_WITH_FUSED builds (scripts/build_variant.sh
// fused -DCHM_WITH_FUSED; tests/test_zz_variant_builds.py) compile it and accept CHM_OPT_FUSED > 0
#ifdef CHM_WITH_FUSED
#include "chm_fused.h"
#endif

#define CHM_MAXP 1024
static_assert(sizeof(chm_params) % 8 == 0, "chm_params layout");

// ------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { \
  return fail(_e == hipErrorOutOfMemory ? CHM_E_NOMEM : CHM_E_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } } while (0)
#define NCCLCHK(x) do { ncclResult_t _r = (x); if (_r != ncclSuccess) { \
  return fail(CHM_E_RCCL, std::string(#x) + ": " + ncclGetErrorString(_r)); } } while (0)

extern "C" const char* chm_version(void) { return "chimera_hip 0.1.0 (gfx950)"; }
extern "C" const char* chm_last_error(void) { return g_err.c_str(); }
extern "C" int chm_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

// ------------------------------------------------------------------------------------------------------
// evaluation options of a handle (chm_like_set_option / chm_sel_set_option, include/chimera_hip.h)
// ------------------------------------------------------------------------------------------------------
// The release library reads NO environment variable: what a call computes and how it is scheduled depends on the handle's options alone.
// The first group is part of the product; the second group (other kernels for the same quantity, launch geometry, switched-off safeguards)
// exists for same-box A/B runs and for the tests that compare code paths -- set_option refuses it unless the library was built with -DCHM_DIAG
// (scripts/build_variant.sh diag -DCHM_DIAG), and only such a build takes its initial values from CHM_* environment variables, once, when
// a handle is created.
#define CHM_MAX_GROUPS 128     // event groups of one call (CHM_OPT_GROUPS): 4 timing events + 1 fork event each per context
struct Opts {
  int serial = 0;            // 1: every kernel of a call on one stream
  int groups = 0;            // event groups alternating between two streams (0: automatic; 1: one group)
  int fused = 0;             // fused event kernel (chm_fused.h): 0 never, 1 few-draw calls, 2 every call
  int timing = 1;            // 0: no timing events in the streams; 1 (default): the whole evaluation only; 2: per-kernel events too (each record costs ~4 us of stream time)
  int graph_max_nb = 8;      // calls of at most this many draws are replayed from a HIP graph (0: never)
  int spin_wait = 1;         // few-draw calls poll the stream for completion instead of sleeping on an interrupt (the wake-up is part of their latency)
  // ---- diagnostics (-DCHM_DIAG builds only)
  int full_chain = 1;        // full mode: 0 = the general kernel alone
  int no_dense = 0;          // standard GW kernel without the dense redo (shows the limit of the prefix-sum form: WRONG results for extreme weights)
  int marg_generic = 0, samples_generic = 0, selection_generic = 0, no_grid_prep = 0, zf_full = 0;
  int kde_ipw = 0, samp_cpb = 0, self_blocks = 8192, few_nb = 8, no_zero_copy = 0, no_zf_sel = 0, host_prof = 0;
  int fused_nw = 0;          // 16: few-draw calls of the fused event kernel with 16 waves per block (-DCHM_FUSED_NW16 builds; A/B)
  long long poison = 0;      // bit mask of the per-call workspaces overwritten with a finite garbage pattern before every evaluation (like_poison_ws)
  long long epoch = 0;       // bumped by every set_option: part of the key of the captured graph (a replayed graph must not outlive the options it was captured under)
};
#ifdef CHM_DIAG
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static bool env_set(const char* name) { return getenv(name) != nullptr; }
#endif
static void opts_init(Opts& o) {
#ifdef CHM_DIAG
  o.serial = env_set("CHM_SERIAL"); o.groups = env_int("CHM_GROUPS", 0); o.fused = env_int("CHM_FUSED", 0);
  o.timing = env_set("CHM_NO_TIMING") ? 0 : (env_set("CHM_TIMING_ALL") ? 2 : 1);
  o.graph_max_nb = env_int("CHM_GRAPH_MAX_NB", 8); o.spin_wait = env_set("CHM_SYNC_BLOCK") ? 0 : 1;
  o.full_chain = env_int("CHM_FULL_CHAIN", 1); o.no_dense = env_set("CHM_NO_DENSE_NODE");
  o.marg_generic = env_set("CHM_MARG_GENERIC"); o.samples_generic = env_set("CHM_SAMPLES_GENERIC"); o.selection_generic = env_set("CHM_SELECTION_GENERIC");
  o.no_grid_prep = env_set("CHM_NO_GRID_PREP"); o.zf_full = env_set("CHM_ZF_FULL"); o.kde_ipw = env_int("CHM_KDE_IPW", 0);
  o.samp_cpb = env_int("CHM_SAMP_CPB", 0); o.self_blocks = env_int("CHM_SELF_BLOCKS", 8192); o.few_nb = env_int("CHM_FEW_NB", 8);
  o.no_zero_copy = env_set("CHM_NO_ZERO_COPY"); o.no_zf_sel = env_set("CHM_NO_ZF_SEL"); o.host_prof = env_set("CHM_HOST_PROF");
  o.fused_nw = env_int("CHM_FUSED_NW", 0);
#else
  (void)o;
#endif
}
static int opts_set(Opts& o, int32_t option, int64_t value) {
  const int v = (int)value;
  o.epoch++;
  switch (option) {
    case CHM_OPT_SERIAL: o.serial = v != 0; return CHM_OK;
    case CHM_OPT_GROUPS: if (v < 0 || v > CHM_MAX_GROUPS) return fail(CHM_E_ARG, "chm_*_set_option: CHM_OPT_GROUPS must be in [0, 128]"); o.groups = v; return CHM_OK;
    case CHM_OPT_FUSED: if (v < 0 || v > 2) return fail(CHM_E_ARG, "chm_*_set_option: CHM_OPT_FUSED must be 0, 1 or 2");
#ifndef CHM_WITH_FUSED
      if (v != 0) return fail(CHM_E_ARG, "chm_*_set_option: CHM_OPT_FUSED > 0 -- this library was built without -DCHM_WITH_FUSED (the fused event kernel is a variant build)");
#endif
      o.fused = v; return CHM_OK;
    case CHM_OPT_TIMING: if (v < 0 || v > 2) return fail(CHM_E_ARG, "chm_*_set_option: CHM_OPT_TIMING must be 0, 1 or 2"); o.timing = v; return CHM_OK;
    case CHM_OPT_GRAPH_MAX_NB: if (v < 0) return fail(CHM_E_ARG, "chm_*_set_option: CHM_OPT_GRAPH_MAX_NB must be >= 0"); o.graph_max_nb = v; return CHM_OK;
    case CHM_OPT_SPIN_WAIT: o.spin_wait = v != 0; return CHM_OK;
    default: break;
  }
#ifdef CHM_DIAG
  switch (option) {
    case CHM_OPT_DIAG_FULL_CHAIN: o.full_chain = v != 0; return CHM_OK;
    case CHM_OPT_DIAG_NO_DENSE_NODE: o.no_dense = v != 0; return CHM_OK;
    case CHM_OPT_DIAG_MARG_GENERIC: o.marg_generic = v != 0; return CHM_OK;
    case CHM_OPT_DIAG_SAMPLES_GENERIC: o.samples_generic = v != 0; return CHM_OK;
    case CHM_OPT_DIAG_SELECTION_GENERIC: o.selection_generic = v != 0; return CHM_OK;
    case CHM_OPT_DIAG_NO_GRID_PREP: o.no_grid_prep = v != 0; return CHM_OK;
    case CHM_OPT_DIAG_ZF_FULL: o.zf_full = v != 0; return CHM_OK;
    case CHM_OPT_DIAG_KDE_IPW: o.kde_ipw = v; return CHM_OK;
    case CHM_OPT_DIAG_SAMP_CPB: o.samp_cpb = v; return CHM_OK;
    case CHM_OPT_DIAG_SELF_BLOCKS: o.self_blocks = v > 0 ? v : 8192; return CHM_OK;
    case CHM_OPT_DIAG_FEW_NB: o.few_nb = v; return CHM_OK;
    case CHM_OPT_DIAG_NO_ZERO_COPY: o.no_zero_copy = v != 0; return CHM_OK;
    case CHM_OPT_DIAG_NO_ZF_SEL: o.no_zf_sel = v != 0; return CHM_OK;
    case CHM_OPT_DIAG_HOST_PROF: o.host_prof = v != 0; return CHM_OK;
    case CHM_OPT_DIAG_FUSED_NW: o.fused_nw = v; return CHM_OK;
    case CHM_OPT_DIAG_POISON: o.poison = v; return CHM_OK;
    default: break;
  }
#else
  if (option >= CHM_OPT_DIAG_FULL_CHAIN && option <= CHM_OPT_DIAG_POISON)
    return fail(CHM_E_ARG, "chm_*_set_option: diagnostic option -- this library was built without -DCHM_DIAG");
#endif
  return fail(CHM_E_ARG, "chm_*_set_option: unknown option");
}
extern "C" int chm_has_fused(void) {
#ifdef CHM_WITH_FUSED
  return 1;
#else
  return 0;
#endif
}
extern "C" int chm_diag_build(void) {
#ifdef CHM_DIAG
  return 1;
#else
  return 0;
#endif
}

// ------------------------------------------------------------------------------------------------------
// per-device evaluation context: stream, per-draw parameter block + tables, staging buffers
// ------------------------------------------------------------------------------------------------------
struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int nb_cap = 0, TcMax = 0, TmMax = 0;
  DevParams* d_params = nullptr;
  DevParams* h_params = nullptr;       // pinned
  double *zt = nullptr, *It = nullptr, *dLt = nullptr, *mg = nullptr, *cdf = nullptr, *tmp = nullptr;
  double* rec = nullptr;               // (nb, TcMax, 4) node records of the fast sample stage (k_tables)
  double *zt_c = nullptr, *lz_c = nullptr;   // (TcMax) z_grid_interp and log(1 + z) of its nodes for (zc_zmax, zc_Tc): draw-independent (k_znodes)
  double zc_zmax = -1.; int zc_Tc = 0;
  double* d_evpart = nullptr;          // (nb, nblk_ev) block sums of log L_i
  int evpart_cap = 0;
  double* d_partials = nullptr;        // (nb,3)
  double* d_out3 = nullptr;            // (nb,3)
  double* h_out = nullptr;             // pinned (nb,6)
  long long* h_seq = nullptr;          // pinned (1 + nb): [0] the sequence number of the call (host -> device), [1 + b] draw b's completion flag (device -> host), see wait_flags
  long long seq = 0;
  hipStream_t stream2 = nullptr;        // second lane of the event-group pipeline
  hipStream_t stream3 = nullptr;        // selection function
  hipEvent_t evg[4 * CHM_MAX_GROUPS] = {};              // per event-group timing: [4g+0..1] sample stage, [4g+2..3] GW kernel
  hipEvent_t evf[CHM_MAX_GROUPS] = {};              // per event-group fork of the per-z-factor kernel onto the other lane
  double *d_lle = nullptr, *d_nle = nullptr;   // (nb,E) per-event outputs (log L_i, L_i), grown on demand, kept across calls
  size_t lle_cap = 0, nle_cap = 0;
  hipEvent_t ev[8] = {};                // timing on `stream`
  hipEvent_t evb[4] = {};               // fork/join + timing on `stream2`
  double ms[8] = {};
  int t_ngroups = 0; bool t_like = false, t_sel = false, t_valid = false, t_all = false;
  bool t_pending = false;              // the last call returned through its completion flags: its timing events may not have retired yet (chm_last_timing waits)
  // HIP graph of the few-draw call (the reference-shaped scalar call like(**lambda) is launch-bound: ~10 kernels on three streams):
  // the launch sequence of one configuration is captured once and replayed; `gkey` = everything baked into the captured arguments
  hipGraphExec_t gexec = nullptr;
  std::vector<long long> gkey, gwarm;
  bool init = false;
};

static int ctx_init(Ctx& c, int device) {
  c.device = device;
  // (no process-wide device flags: few-draw calls that want to spin for their result poll their own stream, wait_stream below)
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&c.stream2, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&c.stream3, hipStreamNonBlocking));
  for (int i = 0; i < 4 * CHM_MAX_GROUPS; i++) HIPCHK(hipEventCreate(&c.evg[i]));
  for (int i = 0; i < CHM_MAX_GROUPS; i++) HIPCHK(hipEventCreateWithFlags(&c.evf[i], hipEventDisableTiming));
  for (int i = 0; i < 8; i++) HIPCHK(hipEventCreate(&c.ev[i]));
  for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&c.evb[i]));
  c.init = true;
  return CHM_OK;
}

static void ctx_free_tables(Ctx& c) {
  (void)hipFree(c.d_params); (void)hipHostFree(c.h_params);
  (void)hipFree(c.zt); (void)hipFree(c.It); (void)hipFree(c.dLt); (void)hipFree(c.mg); (void)hipFree(c.cdf); (void)hipFree(c.tmp); (void)hipFree(c.rec); c.rec = nullptr;
  (void)hipFree(c.zt_c); (void)hipFree(c.lz_c); c.zt_c = c.lz_c = nullptr; c.zc_zmax = -1.; c.zc_Tc = 0;
  (void)hipFree(c.d_partials); (void)hipFree(c.d_out3); (void)hipHostFree(c.h_out); (void)hipHostFree(c.h_seq); c.h_seq = nullptr; (void)hipFree(c.d_evpart);
  c.d_evpart = nullptr; c.evpart_cap = 0;
  c.d_params = nullptr; c.h_params = nullptr; c.zt = c.It = c.dLt = c.mg = c.cdf = c.tmp = nullptr;
  c.d_partials = c.d_out3 = nullptr; c.h_out = nullptr;
  c.nb_cap = c.TcMax = c.TmMax = 0;
}

static void ctx_destroy(Ctx& c) {
  if (!c.init) return;
  (void)hipSetDevice(c.device);
  if (c.gexec) { (void)hipGraphExecDestroy(c.gexec); c.gexec = nullptr; }
  ctx_free_tables(c);
  for (int i = 0; i < 8; i++) if (c.ev[i]) (void)hipEventDestroy(c.ev[i]);
  for (int i = 0; i < 4; i++) if (c.evb[i]) (void)hipEventDestroy(c.evb[i]);
  for (int i = 0; i < 4 * CHM_MAX_GROUPS; i++) if (c.evg[i]) (void)hipEventDestroy(c.evg[i]);
  for (int i = 0; i < CHM_MAX_GROUPS; i++) if (c.evf[i]) (void)hipEventDestroy(c.evf[i]);
  (void)hipFree(c.d_lle); (void)hipFree(c.d_nle); c.d_lle = c.d_nle = nullptr; c.lle_cap = c.nle_cap = 0;
  if (c.stream3) (void)hipStreamDestroy(c.stream3);
  if (c.stream2) (void)hipStreamDestroy(c.stream2);
  if (c.stream) (void)hipStreamDestroy(c.stream);
  c.init = false;
}

static int ctx_ensure(Ctx& c, int nb, int Tc, int Tm) {
  if (nb <= c.nb_cap && Tc <= c.TcMax && Tm <= c.TmMax) return CHM_OK;
  HIPCHK(hipStreamSynchronize(c.stream));
  int nbn = nb > c.nb_cap ? nb : c.nb_cap, Tcn = Tc > c.TcMax ? Tc : c.TcMax, Tmn = Tm > c.TmMax ? Tm : c.TmMax;
  ctx_free_tables(c);
  size_t T = Tcn > Tmn ? Tcn : Tmn;
  HIPCHK(hipMalloc(&c.d_params, sizeof(DevParams) * nbn));
  HIPCHK(hipHostMalloc(&c.h_params, sizeof(DevParams) * nbn));
  HIPCHK(hipMalloc(&c.zt, sizeof(double) * nbn * Tcn));
  HIPCHK(hipMalloc(&c.It, sizeof(double) * nbn * Tcn));
  HIPCHK(hipMalloc(&c.dLt, sizeof(double) * nbn * Tcn));
  HIPCHK(hipMalloc(&c.mg, sizeof(double) * nbn * Tmn));
  HIPCHK(hipMalloc(&c.cdf, sizeof(double) * nbn * Tmn));
  HIPCHK(hipMalloc(&c.tmp, sizeof(double) * nbn * ((size_t)Tcn + Tmn)));
  HIPCHK(hipMalloc(&c.rec, sizeof(double) * nbn * (size_t)Tcn * 4));
  HIPCHK(hipMalloc(&c.zt_c, sizeof(double) * Tcn));
  HIPCHK(hipMalloc(&c.lz_c, sizeof(double) * Tcn));
  HIPCHK(hipMalloc(&c.d_partials, sizeof(double) * nbn * 3));
  HIPCHK(hipMalloc(&c.d_out3, sizeof(double) * nbn * 3));
  HIPCHK(hipHostMalloc(&c.h_out, sizeof(double) * nbn * 6, hipHostMallocCoherent | hipHostMallocMapped));
  HIPCHK(hipHostMalloc(&c.h_seq, sizeof(long long) * (nbn + 1), hipHostMallocCoherent | hipHostMallocMapped));
  memset(c.h_seq, 0, sizeof(long long) * (nbn + 1));
  c.nb_cap = nbn; c.TcMax = Tcn; c.TmMax = Tmn;
  return CHM_OK;
}

static int check_params(const chm_params* p) {
  if (p->cosmo_model < 0 || p->cosmo_model > 1) return fail(CHM_E_ARG, "chm_params.cosmo_model out of range");
  if (p->mass_model < 0 || p->mass_model > 2) return fail(CHM_E_ARG, "chm_params.mass_model out of range");
  if (p->rate_model < 0 || p->rate_model > 3) return fail(CHM_E_ARG, "chm_params.rate_model out of range");
  if (p->z_grid_res < 3 || p->z_grid_res > (1 << 22)) return fail(CHM_E_ARG, "chm_params.z_grid_res must be in [3, 2^22]");
  if (p->mass_grid_res < 3 || p->mass_grid_res > (1 << 22)) return fail(CHM_E_ARG, "chm_params.mass_grid_res must be in [3, 2^22]");
  return CHM_OK;
}

// an infinite rate parameter the model reads (gamma; kappa and z_p of the Madau-Dickinson forms): rate.py:96-122 under NumPy / XLA has a value
// class of its own there (merger_rate_special)
static bool params_rate_special(const chm_params* p) {
  const bool md = p->rate_model == 1 || p->rate_model == 3;
  return std::isinf(p->rate[0]) || (md && (std::isinf(p->rate[1]) || std::isinf(p->rate[2])));
}
static void fill_dev_params(const chm_params* p, DevParams* d) {
  memset(d, 0, sizeof(DevParams));
  d->cosmo_model = p->cosmo_model; d->mass_model = p->mass_model; d->rate_model = p->rate_model;
  d->Tc = p->z_grid_res; d->Tm = p->mass_grid_res; d->scale_free = p->scale_free; d->has_catalog = p->has_catalog;
  d->z_max = p->z_max;
  d->H0 = p->cosmo[CHM_C_H0]; d->Om0 = p->cosmo[CHM_C_OM0]; d->Ok0 = p->cosmo[CHM_C_OK0]; d->Or0 = p->cosmo[CHM_C_OR0];
  d->w0 = p->cosmo[CHM_C_W0]; d->wa = p->cosmo[CHM_C_WA]; d->Xi0 = p->cosmo[CHM_C_XI0]; d->n_mg = p->cosmo[CHM_C_N];
  d->Ode0 = 1.0 - d->Om0 - d->Or0 - d->Ok0;                 // cosmo.py:79-81
  d->dH = 299792.458e-3 / d->H0;                            // cosmo.py:82-84
  for (int i = 0; i < 8; i++) d->m[i] = p->mass[i];
  for (int i = 0; i < 4; i++) d->r[i] = p->rate[i];
  d->R0 = p->R0; d->Tobs = p->Tobs; d->zc0 = p->compl_z0; d->zc1 = p->compl_z1;
  d->rate_special = params_rate_special(p) ? 1 : 0;
  // End nodes of the mass grid, jnp.logspace(log10 m_low, log10 m_high)[0], [-1] (mass.py:46).  Whether they pass the
  // `m_low <= m <= m_high` test of tpl_notnorm (mass.py:240-245) decides if the first / last trapezoid node of cdf_m2 and
  // norm_p_m1 counts -- an O(1e-3) effect hanging on the last bit of pow().  The host's libm forms them (as NumPy / the CPU
  // back-end of the reference does), so the decision is the CPU reference's for every (m_low, m_high), not only the defaults.
  // ([r6] memoised per host thread: the draws of a scan or of a chain's batch mostly share m_low / m_high, and the four libm calls per draw were ~15 us of
  //  host time in front of the first kernel of a 128-draw call)
  static thread_local double memo_in[2] = { -1., -1. }, memo_out[2] = { 0., 0. };
  const double in[2] = { p->mass[CHM_M_MLOW], p->mass[CHM_M_MHIGH] };
  for (int i = 0; i < 2; i++) {
    if (memcmp(&in[i], &memo_in[i], sizeof(double)) != 0) { memo_out[i] = pow(10., log10(in[i])); memo_in[i] = in[i]; }
  }
  d->mg_first = memo_out[0]; d->mg_last = memo_out[1];
}

// host part of the per-draw tables: parameter checks, buffers, the draws packed into the pinned staging block
static int ctx_tables_host(Ctx& c, const chm_params* params, int nb, const double* fR_given, int* Tc_out, int* Tm_out) {
  int Tc = 0, Tm = 0;
  for (int b = 0; b < nb; b++) {
    int rc = check_params(&params[b]); if (rc) return rc;
    Tc = params[b].z_grid_res > Tc ? params[b].z_grid_res : Tc;
    Tm = params[b].mass_grid_res > Tm ? params[b].mass_grid_res : Tm;
  }
  int rc = ctx_ensure(c, nb, Tc, Tm); if (rc) return rc;
  for (int b = 0; b < nb; b++) {
    fill_dev_params(&params[b], &c.h_params[b]);
    if (fR_given) { c.h_params[b].fR = fR_given[b]; c.h_params[b].fR_given = 1.; }     // plug-in completeness (chm_tab.fR)
  }
  *Tc_out = Tc; *Tm_out = Tm;
  return CHM_OK;
}

// device part: upload nb draws and build their tables on c.stream
// zero_copy (few draws per call -- the scalar call): k_tables reads the pinned host copy of the parameters itself instead of a copy
// node in front of it (one graph node and ~8 us of stream time less)
static int ctx_tables_enqueue(Ctx& c, int nb, int Tc, int Tm, LutDesc lutA = LutDesc{}, LutDesc lutB = LutDesc{},
                              const double* tab_zt = nullptr, const double* tab_dLt = nullptr, bool zero_copy = false, bool znodes = false) {
  const double* ztc = znodes ? c.zt_c : nullptr; const double* lzc = znodes ? c.lz_c : nullptr;
  const size_t tl = sizeof(double) * 3 * (size_t)(Tc > Tm ? Tc : Tm);
  if (tl > 112 * 1024) zero_copy = false;                   // the long-table variant of k_tables keeps the copy node
  const DevParams* hsrc = zero_copy ? c.h_params : nullptr;
  if (!zero_copy) HIPCHK(hipMemcpyAsync(c.d_params, c.h_params, sizeof(DevParams) * nb, hipMemcpyHostToDevice, c.stream));
  if (tl <= 112 * 1024) {                                   // + 33 KB of static LDS (build_lut scratch, parameter block)
    static size_t tl_allowed = 0;                             // the kernel also holds 33 KB of static LDS: ask as soon as the sum passes 48 KB
    if (tl > 14 * 1024 && tl > tl_allowed) { (void)hipFuncSetAttribute((const void*)k_tables<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl); tl_allowed = tl; }
    // blocks per draw: the scalar call's form splits the cosmology tail over two blocks (grid (nb, 3)); a call of many draws runs cosmology | mass (grid (nb, 2):
    // 256 blocks of 1024 threads at 128 draws, one per CU -- 26 -> 21 us, -0.5 % of the 125-event shard's step: profiles/r06/ab_shard_step_r06.txt)
    hipLaunchKernelGGL(k_tables<true>, dim3(nb, nb > 8 ? 2 : 3), dim3(1024), tl, c.stream, c.d_params, c.zt, c.It, c.dLt, c.mg, c.cdf, c.tmp, c.TcMax, c.TmMax, lutA, lutB, (lutA.nk > 0 || lutB.nk > 0) ? c.rec : nullptr, tab_zt, tab_dLt, hsrc, ztc, lzc);
  } else {
    hipLaunchKernelGGL(k_tables<false>, dim3(nb, 2), dim3(CHM_TABLES_LONG_NT), 0, c.stream, c.d_params, c.zt, c.It, c.dLt, c.mg, c.cdf, c.tmp, c.TcMax, c.TmMax, lutA, lutB, (lutA.nk > 0 || lutB.nk > 0) ? c.rec : nullptr, tab_zt, tab_dLt, hsrc, ztc, lzc);
  }
  HIPCHK(hipGetLastError());
  return CHM_OK;
}

static int ctx_tables(Ctx& c, const chm_params* params, int nb, const double* fR_given = nullptr, LutDesc lutA = LutDesc{}, LutDesc lutB = LutDesc{}) {
  int Tc = 0, Tm = 0;
  int rc = ctx_tables_host(c, params, nb, fR_given, &Tc, &Tm); if (rc) return rc;
  return ctx_tables_enqueue(c, nb, Tc, Tm, lutA, lutB);
}

// ------------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------------
// Device arrays uploaded once by *_create.  A handle and its clones (chm_like_clone / chm_sel_clone: a second evaluation lane on the same
// resident data, with its own streams, tables and workspaces) share them; the last one to be destroyed frees them.
struct OwnedArrays {
  std::vector<void*> v;
  int device = 0;
  ~OwnedArrays() { (void)hipSetDevice(device); for (void* p : v) (void)hipFree(p); }
  void push_back(void* p) { v.push_back(p); }
};
struct chm_like {
  Ctx ctx;
  Opts opts;
  LikeDev L;
  std::shared_ptr<OwnedArrays> owned_sp = std::make_shared<OwnedArrays>();
  int nb_ws = 0;
  bool ws_dump = false;
  // k_samples_fast: sample tiles (uploaded once), key range of the shard's distances, per-call direct-index tables
  SampFast F = {};
  bool fast_ok = false;
  double dl_gmin = 0., dl_gmax = 0.;
  unsigned short* d_lut = nullptr; int* d_lutinfo = nullptr;
  // draw-independent part of the per-z factors (k_grid_prep), made for the (z_max, z_grid_res) below
  int* d_zg_i = nullptr; double *d_zg_t = nullptr, *d_zg_lz = nullptr;
  double zg_zmax = -1.; int zg_Tc = 0;
  // k_marg_fused (chm_fused.h): pixel of every sorted sample, largest distance of every pixel, plain-event flags (uploaded once, marginalized mode);
  // widest event in octaves of distance / in keys of the direct-index table (LDS reserved per block)
#ifdef CHM_WITH_FUSED
  FusedDesc FD = {};
#endif
  bool fused_ok = false;
  const unsigned char* d_ev_bad = nullptr;   // (E) 1 = an input that multiplies EVERY grid point of the event's integrand holds a NaN (p_cat of a live pixel, P_compl, gw_loc2d_pdf, z_grids):
                                             // the reference's trapz sums 0 * NaN = NaN also where p_gw is zero -> L_i = NaN for every draw; the kernels skip those points, the reductions apply the flag
  bool neg_prior = false;    // some pe_prior < 0: negative sample weights -- the standard GW kernel's rounding bound and empty-bin shortcuts assume weights >= 0
  double ev_oct_max = 0.; int ev_nk_max = 0;
  int* d_redo = nullptr;          // (shared with the clones) count of dense redos, diagnostics
#ifdef CHM_PROBE
  LutDesc probe_lut = {}; size_t probe_lds_fast = 0;      // direct-index table and LDS size of the last call's fast sample stage (scripts/gw_loop_probe.hip)
#endif
};
struct chm_sel {
  Ctx ctx;
  Opts opts;
  SelDev S;
  std::shared_ptr<OwnedArrays> owned_sp = std::make_shared<OwnedArrays>();
  int nb_ws = 0;
  // k_selection_fast: key range of the shard's distances, per-call direct-index tables (as chm_like::F)
  LutDesc lut = {};
  bool fast_ok = false;
  double dl_gmin = 0., dl_gmax = 0.;
  unsigned short* d_lut = nullptr; int* d_lutinfo = nullptr;
};
struct chm_comm {
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;
  double* d_buf = nullptr; int cap = 0;
  long long ticket = -1;               // >= 0: the next chm_eval on this communicator enqueues its collective in ticket order (chm_comm_set_ticket)
};

// Collective sequencer of the process.  Several evaluation lanes of one rank -- each with its own communicator, each driven by its own host
// thread -- must hand their collectives to the device in the SAME order on every rank: the streams of the lanes share a handful of
// hardware queues, a queue runs its kernels in submission order, and an RCCL kernel spins until its peers on the other ranks run; rank A
// queueing lane 0's all-reduce in front of lane 1's while rank B queues them the other way round can leave each waiting for a kernel that sits
// behind the other.  The caller numbers the calls (chm_comm_set_ticket: step k of the job carries ticket k on every rank); a ticketed call
// enqueues its all-reduce only when every lower ticket has enqueued its own.
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <set>
#include <atomic>
// `next`: the lowest ticket that has neither enqueued its collective nor been forfeited.  A served or forfeited ticket k advances `next` only while
// every ticket below it is served or forfeited too ([r6], ADVICE r5: a skip of ticket k used to set next = k + 1 outright, letting ticket k + 1 pass
// ticket k - 1 on this rank only -- the very reordering the tickets exist to prevent); tickets closed ahead of their turn wait in `closed`.
static struct CollSeq {
  std::mutex m; std::condition_variable cv; long long next = 0; std::set<long long> closed;
  std::atomic<long long> timeout_ms{120000};
  void close(long long k) {                                   // (m held) ticket k is served or forfeited
    if (k < next) return;
    closed.insert(k);
    while (!closed.empty() && *closed.begin() == next) { closed.erase(closed.begin()); next++; }
  }
} g_seq;
// wait for the turn of `ticket`: true when every lower ticket is served or forfeited; false after the timeout (chm_comm_ticket_timeout, 120 s by default:
// a lane whose host thread raised before chm_eval, a call made with collective=False) -- the ticket is then forfeited, so that the lanes behind it go on
static bool ticket_wait(long long ticket) {
  std::unique_lock<std::mutex> lk(g_seq.m);
  const long long ms = g_seq.timeout_ms.load();
  if (g_seq.cv.wait_for(lk, std::chrono::milliseconds(ms), [&] { return g_seq.next >= ticket; })) return true;
  g_seq.close(ticket);
  lk.unlock();
  g_seq.cv.notify_all();
  return false;
}
static void ticket_close(long long ticket) {
  { std::lock_guard<std::mutex> lk(g_seq.m); g_seq.close(ticket); }
  g_seq.cv.notify_all();
}
struct TicketTurn {
  chm_comm* c; bool held = false;
  explicit TicketTurn(chm_comm* c_) : c(c_ && c_->ticket >= 0 ? c_ : nullptr) {}
  // false: the lower tickets did not come in time: the caller fails with CHM_E_RCCL instead of hanging this lane and its RCCL peers; the turn is given
  // up for good (the destructor does not wait a second time)
  bool acquire() {
    if (c && !held) {
      if (!ticket_wait(c->ticket)) { c->ticket = -1; c = nullptr; return false; }
      held = true;
    }
    return true;
  }
  void release() { if (c && held) { const long long k = c->ticket; c->ticket = -1; held = false; c = nullptr; ticket_close(k); } }
  // a call that failed BEFORE its collective forfeits its ticket at once (nothing of it will reach the device: waiting for the lower tickets first
  // would only delay the lanes behind it)
  ~TicketTurn() { if (c) { const long long k = c->ticket; c->ticket = -1; ticket_close(k); } }
};
extern "C" int chm_comm_set_ticket(chm_comm* c, int64_t ticket) {
  if (!c) return fail(CHM_E_ARG, "chm_comm_set_ticket: null communicator");
  c->ticket = (long long)ticket;
  return CHM_OK;
}
// forfeit ticket k (a step that will not reach its collective: the lanes behind it go on once the tickets below k are through) -- e.g. from a
// try / finally around a failed step
extern "C" int chm_comm_ticket_skip(int64_t ticket) { ticket_close((long long)ticket); return CHM_OK; }
extern "C" int chm_comm_ticket_reset(int64_t next) {
  { std::lock_guard<std::mutex> lk(g_seq.m); g_seq.next = (long long)next; g_seq.closed.clear(); }
  g_seq.cv.notify_all();
  return CHM_OK;
}
// how long a ticketed call waits for the lower tickets before it fails with CHM_E_RCCL (milliseconds; default 120 000).  A job whose steps may
// legitimately take longer (first-call allocations of a large full-mode shard) raises it; tests lower it.
extern "C" int chm_comm_ticket_timeout(int64_t milliseconds) {
  if (milliseconds <= 0) return fail(CHM_E_ARG, "chm_comm_ticket_timeout: need a positive number of milliseconds");
  g_seq.timeout_ms.store((long long)milliseconds);
  return CHM_OK;
}
// the sequencer on its own (host only; what a ticketed chm_eval does around its all-reduce): wait for the turn of `ticket` / pass it on.  For
// callers that issue collectives of their own between evaluations, and for tests of the ordering rules that need no GPU.
extern "C" int chm_comm_ticket_wait(int64_t ticket) {
  if (!ticket_wait((long long)ticket)) return fail(CHM_E_RCCL, "chm_comm_ticket_wait: the calls with lower tickets never passed their turn on (timeout; the ticket is forfeited)");
  return CHM_OK;
}
extern "C" int chm_comm_ticket_done(int64_t ticket) { ticket_close((long long)ticket); return CHM_OK; }

template <class T, class Own>
static int upload(Own& owned, const T* host, size_t n, const T** dev, hipStream_t s) {
  *dev = nullptr;
  if (!host || n == 0) return CHM_OK;
  T* d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(T) * n));
  owned.push_back(d);
  HIPCHK(hipMemcpyAsync(d, host, sizeof(T) * n, hipMemcpyHostToDevice, s));
  *dev = d;
  return CHM_OK;
}

// Stable sort of one event's samples by pixel index (samples in no pixel last); fills perm and seg (P+1 offsets).
static void pixel_sort(const int32_t* pix, int S, int P, std::vector<int>& perm, int* seg) {
  std::vector<int> cnt(P + 2, 0);
  for (int s = 0; s < S; s++) { int q = pix[s]; cnt[(q >= 0 && q < P) ? q + 1 : P + 1]++; }
  // cnt[q+1] = number in pixel q; cnt[P+1] = outside
  std::vector<int> start(P + 1, 0);
  for (int q = 0; q < P; q++) start[q + 1] = start[q] + cnt[q + 1];
  for (int q = 0; q <= P; q++) seg[q] = start[q];
  std::vector<int> cur(start);
  int out_cur = start[P];
  perm.resize(S);
  for (int s = 0; s < S; s++) {
    int q = pix[s];
    if (q >= 0 && q < P) perm[cur[q]++] = s; else perm[out_cur++] = s;
  }
}

extern "C" int chm_like_create(const chm_like_desc* d, chm_like** out) {
  if (!d || !out) return fail(CHM_E_ARG, "chm_like_create: null argument");
  *out = nullptr;
  if (d->E <= 0 || d->S <= 0 || d->Z < 2) return fail(CHM_E_ARG, "chm_like_create: need E > 0, S > 0, Z >= 2");
  if (d->mode < 0 || d->mode > 3) return fail(CHM_E_ARG, "chm_like_create: bad mode");
  if (d->kernel < 0 || d->kernel > 1) return fail(CHM_E_ARG, "chm_like_create: bad kernel");
  if (d->bw_method < 0 || d->bw_method > 2) return fail(CHM_E_ARG, "chm_like_create: bad bw_method");
  if (!d->dL || !d->m1det || !d->m2det || !d->pe_prior || !d->z_grids) return fail(CHM_E_ARG, "chm_like_create: missing sample arrays / z_grids");
  const bool pixelated = d->mode != CHM_MODE_1D;
  if (pixelated) {
    if (d->P <= 0 || d->P > CHM_MAXP) return fail(CHM_E_ARG, "chm_like_create: pixelated modes need 0 < P <= 1024");
    if (!d->p_cat || !d->P_compl || !d->gw_loc2d_pdf || !d->neff_pixels) return fail(CHM_E_ARG, "chm_like_create: missing p_cat / P_compl / gw_loc2d_pdf / neff_pixels");
    if (d->mode == CHM_MODE_MARG && !d->pix_of_sample) return fail(CHM_E_ARG, "chm_like_create: marginalized mode needs pix_of_sample");
    if (d->mode == CHM_MODE_FULL && (!d->ra || !d->dec || !d->ra_pix || !d->dec_pix)) return fail(CHM_E_ARG, "chm_like_create: full mode needs ra, dec, ra_pix, dec_pix");
    if (d->mode == CHM_MODE_FULL && std::isnan(d->cut_grid)) return fail(CHM_E_ARG, "chm_like_create: full mode needs cut_grid");
  }
  if (d->mode != CHM_MODE_FULL && d->binning && d->num_bins < 1) return fail(CHM_E_ARG, "chm_like_create: num_bins must be >= 1");
  int e0 = d->ev_begin, e1 = d->ev_end;
  if (e0 == 0 && e1 == 0) e1 = d->E;
  if (e0 < 0 || e1 > d->E || e0 >= e1) return fail(CHM_E_ARG, "chm_like_create: bad event range");
  int ndev = chm_device_count();
  if (d->device < 0 || d->device >= ndev) return fail(CHM_E_HIP, "chm_like_create: no such HIP device (is a GPU visible?)");

  chm_like* h = new chm_like();
  opts_init(h->opts);
  int rc = ctx_init(h->ctx, d->device);
  if (rc) { delete h; return rc; }
  h->owned_sp->device = d->device;
  hipStream_t s = h->ctx.stream;
  LikeDev& L = h->L;
  memset(&L, 0, sizeof(L));
  const size_t E = e1 - e0, S = d->S, Z = d->Z, P = pixelated ? d->P : 0;
  L.E = (int)E; L.S = d->S; L.Z = d->Z; L.P = (int)P;
  L.mode = d->mode; L.kernel = d->kernel; L.bw_method = d->bw_method; L.binning = d->binning ? 1 : 0; L.num_bins = d->num_bins;
  L.has_cut = std::isnan(d->cut_grid) ? 0 : 1;
  L.G = L.has_cut ? d->Z / 2 : d->Z;                         // likelihood.py:121,188
  L.NC = (int)((S + SAMPLE_CHUNK - 1) / SAMPLE_CHUNK) * SAMPLE_WPB;       // partial records per event: one per chunk and wave of k_samples
  L.bw_scalar = d->bw_scalar; L.cut_grid = d->cut_grid; L.pe_neff = d->pe_neff;
  { const double B = (double)(d->num_bins > 0 ? d->num_bins : 1); L.inv_B = 1. / B; L.std_unit = sqrt((B * B - 1.) / 12.) / B; }   // math.py:67 on uniform centres
  if (L.mode != CHM_MODE_FULL && L.G < 2) { chm_like_destroy(h); return fail(CHM_E_ARG, "chm_like_create: Z//2 must be >= 2 when cut_grid is set"); }
#define UP(field, src, n) do { rc = upload(*h->owned_sp, (src) ? (src) + (size_t)e0 * (n) : (src), (size_t)E * (n), &L.field, s); if (rc) { chm_like_destroy(h); return rc; } } while (0)
  std::vector<double> tmp;                                   // must outlive the async copies below
  std::vector<std::vector<double>> sorted, logs;
  std::vector<int> seg, perm_all;                            // perm_all: original index of every pixel-sorted sample (for chm_tab)
  std::vector<unsigned char> fused_pix, fused_plain;         // (alive until the stream is synchronised below)
  std::vector<double> fused_dlmax, fused_lm1;
  if (d->mode == CHM_MODE_MARG) {
    // marginalized: store every event's samples sorted by pixel, so that each (event, pixel) wave reads one contiguous
    // segment (the device-side form of `pe_pix == pixels[i]`, likelihood.py:179)
    seg.resize(E * (P + 1));
    const double* src[4] = { d->dL, d->m1det, d->m2det, d->pe_prior };
    sorted.assign(4, std::vector<double>(E * S));
    std::vector<int> perm;
    perm_all.resize(E * S);
    for (size_t e = 0; e < E; e++) {
      pixel_sort(d->pix_of_sample + (size_t)(e0 + e) * S, (int)S, (int)P, perm, &seg[e * (P + 1)]);
      for (size_t k = 0; k < S; k++) perm_all[e * S + k] = perm[k];
      for (int a = 0; a < 4; a++) {
        const double* in = src[a] + (size_t)(e0 + e) * S;
        double* o = sorted[a].data() + e * S;
        if (a == 3) { for (size_t k = 0; k < S; k++) { o[k] = 1. / in[perm[k]]; if (in[perm[k]] < 0.) { h->neg_prior = true; L.neg_w = 1; } } }     // the device keeps 1/pe_prior
        else { for (size_t k = 0; k < S; k++) o[k] = in[perm[k]]; }
      }
    }
    const double** dst[4] = { &L.dL, &L.m1det, &L.m2det, &L.pe_prior };
    for (int a = 0; a < 4; a++) { rc = upload(*h->owned_sp, (const double*)sorted[a].data(), E * S, dst[a], s); if (rc) { chm_like_destroy(h); return rc; } }
    logs.assign(2, std::vector<double>(E * S));
    for (size_t k = 0; k < E * S; k++) { logs[0][k] = std::log(sorted[1][k]); logs[1][k] = std::log(sorted[2][k]); }
    rc = upload(*h->owned_sp, (const int*)seg.data(), E * (P + 1), &L.seg_off, s); if (rc) { chm_like_destroy(h); return rc; }
    rc = upload(*h->owned_sp, (const int*)perm_all.data(), E * S, &L.perm, s); if (rc) { chm_like_destroy(h); return rc; }
    // inputs of the fused event kernel (chm_fused.h): the local pixel of every sorted sample (255: none), the largest distance of every
    // pixel's samples, and which events hold only finite positive distances (their min / max z follow from the extreme distances)
#ifdef CHM_WITH_FUSED
    if (P <= 64 && (S & 1) == 0) {
      const size_t NTl = (S + SF_TILE - 1) / SF_TILE;
      fused_pix.assign(E * NTl * SF_TILE, (unsigned char)255);
      fused_dlmax.assign(E * P, NAN);
      fused_plain.assign(E, 1);
      fused_lm1.assign(2 * E, NAN);
      double oct_max = 0.; int nk_max = 0;
      for (size_t e = 0; e < E; e++) {
        const double* x = sorted[0].data() + e * S;
        const int* sg = &seg[e * (P + 1)];
        double lo = INFINITY, hi = -INFINITY;
        for (size_t k = 0; k < S; k++) {
          if (!(std::isfinite(x[k]) && x[k] >= 2.2250738585072014e-308)) fused_plain[e] = 0;
          else { lo = x[k] < lo ? x[k] : lo; hi = x[k] > hi ? x[k] : hi; }
        }
        for (size_t q = 0; q < P; q++) {
          double m = -INFINITY;
          for (int k = sg[q]; k < sg[q + 1]; k++) { fused_pix[e * NTl * SF_TILE + k] = (unsigned char)q; if (std::isfinite(x[k]) && x[k] > m) m = x[k]; }
          if (m > -INFINITY) fused_dlmax[e * P + q] = m;
        }
        {                                                   // log of the extreme primary masses, a hair outwards (the device forms its own logs)
          const double* m1 = sorted[1].data() + e * S;
          double a = INFINITY, bq = 0.;
          for (size_t k = 0; k < S; k++) if (std::isfinite(m1[k]) && m1[k] > 0.) { a = m1[k] < a ? m1[k] : a; bq = m1[k] > bq ? m1[k] : bq; }
          if (a <= bq) { fused_lm1[2 * e] = std::log(a) - 1e-9; fused_lm1[2 * e + 1] = std::log(bq) + 1e-9; }
        }
        if (fused_plain[e] && lo <= hi) {
          int64_t b0, b1; memcpy(&b0, &lo, 8); memcpy(&b1, &hi, 8);
          const int nk = (int)(b1 >> (32 + LUT_SHIFT)) - (int)(b0 >> (32 + LUT_SHIFT)) + 1;
          nk_max = nk > nk_max ? nk : nk_max;
          const double oc = std::log2(hi / lo);
          oct_max = oc > oct_max ? oc : oct_max;
        }
      }
      h->ev_oct_max = oct_max; h->ev_nk_max = nk_max;
      rc = upload(*h->owned_sp, (const unsigned char*)fused_pix.data(), fused_pix.size(), &h->FD.pix_id, s); if (rc) { chm_like_destroy(h); return rc; }
      rc = upload(*h->owned_sp, (const double*)fused_dlmax.data(), fused_dlmax.size(), &h->FD.pix_dlmax, s); if (rc) { chm_like_destroy(h); return rc; }
      rc = upload(*h->owned_sp, (const unsigned char*)fused_plain.data(), fused_plain.size(), &h->FD.ev_plain, s); if (rc) { chm_like_destroy(h); return rc; }
      { const double* dlm = nullptr; rc = upload(*h->owned_sp, (const double*)fused_lm1.data(), fused_lm1.size(), &dlm, s); if (rc) { chm_like_destroy(h); return rc; }
        h->FD.lm1 = reinterpret_cast<const double2*>(dlm); }
      int* rc_ = nullptr;
      if (hipMalloc(&rc_, sizeof(int)) == hipSuccess) { h->owned_sp->push_back(rc_); (void)hipMemsetAsync(rc_, 0, sizeof(int), s); h->d_redo = rc_; }
      h->FD.redo_count = h->d_redo;
      h->fused_ok = true;
    }
#endif
  } else {
    UP(dL, d->dL, S); UP(m1det, d->m1det, S); UP(m2det, d->m2det, S);
    tmp.resize(E * S);
    for (size_t k = 0; k < E * S; k++) { tmp[k] = 1. / d->pe_prior[(size_t)e0 * S + k]; if (d->pe_prior[(size_t)e0 * S + k] < 0.) { h->neg_prior = true; L.neg_w = 1; } }          // the device keeps 1/pe_prior
    rc = upload(*h->owned_sp, (const double*)tmp.data(), E * S, &L.pe_prior, s); if (rc) { chm_like_destroy(h); return rc; }
    logs.assign(2, std::vector<double>(E * S));
    for (size_t k = 0; k < E * S; k++) { logs[0][k] = std::log(d->m1det[(size_t)e0 * S + k]); logs[1][k] = std::log(d->m2det[(size_t)e0 * S + k]); }
  }
  // k_samples_fast reads the six per-sample inputs from ONE array of tiles (E, NT, 6, 128): dL, m1det, m2det, 1/pe_prior,
  // log m1det, log m2det of 128 consecutive samples (the device order: pixel-sorted in marginalized mode); the last tile of an
  // event is padded with harmless values
  std::vector<double> tiles;
  {
    const size_t NT = (S + SF_TILE - 1) / SF_TILE;
    tiles.assign(E * NT * 6 * SF_TILE, 1.0);
    const bool mg = d->mode == CHM_MODE_MARG;
    for (size_t e = 0; e < E; e++) {
      const double* a[6];
      if (mg) { a[0] = sorted[0].data() + e * S; a[1] = sorted[1].data() + e * S; a[2] = sorted[2].data() + e * S; a[3] = sorted[3].data() + e * S; }
      else { a[0] = d->dL + (size_t)(e0 + e) * S; a[1] = d->m1det + (size_t)(e0 + e) * S; a[2] = d->m2det + (size_t)(e0 + e) * S; a[3] = tmp.data() + e * S; }
      a[4] = logs[0].data() + e * S; a[5] = logs[1].data() + e * S;
      for (size_t k = 0; k < S; k++) {
        double* o = tiles.data() + ((e * NT + k / SF_TILE) * 6) * SF_TILE + k % SF_TILE;
        for (int q = 0; q < 6; q++) o[q * SF_TILE] = a[q][k];
      }
      for (size_t k = S; k < NT * SF_TILE; k++) { double* o = tiles.data() + ((e * NT + k / SF_TILE) * 6) * SF_TILE + k % SF_TILE; o[4 * SF_TILE] = 0.; o[5 * SF_TILE] = 0.; }
    }
    rc = upload(*h->owned_sp, (const double*)tiles.data(), tiles.size(), &h->F.tiles, s); if (rc) { chm_like_destroy(h); return rc; }
    h->F.NT = (int)NT;
  }
  // smallest / largest finite distance of every event: the bracket of its table searches (k_samples)
  std::vector<double> dlo(E), dhi(E);
  for (size_t e = 0; e < E; e++) {
    const double* x = d->dL + (size_t)(e0 + e) * S;
    double lo = INFINITY, hi = -INFINITY;
    for (size_t k = 0; k < S; k++) if (std::isfinite(x[k])) { lo = x[k] < lo ? x[k] : lo; hi = x[k] > hi ? x[k] : hi; }
    if (!(lo <= hi)) { lo = NAN; hi = NAN; }                                     // no finite sample: the kernel searches the whole table
    dlo[e] = lo; dhi[e] = hi;
  }
  {                                                         // key range of the shard's positive finite distances (direct-index table)
    double gmin = INFINITY, gmax = 0.;
    for (size_t e = 0; e < E; e++) {
      const double* x = d->dL + (size_t)(e0 + e) * S;
      for (size_t k = 0; k < S; k++) if (std::isfinite(x[k]) && x[k] >= 2.2250738585072014e-308) { gmin = x[k] < gmin ? x[k] : gmin; gmax = x[k] > gmax ? x[k] : gmax; }
    }
    h->dl_gmin = gmin; h->dl_gmax = gmax;
    h->fast_ok = gmin <= gmax;
    if (h->fast_ok) {
      int64_t hi0, hi1; memcpy(&hi0, &gmin, 8); memcpy(&hi1, &gmax, 8);
      const int k0 = (int)(hi0 >> (32 + LUT_SHIFT)), k1 = (int)(hi1 >> (32 + LUT_SHIFT));
      h->F.lut.key0 = k0; h->F.lut.nk = k1 - k0 + 1;
      if (h->F.lut.nk > LUT_MAXKEYS) h->fast_ok = false;     // distances spanning > 64 octaves: the general kernel
    }
  }
  rc = upload(*h->owned_sp, (const double*)dlo.data(), E, &L.dl_lo, s); if (rc) { chm_like_destroy(h); return rc; }
  rc = upload(*h->owned_sp, (const double*)dhi.data(), E, &L.dl_hi, s); if (rc) { chm_like_destroy(h); return rc; }
  rc = upload(*h->owned_sp, (const double*)logs[0].data(), E * S, &L.lm1det, s); if (rc) { chm_like_destroy(h); return rc; }
  rc = upload(*h->owned_sp, (const double*)logs[1].data(), E * S, &L.lm2det, s); if (rc) { chm_like_destroy(h); return rc; }
  if (d->mode == CHM_MODE_FULL) { UP(ra, d->ra, S); UP(dec, d->dec, S); }
  UP(z_grids, d->z_grids, Z);
  { hipError_t e1 = hipMalloc(&h->d_zg_i, sizeof(int) * E * Z), e2 = hipMalloc(&h->d_zg_t, sizeof(double) * E * Z), e3 = hipMalloc(&h->d_zg_lz, sizeof(double) * E * Z);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { chm_like_destroy(h); return fail(CHM_E_NOMEM, "chm_like_create: grid bracket arrays"); } }
  // step fractions of jnp.linspace (i/div), shared by every event: bin edges (math.py:37) and effective grid (likelihood.py:188)
  std::vector<double> fracB(L.num_bins > 0 ? L.num_bins + 1 : 1), fracG(L.G > 0 ? L.G : 1);
  for (size_t i = 0; i < fracB.size(); i++) fracB[i] = (double)i / (double)(L.num_bins > 0 ? L.num_bins : 1);
  for (size_t i = 0; i < fracG.size(); i++) fracG[i] = (double)i / (double)(L.G > 1 ? L.G - 1 : 1);
  rc = upload(*h->owned_sp, (const double*)fracB.data(), fracB.size(), &L.fracB, s); if (rc) { chm_like_destroy(h); return rc; }
  rc = upload(*h->owned_sp, (const double*)fracG.data(), fracG.size(), &L.fracG, s); if (rc) { chm_like_destroy(h); return rc; }
  if (pixelated) {
    UP(p_cat, d->p_cat, P * Z); UP(P_compl, d->P_compl, Z); UP(gw_pdf, d->gw_loc2d_pdf, P);
    if (d->ra_pix) UP(ra_pix, d->ra_pix, P);
    if (d->dec_pix) UP(dec_pix, d->dec_pix, P);
    UP(neff_pixels, d->neff_pixels, 1);
  }
  // [r6] the k-range of an event (event_stats_from: verified at its two ends) and the early end of the standard GW kernel's grid loop assume
  // an ASCENDING event grid -- the reference's (pop_wrapper.py:207: linspace).  A caller-supplied grid with a descending step anywhere (NaNs aside) is
  // evaluated point by point over the whole grid by the general kernels, as the reference's arithmetic would
  for (size_t e = 0; e < E && !L.grid_unsorted; e++) {
    const double* zg = d->z_grids + ((size_t)e0 + e) * Z;
    double prev = -INFINITY;
    for (size_t k = 0; k < Z; k++) { const double z = zg[k]; if (z != z) continue; if (z < prev) { L.grid_unsorted = 1; break; } prev = z; }
  }
  {                                                          // NaNs in the per-event inputs of the integrand (see d_ev_bad)
    std::vector<unsigned char> bad(E, 0);
    bool any = false;
    for (size_t e = 0; e < E; e++) {
      const size_t ge = (size_t)e0 + e;
      bool b = false;
      const double* zg = d->z_grids + ge * Z;
      for (size_t k = 0; k < Z && !b; k++) b = zg[k] != zg[k];
      if (pixelated && !b) {
        const size_t np_ = (size_t)std::min<long long>(std::max<long long>(d->neff_pixels[ge], 0), (long long)P);
        const double* pc = d->p_cat + ge * P * Z;
        for (size_t q = 0; q < np_ * Z && !b; q++) b = pc[q] != pc[q];
        const double* cm = d->P_compl + ge * Z;
        for (size_t k = 0; k < Z && !b; k++) b = cm[k] != cm[k];
        const double* gp = d->gw_loc2d_pdf + ge * P;
        for (size_t q = 0; q < np_ && !b; q++) b = gp[q] != gp[q];
      }
      bad[e] = b ? 1 : 0; any = any || b;
    }
    if (any) { rc = upload(*h->owned_sp, (const unsigned char*)bad.data(), E, &h->d_ev_bad, s); if (rc) { chm_like_destroy(h); return rc; } HIPCHK(hipStreamSynchronize(s)); }
  }
#undef UP
  // the logs of the detector-frame masses as the device forms them (k_fill_logs): over the host's std::log values in both copies
  hipLaunchKernelGGL(k_fill_logs, dim3(1024), dim3(256), 0, s, L.m1det, L.m2det, const_cast<double*>(L.lm1det), const_cast<double*>(L.lm2det),
                     const_cast<double*>(h->F.tiles), (int)E, (int)S, h->F.NT);
  hipError_t he = hipGetLastError();
  if (he == hipSuccess) he = hipStreamSynchronize(s);
  if (he != hipSuccess) { chm_like_destroy(h); return fail(CHM_E_HIP, std::string("chm_like_create: ") + hipGetErrorString(he)); }
  *out = h;
  return CHM_OK;
}

static void like_free_ws(chm_like* h) {
  LikeDev& L = h->L;
  (void)hipFree(L.ws_z); (void)hipFree(L.ws_w); (void)hipFree(L.part); (void)hipFree(L.jac); (void)hipFree(L.prate); (void)hipFree(L.bkgA);
  (void)hipFree(h->d_lut); (void)hipFree(h->d_lutinfo); h->d_lut = nullptr; h->d_lutinfo = nullptr;
  (void)hipFree(L.ev_rbad); L.ev_rbad = nullptr;
  (void)hipFree(L.err_pix); L.err_pix = nullptr; (void)hipFree(L.ev_li); (void)hipFree(L.ev_ll); L.ev_li = L.ev_ll = nullptr; (void)hipFree(L.full_todo); L.full_todo = nullptr; (void)hipFree(L.full_ev); (void)hipFree(L.full_s); L.full_ev = L.full_s = nullptr;
  (void)hipFree(L.pgw1d); (void)hipFree(L.like_pix); (void)hipFree(L.p_gw_dump); (void)hipFree(L.Aw); (void)hipFree(L.evstat); (void)hipFree(L.effg); (void)hipFree(L.krange); L.krange = nullptr;
  L.ws_z = L.ws_w = L.part = L.jac = L.prate = L.bkgA = L.pgw1d = L.like_pix = L.p_gw_dump = L.Aw = L.evstat = L.effg = nullptr;
  h->nb_ws = 0; h->ws_dump = false;
}

extern "C" int chm_like_destroy(chm_like* h) {
  if (!h) return CHM_OK;
  (void)hipSetDevice(h->ctx.device);
  if (h->ctx.stream) (void)hipStreamSynchronize(h->ctx.stream);
  if (h->ctx.stream2) (void)hipStreamSynchronize(h->ctx.stream2);
  if (h->ctx.stream3) (void)hipStreamSynchronize(h->ctx.stream3);
  like_free_ws(h);
  (void)hipFree(h->d_zg_i); (void)hipFree(h->d_zg_t); (void)hipFree(h->d_zg_lz);
  ctx_destroy(h->ctx);
  delete h;                                                 // the uploaded arrays go with the last handle that shares them
  return CHM_OK;
}

// A second evaluation lane on the SAME resident data: the clone shares every array chm_like_create
// uploaded and owns its streams, per-draw tables, workspaces, grid brackets and graph.  Two host threads, each calling chm_eval on its
// own lane, keep two evaluations in flight on one GPU: the tables, launch path and reduction tail of one call run under the kernels of
// the other (per-call fixed costs are ~0.15 ms of a 125-event shard's 1.6 ms step).  A lane is as thread-unsafe as any handle.
extern "C" int chm_like_clone(const chm_like* src, chm_like** out) {
  if (!src || !out) return fail(CHM_E_ARG, "chm_like_clone: null argument");
  *out = nullptr;
  chm_like* h = new chm_like();
  h->opts = src->opts;
  int rc = ctx_init(h->ctx, src->ctx.device);
  if (rc) { delete h; return rc; }
  h->owned_sp = src->owned_sp;
  h->L = src->L;
  LikeDev& L = h->L;                                        // workspaces are the clone's own (allocated by the first call)
  L.ws_z = L.ws_w = L.part = L.jac = L.prate = L.bkgA = L.Aw = L.evstat = L.effg = L.pgw1d = L.like_pix = L.err_pix = L.p_gw_dump = L.ev_li = L.ev_ll = nullptr; L.full_todo = nullptr; L.full_ev = L.full_s = nullptr;
  L.krange = nullptr; L.tab_pm = L.tab_rate = L.tab_bkg = L.tab_jac = nullptr; L.zg_i = nullptr; L.zg_t = L.zg_lz = nullptr; L.ev_rbad = nullptr;
  h->F = src->F; h->fast_ok = src->fast_ok; h->dl_gmin = src->dl_gmin; h->dl_gmax = src->dl_gmax;
  h->neg_prior = src->neg_prior; h->d_ev_bad = src->d_ev_bad;
#ifdef CHM_WITH_FUSED
  h->FD = src->FD;
#endif
  h->fused_ok = src->fused_ok; h->ev_oct_max = src->ev_oct_max; h->ev_nk_max = src->ev_nk_max; h->d_redo = src->d_redo;
  const size_t EZ = (size_t)L.E * L.Z;
  hipError_t e1 = hipMalloc(&h->d_zg_i, sizeof(int) * EZ), e2 = hipMalloc(&h->d_zg_t, sizeof(double) * EZ), e3 = hipMalloc(&h->d_zg_lz, sizeof(double) * EZ);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { chm_like_destroy(h); return fail(CHM_E_NOMEM, "chm_like_clone: grid bracket arrays"); }
  *out = h;
  return CHM_OK;
}

static int like_ensure_ws(chm_like* h, int nb, bool dump) {
  if (nb <= h->nb_ws && (!dump || h->ws_dump)) return CHM_OK;
  HIPCHK(hipStreamSynchronize(h->ctx.stream));
  like_free_ws(h);
  LikeDev& L = h->L;
  size_t E = L.E, S = L.S, Z = L.Z, Pd = L.P > 0 ? L.P : 1, n = nb;
  HIPCHK(hipMalloc(&L.ws_z, sizeof(double) * (n * E * S + CHM_WS_PAD)));      // (padded: the standard GW kernel reads whole register rounds past a pixel's segment)
  HIPCHK(hipMalloc(&L.ws_w, sizeof(double) * (n * E * S + CHM_WS_PAD)));
  HIPCHK(hipMemsetAsync(L.ws_z + n * E * S, 0, sizeof(double) * CHM_WS_PAD, h->ctx.stream));
  HIPCHK(hipMemsetAsync(L.ws_w + n * E * S, 0, sizeof(double) * CHM_WS_PAD, h->ctx.stream));
  HIPCHK(hipMalloc(&L.part, sizeof(double) * n * E * L.NC * NPART));
  HIPCHK(hipMalloc(&L.jac, sizeof(double) * n * E * Z));
  HIPCHK(hipMalloc(&L.prate, sizeof(double) * n * E * Z));
  HIPCHK(hipMalloc(&L.bkgA, sizeof(double) * n * E * Z));
  if (L.mode == CHM_MODE_1D || L.mode == CHM_MODE_APPROX) {
    HIPCHK(hipMalloc(&L.pgw1d, sizeof(double) * n * E * Z));
    HIPCHK(hipMalloc(&L.Aw, sizeof(double) * n * E * Z));
    HIPCHK(hipMalloc(&L.krange, sizeof(int) * n * E * 2));
  }
  if (L.mode == CHM_MODE_MARG) { HIPCHK(hipMalloc(&L.err_pix, sizeof(double) * n * E * Pd)); HIPCHK(hipMalloc(&L.ev_li, sizeof(double) * n * E)); HIPCHK(hipMalloc(&L.ev_ll, sizeof(double) * n * E)); HIPCHK(hipMalloc(&L.Aw, sizeof(double) * n * E * Z)); HIPCHK(hipMalloc(&L.evstat, sizeof(double) * n * E * NEVSTAT));
                                 HIPCHK(hipMalloc(&L.effg, sizeof(double) * n * E * L.G)); }
  HIPCHK(hipMalloc(&L.like_pix, sizeof(double) * n * E * Pd));
  HIPCHK(hipMalloc(&L.ev_rbad, n * E)); HIPCHK(hipMemsetAsync(L.ev_rbad, 0, n * E, h->ctx.stream));
  if (L.mode == CHM_MODE_FULL) {
    HIPCHK(hipMalloc(&L.full_todo, sizeof(int) * n * E * Pd));
    HIPCHK(hipMalloc(&L.full_ev, sizeof(double) * n * E * FULLEV));
    HIPCHK(hipMalloc(&L.full_s, sizeof(double) * 5 * n * E * S));
    L.nb_alloc = (int)n;
  }
  if (h->fast_ok) { HIPCHK(hipMalloc(&h->d_lut, sizeof(unsigned short) * n * (h->F.lut.nk + 1))); HIPCHK(hipMalloc(&h->d_lutinfo, sizeof(int) * n * 4)); }
  if (dump && L.mode != CHM_MODE_1D) HIPCHK(hipMalloc(&L.p_gw_dump, sizeof(double) * n * E * Pd * Z));
  h->nb_ws = nb; h->ws_dump = dump;
  return CHM_OK;
}

#ifdef CHM_DIAG
// CHM_OPT_DIAG_POISON: the per-call workspaces named by `mask` are filled with the byte 0x3F (doubles of ~4.8e-4: finite, so that the value classes of a
// result survive) before an evaluation.  Every kernel is supposed to write what a later kernel reads within the same evaluation: a result that
// changes under a poisoned buffer names a read of stale memory (found that way in round 4: see DESIGN section 11).  Buffers of indices stay untouched.
static int like_poison_ws(chm_like* h, long long mask, int nb, hipStream_t s) {
  LikeDev& L = h->L;
  const size_t E = L.E, S = L.S, Z = L.Z, Pd = L.P > 0 ? L.P : 1, n = (size_t)h->nb_ws;
  (void)nb;
  struct { double* p; size_t bytes; } B[] = {
    { L.ws_z, 8 * n * E * S }, { L.ws_w, 8 * n * E * S }, { L.part, 8 * n * E * L.NC * NPART }, { L.jac, 8 * n * E * Z }, { L.prate, 8 * n * E * Z },
    { L.bkgA, 8 * n * E * Z }, { L.Aw, 8 * n * E * Z }, { L.err_pix, 8 * n * E * Pd }, { L.ev_li, 8 * n * E }, { L.ev_ll, 8 * n * E },
    { L.evstat, 8 * n * E * NEVSTAT }, { L.effg, 8 * n * E * (size_t)L.G }, { L.like_pix, 8 * n * E * Pd }, { L.pgw1d, 8 * n * E * Z },
    { L.full_ev, 8 * n * E * FULLEV }, { L.full_s, 8 * 5 * n * E * S } };
  for (size_t i = 0; i < sizeof(B) / sizeof(B[0]); i++)
    if (((mask >> i) & 1) && B[i].p) HIPCHK(hipMemsetAsync(B[i].p, 0x3F, B[i].bytes, s));
  HIPCHK(hipStreamSynchronize(s));
  return CHM_OK;
}
#endif

extern "C" int chm_sel_create(const chm_sel_desc* d, chm_sel** out) {
  if (!d || !out) return fail(CHM_E_ARG, "chm_sel_create: null argument");
  *out = nullptr;
  if (d->I <= 0 || !d->dL || !d->m1det || !d->m2det || !d->p_draw) return fail(CHM_E_ARG, "chm_sel_create: need I > 0 and dL, m1det, m2det, p_draw");
  if (!(d->N_inj > 0)) return fail(CHM_E_ARG, "chm_sel_create: N_inj must be > 0");
  long long i0 = d->inj_begin, i1 = d->inj_end;
  if (i0 == 0 && i1 == 0) i1 = d->I;
  if (i0 < 0 || i1 > d->I || i0 > i1) return fail(CHM_E_ARG, "chm_sel_create: bad injection range");
  int ndev = chm_device_count();
  if (d->device < 0 || d->device >= ndev) return fail(CHM_E_HIP, "chm_sel_create: no such HIP device (is a GPU visible?)");
  chm_sel* h = new chm_sel();
  opts_init(h->opts);
  int rc = ctx_init(h->ctx, d->device);
  if (rc) { delete h; return rc; }
  h->owned_sp->device = d->device;
  SelDev& S = h->S;
  memset(&S, 0, sizeof(S));
  size_t n = (size_t)(i1 - i0);
  S.I = (long long)n; S.N_inj = d->N_inj; S.has_neff = std::isnan(d->N_eff) ? 0 : 1; S.N_eff = d->N_eff;
  hipStream_t s = h->ctx.stream;
#define UP(field, src) do { rc = upload(*h->owned_sp, (src) + i0, n, &S.field, s); if (rc) { chm_sel_destroy(h); return rc; } } while (0)
  UP(dL, d->dL); UP(m1det, d->m1det); UP(m2det, d->m2det);
#undef UP
  std::vector<double> ipd(n);
  for (size_t k = 0; k < n; k++) ipd[k] = 1. / d->p_draw[i0 + k];                                 // the device keeps 1/p_draw
  rc = upload(*h->owned_sp, (const double*)ipd.data(), n, &S.p_draw, s); if (rc) { chm_sel_destroy(h); return rc; }
  std::vector<double> l1(n), l2(n);                                                               // log(m_det), once (alive until the sync below)
  for (size_t k = 0; k < n; k++) { l1[k] = log(d->m1det[i0 + k]); l2[k] = log(d->m2det[i0 + k]); }
  rc = upload(*h->owned_sp, (const double*)l1.data(), n, &S.lm1det, s); if (rc) { chm_sel_destroy(h); return rc; }
  rc = upload(*h->owned_sp, (const double*)l2.data(), n, &S.lm2det, s); if (rc) { chm_sel_destroy(h); return rc; }
  long long nblk = (S.I + SEL_TILE / 2 - 1) / (SEL_TILE / 2);       // records of block partial sums per draw (tiles of 512 injections at the finest)
  S.nblocks = (int)(nblk < 1 ? 1 : (nblk > 2048 ? 2048 : nblk));
  S.tile = SEL_TILE / 2;                                    // k_selection_fast: one pass of 256 threads x 2 injections per tile (the same grouping of the sums for every call size)
  {                                                         // key range of the shard's positive finite distances (direct-index table)
    double gmin = INFINITY, gmax = 0.;
    for (size_t k = 0; k < n; k++) { const double x = d->dL[i0 + k]; if (std::isfinite(x) && x >= 2.2250738585072014e-308) { gmin = x < gmin ? x : gmin; gmax = x > gmax ? x : gmax; } }
    h->dl_gmin = gmin; h->dl_gmax = gmax;
    h->fast_ok = gmin <= gmax;
    if (h->fast_ok) {
      int64_t hi0, hi1; memcpy(&hi0, &gmin, 8); memcpy(&hi1, &gmax, 8);
      const int k0 = (int)(hi0 >> (32 + LUT_SHIFT)), k1 = (int)(hi1 >> (32 + LUT_SHIFT));
      h->lut.key0 = k0; h->lut.nk = k1 - k0 + 1;
      if (h->lut.nk > LUT_MAXKEYS) h->fast_ok = false;
    }
  }
  hipError_t he = hipStreamSynchronize(s);
  if (he != hipSuccess) { chm_sel_destroy(h); return fail(CHM_E_HIP, std::string("chm_sel_create: ") + hipGetErrorString(he)); }
  *out = h;
  return CHM_OK;
}

extern "C" int chm_sel_destroy(chm_sel* h) {
  if (!h) return CHM_OK;
  (void)hipSetDevice(h->ctx.device);
  if (h->ctx.stream) (void)hipStreamSynchronize(h->ctx.stream);
  if (h->ctx.stream2) (void)hipStreamSynchronize(h->ctx.stream2);
  if (h->ctx.stream3) (void)hipStreamSynchronize(h->ctx.stream3);
  (void)hipFree(h->S.partial); (void)hipFree(h->d_lut); (void)hipFree(h->d_lutinfo);
  ctx_destroy(h->ctx);
  delete h;
  return CHM_OK;
}

// second evaluation lane on the same resident injections (see chm_like_clone)
extern "C" int chm_sel_clone(const chm_sel* src, chm_sel** out) {
  if (!src || !out) return fail(CHM_E_ARG, "chm_sel_clone: null argument");
  *out = nullptr;
  chm_sel* h = new chm_sel();
  h->opts = src->opts;
  int rc = ctx_init(h->ctx, src->ctx.device);
  if (rc) { delete h; return rc; }
  h->owned_sp = src->owned_sp;
  h->S = src->S;
  h->S.partial = nullptr; h->S.tab_pm = h->S.tab_rate = h->S.tab_bkg = h->S.tab_jac = nullptr;
  h->lut = src->lut; h->fast_ok = src->fast_ok; h->dl_gmin = src->dl_gmin; h->dl_gmax = src->dl_gmax;
  *out = h;
  return CHM_OK;
}

extern "C" int chm_like_set_option(chm_like* h, int32_t option, int64_t value) {
  if (!h) return fail(CHM_E_ARG, "chm_like_set_option: null handle");
  return opts_set(h->opts, option, value);
}
extern "C" int chm_sel_set_option(chm_sel* h, int32_t option, int64_t value) {
  if (!h) return fail(CHM_E_ARG, "chm_sel_set_option: null handle");
  return opts_set(h->opts, option, value);
}

static int sel_ensure_ws(chm_sel* h, int nb) {
  if (nb <= h->nb_ws) return CHM_OK;
  HIPCHK(hipStreamSynchronize(h->ctx.stream));
  (void)hipFree(h->S.partial); h->S.partial = nullptr;
  HIPCHK(hipMalloc(&h->S.partial, sizeof(double) * (size_t)nb * h->S.nblocks * 2));
  (void)hipFree(h->d_lut); (void)hipFree(h->d_lutinfo); h->d_lut = nullptr; h->d_lutinfo = nullptr;
  if (h->fast_ok) { HIPCHK(hipMalloc(&h->d_lut, sizeof(unsigned short) * (size_t)nb * (h->lut.nk + 1))); HIPCHK(hipMalloc(&h->d_lutinfo, sizeof(int) * (size_t)nb * 4)); }
  h->nb_ws = nb;
  return CHM_OK;
}

// ------------------------------------------------------------------------------------------------------
// evaluation
// ------------------------------------------------------------------------------------------------------
template <class K>
static void allow_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// device copies of the caller's tables for one call (plug-in models, chm_tab); freed when the call returns
struct TabDev {
  double *pm_s = nullptr, *pm_i = nullptr, *rate_g = nullptr, *rate_i = nullptr, *bkg_g = nullptr, *bkg_i = nullptr;
  double *zt = nullptr, *dLt = nullptr, *jac_g = nullptr, *jac_i = nullptr;      // plug-in cosmology
  ~TabDev() { (void)hipFree(pm_s); (void)hipFree(pm_i); (void)hipFree(rate_g); (void)hipFree(rate_i); (void)hipFree(bkg_g); (void)hipFree(bkg_i);
              (void)hipFree(zt); (void)hipFree(dLt); (void)hipFree(jac_g); (void)hipFree(jac_i); }
};
static int tab_upload(double** dst, const double* src, size_t n, hipStream_t s) {
  if (!src || n == 0) return CHM_OK;
  HIPCHK(hipMalloc(dst, sizeof(double) * n));
  HIPCHK(hipMemcpyAsync(*dst, src, sizeof(double) * n, hipMemcpyHostToDevice, s));
  return CHM_OK;
}

static int eval_impl(chm_like* like, chm_sel* sel, chm_comm* comm, const chm_params* params, int32_t nb,
                     int64_t E_total, const chm_tab* tab, chm_out* out);

// CHM_HOST_PROF=1 (diagnostics): host-side time of the graph-replayed scalar call, printed at exit -- before the launch (argument checks,
// key, parameter packing), inside hipGraphLaunch, inside hipStreamSynchronize
#include <chrono>
#include <algorithm>
// Completion of a call: few-draw calls (the reference-shaped scalar call waits ~0.15 ms per evaluation) poll their own stream -- the default
// wait sleeps on an interrupt whose wake-up is part of the call's latency.  Per call and per handle: no process-wide device flag.
static hipError_t wait_stream(hipStream_t s, bool spin) {
  if (spin) { hipError_t e; while ((e = hipStreamQuery(s)) == hipErrorNotReady) {} return e; }
  return hipStreamSynchronize(s);
}
// [r5] Completion of a zero-copy call through the flags the last kernel stores behind the results (completion_flag, chm_kernels.h): the host spins on
// pinned memory; every 4096 looks it asks the stream, so that a kernel that faulted (the flags never come) surfaces as an error instead of a hang.
// [r6] ... and the RESULTS prove their own arrival: the host fills the result block with CHM_PENDING (a NaN bit pattern no result can have:
// combine_one stores canonical NaNs) before the launch and, once the flags are there, waits until none of the 3 nb doubles holds it any more.  A
// fence on the device orders the kernel's stores as it issues them; it does not order their ARRIVAL in host memory -- results and flags live in
// different allocations and travel as independent posted writes.  Found by the round's fuzz campaign with the flags on every call: 2 of 6648
// configurations (~20 000 flagged calls) read a result block the flag had overtaken (log_hyper = 0.0 of freshly allocated pages; not reproducible in
// isolation: profiles/r06/fuzz_r06.txt).
static const unsigned long long CHM_PENDING = 0x7ff8dead5eedbeefULL;
static void mark_pending(double* h_out, int nb) {
  volatile unsigned long long* r = reinterpret_cast<volatile unsigned long long*>(h_out);
  for (int i = 0; i < 3 * nb; i++) r[i] = CHM_PENDING;
}
static hipError_t wait_flags(const long long* h_seq, const double* h_out, int nb, long long seq, hipStream_t s) {
  const volatile long long* f = h_seq + 1;
  const volatile unsigned long long* r = reinterpret_cast<const volatile unsigned long long*>(h_out);
  auto arrived = [&]() {
    for (int b = 0; b < nb; b++) if (f[b] != seq) return false;
    for (int i = 0; i < 3 * nb; i++) if (r[i] == CHM_PENDING) return false;
    return true;
  };
  for (unsigned it = 1;; it++) {
    if (arrived()) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return hipSuccess; }
    if ((it & 0xFFFu) == 0u) {
      hipError_t e = hipStreamQuery(s);
      if (e == hipErrorNotReady) continue;
      if (e != hipSuccess) return e;
      // the stream drained: what it wrote is on its way at the latest now -- a bounded grace, then an error (never a hang)
      for (unsigned g = 0; g < 2000000u; g++) if (arrived()) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return hipSuccess; }
      return hipErrorUnknown;
    }
  }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static thread_local bool g_host_prof = false;               // (CHM_OPT_DIAG_HOST_PROF of the handle being evaluated)
static bool host_prof_on() { return g_host_prof; }
static struct HostProf { std::vector<double> pre, launch, sync;
  static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
  ~HostProf() { if (!pre.empty()) fprintf(stderr, "[chm host prof] %zu replayed calls: before launch %.1f us, hipGraphLaunch %.1f us, hipStreamSynchronize %.1f us (medians)\n",
                                          pre.size(), med(pre), med(launch), med(sync)); } } g_hp;

extern "C" int chm_eval(chm_like* like, chm_sel* sel, chm_comm* comm, const chm_params* params, int32_t nb,
                        int64_t E_total, chm_out* out) {
  return eval_impl(like, sel, comm, params, nb, E_total, nullptr, out);
}

extern "C" int chm_eval_tabulated(chm_like* like, chm_sel* sel, chm_comm* comm, const chm_params* params, int32_t nb,
                                  int64_t E_total, const chm_tab* tab, chm_out* out) {
  return eval_impl(like, sel, comm, params, nb, E_total, tab, out);
}

static int eval_impl(chm_like* like, chm_sel* sel, chm_comm* comm, const chm_params* params, int32_t nb,
                     int64_t E_total, const chm_tab* tab, chm_out* out) {
  TicketTurn turn(comm);                                      // (ticketed calls: the collective is enqueued in ticket order, see CollSeq; every exit passes the turn on)
  if ((!like && !sel) || !params || !out || nb <= 0) return fail(CHM_E_ARG, "chm_eval: need a handle, params, out and nb > 0");
  if (like && sel && like->ctx.device != sel->ctx.device) return fail(CHM_E_ARG, "chm_eval: like and sel live on different devices");
  Ctx& c = like ? like->ctx : sel->ctx;
  const Opts& o = like ? like->opts : sel->opts;             // the options of the call: the event handle's when there is one
  g_host_prof = o.host_prof != 0;
  const double hp0 = host_prof_on() ? now_us() : 0.;
  if (comm && comm->device != c.device) return fail(CHM_E_ARG, "chm_eval: comm lives on a different device");
  HIPCHK(hipSetDevice(c.device));
  const bool serial = o.serial != 0;                        // everything on one stream (bench.py times the kernels on their own after its timed region)
  hipStream_t sA = c.stream, sB = serial ? c.stream : c.stream2, sC = serial ? c.stream : c.stream3;      // (all three = sA for fused few-draw calls, below)
  const bool want_dump = like && out->p_gw != nullptr;
  int rc;
  if (like) { rc = like_ensure_ws(like, nb, want_dump); if (rc) return rc; }
  if (sel) { rc = sel_ensure_ws(sel, nb); if (rc) return rc; }
#ifdef CHM_DIAG
  if (like && o.poison) { rc = like_poison_ws(like, o.poison, nb, sA); if (rc) return rc; }
#endif

  TabDev td;
  if (tab) {                                                  // plug-in models: the caller's tables of this call, host -> device
    if ((tab->z_table != nullptr) != (tab->dL_table != nullptr)) return fail(CHM_E_ARG, "chm_eval_tabulated: z_table and dL_table go together");
    if (tab->z_table) {                                       // plug-in cosmology: everything cosmological must come from the caller
      for (int b = 1; b < nb; b++) if (params[b].z_grid_res != params[0].z_grid_res) return fail(CHM_E_ARG, "chm_eval_tabulated: a plug-in cosmology needs the same z_grid_res for every draw");
      if (like && (!tab->jac_grid || !tab->bkg_grid)) return fail(CHM_E_ARG, "chm_eval_tabulated: a plug-in cosmology needs jac_grid and bkg_grid");
      if (like && params[0].has_catalog && !tab->fR) return fail(CHM_E_ARG, "chm_eval_tabulated: a plug-in cosmology with a catalogue needs fR");
      if (sel && (!tab->jac_inj || !tab->bkg_inj)) return fail(CHM_E_ARG, "chm_eval_tabulated: a plug-in cosmology needs jac_inj and bkg_inj");
      const size_t NT = (size_t)nb * params[0].z_grid_res;
      rc = tab_upload(&td.zt, tab->z_table, NT, sA); if (rc) return rc;
      rc = tab_upload(&td.dLt, tab->dL_table, NT, sA); if (rc) return rc;
    }
    if (like) {
      const size_t ES = (size_t)nb * like->L.E * like->L.S, EZ = (size_t)nb * like->L.E * like->L.Z;
      rc = tab_upload(&td.pm_s, tab->pm_samples, ES, sA); if (rc) return rc;
      rc = tab_upload(&td.rate_g, tab->rate_grid, EZ, sA); if (rc) return rc;
      rc = tab_upload(&td.bkg_g, tab->bkg_grid, EZ, sA); if (rc) return rc;
      rc = tab_upload(&td.jac_g, tab->jac_grid, EZ, sA); if (rc) return rc;
    }
    if (sel) {
      const size_t NI = (size_t)nb * (size_t)sel->S.I;
      rc = tab_upload(&td.pm_i, tab->pm_inj, NI, sA); if (rc) return rc;
      rc = tab_upload(&td.rate_i, tab->rate_inj, NI, sA); if (rc) return rc;
      rc = tab_upload(&td.bkg_i, tab->bkg_inj, NI, sA); if (rc) return rc;
      rc = tab_upload(&td.jac_i, tab->jac_inj, NI, sA); if (rc) return rc;
    }
  }
  const bool timing_env = o.timing > 0;                       // CHM_OPT_TIMING 0: no timing events in the streams (chm_last_timing returns zeros)
  // Few draws per call (the reference's scalar call): the launch sequence is replayed from a HIP graph -- no timing events, no
  // per-event outputs, no caller tables.  A configuration runs eagerly the first time it is seen (function attributes, workspaces), is
  // captured the second time and replayed afterwards.  [r3] With a communicator the graph ends at the rank's partial sums; the RCCL
  // all-reduce and k_combine follow it on the same stream (the 8-GPU job keeps the replayed path of the scalar call).
  // [r5] a draw with an infinite rate parameter: the whole call takes the kernels that multiply every grid point out (the reference's 0 * inf = NaN
  // where p_gw vanishes must come out of the arithmetic) and the general selection kernel (the reference's order of quotients)
  bool rate_special_call = false;
  for (int b = 0; b < nb; b++) rate_special_call = rate_special_call || params_rate_special(&params[b]);
  const bool opt_zf_full = o.zf_full || rate_special_call, opt_marg_generic = o.marg_generic || rate_special_call;
  const int graph_max_nb = o.graph_max_nb;
  const bool zc_env = !o.no_zero_copy;
  // k_tables reads the draws from pinned host memory itself (no copy node in front of it): [r6] calls of every size -- each block fetches its draw's 400 bytes;
  // the H2D copy of a 128-draw block in front of the first kernel was 15 us of the 125-event shard's step (same-box A/B, four repetitions: 1.1800 -> 1.1630 ms,
  // profiles/r06/ab_shard_step_r06.txt; until round 5 only calls of <= 8 draws took this path)
  const bool zero_copy = zc_env;
  // [r6] results of EVERY call size are written to pinned host memory by the last kernel (the trailing D2H copy of 3 nb doubles was ~4 us of copy kernel
  // + its launch behind every batched call: profiles/r05/timeline_shard125_batched.txt)
  const bool zc_out = zc_env;
  const bool graph_ok = nb <= graph_max_nb && !tab && !want_dump && !out->log_like_evs && !out->numlike_evs;
  int Tc_host = 0, Tm_host = 0;
  rc = ctx_tables_host(c, params, nb, tab ? tab->fR : nullptr, &Tc_host, &Tm_host); if (rc) return rc;
  if (like && (like->L.E + 255) / 256 * nb > c.evpart_cap) {       // block sums of log L_i for shards beyond 4096 events (k_reduce_events)
    HIPCHK(hipStreamSynchronize(sA));
    (void)hipFree(c.d_evpart); c.d_evpart = nullptr;
    HIPCHK(hipMalloc(&c.d_evpart, sizeof(double) * nb * ((like->L.E + 255) / 256)));
    c.evpart_cap = nb * ((like->L.E + 255) / 256);
  }
  // k_samples_fast (built-in mass model, the same for every draw of the call, tables of at most 65535 entries): the direct-index
  // table of the dL table is built by k_tables for the key range of this shard's distances; LDS capacity of the table slice from the
  // range (entries per octave of the reference's logspace z grid, cosmo.py:43-46, with a margin; a draw whose slice does not fit
  // takes the general searches inside the same kernel)
  LutDesc lutA = {};
  bool use_fast = like && like->fast_ok && !td.pm_s && !o.samples_generic;
  size_t lds_fast = 0;
  if (use_fast) {
    int Tc_call = 0, Tm_call = 0;
    double zmax_min = INFINITY;
    for (int b = 0; b < nb; b++) {
      if (params[b].mass_model != params[0].mass_model) use_fast = false;
      Tc_call = params[b].z_grid_res > Tc_call ? params[b].z_grid_res : Tc_call;
      Tm_call = params[b].mass_grid_res > Tm_call ? params[b].mass_grid_res : Tm_call;
      zmax_min = params[b].z_max < zmax_min ? params[b].z_max : zmax_min;
    }
    if (Tc_call > 65535 || !(zmax_min > 0.)) use_fast = false;
    if (use_fast) {
      const double n_oct = std::log2(like->dl_gmax / like->dl_gmin) + 1.;
      const double per_oct = (double)(Tc_call - 2) / (std::log2(zmax_min) + 33.22);
      long long cap = (long long)(n_oct * per_oct * 1.3) + 64;
      cap = cap > Tc_call ? Tc_call : cap;
      cap = (cap + 7) / 8 * 8;
      lutA = like->F.lut;
      lutA.cap = (int)cap; lutA.lut = like->d_lut; lutA.info = like->d_lutinfo;
      lds_fast = sizeof(double) * (CHM_EXPTAB_N + 4 * (size_t)cap + 2 * (size_t)Tm_call) + ((size_t)(lutA.nk + 1) * 2 + 15) / 16 * 16;
      if (lds_fast > 96 * 1024) use_fast = false;
    }
  }
#ifdef CHM_PROBE
  if (like) { like->probe_lut = use_fast ? lutA : LutDesc{}; like->probe_lds_fast = use_fast ? lds_fast : 0; }
#endif
  // k_marg_fused (chm_fused.h): the standard marginalized configuration in ONE kernel per (event, draw) -- few-draw calls by default
  // (CHM_OPT_FUSED 1: few-draw calls, 2: calls of any size; default off).  LDS per block: P histograms of num_bins + 1 doubles, the overlay region
  // (the draw's mass tables + the widest event's slice of the distance tables + NW - 1 boundary rows | 4 NW prefix arrays) and ~2 KB.
#ifdef CHM_WITH_FUSED
  FusedDesc FDc = {};
#else
  struct { int cap_rec = 0, cap_keys = 0, cap_m = 0; } FDc;      // (keeps the graph key's layout)
#endif
  size_t lds_fused = 0;
  // 4 waves per block, two blocks per CU.  (16 waves per block -- one block per CU, 154 KB of LDS with all 32 pixels' prefix arrays at once, 128
  // VGPRs -- was measured for few-draw calls: 54 spilled registers, 0.294 against 0.219 ms per scalar call; profiles/r04/ab_fused_event_kernel.txt.
  // -DCHM_FUSED_NW16 + CHM_OPT_DIAG_FUSED_NW 16 bring it back for A/B runs.)
#ifdef CHM_FUSED_NW16
  const int fused_nw = (nb <= o.few_nb && o.fused_nw == 16) ? 16 : 4;
#else
  const int fused_nw = 4;
#endif
  bool use_fused = false;
#ifdef CHM_WITH_FUSED
  {
    const int fmode = o.fused;                              // off by default: measured slower than the separate kernels at every call size (profiles/r04/ab_fused_event_kernel.txt)
    const int few_nb_f = o.few_nb;
    if (like && like->fused_ok && use_fast && !tab && !want_dump && fmode > 0 && (nb <= few_nb_f || fmode >= 2) &&
        like->L.mode == CHM_MODE_MARG && like->L.binning && like->L.has_cut && (like->L.Z & 1) == 0 && like->L.num_bins == 200 &&
        !opt_marg_generic && !opt_zf_full && !like->neg_prior && !like->L.grid_unsorted) {
      int Tc_call = 0, Tm_call = 0;
      double zmax_min = INFINITY;
      for (int b = 0; b < nb; b++) {
        Tc_call = params[b].z_grid_res > Tc_call ? params[b].z_grid_res : Tc_call;
        Tm_call = params[b].mass_grid_res > Tm_call ? params[b].mass_grid_res : Tm_call;
        zmax_min = params[b].z_max < zmax_min ? params[b].z_max : zmax_min;
      }
      const double per_oct = (double)(Tc_call - 2) / (std::log2(zmax_min) + 33.22);
      long long cap = (long long)(like->ev_oct_max * per_oct * 1.3) + 16;
      cap = cap > Tc_call ? Tc_call : cap;
      cap = (cap + 1) / 2 * 2;
      const int capk = (like->ev_nk_max + 2 + 3) / 4 * 4;
      const long long HS = like->L.num_bins + 1, Hrows = like->L.P + (like->L.P & 1);
      // the window of the mass tables gets what the pixel pass's prefix arrays leave beside the other tables (at least 128 entries; all of a short table)
      const long long ov_fix = CHM_EXPTAB_N + 4 * cap + capk / 4 + (fused_nw - 1) * HS, ov_p = 4LL * fused_nw * HS;
      long long cap_m = (ov_p - ov_fix) / 2;
      cap_m = cap_m < 128 ? 128 : cap_m;
      cap_m = cap_m > Tm_call ? Tm_call : cap_m;
      cap_m = (cap_m + 1) / 2 * 2;
      const long long ov_s = ov_fix + 2 * cap_m;
      const long long ov = ov_s > ov_p ? ov_s : ov_p;
      FDc = like->FD;
      FDc.cap_rec = (int)cap; FDc.cap_keys = capk; FDc.cap_m = (int)cap_m; FDc.overlay_doubles = (int)ov; FDc.tol = 3e-10;
      lds_fused = sizeof(double) * (size_t)(Hrows * HS + 16 + 4 * fused_nw + 5 * Hrows + fused_nw + ov);
      const size_t Nk = like->L.num_bins;
      const size_t lds_redo = sizeof(double) * (2 * Nk + 3 * (Nk + 1) + 2 * (size_t)like->L.G);      // the dense redo runs the general kernel's body in the same LDS
      if (lds_redo > lds_fused) lds_fused = lds_redo;
      use_fused = lds_fused <= 160 * 1024;
    }
  }
#endif
  // k_selection_fast: built-in models of an FLRW draw (cosmo_model 0), the same mass model for every draw; table slice capacity as above
  LutDesc lutB = {};
  bool sel_fast = sel && sel->fast_ok && !td.pm_i && !td.rate_i && !td.bkg_i && !td.jac_i && !td.zt && !o.selection_generic && !rate_special_call;
  size_t lds_sel = 0;
  if (sel_fast) {
    int Tc_call = 0, Tm_call = 0;
    double zmax_min = INFINITY;
    for (int b = 0; b < nb; b++) {
      if (params[b].mass_model != params[0].mass_model || params[b].cosmo_model != params[0].cosmo_model) sel_fast = false;      // [r5] mg_flrw too (MG instantiation)
      Tc_call = params[b].z_grid_res > Tc_call ? params[b].z_grid_res : Tc_call;
      Tm_call = params[b].mass_grid_res > Tm_call ? params[b].mass_grid_res : Tm_call;
      zmax_min = params[b].z_max < zmax_min ? params[b].z_max : zmax_min;
    }
    if (Tc_call > 65535 || !(zmax_min > 0.)) sel_fast = false;
    if (sel_fast) {
      const double n_oct = std::log2(sel->dl_gmax / sel->dl_gmin) + 1.;
      const double per_oct = (double)(Tc_call - 2) / (std::log2(zmax_min) + 33.22);
      long long cap = (long long)(n_oct * per_oct * 1.3) + 64;
      cap = cap > Tc_call ? Tc_call : cap;
      cap = (cap + 7) / 8 * 8;
      lutB = sel->lut;
      lutB.cap = (int)cap; lutB.lut = sel->d_lut; lutB.info = sel->d_lutinfo;
      lds_sel = sizeof(double) * (CHM_EXPTAB_N + 4 * (size_t)cap + 2 * (size_t)Tm_call) + ((size_t)(lutB.nk + 1) * 2 + 15) / 16 * 16;
      if (lds_sel > 64 * 1024) sel_fast = false;
    }
  }
  // Few draws per call (the scalar call), standard marginalized configuration, fast selection kernel: the per-z factors, the event
  // statistics and the selection sums share ONE launch (k_zf_sel) and the whole call runs on one stream -- the captured graph is a plain
  // chain (hipGraphLaunch 10 instead of 39 us) and the selection kernel hides behind the per-z factors.
  const int few_nb = o.few_nb;
  const bool fuse_env = !o.no_zf_sel;
  const size_t lds_zfac_call = sizeof(double) * (size_t)2 * c.TcMax;
  const bool fuse_sel = fuse_env && !use_fused && !serial && like && sel && sel_fast && params[0].cosmo_model == 0 && nb <= few_nb && !td.rate_g && !td.bkg_g && !td.jac_g &&
                        like->L.mode == CHM_MODE_MARG && like->L.binning && like->L.has_cut && (like->L.Z & 1) == 0 && !opt_marg_generic && !opt_zf_full &&
                        !(o.groups > 1) && lds_zfac_call <= 64 * 1024 && like->L.E <= 65535;
  const bool one_stream = serial || fuse_sel;
  if (fuse_sel) { sB = sA; sC = sA; }
  // draw-independent brackets of the event grids on the z table: usable when every draw of the call has one (z_max, z_grid_res) and the
  // cosmology is built in; (re)made by k_grid_prep after k_tables when that pair changes
  bool zg_use = false, zg_make = false;
  if (like && !td.zt && !o.no_grid_prep) {
    zg_use = true;
    for (int b = 1; b < nb; b++) if (params[b].z_max != params[0].z_max || params[b].z_grid_res != params[0].z_grid_res) zg_use = false;
    if (zg_use && (like->zg_zmax != params[0].z_max || like->zg_Tc != params[0].z_grid_res)) zg_make = true;
  }
  // [r4] the nodes of z_grid_interp and their log(1 + z) depend on (z_max, z_grid_res) alone: formed once by k_znodes, read by k_tables while every
  // draw of a call has the pair they were made for (an H0 scan, a chain); re-made when the pair changes
  bool zc_use = !td.zt, zc_make = false;
  for (int b = 1; b < nb && zc_use; b++) if (params[b].z_max != params[0].z_max || params[b].z_grid_res != params[0].z_grid_res) zc_use = false;
  if (zc_use && !(params[0].z_max > 0.)) zc_use = false;
  if (zc_use && (c.zc_zmax != params[0].z_max || c.zc_Tc != params[0].z_grid_res)) zc_make = true;
  // ---- graph bookkeeping: the key lists everything the captured launch arguments depend on
  // completion through the flags in pinned memory (wait_flags): zero-copy results, spinning wait, nothing copied back behind the last kernel
  // [r6] calls of any size (round 5: few-draw calls only): a batched call of a small shard is ~1.3 ms, of which the interrupt-driven wake-up of
  // hipStreamSynchronize was 20-40 us (profiles/r06/ab_shard_step_r06.txt)
  const bool use_flags = zc_out && o.spin_wait != 0 && !out->partials && !out->log_like_evs && !out->numlike_evs && !want_dump && !tab;
  std::vector<long long> key;
  bool capturing = false;
  long long zmax_bits = 0; { const double zm = params[0].z_max; memcpy(&zmax_bits, &zm, 8); }
  if (graph_ok) {
    key = { (long long)(intptr_t)like, (long long)(intptr_t)sel, nb, (long long)E_total, like ? like->nb_ws : 0, sel ? sel->nb_ws : 0, c.nb_cap, c.TcMax, c.TmMax,
            Tc_host, Tm_host, use_fast, lutA.key0, lutA.nk, lutA.cap, (long long)lds_fast, params[0].mass_model, out->partials != nullptr,
            sel_fast, lutB.key0, lutB.nk, lutB.cap, (long long)lds_sel, (long long)(intptr_t)lutB.lut, fuse_sel,
            (long long)(intptr_t)c.d_evpart, (long long)(intptr_t)(like ? like->L.ws_z : nullptr), (long long)(intptr_t)(sel ? sel->S.partial : nullptr),
            zg_use, zg_make, (long long)(intptr_t)comm, use_fused, fused_nw, (long long)lds_fused, FDc.cap_rec, FDc.cap_keys, FDc.cap_m, zc_use, zc_make,
            // [r5] k_znodes takes z_max by value and k_tables reads the cached nodes: a graph captured for one z_max must never be replayed
            // for another (a z_max scan with scalar calls: A, B, A replayed B's nodes under A's parameters) -- the bit pattern of z_max is part of the key;
            // so are the options of both handles (serial, groups, diagnostics: a replay would ignore a set_option made after the capture)
            zmax_bits, like ? like->opts.epoch : -1, sel ? sel->opts.epoch : -1, params[0].cosmo_model, rate_special_call, use_flags, (long long)(intptr_t)c.h_seq };
    if (c.gexec && key == c.gkey) {                           // replay
      const double hp1 = host_prof_on() ? now_us() : 0.;
      if (use_flags) { mark_pending(c.h_out, nb); c.h_seq[0] = ++c.seq; __atomic_thread_fence(__ATOMIC_RELEASE); }
      HIPCHK(hipGraphLaunch(c.gexec, sA));
      if (comm) {                                             // the graph ends at the rank's partials: all-reduce + combination behind it
        if (!turn.acquire()) return fail(CHM_E_RCCL, "chm_eval: the calls with lower tickets never enqueued their collectives (chm_comm_set_ticket; timeout)");
        NCCLCHK(ncclAllReduce(c.d_partials, c.d_partials, (size_t)nb * 3, ncclDouble, ncclSum, comm->comm, sA));
        turn.release();
        hipLaunchKernelGGL(k_combine, dim3((nb + 63) / 64), dim3(64), 0, sA, nb, (const DevParams*)c.d_params, (const double*)c.d_partials,
                           comm ? (double)E_total : (like ? (double)like->L.E : 0.), sel ? sel->S.N_inj : 1., sel ? sel->S.N_eff : 0.,
                           sel ? sel->S.has_neff : 0, like ? 1 : 0, sel ? 1 : 0, zc_out ? c.h_out : c.d_out3,
                           (const long long*)c.h_seq, use_flags ? c.h_seq + 1 : (long long*)nullptr);
        HIPCHK(hipGetLastError());
        if (!zc_out) HIPCHK(hipMemcpyAsync(c.h_out, c.d_out3, sizeof(double) * nb * 3, hipMemcpyDeviceToHost, sA));
      }
      const double hp2 = host_prof_on() ? now_us() : 0.;
      if (use_flags) HIPCHK(wait_flags(c.h_seq, c.h_out, nb, c.seq, sA)); else
      HIPCHK(wait_stream(sA, o.spin_wait != 0));
      if (host_prof_on()) { const double hp3 = now_us(); g_hp.pre.push_back(hp1 - hp0); g_hp.launch.push_back(hp2 - hp1); g_hp.sync.push_back(hp3 - hp2); }
      for (int b = 0; b < nb; b++) {
        if (out->log_hyper) out->log_hyper[b] = c.h_out[b * 3];
        if (out->log_num) out->log_num[b] = c.h_out[b * 3 + 1];
        if (out->N_exp) out->N_exp[b] = c.h_out[b * 3 + 2];
        if (out->partials) for (int k = 0; k < 3; k++) out->partials[b * 3 + k] = c.h_out[3 * nb + b * 3 + k];
      }
      c.t_valid = false;
      return CHM_OK;
    }
    if (key == c.gwarm) {                                     // second sight of this configuration: capture it
      // (a call that completed through its flags may still be retiring on the stream: drain it before the graph it runs from goes -- ADVICE r5)
      if (c.gexec) { HIPCHK(hipStreamSynchronize(sA)); (void)hipGraphExecDestroy(c.gexec); c.gexec = nullptr; c.gkey.clear(); }
      HIPCHK(hipStreamBeginCapture(sA, hipStreamCaptureModeThreadLocal));
      capturing = true;
    } else c.gwarm = key;
  }
  const bool timing = timing_env && !capturing;
  // (an eager call that carries timing events completes through the flags too; chm_last_timing waits for the call's last event before it reads them)
  const bool flags_now = use_flags;
  if (flags_now) { mark_pending(c.h_out, nb); c.h_seq[0] = ++c.seq; __atomic_thread_fence(__ATOMIC_RELEASE); }
  // with a communicator (multi-GPU shards: short calls) only the whole evaluation and the GW kernel are timed: each event record
  // costs ~3 us of stream time (measured: 35 us per call for the full set); CHM_TIMING_ALL=1 keeps the full set (diagnosing a multi-GPU line)
  // [r6] per-kernel events (sample stage, GW kernel, selection, reduction) only on request (CHM_OPT_TIMING 2: bench.py's pass after its timed region,
  // chm_last_timing's consumers): the default call carries the two events of the whole evaluation -- the full set was ~40 us of a 1.24 ms
  // step of a 125-event shard (profiles/r06/ab_shard_step_r06.txt: 1.24 ms without a communicator against 1.20 with one, which had the reduced set)
  const bool timing_all = timing && o.timing >= 2;
  // an error inside a capture must end it before returning
  struct CaptureGuard { hipStream_t s; bool* on; ~CaptureGuard() { if (*on) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(s, &g); if (g) (void)hipGraphDestroy(g); } } } cguard{sA, &capturing};
  if (timing) HIPCHK(hipEventRecord(c.ev[0], sA));
  if (zc_make) {
    hipLaunchKernelGGL(k_znodes, dim3((params[0].z_grid_res + 255) / 256), dim3(256), 0, sA, params[0].z_max, params[0].z_grid_res, c.zt_c, c.lz_c);
    HIPCHK(hipGetLastError());
    c.zc_zmax = params[0].z_max; c.zc_Tc = params[0].z_grid_res;
  }
  rc = ctx_tables_enqueue(c, nb, Tc_host, Tm_host, use_fast ? lutA : LutDesc{}, sel_fast ? lutB : LutDesc{}, td.zt, td.dLt, zero_copy, zc_use); if (rc) return rc;
  if (zg_make) {                                            // the table of draw 0 stands for all of them
    const size_t n = (size_t)like->L.E * like->L.Z;
    hipLaunchKernelGGL(k_grid_prep, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, sA, like->L.E, like->L.Z, like->L.z_grids,
                       (const double*)c.zt, params[0].z_grid_res, like->d_zg_i, like->d_zg_t, like->d_zg_lz);
    HIPCHK(hipGetLastError());
    like->zg_zmax = params[0].z_max; like->zg_Tc = params[0].z_grid_res;
  }
  HIPCHK(hipEventRecord(c.ev[1], sA));
  bool used_sB = false;                                      // (the fork of lane B is made when its first kernel is about to be enqueued)
  auto fork_sB = [&]() -> int { if (!one_stream && !used_sB) { HIPCHK(hipStreamWaitEvent(sB, c.ev[1], 0)); used_sB = true; } return CHM_OK; };

  const int Tc = c.TcMax, Tm = c.TmMax;
  const DevParams* dp = c.d_params;
  const size_t lds_samp = sizeof(double) * ((size_t)2 * Tc + (size_t)2 * Tm);      // zt, dLt, mg, cdf
  const size_t lds_zfac = sizeof(double) * (size_t)2 * Tc;                          // zt, It
  const bool tab_samp = lds_samp <= 64 * 1024, tab_zfac = lds_zfac <= 64 * 1024;

  // ---- events: groups of events alternate between two streams, so that the (VALU-bound) sample stage of one group
  //      overlaps the (latency-bound) GW-kernel stage of the previous one
  int nblk_ev = 0, ngroups = 0;
  bool ev_from_fixup = false;                                // k_marg_fixup formed L_i and log L_i of every event (standard marginalized kernel)
  if (like) {
    const LikeDev& L0 = like->L;
    const int Pd = L0.P > 0 ? L0.P : 1;
    const size_t N = L0.binning ? L0.num_bins : L0.S;
    const size_t lds_kde = sizeof(double) * (2 * N + (L0.binning ? 3 * (N + 1) : 0) + 2 * (size_t)L0.G);
    if (L0.mode != CHM_MODE_FULL && lds_kde > 150 * 1024)
      return fail(CHM_E_ARG, "chm_eval: KDE working set exceeds the LDS (binning=False needs 2*S + 2*G doubles <= 150 KiB)");
    // [r3] Event groups alternate between two streams: the sample stage of one group runs beside the GW kernel of the previous one.  Both
    // are VALU-issue bound at ~80 % of the slots with the chip to themselves; side by side they fill each other's stalls: 9.74 -> 9.43 ms per
    // 128-draw step at C3 with 4 - 8 groups (profiles/r03/ab_event_groups_and_lanes.txt; 12 / 16 groups: 9.6).  One group per 250 events, at
    // most 8; CHM_GROUPS=n overrides (1: one group).  Few-draw calls stay a single chain (one_stream).
    const int env_groups = o.groups;
    ngroups = env_groups > 0 ? env_groups : (nb > few_nb ? (L0.E / 250 < 1 ? 1 : (L0.E / 250 > 8 ? 8 : L0.E / 250)) : 1);
    if (ngroups > CHM_MAX_GROUPS) ngroups = CHM_MAX_GROUPS;
    if (ngroups > L0.E) ngroups = L0.E;
    if (one_stream) ngroups = 1;
    if ((L0.E + ngroups - 1) / ngroups > 65535) ngroups = (L0.E + 65534) / 65535;      // the GW kernel's grid carries the event in blockIdx.z
    if (ngroups > CHM_MAX_GROUPS) return fail(CHM_E_ARG, "chm_eval: more than 128 x 65535 events in one shard");
    for (int g = 0; g < ngroups; g++) {
      hipStream_t sg = (g & 1) ? sB : sA;
      if (sg == sB && sB != sA) { rc = fork_sB(); if (rc) return rc; }
      LikeDev L = like->L;
      L.tab_pm = td.pm_s; L.tab_rate = td.rate_g; L.tab_bkg = td.bkg_g; L.tab_jac = td.jac_g;
      if (zg_use) { L.zg_i = like->d_zg_i; L.zg_t = like->d_zg_t; L.zg_lz = like->d_zg_lz; }
      L.no_dense = o.no_dense ? 1 : 0;                      // diagnostics (-DCHM_DIAG): no dense-sum fallback in the standard GW kernel
      if (!want_dump) L.p_gw_dump = nullptr;
      const int eb = (int)((long long)L0.E * g / ngroups), ee = (int)((long long)L0.E * (g + 1) / ngroups);
      L.e_off = eb; L.E_cnt = ee - eb; L.nb = nb;
      L.zw_stream = (size_t)nb * (size_t)(ee - eb) * (size_t)L.S * 16 > ((size_t)256 << 20) ? 1 : 0;      // (256 MB: the memory-side cache)
      // per-z factors of the group's events: on the other lane, concurrently with the sample stage -- except in marginalized
      // mode, where they follow k_event_prep on the group's own lane and cover only the support of each event's KDE
#ifdef CHM_WITH_FUSED
      if (use_fused) {                                      // the whole event side of the call in one kernel (chm_fused.h)
        SampFast Fq = like->F; Fq.lut = lutA;
        L.ev_publish = 1;
        if (timing_all) { HIPCHK(hipEventRecord(c.evg[4 * g], sg)); HIPCHK(hipEventRecord(c.evg[4 * g + 1], sg)); }
        if (timing_all) HIPCHK(hipEventRecord(c.evg[4 * g + 2], sg));
#define LAUNCH_FUSED_(M, NWV, NTL) do { allow_lds((k_marg_fused<M, NWV, 200, NTL>), lds_fused); \
          hipLaunchKernelGGL((k_marg_fused<M, NWV, 200, NTL>), dim3((unsigned)L.E_cnt * nb), dim3(64 * NWV), lds_fused, sg, L, Fq, FDc, dp, c.zt, c.It, c.dLt, c.mg, c.cdf, c.rec, c.TcMax, c.TmMax); } while (0)
#ifdef CHM_FUSED_NW16
#define LAUNCH_FUSED(M) do { if (nb <= few_nb) { if (fused_nw == 16) LAUNCH_FUSED_(M, 16, true); else LAUNCH_FUSED_(M, 4, true); } else LAUNCH_FUSED_(M, 4, false); } while (0)
#else
#define LAUNCH_FUSED(M) do { if (nb <= few_nb) LAUNCH_FUSED_(M, 4, true); else LAUNCH_FUSED_(M, 4, false); } while (0)
#endif
        const int mm = params[0].mass_model;
        if (mm == 0) LAUNCH_FUSED(0); else if (mm == 1) LAUNCH_FUSED(1); else LAUNCH_FUSED(2);
#undef LAUNCH_FUSED
#undef LAUNCH_FUSED_
        HIPCHK(hipGetLastError());
        if (timing_all) HIPCHK(hipEventRecord(c.evg[4 * g + 3], sg));
        ev_from_fixup = true;
        continue;
      }
#endif
      const bool zf_ranged = (L.mode == CHM_MODE_MARG || L.mode == CHM_MODE_1D || L.mode == CHM_MODE_APPROX) && !opt_zf_full;
      // standard configuration (binning, cut_grid set) -> k_kde_marg_sub<32>, two pixels per wave (16 lanes per pixel measured
      // 30 % slower: 19 KB of LDS per wave halve the occupancy); anything else, or CHM_MARG_GENERIC=1, -> the general kernel
      // (a negative pe_prior -- negative weights, which the reference's arithmetic takes as they come -- goes to the general kernel: the standard
      //  one bounds its rounding error and skips empty bins on the assumption of weights >= 0; scripts/fuzz_parity.py, round 4)
      const bool marg_std = L.mode == CHM_MODE_MARG && L.binning && L.has_cut && (L.Z & 1) == 0 && !opt_marg_generic && !like->neg_prior && !L.grid_unsorted;
      const int zf_mode = !zf_ranged ? 0 : (L.mode == CHM_MODE_MARG ? 1 : 2);
      hipStream_t sz = (one_stream || zf_ranged) ? sg : ((g & 1) ? sA : sB);
      if (sz == sB && sB != sA) { rc = fork_sB(); if (rc) return rc; }
      // ~2048 blocks in all: each stages the draw's (zt, It) tables in LDS once and walks over E_cnt / gridDim.x events
#ifndef CHM_ZF_TARGET
#define CHM_ZF_TARGET 2048     // (A/B, profiles/r04/ab_zf_target.txt: 1536 = one full round of 6 blocks per CU, and 1024: -0.3 .. -0.8 % of the step -- inside the run-to-run spread; unchanged)
#endif
      const int zf_target = CHM_ZF_TARGET / nb > 1 ? CHM_ZF_TARGET / nb : 1;
      // few draws per call, standard marginalized configuration: the per-z-factor kernel forms the event statistics itself
      const bool zf_stats = zf_mode == 1 && marg_std && tab_zfac && nb <= few_nb;
      const int zf_epb = zf_stats ? 4 / CHM_ZF_WPE_FEW : 4;       // ranged: events per block pass (a wave per event; CHM_ZF_WPE_FEW waves per event for few draws)
      const int zf_units = zf_mode ? (L.E_cnt + zf_epb - 1) / zf_epb : L.E_cnt;
      const int zf_blocks = zf_units < zf_target ? zf_units : zf_target;
      // (the selection blocks inside the SAMPLE-stage launch instead were measured slower: profiles/r05/ab_scalar_call_r05.txt, docs/history/pruned_ab_arms_r06.patch)
      auto launch_zfactors = [&]() {
        if (fuse_sel) {                  // + the selection sums: blocks [zf_blocks, zf_blocks + gx)
          SelDev S = sel->S;
          int gx = 8192 / nb; gx = gx < 1 ? 1 : (gx > S.nblocks ? S.nblocks : gx);
          const size_t lds_f = lds_zfac > lds_sel ? lds_zfac : lds_sel;
#define LAUNCH_ZS(M) do { allow_lds(k_zf_sel<M>, lds_f); \
            hipLaunchKernelGGL((k_zf_sel<M>), dim3(zf_blocks + gx, nb), dim3(256), lds_f, sz, L, S, lutB, dp, c.zt, c.It, c.dLt, c.mg, c.cdf, c.rec, Tc, Tm, zf_blocks); } while (0)
          const int mm = params[0].mass_model;
          if (mm == 0) LAUNCH_ZS(0); else if (mm == 1) LAUNCH_ZS(1); else LAUNCH_ZS(2);
#undef LAUNCH_ZS
        } else if (zf_stats) { allow_lds((k_zfactors<true, true>), lds_zfac);
          hipLaunchKernelGGL((k_zfactors<true, true>), dim3(zf_blocks, nb), dim3(256), lds_zfac, sz, L, dp, c.zt, c.It, Tc, zf_mode);
        } else if (tab_zfac) { allow_lds(k_zfactors<true>, lds_zfac);
          hipLaunchKernelGGL(k_zfactors<true>, dim3(zf_blocks, nb), dim3(256), lds_zfac, sz, L, dp, c.zt, c.It, Tc, zf_mode);
        } else hipLaunchKernelGGL(k_zfactors<false>, dim3(zf_blocks, nb), dim3(256), 0, sz, L, dp, c.zt, c.It, Tc, zf_mode);
      };
      if (!zf_ranged) {
        launch_zfactors();
        HIPCHK(hipGetLastError());
        if (rate_special_call) {                            // [r5] rewrite the rate factors of the draws with an infinite rate parameter, flag the poisoned grids
          hipLaunchKernelGGL(k_rate_special, dim3(L.E_cnt, nb), dim3(256), 0, sz, L, dp);
          HIPCHK(hipGetLastError());
        }
        if (sz != sg) HIPCHK(hipEventRecord(c.evf[g], sz));
      }
      // sample stage
      if (timing_all) HIPCHK(hipEventRecord(c.evg[4 * g], sg));
      const int nchunk = L.E_cnt * (L.NC / SAMPLE_WPB);
      // blocks stage the draw's tables in LDS (40 KB) once and walk over their chunks of SAMPLE_CHUNK = 4096 samples (one record
      // of statistics per wave and chunk; 1024-sample chunks cost 3.68 ms at C3 / 64 draws, 2048: 3.33 ms, 4096: 2.98 ms);
      // four chunks per block amortise the staging while leaving enough blocks for dynamic balance (1 / 2 / 4 / 8 / 16 chunks per
      // block at C3 / 128 draws, barrier-free chunk loop: 5.89 / 5.70 / 5.61 / 5.67 / 5.78 ms), fewer when few draws leave fewer
      // than ~2048 blocks
      const int cpb_env = o.samp_cpb;
      int cpb = cpb_env > 0 ? cpb_env : 4;
      if (cpb_env <= 0) {
        long long want = (long long)nchunk * nb / 2048;
        cpb = want < 1 ? 1 : (want > 4 ? 4 : (int)want);
      }
      int samp_blocks = (nchunk + cpb - 1) / cpb;
      samp_blocks = samp_blocks < 1 ? 1 : (samp_blocks > 1024 ? 1024 : samp_blocks);
      dim3 g1(samp_blocks * nb, 1);
      const bool fullm = L.mode == CHM_MODE_FULL;
      if (use_fast) {
        SampFast F = like->F; F.lut = lutA;
#define LAUNCH_FAST_(M, FU, NTL) do { allow_lds((k_samples_fast<M, FU, NTL>), lds_fast); \
          hipLaunchKernelGGL((k_samples_fast<M, FU, NTL>), g1, dim3(64 * CHM_SF_WAVES), lds_fast, sg, L, F, dp, c.zt, c.dLt, c.mg, c.cdf, c.rec, Tc, Tm); } while (0)
#define LAUNCH_FAST(M, FU) do { if (nb <= few_nb && !(FU)) LAUNCH_FAST_(M, FU, true); else LAUNCH_FAST_(M, FU, false); } while (0)    /* few draws: tiles streamed non-temporally */
        const int mm = params[0].mass_model;
        if (fullm) { if (mm == 0) LAUNCH_FAST(0, true); else if (mm == 1) LAUNCH_FAST(1, true); else LAUNCH_FAST(2, true); }
        else { if (mm == 0) LAUNCH_FAST(0, false); else if (mm == 1) LAUNCH_FAST(1, false); else LAUNCH_FAST(2, false); }
#undef LAUNCH_FAST
#undef LAUNCH_FAST_
      } else if (tab_samp) {
        if (fullm) { allow_lds(k_samples<true, true>, lds_samp);
          hipLaunchKernelGGL((k_samples<true, true>), g1, dim3(64 * SAMPLE_WPB), lds_samp, sg, L, dp, c.zt, c.It, c.dLt, c.mg, c.cdf, Tc, Tm);
        } else { allow_lds(k_samples<true, false>, lds_samp);
          hipLaunchKernelGGL((k_samples<true, false>), g1, dim3(64 * SAMPLE_WPB), lds_samp, sg, L, dp, c.zt, c.It, c.dLt, c.mg, c.cdf, Tc, Tm); }
      } else {
        if (fullm) hipLaunchKernelGGL((k_samples<false, true>), g1, dim3(64 * SAMPLE_WPB), 0, sg, L, dp, c.zt, c.It, c.dLt, c.mg, c.cdf, Tc, Tm);
        else hipLaunchKernelGGL((k_samples<false, false>), g1, dim3(64 * SAMPLE_WPB), 0, sg, L, dp, c.zt, c.It, c.dLt, c.mg, c.cdf, Tc, Tm);
      }
      HIPCHK(hipGetLastError());
      if (timing_all) HIPCHK(hipEventRecord(c.evg[4 * g + 1], sg));
      // GW kernel + integrand (needs the per-z factors)
      if (sz != sg) HIPCHK(hipStreamWaitEvent(sg, c.evf[g], 0));
      if (L.mode == CHM_MODE_FULL) {
        if (timing_all) HIPCHK(hipEventRecord(c.evg[4 * g + 2], sg));
        // [r3] sample-stationary kernel first; the pixels it cannot do (non-uniform stretch of the grid, very coarse grid, > 4096 samples) are
        // flagged in full_todo and done by the general kernel, whose other blocks return at once.  CHM_FULL_CHAIN=0: general kernel only.
        const bool full_chain = o.full_chain != 0 && !rate_special_call;         // (CHM_OPT_DIAG_FULL_CHAIN 0: the general kernel alone; tests compare the two)
        if (full_chain) {
          hipLaunchKernelGGL(k_full_prep, dim3(L.E_cnt, nb), dim3(256), 0, sg, L); HIPCHK(hipGetLastError());
          hipLaunchKernelGGL(k_full_kde_chain, dim3(L.E_cnt * Pd, nb), dim3(64 * FULLC_NW), 0, sg, L, dp, L.full_todo); HIPCHK(hipGetLastError());
        }
        hipLaunchKernelGGL(k_full_kde, dim3(L.E_cnt * Pd, nb), dim3(256), 0, sg, L, dp, full_chain ? (const int*)L.full_todo : (const int*)nullptr);
      } else if (L.mode == CHM_MODE_MARG) {
        if (marg_std) { if (!zf_stats && !fuse_sel) hipLaunchKernelGGL(k_event_stats, dim3((L.E_cnt + 255) / 256, nb), dim3(256), 0, sg, L); }
        else hipLaunchKernelGGL(k_event_prep, dim3((L.E_cnt + 3) / 4, nb), dim3(256), 0, sg, L, 1);
        HIPCHK(hipGetLastError());
        if (zf_ranged) { launch_zfactors(); HIPCHK(hipGetLastError()); }
        if (timing_all) HIPCHK(hipEventRecord(c.evg[4 * g + 2], sg));
        const bool fast = marg_std;
        if (fast) {
          // several pixel groups (pairs of pixels) of the same (event, draw) per wave, one after the other: event statistics and segment
          // offsets once (four items per wave: 5.45 ms at C3 / 128 draws, two: 5.51, one: 5.66)
          // (two pixels per wave, 32 lanes each; one pixel per wave was measured 40 % slower at equal occupancy: profiles/r05/ab_gw_kernel_r05.txt)
          constexpr int GW_SW = 32, GW_NPW = 64 / GW_SW;
          const int PG2 = (Pd + GW_NPW - 1) / GW_NPW;
          const int ipw_env = o.kde_ipw;                    // diagnostics: 2 or 4 items per wave
          // (few draws per call: two items per wave -- twice the waves, half the serial chain of each: 0.238 -> 0.229 ms for the scalar call at C3)
          const int ipw = ipw_env == 2 || ipw_env == 4 ? ipw_env : ((PG2 >= 4 && nb > 8) ? 4 : 2);
          // (compile-time bin count: five slots of QS = B + 1 + 7 + 1 doubles and one of B + 1 per wave, see kde_sub_item: SPLIT)
          const size_t lds_sub = (L.num_bins == 200 && !L.p_gw_dump) ? sizeof(double) * (5 * (200 + 1 + 7 + 1) + 201) : sizeof(double) * (3 * N + 3) * GW_NPW;
#define LAUNCH_SUB2(I, BN, DU) hipLaunchKernelGGL((k_kde_marg_sub2<GW_SW, I, BN, DU>), dim3(nb, (PG2 + I - 1) / I, L.E_cnt), dim3(64), lds_sub, sg, L, dp)
          if (L.p_gw_dump) { if (ipw == 4) LAUNCH_SUB2(4, 0, true); else LAUNCH_SUB2(2, 0, true); }    // p_gw3d requested (tests, hyperlikelihood.p_gw3d): the instantiation that stores it
          else if (L.num_bins == 200) { if (ipw == 4) LAUNCH_SUB2(4, 200, false); else LAUNCH_SUB2(2, 200, false); }      // the reference's default bin count (likelihood.py:59): compile-time
          else { if (ipw == 4) LAUNCH_SUB2(4, 0, false); else LAUNCH_SUB2(2, 0, false); }
#undef LAUNCH_SUB2
          // events whose summed rounding bound matters against L_i (3e-10; the stated tolerance on L_i is 1e-9) get their heavy pixels redone with dense sums
          // few draws per call: the kernel also forms the per-event L_i / log L_i the reduction then reads (with CHM_NO_DENSE_NODE=1 it redoes nothing)
          L.ev_publish = nb <= few_nb ? 1 : 0;
          if (!L.no_dense || L.ev_publish) { HIPCHK(hipGetLastError()); allow_lds(k_marg_fixup, lds_kde); hipLaunchKernelGGL(k_marg_fixup, dim3(L.E_cnt, nb), dim3(64), lds_kde, sg, L, dp, 3e-10); }
          ev_from_fixup = L.ev_publish != 0;
        }
        else { allow_lds(k_kde_marg, lds_kde); hipLaunchKernelGGL(k_kde_marg, dim3(L.E_cnt * Pd, nb), dim3(64), lds_kde, sg, L, dp); }
      } else {
        if (timing_all) HIPCHK(hipEventRecord(c.evg[4 * g + 2], sg));
        allow_lds(k_kde1d, lds_kde);
        hipLaunchKernelGGL(k_kde1d, dim3(L.E_cnt, nb), dim3(256), lds_kde, sg, L, dp);
        HIPCHK(hipGetLastError());
        if (zf_ranged) { launch_zfactors(); HIPCHK(hipGetLastError()); }
        hipLaunchKernelGGL(k_integrate_1d, dim3(L.E_cnt * nb, 1), dim3(256), 0, sg, L, dp);
      }
      HIPCHK(hipGetLastError());
      if (timing_all) HIPCHK(hipEventRecord(c.evg[4 * g + 3], sg));
    }
    // join the two lanes -- [r6] only when the second one carried anything (one event group with ranged per-z factors runs on lane A alone: the record on
    // an idle stream + the barrier packet cost ~10 us between the fix-up and the reduction of a small shard)
    if (!one_stream && used_sB) { HIPCHK(hipEventRecord(c.evb[0], sB)); HIPCHK(hipStreamWaitEvent(sA, c.evb[0], 0)); }
    if (timing_all) HIPCHK(hipEventRecord(c.ev[3], sA));
    // per-event log-likelihoods and their block sums
    nblk_ev = (L0.E + 255) / 256;
  } else {
    if (timing_all) HIPCHK(hipEventRecord(c.ev[3], sA));
  }
  // ---- selection function on its own stream, forked after the tables.  Enqueued AFTER the event kernels: every API call between the
  //      table kernel and the sample stage is stream time the GPU idles (k_tables is 19 us; the fork used to cost 23 us there).
  //      ([r6] Enqueued right behind the sample stage and joined in FRONT of the GW kernel for one-group calls -- the kernel trace shows ~20 us between
  //      the fix-up and the reduction, the cross-queue dependency -- the step of the 125-event shard did not change: 1.1963 against 1.1967 ms, mean of
  //      three, same box; profiles/r06/ab_shard_step_r06.txt.  Not adopted.)
  if (sel && !fuse_sel) {
    HIPCHK(hipStreamWaitEvent(sC, c.ev[1], 0));
    SelDev S = sel->S;
    S.tab_pm = td.pm_i; S.tab_rate = td.rate_i; S.tab_bkg = td.bkg_i; S.tab_jac = td.jac_i;
    if (timing_all) HIPCHK(hipEventRecord(c.evb[1], sC));
    if (sel_fast) {
#define LAUNCH_SELF_(M, G) do { allow_lds((k_selection_fast<M, G>), lds_sel); \
        hipLaunchKernelGGL((k_selection_fast<M, G>), dim3(gx, nb), dim3(256), lds_sel, sC, S, lutB, dp, c.zt, c.dLt, c.mg, c.cdf, c.rec, Tc, Tm); } while (0)
#define LAUNCH_SELF(M) do { if (params[0].cosmo_model == 1) LAUNCH_SELF_(M, true); else LAUNCH_SELF_(M, false); } while (0)
      // every block stages the draw's table slice (tens of KB): ~8192 blocks in all, each walking over several tiles of injections
      const int self_blocks = (sel ? sel->opts.self_blocks : 8192) > 0 ? (sel ? sel->opts.self_blocks : 8192) : 8192;
      int gx = self_blocks / nb;
      gx = gx < 1 ? 1 : (gx > S.nblocks ? S.nblocks : gx);
      const int mm = params[0].mass_model;
      if (mm == 0) LAUNCH_SELF(0); else if (mm == 1) LAUNCH_SELF(1); else LAUNCH_SELF(2);
#undef LAUNCH_SELF
#undef LAUNCH_SELF_
    } else if (rate_special_call) {                          // a draw with an infinite rate parameter: the reference's own operations for the rate (merger_rate_special)
      hipLaunchKernelGGL((k_selection<false, true>), dim3(S.nblocks, nb), dim3(256), 0, sC, S, dp, c.zt, c.It, c.dLt, c.mg, c.cdf, Tc, Tm);
    } else if (tab_samp) { allow_lds(k_selection<true>, lds_samp);
      hipLaunchKernelGGL(k_selection<true>, dim3(S.nblocks, nb), dim3(256), lds_samp, sC, S, dp, c.zt, c.It, c.dLt, c.mg, c.cdf, Tc, Tm);
    } else hipLaunchKernelGGL(k_selection<false>, dim3(S.nblocks, nb), dim3(256), 0, sC, S, dp, c.zt, c.It, c.dLt, c.mg, c.cdf, Tc, Tm);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c.evb[2], sC));
  }

  double* d_lle = nullptr; double* d_nle = nullptr;
  const size_t El = like ? like->L.E : 0;
  // per-event outputs: buffers of the context, grown on demand (no allocation in the steady state of a compute_all loop)
  auto grow = [&](double*& buf, size_t& cap, size_t n) -> int {
    if (n <= cap) return CHM_OK;
    HIPCHK(hipStreamSynchronize(sA));
    (void)hipFree(buf); buf = nullptr; cap = 0;
    HIPCHK(hipMalloc(&buf, sizeof(double) * n));
    cap = n;
    return CHM_OK;
  };
  if (like && out->log_like_evs) { rc = grow(c.d_lle, c.lle_cap, (size_t)nb * El); if (rc) return rc; d_lle = c.d_lle; }
  if (like && out->numlike_evs) { rc = grow(c.d_nle, c.nle_cap, (size_t)nb * El); if (rc) return rc; d_nle = c.d_nle; }
  const bool multi = comm != nullptr;       // with a communicator the partials always go through ncclAllReduce + k_combine (also for
                                            // one rank, so that a single GPU exercises the very path the multi-GPU run takes)
  double Etot = comm ? (double)E_total : (like ? (double)like->L.E : 0.);
  const bool one_kernel = !like || like->L.E <= 4096;
  double* out3 = zc_out ? c.h_out : c.d_out3;              // zc_out: the last kernel stores the 3 doubles per draw in pinned host memory
  if (like && !one_kernel) {
    hipLaunchKernelGGL(k_reduce_events, dim3(nblk_ev, nb), dim3(256), 0, sA, like->L.E, like->L.P > 0 ? like->L.P : 1,
                       (const double*)like->L.like_pix, c.d_evpart, d_lle, d_nle, ev_from_fixup ? (const double*)like->L.ev_li : nullptr,
                       ev_from_fixup ? (const double*)like->L.ev_ll : nullptr, like->d_ev_bad);
    HIPCHK(hipGetLastError());
  }
  if (sel && !fuse_sel) HIPCHK(hipStreamWaitEvent(sA, c.evb[2], 0));     // join: selection sums
  if (one_kernel) {
    hipLaunchKernelGGL(k_reduce_final, dim3(nb), dim3(1024), 0, sA, like ? like->L.E : 0, like ? (like->L.P > 0 ? like->L.P : 1) : 1,
                       like ? (const double*)like->L.like_pix : nullptr, sel ? sel->S.nblocks : 0,
                       sel ? (const double*)sel->S.partial : nullptr, c.d_partials, dp, Etot, sel ? sel->S.N_inj : 1.,
                       sel ? sel->S.N_eff : 0., sel ? sel->S.has_neff : 0, like ? 1 : 0, sel ? 1 : 0, multi ? 0 : 1, out3, d_lle, d_nle,
                       ev_from_fixup ? (const double*)like->L.ev_li : nullptr, ev_from_fixup ? (const double*)like->L.ev_ll : nullptr,
                       like ? like->d_ev_bad : (const unsigned char*)nullptr, (const long long*)c.h_seq, (flags_now && !multi) ? c.h_seq + 1 : (long long*)nullptr);
  } else {
    hipLaunchKernelGGL(k_final, dim3(nb), dim3(256), 0, sA, nblk_ev, (const double*)c.d_evpart, sel ? sel->S.nblocks : 0,
                       sel ? (const double*)sel->S.partial : nullptr, c.d_partials, dp, Etot, sel ? sel->S.N_inj : 1.,
                       sel ? sel->S.N_eff : 0., sel ? sel->S.has_neff : 0, like ? 1 : 0, sel ? 1 : 0, multi ? 0 : 1, out3,
                       (const long long*)c.h_seq, (flags_now && !multi) ? c.h_seq + 1 : (long long*)nullptr);
  }
  HIPCHK(hipGetLastError());
  if (out->partials) HIPCHK(hipMemcpyAsync(c.h_out + 3 * nb, c.d_partials, sizeof(double) * nb * 3, hipMemcpyDeviceToHost, sA));
  auto end_capture = [&]() -> int {                           // instantiate what was captured so far and launch it
    hipGraph_t graph = nullptr;
    capturing = false;
    HIPCHK(hipStreamEndCapture(sA, &graph));
    hipError_t ge = hipGraphInstantiate(&c.gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ge != hipSuccess) { c.gexec = nullptr; return fail(CHM_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ge)); }
    c.gkey = key;
    HIPCHK(hipGraphLaunch(c.gexec, sA));
    return CHM_OK;
  };
  if (multi) {
    if (capturing) { rc = end_capture(); if (rc) return rc; }      // the collective stays outside the graph
    if (!turn.acquire()) return fail(CHM_E_RCCL, "chm_eval: the calls with lower tickets never enqueued their collectives (chm_comm_set_ticket; timeout)");
    NCCLCHK(ncclAllReduce(c.d_partials, c.d_partials, (size_t)nb * 3, ncclDouble, ncclSum, comm->comm, sA));
    turn.release();
    hipLaunchKernelGGL(k_combine, dim3((nb + 63) / 64), dim3(64), 0, sA, nb, dp, (const double*)c.d_partials, Etot,
                       sel ? sel->S.N_inj : 1., sel ? sel->S.N_eff : 0., sel ? sel->S.has_neff : 0, like ? 1 : 0, sel ? 1 : 0, out3,
                       (const long long*)c.h_seq, flags_now ? c.h_seq + 1 : (long long*)nullptr);
    HIPCHK(hipGetLastError());
  }
  if (!zc_out) HIPCHK(hipMemcpyAsync(c.h_out, c.d_out3, sizeof(double) * nb * 3, hipMemcpyDeviceToHost, sA));
  if (timing) HIPCHK(hipEventRecord(c.ev[5], sA));
  if (d_lle) HIPCHK(hipMemcpyAsync(out->log_like_evs, d_lle, sizeof(double) * nb * El, hipMemcpyDeviceToHost, sA));
  if (d_nle) HIPCHK(hipMemcpyAsync(out->numlike_evs, d_nle, sizeof(double) * nb * El, hipMemcpyDeviceToHost, sA));
  if (want_dump) {
    size_t Pd = like->L.P > 0 ? like->L.P : 1;
    const double* src = like->L.mode == CHM_MODE_1D ? like->L.pgw1d : like->L.p_gw_dump;
    HIPCHK(hipMemcpyAsync(out->p_gw, src, sizeof(double) * nb * El * Pd * like->L.Z, hipMemcpyDeviceToHost, sA));
  }
  if (capturing) { rc = end_capture(); if (rc) return rc; }
  if (flags_now) HIPCHK(wait_flags(c.h_seq, c.h_out, nb, c.seq, sA)); else
  HIPCHK(wait_stream(sA, o.spin_wait != 0 && nb <= few_nb));
  for (int b = 0; b < nb; b++) {
    if (out->log_hyper) out->log_hyper[b] = c.h_out[b * 3];
    if (out->log_num) out->log_num[b] = c.h_out[b * 3 + 1];
    if (out->N_exp) out->N_exp[b] = c.h_out[b * 3 + 2];
    if (out->partials) for (int k = 0; k < 3; k++) out->partials[b * 3 + k] = c.h_out[3 * nb + b * 3 + k];
  }
  c.t_ngroups = ngroups; c.t_like = like != nullptr; c.t_sel = sel != nullptr && !fuse_sel; c.t_valid = timing; c.t_all = timing_all; c.t_pending = timing && flags_now;      // (fused: the selection sums have no span of their own)
  return CHM_OK;
}

extern "C" int chm_like_full_general_pixels(chm_like* like, int32_t nb, int64_t* count) {
  if (!like || !count || nb < 1) return fail(CHM_E_ARG, "chm_like_full_general_pixels: null argument");
  *count = 0;
  const LikeDev& L = like->L;
  if (L.mode != CHM_MODE_FULL || !L.full_todo || nb > like->nb_ws) return CHM_OK;
  HIPCHK(hipSetDevice(like->ctx.device));
  HIPCHK(hipDeviceSynchronize());
  std::vector<int> f((size_t)nb * L.E * L.P);
  HIPCHK(hipMemcpy(f.data(), L.full_todo, sizeof(int) * f.size(), hipMemcpyDeviceToHost));
  for (int v : f) *count += v != 0;
  return CHM_OK;
}

extern "C" int chm_last_timing(chm_like* like, chm_sel* sel, double msout[8]) {
  if (!msout || (!like && !sel)) return fail(CHM_E_ARG, "chm_last_timing: null argument");
  Ctx& c = like ? like->ctx : sel->ctx;
  float ms = 0.f;
  for (int i = 0; i < 8; i++) c.ms[i] = 0.;
  if (c.t_valid) {                                                                      // elapsed times are formed on demand
    if (c.t_pending) { HIPCHK(hipSetDevice(c.device)); HIPCHK(hipEventSynchronize(c.ev[5])); c.t_pending = false; }
    if (hipEventElapsedTime(&ms, c.ev[0], c.ev[5]) == hipSuccess) c.ms[0] = ms;       // whole evaluation
    if (hipEventElapsedTime(&ms, c.ev[0], c.ev[1]) == hipSuccess) c.ms[1] = ms;       // tables
    for (int g = 0; g < c.t_ngroups; g++) {                                            // summed over the event groups
      if (c.t_all && hipEventElapsedTime(&ms, c.evg[4 * g], c.evg[4 * g + 1]) == hipSuccess) c.ms[2] += ms;       // sample stage
      if (c.t_all && hipEventElapsedTime(&ms, c.evg[4 * g + 2], c.evg[4 * g + 3]) == hipSuccess) c.ms[3] += ms;   // GW kernel + integrand
    }
    if (c.t_all && c.t_sel && hipEventElapsedTime(&ms, c.evb[1], c.evb[2]) == hipSuccess) c.ms[4] = ms;   // selection (own stream)
    if (c.t_all && hipEventElapsedTime(&ms, c.ev[3], c.ev[5]) == hipSuccess) c.ms[5] = ms;       // reduce + combine (+ all-reduce)
    if (c.t_all && c.t_like && hipEventElapsedTime(&ms, c.ev[1], c.ev[3]) == hipSuccess) c.ms[6] = ms;    // all event groups, wall
    c.ms[7] = (double)c.t_ngroups;
  }
  for (int i = 0; i < 8; i++) msout[i] = c.ms[i];
  return CHM_OK;
}

// ------------------------------------------------------------------------------------------------------
// elementwise model functions and tables (setup path; not performance critical)
// ------------------------------------------------------------------------------------------------------
static int with_tables(const chm_params* p, int device, Ctx& c) {
  if (!p) return fail(CHM_E_ARG, "null chm_params");
  int ndev = chm_device_count();
  if (device < 0 || device >= ndev) return fail(CHM_E_HIP, "no such HIP device (is a GPU visible?)");
  int rc = ctx_init(c, device); if (rc) return rc;
  rc = ctx_tables(c, p, 1);
  return rc;
}

extern "C" int chm_model_eval(const chm_params* p, int32_t func, const double* a, const double* b, int64_t n,
                              double* out, int32_t device) {
  if (!a || !out || n < 0) return fail(CHM_E_ARG, "chm_model_eval: null argument");
  if (func < 0 || func > CHM_F_TRUNC_GAUSSIAN) return fail(CHM_E_ARG, "chm_model_eval: unknown function id");
  if ((func == CHM_F_PM1M2 || func == CHM_F_SECONDARY || func == CHM_F_PM1M2_FUSED) && !b) return fail(CHM_E_ARG, "chm_model_eval: function needs two inputs");
  if (n == 0) return CHM_OK;
  Ctx c;
  int rc = with_tables(p, device, c);
  if (rc) { ctx_destroy(c); return rc; }
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  auto cleanup = [&]() { (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout); ctx_destroy(c); };
#define CK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { cleanup(); return fail(_e == hipErrorOutOfMemory ? CHM_E_NOMEM : CHM_E_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } } while (0)
  CK(hipMalloc(&da, sizeof(double) * n));
  CK(hipMalloc(&dout, sizeof(double) * n));
  CK(hipMemcpyAsync(da, a, sizeof(double) * n, hipMemcpyHostToDevice, c.stream));
  if (b) { CK(hipMalloc(&db, sizeof(double) * n)); CK(hipMemcpyAsync(db, b, sizeof(double) * n, hipMemcpyHostToDevice, c.stream)); }
  TablePtrs g = { c.zt, c.It, c.dLt, c.mg, c.cdf };
  long long nblk = (n + 255) / 256; if (nblk > 4096) nblk = 4096;
  hipLaunchKernelGGL(k_model_eval, dim3((unsigned)nblk), dim3(256), 0, c.stream, (const DevParams*)c.d_params, g, func, (const double*)da, (const double*)db, (long long)n, dout);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
#undef CK
  cleanup();
  return CHM_OK;
}

extern "C" int chm_model_tables(const chm_params* p, double* zt, double* It, double* dLt, double* mgrid,
                                double* cdf_m2, double* scalars, int32_t device) {
  Ctx c;
  int rc = with_tables(p, device, c);
  if (rc) { ctx_destroy(c); return rc; }
  auto cleanup = [&]() { ctx_destroy(c); };
#define CK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { cleanup(); return fail(CHM_E_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } } while (0)
  size_t Tc = p->z_grid_res, Tm = p->mass_grid_res;
  if (zt) CK(hipMemcpyAsync(zt, c.zt, sizeof(double) * Tc, hipMemcpyDeviceToHost, c.stream));
  if (It) CK(hipMemcpyAsync(It, c.It, sizeof(double) * Tc, hipMemcpyDeviceToHost, c.stream));
  if (dLt) CK(hipMemcpyAsync(dLt, c.dLt, sizeof(double) * Tc, hipMemcpyDeviceToHost, c.stream));
  if (mgrid) CK(hipMemcpyAsync(mgrid, c.mg, sizeof(double) * Tm, hipMemcpyDeviceToHost, c.stream));
  if (cdf_m2) CK(hipMemcpyAsync(cdf_m2, c.cdf, sizeof(double) * Tm, hipMemcpyDeviceToHost, c.stream));
  DevParams hp;
  CK(hipMemcpyAsync(&hp, c.d_params, sizeof(DevParams), hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
#undef CK
  if (scalars) { scalars[0] = hp.norm_p_m1; scalars[1] = hp.fR; }
  cleanup();
  return CHM_OK;
}

// ------------------------------------------------------------------------------------------------------
// catalogue term (setup path)
// ------------------------------------------------------------------------------------------------------
extern "C" int chm_pcat_compute(const chm_params* cosmo, const chm_pcat_desc* d, double* p_cat) {
  if (!cosmo || !d || !p_cat) return fail(CHM_E_ARG, "chm_pcat_compute: null argument");
  if (d->E <= 0 || d->P <= 0 || d->Z < 2 || !d->z_grids || !d->offsets) return fail(CHM_E_ARG, "chm_pcat_compute: need E, P > 0, Z >= 2, z_grids, offsets");
  const size_t npix = (size_t)d->E * d->P;
  const long long nnz = d->offsets[npix];
  if (d->offsets[0] != 0 || nnz < 0) return fail(CHM_E_ARG, "chm_pcat_compute: offsets must start at 0 and be non-decreasing");
  for (size_t i = 0; i < npix; i++) if (d->offsets[i + 1] < d->offsets[i]) return fail(CHM_E_ARG, "chm_pcat_compute: offsets must be non-decreasing");
  if (nnz > 0 && (!d->gal_z || !d->gal_sig || !d->gal_w)) return fail(CHM_E_ARG, "chm_pcat_compute: missing galaxy arrays");
  const size_t lds = sizeof(double) * 3 * (size_t)d->Z;
  if (lds > 150 * 1024) return fail(CHM_E_ARG, "chm_pcat_compute: Z too large for the LDS working set (3 Z doubles <= 150 KiB)");
  Ctx c;
  int rc = with_tables(cosmo, d->device, c);
  if (rc) { ctx_destroy(c); return rc; }
  std::vector<void*> owned;
  auto cleanup = [&]() { for (void* q : owned) (void)hipFree(q); ctx_destroy(c); };
  PcatDev D; memset(&D, 0, sizeof(D));
  D.E = d->E; D.P = d->P; D.Z = d->Z;
  const long long* offs = nullptr;
#define CKR(x) do { int _r = (x); if (_r) { cleanup(); return _r; } } while (0)
  CKR(upload(owned, d->z_grids, (size_t)d->E * d->Z, &D.z_grids, c.stream));
  CKR(upload(owned, (const long long*)d->offsets, npix + 1, &offs, c.stream));
  D.offsets = offs;
  if (nnz > 0) {
    CKR(upload(owned, d->gal_z, (size_t)nnz, &D.gal_z, c.stream));
    CKR(upload(owned, d->gal_sig, (size_t)nnz, &D.gal_sig, c.stream));
    CKR(upload(owned, d->gal_w, (size_t)nnz, &D.gal_w, c.stream));
    if (d->weight_grid) CKR(upload(owned, d->weight_grid, (size_t)d->E * d->Z, &D.weight_grid, c.stream));
  }
#undef CKR
  hipError_t he = hipMalloc(&D.p_cat, sizeof(double) * npix * d->Z);
  if (he != hipSuccess) { cleanup(); return fail(CHM_E_NOMEM, std::string("chm_pcat_compute: ") + hipGetErrorString(he)); }
  owned.push_back(D.p_cat);
  TablePtrs g = { c.zt, c.It, c.dLt, c.mg, c.cdf };
  allow_lds(k_pcat, lds);
  hipLaunchKernelGGL(k_pcat, dim3((unsigned)npix), dim3(256), lds, c.stream, D, (const DevParams*)c.d_params, g);
  he = hipGetLastError();
  if (he == hipSuccess) he = hipMemcpyAsync(p_cat, D.p_cat, sizeof(double) * npix * d->Z, hipMemcpyDeviceToHost, c.stream);
  if (he == hipSuccess) he = hipStreamSynchronize(c.stream);
  cleanup();
  if (he != hipSuccess) return fail(CHM_E_HIP, std::string("chm_pcat_compute: ") + hipGetErrorString(he));
  return CHM_OK;
}

#ifdef CHM_PHASE_PROF
// diagnostic builds only: read (and clear) the phase-cycle sums of k_kde_marg_sub
extern "C" int chm_debug_phase(double out[8]) {
  unsigned long long h[8];
  HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)));
  for (int i = 0; i < 8; i++) out[i] = (double)h[i];
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)));
  return CHM_OK;
}
extern "C" int chm_debug_phase_samples(double out[8]) {
  unsigned long long h[8];
  HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_s), sizeof(h)));
  for (int i = 0; i < 8; i++) out[i] = (double)h[i];
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_s), z, sizeof(z)));
  return CHM_OK;
}
#endif

// ---- stand-alone forms of CHIMERA/utils/math.py (host arrays in, host arrays out; one call = one array) ----
namespace {
struct DevScope {                                             // device buffers of one call, freed when it returns
  std::vector<void*> owned;
  ~DevScope() { for (void* q : owned) (void)hipFree(q); }
  template <class T> int up(const T* host, size_t n, const T** dev) { return upload(owned, host, n, dev, (hipStream_t)0); }
  int alloc(double** dev, size_t n) {
    *dev = nullptr;
    HIPCHK(hipMalloc(dev, sizeof(double) * (n ? n : 1)));
    owned.push_back(*dev);
    return CHM_OK;
  }
};
int math_device(const char* who, int32_t device) {
  int ndev = chm_device_count();
  if (device < 0 || device >= ndev) return fail(CHM_E_HIP, std::string(who) + ": no such HIP device (is a GPU visible?)");
  HIPCHK(hipSetDevice(device));
  return CHM_OK;
}
}  // namespace
#define MCK(x) do { int _r = (x); if (_r) return _r; } while (0)

extern "C" int chm_kde1d(const double* dataset, const double* weights, int64_t N, const double* grid, int64_t G, int32_t kernel,
                         int32_t bw_method, double bw_scalar, double* out, int32_t device) {
  if (!dataset || !grid || !out || N <= 0 || G <= 0 || kernel < 0 || kernel > 1 || bw_method < 0 || bw_method > 2)
    return fail(CHM_E_ARG, "chm_kde1d: need dataset (N > 0), grid (G > 0), out, kernel in {0,1}, bw_method in {0,1,2}");
  MCK(math_device("chm_kde1d", device));
  DevScope sc;
  const double *d_x = nullptr, *d_w = nullptr, *d_g = nullptr; double *d_st = nullptr, *d_o = nullptr;
  MCK(sc.up(dataset, (size_t)N, &d_x)); MCK(sc.up(weights, weights ? (size_t)N : 0, &d_w)); MCK(sc.up(grid, (size_t)G, &d_g));
  MCK(sc.alloc(&d_st, 2)); MCK(sc.alloc(&d_o, (size_t)G));
  hipLaunchKernelGGL(k_math_kde1d_setup, dim3(1), dim3(1024), 0, (hipStream_t)0, d_x, d_w, (long long)N, bw_method, bw_scalar, d_st);
  hipLaunchKernelGGL(k_math_kde1d_eval, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, (hipStream_t)0, d_x, d_w, (long long)N, d_g, (long long)G,
                     kernel == 0 ? 1 : 0, (const double*)d_st, d_o);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, d_o, sizeof(double) * (size_t)G, hipMemcpyDeviceToHost));
  return CHM_OK;
}

extern "C" int chm_binning1d(const double* dataset, const double* weights, int64_t N, int32_t num_bins, double* centers, double* counts,
                             int32_t device) {
  if (!dataset || !weights || !centers || !counts || N <= 0 || num_bins <= 0)
    return fail(CHM_E_ARG, "chm_binning1d: need dataset, weights (N > 0), num_bins > 0, centers, counts");
  MCK(math_device("chm_binning1d", device));
  DevScope sc;
  const double *d_x = nullptr, *d_w = nullptr; double *d_c = nullptr, *d_n = nullptr;
  MCK(sc.up(dataset, (size_t)N, &d_x)); MCK(sc.up(weights, (size_t)N, &d_w));
  MCK(sc.alloc(&d_c, (size_t)num_bins)); MCK(sc.alloc(&d_n, (size_t)num_bins));
  hipLaunchKernelGGL(k_math_binning1d, dim3(1), dim3(64), 0, (hipStream_t)0, d_x, d_w, (long long)N, (int)num_bins, d_c, d_n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(centers, d_c, sizeof(double) * (size_t)num_bins, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(counts, d_n, sizeof(double) * (size_t)num_bins, hipMemcpyDeviceToHost));
  return CHM_OK;
}

static int gkde_nd_impl(const double* dataset, const double* weights, int32_t d, int64_t N, const double* points, int64_t M,
                        int32_t bw_method, double bw_scalar, double* out, int32_t device, bool in_log) {
  if (!dataset || !points || !out || d <= 0 || d > CHM_GKDE_MAXD || N <= 1 || M <= 0 || bw_method < 0 || bw_method > 2)
    return fail(CHM_E_ARG, "chm_gkde_nd: need dataset (d in 1..4, N > 1), points (M > 0), out, bw_method in {0,1,2}");
  MCK(math_device("chm_gkde_nd", device));
  DevScope sc;
  const double *d_x = nullptr, *d_w = nullptr, *d_p = nullptr; double *d_st = nullptr, *d_o = nullptr;
  MCK(sc.up(dataset, (size_t)d * N, &d_x)); MCK(sc.up(weights, weights ? (size_t)N : 0, &d_w)); MCK(sc.up(points, (size_t)d * M, &d_p));
  MCK(sc.alloc(&d_st, 2 + CHM_GKDE_MAXD * CHM_GKDE_MAXD)); MCK(sc.alloc(&d_o, (size_t)M));
  hipLaunchKernelGGL(k_math_gkde_setup, dim3(1), dim3(1024), 0, (hipStream_t)0, d_x, d_w, (int)d, (long long)N, bw_method, bw_scalar, d_st);
  if (in_log) hipLaunchKernelGGL(k_math_gkde_eval<true>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)0, d_x, d_w, (int)d, (long long)N, d_p,
                                 (long long)M, (const double*)d_st, d_o);
  else hipLaunchKernelGGL(k_math_gkde_eval<false>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)0, d_x, d_w, (int)d, (long long)N, d_p,
                          (long long)M, (const double*)d_st, d_o);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, d_o, sizeof(double) * (size_t)M, hipMemcpyDeviceToHost));
  return CHM_OK;
}
extern "C" int chm_gkde_nd(const double* dataset, const double* weights, int32_t d, int64_t N, const double* points, int64_t M,
                           int32_t bw_method, double bw_scalar, double* out, int32_t device) {
  return gkde_nd_impl(dataset, weights, d, N, points, M, bw_method, bw_scalar, out, device, false);
}
extern "C" int chm_gkde_nd_log(const double* dataset, const double* weights, int32_t d, int64_t N, const double* points, int64_t M,
                               int32_t bw_method, double bw_scalar, double* out, int32_t device) {
  return gkde_nd_impl(dataset, weights, d, N, points, M, bw_method, bw_scalar, out, device, true);
}

extern "C" int chm_trapz(const double* y, const double* x, int64_t rows, int32_t n, int32_t x_per_row, double* out, int32_t device) {
  if (!y || !x || !out || rows <= 0 || n <= 0) return fail(CHM_E_ARG, "chm_trapz: need y (rows x n), x, out, rows > 0, n > 0");
  MCK(math_device("chm_trapz", device));
  DevScope sc;
  const double *d_y = nullptr, *d_x = nullptr; double* d_o = nullptr;
  MCK(sc.up(y, (size_t)rows * n, &d_y)); MCK(sc.up(x, x_per_row ? (size_t)rows * n : (size_t)n, &d_x)); MCK(sc.alloc(&d_o, (size_t)rows));
  hipLaunchKernelGGL(k_math_trapz, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)0, d_y, d_x, (long long)rows, (int)n, (int)x_per_row, d_o);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, d_o, sizeof(double) * (size_t)rows, hipMemcpyDeviceToHost));
  return CHM_OK;
}

extern "C" int chm_cumtrapz(const double* y, const double* x, int32_t n, double* out, int32_t device) {
  if (!y || !x || !out || n <= 0) return fail(CHM_E_ARG, "chm_cumtrapz: need y, x, out, n > 0");
  MCK(math_device("chm_cumtrapz", device));
  DevScope sc;
  const double *d_y = nullptr, *d_x = nullptr; double* d_o = nullptr;
  MCK(sc.up(y, (size_t)n, &d_y)); MCK(sc.up(x, (size_t)n, &d_x)); MCK(sc.alloc(&d_o, (size_t)n));
  hipLaunchKernelGGL(k_math_cumtrapz, dim3(1), dim3(1024), 0, (hipStream_t)0, d_y, d_x, (int)n, d_o);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, d_o, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  return CHM_OK;
}
#undef MCK

extern "C" int chm_kde2d_pixels(int32_t E, int32_t S, int32_t P, const double* ra, const double* dec, const double* ra_pix,
                                const double* dec_pix, const int32_t* npix, double* out, int32_t device) {
  if (E <= 0 || S <= 1 || P <= 0 || !ra || !dec || !ra_pix || !dec_pix || !npix || !out)
    return fail(CHM_E_ARG, "chm_kde2d_pixels: need E > 0, S > 1, P > 0 and all arrays");
  int ndev = chm_device_count();
  if (device < 0 || device >= ndev) return fail(CHM_E_HIP, "chm_kde2d_pixels: no such HIP device (is a GPU visible?)");
  HIPCHK(hipSetDevice(device));
  std::vector<void*> owned;
  auto cleanup = [&]() { for (void* q : owned) (void)hipFree(q); };
  const double *d_ra = nullptr, *d_dec = nullptr, *d_rp = nullptr, *d_dp = nullptr; const int* d_np = nullptr;
#define CKR(x) do { int _r = (x); if (_r) { cleanup(); return _r; } } while (0)
  CKR(upload(owned, ra, (size_t)E * S, &d_ra, (hipStream_t)0));
  CKR(upload(owned, dec, (size_t)E * S, &d_dec, (hipStream_t)0));
  CKR(upload(owned, ra_pix, (size_t)E * P, &d_rp, (hipStream_t)0));
  CKR(upload(owned, dec_pix, (size_t)E * P, &d_dp, (hipStream_t)0));
  CKR(upload(owned, (const int*)npix, (size_t)E, &d_np, (hipStream_t)0));
#undef CKR
  double* d_out = nullptr;
  hipError_t he = hipMalloc(&d_out, sizeof(double) * (size_t)E * P);
  if (he != hipSuccess) { cleanup(); return fail(CHM_E_NOMEM, std::string("chm_kde2d_pixels: ") + hipGetErrorString(he)); }
  owned.push_back(d_out);
  he = hipMemcpy(d_out, out, sizeof(double) * (size_t)E * P, hipMemcpyHostToDevice);     // keeps the caller's padding
  if (he == hipSuccess) {
    hipLaunchKernelGGL(k_kde2d, dim3(E), dim3(256), 0, (hipStream_t)0, S, P, d_ra, d_dec, d_rp, d_dp, d_np, d_out);
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipMemcpy(out, d_out, sizeof(double) * (size_t)E * P, hipMemcpyDeviceToHost);
  cleanup();
  if (he != hipSuccess) return fail(CHM_E_HIP, std::string("chm_kde2d_pixels: ") + hipGetErrorString(he));
  return CHM_OK;
}

// ------------------------------------------------------------------------------------------------------
// RCCL communicator (one process per GPU)
// ------------------------------------------------------------------------------------------------------
static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId larger than the ABI's 128 bytes");

extern "C" int chm_comm_unique_id(char id[128]) {
  if (!id) return fail(CHM_E_ARG, "chm_comm_unique_id: null argument");
  ncclUniqueId u;
  NCCLCHK(ncclGetUniqueId(&u));
  memset(id, 0, 128);
  memcpy(id, &u, sizeof(u));
  return CHM_OK;
}

extern "C" int chm_comm_init_rank(const char id[128], int32_t nranks, int32_t rank, int32_t device, chm_comm** out) {
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(CHM_E_ARG, "chm_comm_init_rank: bad argument");
  *out = nullptr;
  int ndev = chm_device_count();
  if (device < 0 || device >= ndev) return fail(CHM_E_HIP, "chm_comm_init_rank: no such HIP device");
  HIPCHK(hipSetDevice(device));
  chm_comm* c = new chm_comm();
  c->nranks = nranks; c->rank = rank; c->device = device;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
  if (r != ncclSuccess) { delete c; return fail(CHM_E_RCCL, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
  hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (he != hipSuccess) { ncclCommDestroy(c->comm); delete c; return fail(CHM_E_HIP, "chm_comm_init_rank: stream create failed"); }
  *out = c;
  return CHM_OK;
}

extern "C" int chm_comm_destroy(chm_comm* c) {
  if (!c) return CHM_OK;
  (void)hipSetDevice(c->device);
  if (c->d_buf) (void)hipFree(c->d_buf);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->comm) ncclCommDestroy(c->comm);
  delete c;
  return CHM_OK;
}

extern "C" int chm_comm_nranks(chm_comm* c) {
  if (!c || !c->comm) return 0;
  int n = 0;
  if (ncclCommCount(c->comm, &n) != ncclSuccess) return 0;     // what RCCL itself says, not what the caller passed
  return n;
}

// PCI bus id of a device ("0000:c1:00.0"): an N-rank job records one per rank -- N distinct ids prove N ranks on N GPUs (bench.py: multi_gpu.pci_bus_ids)
extern "C" int chm_device_pci_bus_id(int32_t device, char* out, int32_t len) {
  if (!out || len < 16) return fail(CHM_E_ARG, "chm_device_pci_bus_id: need a buffer of at least 16 bytes");
  int ndev = chm_device_count();
  if (device < 0 || device >= ndev) return fail(CHM_E_HIP, "chm_device_pci_bus_id: no such HIP device");
  HIPCHK(hipDeviceGetPCIBusId(out, len, device));
  return CHM_OK;
}

extern "C" int chm_device_synchronize(int32_t device) {
  int ndev = chm_device_count();
  if (device < 0 || device >= ndev) return fail(CHM_E_HIP, "chm_device_synchronize: no such HIP device");
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipDeviceSynchronize());
  return CHM_OK;
}

extern "C" int chm_comm_allreduce_sum(chm_comm* c, double* buf, int32_t n) {
  if (!c || !buf || n <= 0) return fail(CHM_E_ARG, "chm_comm_allreduce_sum: bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (n > c->cap) { if (c->d_buf) (void)hipFree(c->d_buf); c->d_buf = nullptr; HIPCHK(hipMalloc(&c->d_buf, sizeof(double) * n)); c->cap = n; }
  HIPCHK(hipMemcpyAsync(c->d_buf, buf, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  NCCLCHK(ncclAllReduce(c->d_buf, c->d_buf, (size_t)n, ncclDouble, ncclSum, c->comm, c->stream));
  HIPCHK(hipMemcpyAsync(buf, c->d_buf, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return CHM_OK;
}

#ifdef CHM_PROBE
// measured ceilings of the two hot kernels (diagnostic builds only): the production bodies on a cache-resident workload
#include "../../scripts/sample_body_probe.hip"
#include "../../scripts/gw_loop_probe.hip"
#endif
