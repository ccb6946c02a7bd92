// chm_kernels.h -- HIP kernels of the hyper-likelihood path (gfx950 / CDNA4, wave64, fp64).
//
// One evaluation (nb = draws in the batch, blockIdx.y = draw).  Stream A: k_tables -> k_samples -> GW-kernel -> reduce;
// stream B (forked after k_tables): k_zfactors, k_selection.
//   k_tables        1 block / draw            per-draw tables + constants            cosmo.py:43-46,263, mass.py:45-52
//   k_samples       blocks walk over chunks   det->src, weights, partial statistics  pop_wrapper.py:67-80, likelihood.py:111-118
//   k_zfactors      1 block / event           per-z factors of the integrand          likelihood.py:270-272, pop_wrapper.py:82-90
//   k_kde_marg      1 wave  / (event,pixel)   histogram, KDE, interp, integrand, trapz  math.py:32-89, likelihood.py:160-205,266-281
//   k_kde1d         1 block / event           1-D GW kernel p_gw(z)                   likelihood.py:105-144
//   k_integrate_1d  1 wave  / (event,pixel)   integrand + trapz for 1d / approximate  likelihood.py:150-154,266-292
//   k_full_kde      1 block / (event,pixel)   3-D Gaussian KDE + integrand            math.py:154-229, likelihood.py:211-260
//   k_selection     grid-stride               dN/dtheta per injection + 2 sums        pop_wrapper.py:102-111
//   k_reduce_events / k_final / k_combine     log, nan_to_num, sums, N_exp guard      likelihood.py:294-338, selection_function.py:38-48
//
// HBM layout: sample arrays are (E,S) row-major fp64; in marginalized mode the samples of an event are stored SORTED
// BY PIXEL at upload time (stable, samples outside every pixel last) with seg_off[e][0..P] giving each pixel's
// contiguous segment, so a (event,pixel) wave reads exactly its own samples; p_cat is (E,P,Z) and is streamed once.
#pragma once
#include <type_traits>
#include "chm_models.h"

#define NPART 16
// per-(draw,event,chunk) partial statistics written by k_samples; d = z - z_ref (z_ref = z of the event's first sample)
enum { PT_SW = 0, PT_SW2, PT_SD1, PT_SD2, PT_ZMIN, PT_ZMAX, PT_WD0, PT_WD1, PT_WD2, PT_W00, PT_W01, PT_W02, PT_W11, PT_W12, PT_W22, PT_ZREF };
#define SAMPLE_CHUNK 4096
#define SAMPLE_WPB 8                   // waves per block of k_samples (512 threads): one partial record per wave and chunk
#define NEVSTAT 12           // doubles per (draw, event) written by k_event_prep

struct LikeDev {                  // device-resident shard of events (see chm_like_desc)
  int E, S, Z, P;
  int mode, kernel, bw_method, binning, num_bins, G, has_cut, NC;
  int e_off, E_cnt;               // event group handled by this launch: events [e_off, e_off + E_cnt)
  int ev_publish;                 // k_marg_fixup writes every event's L_i / log L_i (ev_li, ev_ll): few-draw calls, where the one-block reduction is on the critical path
  int neg_w;                      // some pe_prior < 0 (set at upload): negative sample weights -- the prefix-sum forms of the binned Epanechnikov KDE bound their
                                  // rounding and clamp at 0 on the assumption of weights >= 0; the dense sums of the reference are taken instead
  int nb, no_dense;               // draws in this call; no_dense (diagnostics, CHM_NO_DENSE_NODE=1): the standard GW kernel keeps the prefix differences everywhere
                                  // (hot kernels fold the draw into blockIdx.x, draw fastest, so that the
                                  // blocks working on the same samples / p_cat rows for different draws run together and share L2)
  double bw_scalar, cut_grid, pe_neff;
  double inv_B, std_unit;         // 1/num_bins; std of num_bins uniform bin centres per unit range, sqrt((B^2-1)/12)/B (host, once)
  const double *dL, *m1det, *m2det, *pe_prior, *ra, *dec;
  const double *lm1det, *lm2det;  // log(m1det), log(m2det), formed once at upload: log(m_src) = log(m_det) - log(1+z)
  const int* seg_off;             // (E,P+1) marginalized: pixel segments of the pixel-sorted samples
  const double *z_grids, *p_cat, *P_compl, *gw_pdf, *ra_pix, *dec_pix;
  const int* neff_pixels;
  const double *dl_lo, *dl_hi;    // (E,) smallest / largest finite dL of each event (set at upload): brackets its table searches
  const int* perm;                // (E,S) marginalized: original index of the pixel-sorted sample (for caller-tabulated values)
  const double *tab_pm, *tab_rate, *tab_bkg;   // plug-in models evaluated by the caller (chm_tab), device copies; NULL = built-in
  const double *tab_jac;          // (nb,E,Z) plug-in cosmology: ddL/dz (1+z)^2 on the event grids
  // draw-independent part of the per-z factors (k_grid_prep; valid while every draw has the z_max / z_grid_res it was made for: the
  // z table [0] U logspace(-10, log10 z_max, Tc - 1) depends on nothing else): bracket of each grid point on the table, its
  // interpolation weight (z - zt[i-1])/(zt[i] - zt[i-1]) (codes < 0: see interp_pre) and log(1 + z)
  const int* zg_i; const double* zg_t; const double* zg_lz;
  const double *fracB, *fracG;    // i/num_bins (num_bins+1), i/(G-1) (G): the step fractions of jnp.linspace
  // workspaces (nb-major)
  double *ws_z, *ws_w;            // (nb,E,S)
  double *part;                   // (nb,E,NC,NPART), NC = chunks per event x SAMPLE_WPB
  double *jac, *prate, *bkgA;     // (nb,E,Z)
  double *Aw;                     // (nb,E,Z)  prate/jac * trapezoid weight (marginalized)
  double *evstat;                 // (nb,E,8)  per-event statistics (k_event_prep)
  double *effg;                   // (nb,E,G)  per-event effective grid (k_event_prep)
  int *krange;                    // (nb,E,2)  first / last grid point with p_gw1d != 0 (k_kde1d; 1d / approximate)
  double *pgw1d;                  // (nb,E,Z)   1d / approximate
  double *like_pix;               // (nb,E,max(P,1))
  double *err_pix;                // (nb,E,P)  marginalized, standard kernel: bound on what the prefix-sum form may have lost (k_marg_fixup)
  double *ev_li, *ev_ll;          // (nb,E)    marginalized, standard kernel: L_i and nan_to_num(log L_i) of every event, formed by k_marg_fixup
  double *p_gw_dump;              // optional (nb,E,P,Z) or NULL
  int *full_todo;                 // (nb,E,P)  full mode: 1 = the pixel is left to the general kernel by k_full_kde_chain
  double *full_ev;                // (nb,E,FULLEV) full mode: k_full_prep's record of every (draw, event)
  double *full_s;                 // (5,nb_alloc,E,S) full mode: whitened coordinates a, y1, y2, normalised weight, step factor U of every sample
  int nb_alloc;                   // draws the workspaces are allocated for (the stride of full_s's five planes)
  int grid_unsorted;              // [r6] some row of z_grids is not non-decreasing (set at upload): the k-range of an event is then the whole grid and the
                                  // marginalized mode takes the general kernel (the standard one ends its grid loop at the first pass beyond the KDE's support)
  int zw_stream;                  // [r6] the (z, w) of this launch exceed what the memory-side cache holds (set per launch): the fast sample stage stores them with the streaming hint
  unsigned char *ev_rbad;         // (nb,E) [r5] calls with an infinite rate parameter only: 1 = the rate factor prate/jac is inf or NaN at some point of the event grid
                                  // (k_zfactors, whole-grid launch) -- the reference's trapz then holds a 0 * inf = NaN where p_gw vanishes (event_poisoned)
};

struct EvStats { double zmin, zmax, sd, norm, n_eff, sumw; };
// L_i of (draw, event) is NaN whatever the KDE: NaN distance factors on the event grid (grid_is_poisoned), or -- a draw with an infinite rate
// parameter -- a rate factor that is inf / NaN somewhere on the grid.  The kernels skip grid points where p_gw vanishes; the reference multiplies
// them out (likelihood.py:274-278: 0 * inf = NaN).  (Corner: every such point inside the KDE's support AND the value +inf -> the reference has
// +inf.  k_integrate_1d -- the 1-D and 'approximate' modes, where a Gaussian kernel without cut_grid makes it reachable -- forms the reference's
// products over the whole grid for such a draw and reproduces it; the compact-support modes keep this rule.)
DEVFN bool event_poisoned(const LikeDev& L, const DevParams& P, int b, int e, const double* zg, int Z) {
  return grid_is_poisoned(P.z_bad, zg, Z) || (P.rate_special && L.ev_rbad && L.ev_rbad[(size_t)b * L.E + e]);
}

// Combine the chunk partials of one event.  std via the shifted one-pass form: var = <d^2> - <d>^2 with
// d = z - z_ref (jnp.std two-pass result to ~1e-15 relative; likelihood.py:118,186,222).
DEVFN EvStats combine_stats(const double* part, int NC, int S) {
  double sw = 0., sw2 = 0., sd1 = 0., sd2 = 0.;
  double zmn = part[PT_ZMIN], zmx = part[PT_ZMAX];
  // four records (the waves of one k_samples block) per round: their loads are issued together, the sums run in record order
  for (int c0 = 0; c0 < NC; c0 += 4) {
    double f[4][6];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double* q = part + (size_t)(c0 + i < NC ? c0 + i : NC - 1) * NPART;
      f[i][0] = q[PT_SW]; f[i][1] = q[PT_SW2]; f[i][2] = q[PT_SD1]; f[i][3] = q[PT_SD2]; f[i][4] = q[PT_ZMIN]; f[i][5] = q[PT_ZMAX];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) if (c0 + i < NC) {
      sw += f[i][0]; sw2 += f[i][1]; sd1 += f[i][2]; sd2 += f[i][3];
      zmn = nanmin2(zmn, f[i][4]); zmx = nanmax2(zmx, f[i][5]);
    }
  }
  EvStats s;
  double md = sd1 / (double)S;
  double var = sd2 / (double)S - md * md;
  s.zmin = zmn; s.zmax = zmx;
  s.sd = sqrt(var > 0. ? var : (var != var ? var : 0.));
  s.norm = sw / (double)S;                       // jnp.mean(weights)          likelihood.py:111
  s.n_eff = (sw * sw) / sw2;                     // sum(w)^2 / sum(w^2)        likelihood.py:112
  s.sumw = sw;
  return s;
}

// ------------------------------------------------------------------------------------------------------
// k_tables: per-draw tables; grid (nb, 2): blockIdx.y = 0 cosmology (cosmo.py:43-46, 263; completeness.py:54-58),
//           blockIdx.y = 1 mass normalisations (mass.py:45-52).  1024 threads per block.
// ------------------------------------------------------------------------------------------------------
// exclusive prefix sum of one value per thread over the block (blockDim.x <= 1024); sh: >= 17 doubles
DEVFN double block_excl_scan(double v, double* sh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  double x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { double y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
  __syncthreads();
  if (lane == 63) sh[wid] = x;
  __syncthreads();
  double off = 0.;
  for (int w = 0; w < wid && w < nw; w++) off += sh[w];
  // exclusive prefix = the inclusive value of the previous lane (not x - v: a NaN value of THIS thread -- an unphysical draw's
  // 1/E -- must not reach the entries before it, jnp.cumsum turns NaN only from the first NaN term on)
  double ex = __shfl_up(x, 1, 64);
  if (lane == 0) ex = 0.;
  return off + ex;
}

// cumtrapz(y, x) (math.py:22-26) of n points in global memory by one block: thread t owns a contiguous chunk of
// terms 0.5 (y_k + y_{k+1}) (x_{k+1} - x_k); out[0] = 0.
DEVFN void block_cumtrapz(const double* y, const double* x, double* out, int n, double* sh) {
  const int nt = blockDim.x, t = threadIdx.x;
  const int nterm = n - 1;
  const int per = (nterm + nt - 1) / nt;
  const int k0 = min(t * per, nterm), k1 = min(k0 + per, nterm);
  double acc = 0.;
  for (int k = k0; k < k1; k++) acc += 0.5 * (y[k] + y[k + 1]) * (x[k + 1] - x[k]);
  acc = block_excl_scan(acc, sh);
  if (t == 0) out[0] = 0.;
  for (int k = k0; k < k1; k++) { acc += 0.5 * (y[k] + y[k + 1]) * (x[k + 1] - x[k]); out[k + 1] = acc; }
  __syncthreads();
}

// jnp.interp(x, xp, fp) (clamped) by the whole block: the thread that owns the bracketing interval publishes it
DEVFN double block_interp(double x, const double* xp, const double* fp, int n, double* sh) {
  const int nt = blockDim.x, t = threadIdx.x;
  __syncthreads();
  if (t == 0) sh[0] = -1.;
  __syncthreads();
  for (int i = t + 1; i < n; i += nt) {               // interval (i-1, i): xp[i-1] <= x < xp[i]  (searchsorted right)
    if (xp[i - 1] <= x && (x < xp[i] || i == n - 1)) sh[0] = (double)i;
  }
  __syncthreads();
  int i = (int)sh[0];
  if (i < 1) i = 1;                                   // x < xp[0] (or NaN): jnp.interp clamps to the first interval
  double f0 = fp[i - 1], f1 = fp[i], x0 = xp[i - 1], x1 = xp[i];
  double dx = x1 - x0;
  double f = (fabs(dx) <= 4.930380657631324e-32) ? f0 : f0 + ((x - x0) / dx) * (f1 - f0);
  if (x < xp[0]) f = fp[0];
  if (x > xp[n - 1]) f = fp[n - 1];
  return f;
}

// LDS_ARR: the working arrays (grid, integrand, running integral) live in LDS (3 T doubles) and are written to global
// memory once at the end; otherwise (very long tables) the global arrays are used throughout.
// Data exchanged between the threads of the block goes through LDS (the draw's parameter block `Ps`, the working arrays when
// LDS_ARR) or, for very long tables, through global memory behind an agent-scope fence (gsync).
DEVFN void gsync() { __threadfence(); __syncthreads(); }

// Direct-index table for z_from_dGW (k_samples_fast, k_selection): the key of a distance is the top LUT_KEYBITS of its fp64 bit
// pattern (sign, exponent, 7 mantissa bits: 128 keys per octave), monotone in the distance; lut[k] = #{table entries <= smallest
// double with key key0 + k} (searchsorted side='right'), so a distance with key k has its searchsorted answer in [lut[k], lut[k+1]]
// -- at most `lmax` entries to look at (1 for the reference's 1500-point table) instead of a 7-11 step binary search.
// info[4] per draw: i_lo (first table entry a search or the interpolation can touch), ns (their number), lmax, fits (the draw's
// table is sorted and its slice [i_lo, i_lo + ns) fits the LDS capacity the host reserved).
#define LUT_SHIFT 13
#define LUT_MAXKEYS 8192
struct LutDesc { int key0, nk, cap, pad; unsigned short* lut; int* info; };
DEVFN int lut_key(double x, int key0) { return (__double2hiint(x) >> LUT_SHIFT) - key0; }
DEVFN double lut_key_floor(int key) { return __hiloint2double(key << LUT_SHIFT, 0); }

#ifdef CHM_TABLES_PROF
#define TS_INIT long long ts_[9]; int ts_n = 0
#define TS(i) do { __syncthreads(); ts_[ts_n++] = wall_clock64(); } while (0)
#define TS_PRINT do { if (t == 0 && b == 0) printf("[k_tables phases, x10 ns] parameter block %lld | nodes+1/E %lld | cumtrapz %lld | dL %lld | flags + index tables + records + fR %lld\n", \
  ts_[0] - ts_k0, ts_[1] - ts_[0], ts_[2] - ts_[1], ts_[3] - ts_[2], ts_[4] - ts_[3]); } while (0)
#define TS_PRINT_M do { if (t == 0 && b == 0) printf("[k_tables mass block, x10 ns] parameter block %lld | constants (thread 0) %lld | nodes + secondary %lld | cumtrapz %lld | primary %lld | norm %lld\n", \
  ts_[0] - ts_k0, ts_[1] - ts_[0], ts_[2] - ts_[1], ts_[3] - ts_[2], ts_[4] - ts_[3], ts_[5] - ts_[4]); } while (0)
#else
#define TS_INIT
#define TS(i)
#define TS_PRINT
#define TS_PRINT_M
#endif
// node i of z_grid_interp = [0] U logspace(-10, log10 z_max, Tc - 1) (cosmo.py:43-46), lzmax = log10(z_max)
DEVFN double znode_at(double lzmax, int Tc, int i) { return i == 0 ? 0. : chm_pow10(jnp_linspace_at(-10., lzmax, Tc - 1, i - 1)); }
// k_znodes: the nodes and their log(1 + z) for one (z_max, Tc) -- what k_tables otherwise re-forms for every draw of every call
__global__ void __launch_bounds__(256) k_znodes(double z_max, int Tc, double* zt_c, double* lz_c) {
  const double lzmax = log10(z_max);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Tc; i += gridDim.x * 256) { const double z = znode_at(lzmax, Tc, i); zt_c[i] = z; lz_c[i] = log1p(z); }
}
// (the long-table variant, LDS_ARR = false, runs 512-thread blocks: at 1024 threads it sits at the 128-register limit and any spill there
//  has ended in a memory-aperture fault on the MI355X boxes, see tests/test_abi_and_host.py)
#ifndef CHM_TABLES_LONG_NT
#define CHM_TABLES_LONG_NT 512
#endif
template <bool LDS_ARR>
__global__ void __launch_bounds__(LDS_ARR ? 1024 : CHM_TABLES_LONG_NT) k_tables(DevParams* params, double* zt_all, double* It_all,
                                                  double* dLt_all, double* mg_all, double* cdf_all, double* tmp_all, int TcMax, int TmMax,
                                                  LutDesc lutA, LutDesc lutB, double* rec_all, const double* tab_zt, const double* tab_dLt, const DevParams* hsrc,
                                                  const double* zt_c, const double* lz_c) {
  extern __shared__ double larr[];
#ifdef CHM_TABLES_PROF
  const long long ts_k0 = wall_clock64();
#endif
  __shared__ double sh[32];
  __shared__ int lut_ks[LUT_MAXKEYS + 1];           // searchsorted answer of every key of the direct-index tables
  __shared__ int s_int[8];                          // [0] first non-finite node of the integral, [1], [2] widest key of table A / B, [3], [4] brackets of the completeness limits
  __shared__ DevParams Ps;                          // block-local copy of the draw: constants derived here are shared through LDS
  const int b = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  DevParams& Pg = params[b];
#ifdef CHM_TABLES_FORCE_SCRATCH
  // diagnostics (scripts/probe_tables_scratch.py): a dynamically indexed private array puts CHM_TABLES_FORCE_SCRATCH doubles per lane into the
  // private segment -- the round-2 builds of this kernel that spilled registers at 1024 threads died with a memory-aperture violation
  volatile double pad_[CHM_TABLES_FORCE_SCRATCH];
  for (int i = 0; i < CHM_TABLES_FORCE_SCRATCH; i++) pad_[(i + t) % CHM_TABLES_FORCE_SCRATCH] = (double)(t + i);
#endif
  if (t == 0) { s_int[0] = 0x7fffffff; s_int[1] = 0; s_int[2] = 0; s_int[3] = -1; s_int[4] = -1; }
  if (t < (int)(sizeof(DevParams) / sizeof(double))) {
    // hsrc: the draw comes straight from the pinned host copy (no copy node in front of this kernel); block y = 1 then fills the device
    // copy the later kernels read -- every field except the ones block y = 0 derives below (disjoint stores, no ordering needed)
    // (LDS_ARR only: the long-table variant runs at its register limit)
    const bool from_host = LDS_ARR && hsrc != nullptr;
    const double v = reinterpret_cast<const double*>(from_host ? &hsrc[b] : &Pg)[t];
    reinterpret_cast<double*>(&Ps)[t] = v;
    if (from_host && blockIdx.y == 1) {
      const size_t o = (size_t)t * sizeof(double);
      const bool y0 = o == offsetof(DevParams, md_norm) || o == offsetof(DevParams, tpl_rate_norm) || o == offsetof(DevParams, l1pzp) ||
                      o == offsetof(DevParams, z_bad) || o == offsetof(DevParams, dl_sorted) || o == offsetof(DevParams, fR);
      if (!y0) reinterpret_cast<double*>(&Pg)[t] = v;
    }
  }
  __syncthreads();
  const DevParams& P = Ps;
  TS_INIT; TS(0);
  const int Tc = P.Tc, Tm = P.Tm;
  double* g_zt = zt_all + (size_t)b * TcMax;
  double* g_It = It_all + (size_t)b * TcMax;
  double* dLt = dLt_all + (size_t)b * TcMax;
  double* g_mg = mg_all + (size_t)b * TmMax;
  double* g_cdf = cdf_all + (size_t)b * TmMax;
  double* g_tmp = tmp_all + (size_t)b * (TcMax + TmMax) + (blockIdx.y == 0 ? 0 : TcMax);

  // [r4] LDS_ARR launches run TWO cosmology blocks per draw (grid (nb, 3)): both form the z nodes, the running integral and the dL table in their
  // own LDS (the same bits); block 0 (doA) stores them and goes on to what the sample stage waits for -- its direct-index table and the
  // node records --, block 2 (doB) to the rest: the per-draw flags, the injections' index table, fR.  One block did all of it in 17 us, 8 of them
  // in this tail (1024 threads on one CU are bound by instruction issue, 255 CUs stand idle).  Grid (nb, 2): block 0 does both (doA = doB).
  const bool split = LDS_ARR && gridDim.y == 3;
  const bool doA = blockIdx.y == 0, doB = split ? blockIdx.y == 2 : true;
  if (blockIdx.y != 1) {
    double* zt = LDS_ARR ? larr : g_zt;
    double* tmp = LDS_ARR ? larr + Tc : g_tmp;
    double* It = LDS_ARR ? larr + 2 * Tc : g_It;
    if (t == 0 && doA) {                            // rate constants (used by later kernels only)
      double g = P.r[0], k = P.r[1], zp = P.r[2], zmax = P.r[3];
      Pg.md_norm = 1. + pow(1. + zp, -g - k);                             // rate.py:114
      Pg.tpl_rate_norm = (pow(1. + zmax, g + 1.) - 1.) / (g + 1.);        // rate.py:105
      Pg.l1pzp = chm_log(1. + zp);
    }
    // zt = [0] U logspace(-10, log10 z_max, Tc-1); It = cumtrapz(1/E, zt)     cosmo.py:43-46
    const double lzmax = log10(P.z_max);
    for (int i = t; i < Tc; i += nt) {
      // plug-in cosmology (chm_tab): the caller's z_grid_interp; 1/E is not needed (the Jacobian and p_bkg come tabulated too)
      // [r4] zt_c: the nodes of this (z_max, Tc) as k_znodes formed them for an earlier call (the same expression: the same bits) -- the grid
      // depends on nothing else, and an H0 scan or a chain changes neither
      double z = tab_zt ? tab_zt[(size_t)b * Tc + i] : (zt_c ? zt_c[i] : znode_at(lzmax, Tc, i));
      zt[i] = z;
      // (a cosmological constant -- w0 = -1, wa = 0 -- needs neither w(z)'s quotient nor the power: the same sum with de = 1, the same bits)
      const double zp1 = 1. + z, z2 = zp1 * zp1;
      const double Ez = de_needs_log(P) ? E_at_z(P, z) : sqrt(P.Om0 * (z2 * zp1) + P.Or0 * (z2 * z2) + P.Ok0 * z2 + P.Ode0 * 1.);
      tmp[i] = tab_zt ? 0. : 1. / Ez;
    }
    if (LDS_ARR) __syncthreads(); else gsync();
    TS(1);
    block_cumtrapz(tmp, zt, It, Tc, sh);
    if (!LDS_ARR) gsync();
    TS(2);
    // dL table of z_from_dGW: dL_at_z(cosmo, z_grid_interp) (cosmo.py:263): jnp.interp evaluated AT its own nodes (below; the last
    // node is bracketed from the left: It[Tc-2] + (dx/dx) dI).  The values also go into `tmp` (free
    // after the cumulative integral) for the monotonicity check below.
    for (int i = t; i < Tc; i += nt) {
      double z = zt[i];
      // interp AT node i brackets it with node i+1:  It[i] + (0/dx) (It[i+1] - It[i])  -- It[i] when the next node is finite, NaN
      // when it is not (an unphysical draw: the dL table turns NaN one node before the integral does, as the reference's)
      double ii;
      if (i == Tc - 1) { double dx = zt[i] - zt[i - 1]; ii = It[i - 1] + (dx / dx) * (It[i] - It[i - 1]); }
      else ii = It[i] + (0. / (zt[i + 1] - zt[i])) * (It[i + 1] - It[i]);
      double dl = tab_dLt ? tab_dLt[(size_t)b * Tc + i] : dL_from_dCt(P, dCt_from_dCr(P, P.dH * ii), z);
      if (doA) dLt[i] = dl;
      tmp[i] = dl;
      if (LDS_ARR && doA) { g_zt[i] = z; g_It[i] = It[i]; }
    }
    if (LDS_ARR) __syncthreads(); else gsync();
    TS(3);
    // Everything that only reads the finished tables runs between TWO barriers (each of the ~14 it took before costs ~0.4 us with 16 waves):
    //   per-draw flags   z_bad, the last finite stretch of the cumulative integral (first non-finite node -> grid_is_poisoned), and
    //                    dl_sorted, whether the dL table is non-decreasing (-> z_from_dGW_x2)
    //   direct-index tables of the dL table (`tmp`) for the posterior samples (lutA) and the injections (lutB): one search per key
    //   node records of the fast sample stage: rec[i] = { dL_i, z_i, slope of z(dL) on [node i, node i+1], log(1 + z_i) }
    //   brackets of the completeness limits on the z table (fR = Vc(z1) - Vc(z0), completeness.py:54-58)
    {
      int bad = 0;
      for (int i = t; i < Tc; i += nt) {
        if (doB && !(fabs(It[i]) <= 1.7976931348623157e308)) atomicMin(&s_int[0], i);
        if (i > 0) bad |= ((tmp[i] < tmp[i - 1]) || (tmp[i] != tmp[i])) ? 1 : 0;
      }
      const int nkA = (doA && lutA.nk > 0) ? lutA.nk + 1 : 0, nkB = (doB && lutB.nk > 0) ? lutB.nk + 1 : 0;      // entries of each table (keys + 1) this block makes
      int* ksA = lut_ks; int* ksB = lut_ks + nkA;
      auto search = [&](const LutDesc& D, int* ks, int n) {
        unsigned short* lut = D.lut + (size_t)b * n;
        for (int q = t; q < n; q += nt) {
          const int c0 = searchsorted_right((const double*)tmp, Tc, lut_key_floor(D.key0 + q));
          lut[q] = (unsigned short)c0; ks[q] = c0;
        }
      };
      const bool lut_both = nkA + nkB <= LUT_MAXKEYS + 1;         // both tables' answers fit the scratch (else B follows A, below)
      if (nkA) search(lutA, ksA, nkA);
      if (nkB && lut_both) search(lutB, ksB, nkB);
      if (rec_all && doA) {
        double* rec = rec_all + (size_t)b * TcMax * 4;
        for (int i = t; i < Tc; i += nt) {
          const double x0 = tmp[i], f0 = zt[i];
          double sl = 0.;
          if (i + 1 < Tc) { const double dx = tmp[i + 1] - x0; sl = (fabs(dx) <= 4.930380657631324e-32) ? 0. : (zt[i + 1] - f0) / dx; }
          rec[4 * i] = x0; rec[4 * i + 1] = f0; rec[4 * i + 2] = sl; rec[4 * i + 3] = lz_c ? lz_c[i] : log1p(f0);
        }
      }
      if (doB) for (int i = t + 1; i < Tc; i += nt) {             // interval (i-1, i): zt[i-1] <= x < zt[i]  (searchsorted right), as block_interp
        if (zt[i - 1] <= P.zc0 && (P.zc0 < zt[i] || i == Tc - 1)) s_int[3] = i;
        if (zt[i - 1] <= P.zc1 && (P.zc1 < zt[i] || i == Tc - 1)) s_int[4] = i;
      }
      bad = __syncthreads_or(bad);                                // barrier 1
#ifdef CHM_TABLES_PROF
      const long long ts_b1 = wall_clock64();
#endif
      auto spans = [&](const int* ks, int n, int slot) {
        int lm = 0;
        for (int q = t; q + 1 < n; q += nt) lm = max(lm, ks[q + 1] - ks[q]);
        // [r4] one LDS atomic per wave (a thousand threads on one cell took 2.3 of the kernel's 17 us)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lm = max(lm, __shfl_xor(lm, o, 64));
        if ((t & 63) == 0 && lm > 0) atomicMax(&s_int[slot], lm);
      };
      if (nkA) spans(ksA, nkA, 1);
      if (nkB && lut_both) spans(ksB, nkB, 2);
      __syncthreads();                                            // barrier 2
#ifdef CHM_TABLES_PROF
      const long long ts_b2 = wall_clock64();
      if (t == 0 && b == 0) printf("[k_tables last phase, x10 ns] searches + records + brackets %lld | spans %lld\n", ts_b1 - ts_[3], ts_b2 - ts_b1);
#endif
      auto info = [&](const LutDesc& D, const int* ks, int n, int slot) {
        const int first = ks[0], last = ks[n - 1];
        const int i_lo = first > 0 ? first - 1 : 0, i_hi = last + 1 < Tc ? last + 1 : Tc - 1;
        int* o = D.info + (size_t)b * 4;
        o[0] = i_lo; o[1] = i_hi - i_lo + 1; o[2] = s_int[slot];
        o[3] = (!bad && i_hi - i_lo + 1 <= D.cap && Tc <= 65535) ? 1 : 0;
      };
      auto interp_at = [&](double x, int i) {                     // jnp.interp(x, zt, It) from the published bracket (block_interp's arithmetic)
        if (i < 1) i = 1;
        const double f0 = It[i - 1], f1 = It[i], x0 = zt[i - 1], dx = zt[i] - x0;
        double f = (fabs(dx) <= 4.930380657631324e-32) ? f0 : f0 + ((x - x0) / dx) * (f1 - f0);
        if (x < zt[0]) f = It[0];
        if (x > zt[Tc - 1]) f = It[Tc - 1];
        return f;
      };
      if (t == 0) {
        if (nkA) info(lutA, ksA, nkA, 1);
        if (nkB && lut_both) info(lutB, ksB, nkB, 2);
        if (doB) {
          const int j = s_int[0];
          Pg.z_bad = j < 0x7fffffff ? zt[j > 0 ? j - 1 : 0] : __builtin_inf();
          Pg.dl_sorted = bad ? 0. : 1.;
          const double v0 = Vc_from_dCt(P, dCt_from_dCr(P, P.dH * interp_at(P.zc0, s_int[3])));
          const double v1 = Vc_from_dCt(P, dCt_from_dCr(P, P.dH * interp_at(P.zc1, s_int[4])));
          Pg.fR = P.fR_given != 0. ? P.fR : v1 - v0;    // a plug-in completeness hands its own fR(cosmo) over (chm_tab.fR)
        }
      }
      if (nkB && !lut_both) {                                     // > 64 octaves of distances in all: the second table on its own
        __syncthreads();
        if (t == 0) s_int[2] = 0;
        search(lutB, lut_ks, nkB);
        __syncthreads();
        spans(lut_ks, nkB, 2);
        __syncthreads();
        if (t == 0) info(lutB, lut_ks, nkB, 2);
      }
    }
    TS(4); TS_PRINT;
#ifdef CHM_TABLES_FORCE_SCRATCH
    if (pad_[(t + b) % CHM_TABLES_FORCE_SCRATCH] == -1.) g_tmp[0] = pad_[t % CHM_TABLES_FORCE_SCRATCH];
#endif
  } else {
    double* mg = LDS_ARR ? larr : g_mg;
    double* tmp = LDS_ARR ? larr + Tm : g_tmp;
    double* cdf = LDS_ARR ? larr + 2 * Tm : g_cdf;
    // [r4] the transcendental pieces of the constants by the first lanes of FOUR waves side by side (one thread formed them one after the
    // other in 3 of the block's 12 us): sh[24..29]; thread 0 assembles them below -- the same calls on the same arguments, the same bits
    if (P.mass_model == 2) {
      const double mu = P.m[6], sg = P.m[7], hi = mu + 5. * sg;
      if (t == 128) sh[24] = erf((hi - mu) / (sg * sqrt(2.)));
      if (t == 192) sh[25] = erf((P.m[0] - mu) / (sg * sqrt(2.)));
      if (t == 256) sh[26] = -0.5 * chm_log(2. * CHM_PI) - chm_log(sg);
    }
    if (t == 320) { sh[27] = chm_log(P.m[0]); sh[28] = chm_log(P.m[1]); }
    if (t == 64) { sh[30] = log10(P.m[0]); sh[31] = log10(P.m[1]); }
    if (t == 0 && P.mass_model == 2) sh[29] = tpl_cdf(-P.m[3], P.m[0], P.m[1]);      // mass.py:301
    __syncthreads();
    if (t == 0) {                                   // mass-model constants: into the LDS copy (used below) and to global memory
      double m_low = P.m[0], m_high = P.m[1];
      if (P.mass_model == 2) {
        double mu = P.m[6], sg = P.m[7];
        Ps.plp_plnorm = sh[29];
        Ps.tg_hi = mu + 5. * sg;                                          // mass.py:302
        Ps.tg_norm = 0.5 * sh[24] - 0.5 * sh[25];                         // mass.py:272-274
        Ps.g_c0 = sh[26];                                                 // mass.py:268
        Ps.inv_plnorm = 1. / Ps.plp_plnorm; Ps.inv_tg_norm = 1. / Ps.tg_norm; Ps.inv_2s2 = 1. / (2. * (sg * sg));
        Pg.plp_plnorm = Ps.plp_plnorm; Pg.tg_hi = Ps.tg_hi; Pg.tg_norm = Ps.tg_norm; Pg.g_c0 = Ps.g_c0;
        Pg.inv_plnorm = Ps.inv_plnorm; Pg.inv_tg_norm = Ps.inv_tg_norm; Pg.inv_2s2 = Ps.inv_2s2;
      } else if (P.mass_model == 1) {
        double mb = m_low + P.m[6] * (m_high - m_low);                    // mass.py:291-293
        Ps.bpl_mbreak = mb;
        Ps.bpl_pl1 = tpl_notnorm(mb, -P.m[2], m_low, mb);
        Ps.bpl_pl2 = tpl_notnorm(mb, -P.m[3], mb, m_high);
        Pg.bpl_mbreak = Ps.bpl_mbreak; Pg.bpl_pl1 = Ps.bpl_pl1; Pg.bpl_pl2 = Ps.bpl_pl2;
      }
      Pg.lmg0 = sh[27];
      Pg.inv_dlmg = (double)(P.Tm - 1) / (sh[28] - sh[27]);
    }
    __syncthreads();
    TS(1);
    // m_grid = logspace(log10 m_low, log10 m_high, Tm); cdf = cumtrapz(secondary(m_grid; m_high))   mass.py:45-49
    const double l0 = sh[30], l1 = sh[31];                    // log10(m_low), log10(m_high): formed once by thread 1 above
    for (int i = t; i < Tm; i += nt) {
      double m = i == 0 ? P.mg_first : (i == Tm - 1 ? P.mg_last : chm_pow10(jnp_linspace_at(l0, l1, Tm, i)));   // end nodes: host libm (DevParams)
      mg[i] = m;
      tmp[i] = secondary_notnorm(P, m, P.m[1]);
    }
    if (LDS_ARR) __syncthreads(); else gsync();
    TS(2);
    block_cumtrapz(tmp, mg, cdf, Tm, sh);
    if (!LDS_ARR) gsync();
    TS(3);
    // norm_p_m1 = trapz(primary(m_grid), m_grid) = 0.5 * sum(dx * (y1 + y0))             mass.py:50-52
    for (int i = t; i < Tm; i += nt) {
      tmp[i] = primary_notnorm(P, mg[i]);
      if (LDS_ARR) { g_mg[i] = mg[i]; g_cdf[i] = cdf[i]; }
    }
    if (LDS_ARR) __syncthreads(); else gsync();
    TS(4);
    double acc = 0.;
    for (int k = t; k < Tm - 1; k += nt) acc += (mg[k + 1] - mg[k]) * (tmp[k + 1] + tmp[k]);
    acc = block_reduce<RED_SUM>(acc, sh);
    if (t == 0) { Pg.norm_p_m1 = 0.5 * acc; Pg.inv_norm_p_m1 = 1. / (0.5 * acc); Pg.cdf_last = cdf[Tm - 1]; }
    TS(5); TS_PRINT_M;
  }
}

// ------------------------------------------------------------------------------------------------------
// table staging: copy the per-draw tables into LDS when they fit, else read them from global memory
// ------------------------------------------------------------------------------------------------------
struct TabView { const double *zt, *It, *dLt, *mg, *cdf; };

DEVFN TabView stage_tables(const DevParams& P, const TablePtrs& g, bool use_lds, double* lds, bool need_It) {
  TabView v;
  if (!use_lds) { v.zt = g.zt; v.It = g.It; v.dLt = g.dLt; v.mg = g.mg; v.cdf = g.cdf; return v; }
  int Tc = P.Tc, Tm = P.Tm;
  double* zt = lds; double* dLt = zt + Tc; double* mg = dLt + Tc; double* cdf = mg + Tm; double* It = cdf + Tm;
  // All loads of a round are issued before the first LDS store (the plain copy loop waited for every load in turn: ten dependent
  // memory round trips, 5 us per block): eight entries per thread of each of the four arrays per round.
  const int nt = blockDim.x, t = threadIdx.x;
  const int nmax = Tc > Tm ? Tc : Tm;
  for (int base = 0; base < nmax; base += 8 * nt) {
    double a[8], bq[8], c[8], d[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = base + k * nt + t;
      a[k] = i < Tc ? g.zt[i] : 0.; bq[k] = i < Tc ? g.dLt[i] : 0.;
      c[k] = i < Tm ? g.mg[i] : 0.; d[k] = i < Tm ? g.cdf[i] : 0.;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = base + k * nt + t;
      if (i < Tc) { zt[i] = a[k]; dLt[i] = bq[k]; }
      if (i < Tm) { mg[i] = c[k]; cdf[i] = d[k]; }
    }
  }
  if (need_It) for (int i = t; i < Tc; i += nt) It[i] = g.It[i];
  __syncthreads();
  v.zt = zt; v.It = It; v.dLt = dLt; v.mg = mg; v.cdf = cdf;
  return v;
}

// ------------------------------------------------------------------------------------------------------
// k_samples: blocks of 256 threads stage the per-draw tables in LDS once and then walk over chunks of SAMPLE_CHUNK
// samples (chunk = blockIdx.x, += gridDim.x); one set of partial statistics per (event, chunk)
// ------------------------------------------------------------------------------------------------------
// all-in-one block reduction of NV values: sums for v[0..NS), then one min (v[NS]) and one max (v[NS+1])
template <int NS>
DEVFN void block_reduce_stats(double* v, double* scratch /* 4 x (NS+2) */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int i = 0; i < NS; i++) v[i] = wave_sum_dpp(v[i]);                     // every lane of the block is active here
  v[NS] = wave_min_dpp(v[NS]); v[NS + 1] = wave_max_dpp(v[NS + 1]);         // NaN-ignoring; the caller restores NaN from the sums
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NS + 2; i++) scratch[wid * (NS + 2) + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NS; i++) { double r = 0.; for (int w = 0; w < nw; w++) r += scratch[w * (NS + 2) + i]; v[i] = r; }
  double mn = scratch[NS], mx = scratch[NS + 1];
  for (int w = 1; w < nw; w++) { mn = nanmin2(mn, scratch[w * (NS + 2) + NS]); mx = nanmax2(mx, scratch[w * (NS + 2) + NS + 1]); }
  v[NS] = mn; v[NS + 1] = mx;
}

// CHM_PHASE_PROF (diagnostic builds only, scripts/phase_prof.py): shader-clock cycles between phase marks of k_kde_marg_sub (g_phase) and
// k_samples (g_phase_s), summed over every 64th block's first wave ([7] counts them)
// CHM_CLOCK_STAMP builds (with CHM_PROBE): the shader clock INSIDE the two hot kernels -- every 64th block's first lane adds the core-clock cycles
// (s_memtime) and the 10 ns ticks of the constant 100 MHz counter (s_memrealtime) between the start and the end of its body call
// (scripts/clock_under_load.py).  The four scalar registers held across the body cost the kernels spilled registers (2 / 21 vector registers):
// such a build tells the clock, not the time -- the ceilings come from the plain CHM_PROBE build.
#ifdef CHM_CLOCK_STAMP
__device__ unsigned long long g_clk[4];                    // GW kernel: cycles, ticks; sample stage: cycles, ticks
#define CLK_BEGIN const unsigned long long clk_c0 = clock64(), clk_r0 = wall_clock64()
#define CLK_END(i) do { if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { atomicAdd(&g_clk[i], clock64() - clk_c0); atomicAdd(&g_clk[i + 1], wall_clock64() - clk_r0); } } while (0)
#else
#define CLK_BEGIN
#define CLK_END(i)
#endif
#ifdef CHM_PHASE_PROF
__device__ unsigned long long g_phase[8], g_phase_s[8];
#define PH_INIT unsigned long long ph_prev = clock64(); const bool ph_on = threadIdx.x == 0 && (blockIdx.x & 63) == 0; if (ph_on) atomicAdd(&g_phase[7], 1ull)
#define PHS_INIT unsigned long long ph_prev = clock64(); const bool ph_on = threadIdx.x == 0 && (blockIdx.x & 63) == 0; if (ph_on) atomicAdd(&g_phase_s[7], 1ull)
#define PHS(i) do { __builtin_amdgcn_s_waitcnt(0); unsigned long long ph_t = clock64(); if (ph_on) atomicAdd(&g_phase_s[i], ph_t - ph_prev); ph_prev = clock64(); } while (0)
#define PH(i) do { __builtin_amdgcn_s_waitcnt(0); unsigned long long ph_t = clock64(); if (ph_on) atomicAdd(&g_phase[i], ph_t - ph_prev); ph_prev = clock64(); } while (0)
// [r5] phase marks of the standard GW kernel (scripts/phase_gw.py): the clock travels into kde_sub_item through two extra arguments
#define PHG(i) do { if (ph_prev_p) { __builtin_amdgcn_s_waitcnt(0); unsigned long long ph_t = clock64(); if (ph_on_) atomicAdd(&g_phase[i], ph_t - *ph_prev_p); *ph_prev_p = clock64(); } } while (0)
#else
#define PH_INIT
#define PH(i)
#define PHG(i)
#define PHS_INIT
#define PHS(i)
#endif

template <bool LDS_TAB, bool FULL>
__global__ void __launch_bounds__(64 * SAMPLE_WPB) k_samples(LikeDev L, const DevParams* params, const double* zt_all, const double* It_all,
                                                  const double* dLt_all, const double* mg_all, const double* cdf_all,
                                                  int TcMax, int TmMax) {
#pragma clang fp contract(fast)                  // sums of products may fuse; z comes from jnp_interp_x2 (contract off) untouched
  extern __shared__ double lds[];
  __shared__ double red[4 * 16];
  __shared__ int ired[8];
  const int b = blockIdx.x % L.nb, bx = blockIdx.x / L.nb, nbx = gridDim.x / L.nb, t = threadIdx.x;
  const DevParams P = params[b];      // by value: uniform loads at kernel start -> scalar registers, nothing re-read in the loops
  TablePtrs g = { zt_all + (size_t)b * TcMax, It_all + (size_t)b * TcMax, dLt_all + (size_t)b * TcMax,
                  mg_all + (size_t)b * TmMax, cdf_all + (size_t)b * TmMax };
  PHS_INIT;
  TabView T = stage_tables(P, g, LDS_TAB, lds, false);
  PHS(0);                                                   // tables staged
  const int S = L.S;
  const int NCH = L.NC / SAMPLE_WPB;                        // chunks per event (L.NC counts the partial records: one per wave and chunk)
  const int nchunk = L.E_cnt * NCH;
  const bool vec2 = ((S & 1) == 0);
  for (int ch = bx; ch < nchunk; ch += nbx) {
    const int e = L.e_off + ch / NCH, c = ch % NCH;
    const size_t so = ((size_t)b * L.E + e) * S;
    const size_t eo = (size_t)e * S;
    double* wz = L.ws_z + so;
    double* ww = L.ws_w + so;
    // Bracket of the event's distances on the (sorted) dL table: every sample's  #entries <= dL  lies in [c_lo, c_hi], so its halving
    // search runs over c_hi - c_lo entries (~7 steps) instead of the table (11 steps).  Each wave finds the two ends by its own
    // (uniform) binary searches -- no block barrier anywhere in the chunk loop, the waves of a block drift freely.
    // Reference point of the shifted sums: the table node at the lower end of the bracket (any value near the event's z serves; it
    // travels with the partial records); without a bracket, the event's first sample.
    int s_base = 0, s_len = P.Tc;
    double z_ref;
    if (P.dl_sorted != 0. && L.dl_lo) {
      const double xlo = L.dl_lo[e], xhi = L.dl_hi[e];
      const int c_lo = searchsorted_right(T.dLt, P.Tc, xlo), c_hi = searchsorted_right(T.dLt, P.Tc, xhi);
      if (xlo == xlo && xhi == xhi && c_hi >= c_lo) { s_base = c_lo; s_len = c_hi - c_lo; }
      z_ref = T.zt[c_lo < P.Tc ? c_lo : P.Tc - 1];
    } else z_ref = jnp_interp(L.dL[eo], T.dLt, T.zt, P.Tc, false, 0., 0.);
    PHS(1);                                                 // reference point, bracket of the event on the table
    const double ra_ref = FULL ? L.ra[eo] : 0., dec_ref = FULL ? L.dec[eo] : 0.;
    double v[6] = { 0., 0., 0., 0., __builtin_inf(), -__builtin_inf() };     // sw, sw2, sd1, sd2, zmin, zmax
    double m[9] = { 0., 0., 0., 0., 0., 0., 0., 0., 0. };
    const int s_end = min(S, (c + 1) * SAMPLE_CHUNK);
    // 512 threads x 2 consecutive samples (16 B per lane per array) per pass
#pragma unroll 1
    for (int s = c * SAMPLE_CHUNK + 2 * t; s < s_end; s += 128 * SAMPLE_WPB) {
      double dl[2], md1[2], md2[2], ipr[2], l1[2], l2[2];
      if (vec2) {                                 // s even, S even -> s + 1 < s_end and 16-byte aligned
        double2 a = *reinterpret_cast<const double2*>(L.dL + eo + s), bb = *reinterpret_cast<const double2*>(L.m1det + eo + s);
        double2 cc = *reinterpret_cast<const double2*>(L.m2det + eo + s), dd = *reinterpret_cast<const double2*>(L.pe_prior + eo + s);
        double2 ee = *reinterpret_cast<const double2*>(L.lm1det + eo + s), ff = *reinterpret_cast<const double2*>(L.lm2det + eo + s);
        dl[0] = a.x; dl[1] = a.y; md1[0] = bb.x; md1[1] = bb.y; md2[0] = cc.x; md2[1] = cc.y; ipr[0] = dd.x; ipr[1] = dd.y;
        l1[0] = ee.x; l1[1] = ee.y; l2[0] = ff.x; l2[1] = ff.y;
      } else {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const bool in = s + h < s_end;
          dl[h] = in ? L.dL[eo + s + h] : 1.; md1[h] = in ? L.m1det[eo + s + h] : 1.;
          md2[h] = in ? L.m2det[eo + s + h] : 1.; ipr[h] = in ? L.pe_prior[eo + s + h] : 1.;
          l1[h] = in ? L.lm1det[eo + s + h] : 0.; l2[h] = in ? L.lm2det[eo + s + h] : 0.;
        }
      }
      double zz[2], wv[2];
      // z = z_from_dGW(dL) (cosmo.py:260-264) for the two samples together
      if (P.dl_sorted != 0.) jnp_interp_x2_range(dl[0], dl[1], T.dLt, T.zt, P.Tc, s_base, s_len, zz[0], zz[1]);
      else z_from_dGW_x2(P, dl[0], dl[1], T.dLt, T.zt, zz[0], zz[1]);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        // m_src = m_det/(1+z) (pop_wrapper.py:70); w = p_m1m2 / pe_prior (pop_wrapper.py:79; the device array holds 1/pe_prior)
        double z = zz[h];
        double zp1 = 1. + z;
        double r = chm_div(1., zp1);
        double m1 = md1[h] * r, m2 = md2[h] * r;
        double lz = chm_log_pos(zp1);                             // log(m_src) = log(m_det) - log(1+z): one log for both masses
        double pm;
        if (L.tab_pm) {                                       // plug-in mass model: p_m1m2 tabulated by the caller (original sample order)
          const int si = s + h < s_end ? s + h : s;
          pm = L.tab_pm[so + (L.perm ? L.perm[eo + si] : si)];
        } else pm = p_m1m2_fused(P, m1, m2, l1[h] - lz, l2[h] - lz, T.mg, T.cdf);
        double w = pm * ipr[h];
        wv[h] = w;
        if (s + h < s_end) {
          double d = z - z_ref;
          v[0] += w; v[1] += w * w; v[2] += d; v[3] += d * d;
          v[4] = vmin_f64(v[4], z); v[5] = vmax_f64(v[5], z);                   // a NaN z is caught through sum(d) below
          if (FULL) {                             // un-normalised weighted moments of (z, ra, dec) about the reference
            double d1 = L.ra[eo + s + h] - ra_ref, d2 = L.dec[eo + s + h] - dec_ref;
            m[0] += w * d; m[1] += w * d1; m[2] += w * d2;
            m[3] += w * d * d; m[4] += w * d * d1; m[5] += w * d * d2; m[6] += w * d1 * d1; m[7] += w * d1 * d2; m[8] += w * d2 * d2;
          }
        }
      }
      if (vec2) {
        *reinterpret_cast<double2*>(wz + s) = make_double2(zz[0], zz[1]);
        *reinterpret_cast<double2*>(ww + s) = make_double2(wv[0], wv[1]);
      } else {
        if (s < s_end) { wz[s] = zz[0]; ww[s] = wv[0]; }
        if (s + 1 < s_end) { wz[s + 1] = zz[1]; ww[s + 1] = wv[1]; }
      }
    }
    PHS(2);                                                 // sample loop
    // One partial record per WAVE and chunk (no block barrier: the waves of a block finish their samples at different times and would
    // wait for the slowest twice); combine_stats adds them up.  Every lane is active here (the loop bounds are uniform per wave pair).
    double* q = L.part + (((size_t)b * L.E + e) * L.NC + (size_t)c * SAMPLE_WPB + (t >> 6)) * NPART;
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = wave_sum_dpp(v[i]);
    v[4] = wave_min_dpp(v[4]); v[5] = wave_max_dpp(v[5]);     // NaN-ignoring; NaN restored from the sums:
    if (v[2] != v[2]) { v[4] = v[2]; v[5] = v[2]; }         // jnp.min / jnp.max propagate NaN (any NaN z makes sum(d) NaN)
    if ((t & 63) == 0) { q[PT_SW] = v[0]; q[PT_SW2] = v[1]; q[PT_SD1] = v[2]; q[PT_SD2] = v[3]; q[PT_ZMIN] = v[4]; q[PT_ZMAX] = v[5]; q[PT_ZREF] = z_ref; }
    PHS(3);                                                 // wave reductions
    if (FULL) {
#pragma unroll
      for (int i = 0; i < 9; i++) m[i] = wave_sum_dpp(m[i]);
      if ((t & 63) == 0) { q[PT_WD0] = m[0]; q[PT_WD1] = m[1]; q[PT_WD2] = m[2]; q[PT_W00] = m[3]; q[PT_W01] = m[4]; q[PT_W02] = m[5];
                           q[PT_W11] = m[6]; q[PT_W12] = m[7]; q[PT_W22] = m[8]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// k_samples_fast<MASS, FULL>: the sample stage for the built-in mass models with the per-draw tables in LDS -- same quantities,
// same partial records as k_samples, fewer instructions per sample:
//   * the mass model is a template parameter (one instantiation per model: no run-time model branches, the other models'
//     parameters never reach the scalar registers);
//   * z_from_dGW: the searchsorted answer comes from the draw's direct-index table (LutDesc: one LDS read + `lmax` probes, 1 for
//     the reference's table) instead of a ~7-step binary search; the interpolation itself is jnp.interp's, bit for bit;
//   * only the slice of the (dL, z) table the shard's distances can reach is staged in LDS (info[0..1]); a draw whose table is
//     not sorted or whose slice does not fit takes the table searches of k_samples on the global tables (`slow`);
//   * the six per-sample inputs come from one tile array (E, NT, 6, 128): one base address per pass, 16 B per lane per array.
// ------------------------------------------------------------------------------------------------------
#define SF_TILE 128
struct SampFast {
  const double* tiles;            // (E, NT, 6, SF_TILE): dL, m1det, m2det, 1/pe_prior, log m1det, log m2det of SF_TILE consecutive samples
  int NT, pad;
  LutDesc lut;
};

// log(1 + z) from the node record below z: 1 + z_0 = (1 + z)(1 - v), v = (z - z_0)/(1 + z) in [0, 0.017] for the reference's
// logspace nodes, so log(1 + z) = log(1 + z_0) + v + v^2/2 + ... (8 terms: remainder < 1e-17 relative) -- 10 instructions, no log.
// v > 0.02 (a coarser user table) or a lane without a record: chm_log_pos.
DEVFN double log1pz_from_node(double z, double z0, double lz0, double r, double& v) {
  v = (z - z0) * r;
  double p = 0.125;
  p = FM_FMA(p, v, 1. / 7.); p = FM_FMA(p, v, 1. / 6.); p = FM_FMA(p, v, 0.2); p = FM_FMA(p, v, 0.25);
  p = FM_FMA(p, v, 1. / 3.); p = __builtin_fma(p, v, 0.5); p = __builtin_fma(p, v, 1.0);
  return __builtin_fma(p, v, lz0);
}

// z = z_from_dGW(x) (cosmo.py:260-264) for two distances: searchsorted from the direct-index table (lut: one LDS read, then `lmax`
// probes), then the interpolant of jnp.interp in slope form, z_0 + (x - x_0) s with s = dz/dx of the bracketing interval formed
// once per draw (k_tables) -- the reference's z_0 + ((x - x_0)/dx) dz to <= 2e-18 relative (the increment is < 1.7 % of z).
// rec: LDS slice [i_lo, i_lo + ns) of the node records.  Returns z and the record (z_0, log(1 + z_0)) used; lanes whose key is
// outside the table (NaN, <= 0, inf) report bad.
DEVFN void z_from_lut_x2(double xa, double xb, const double* rec, const unsigned short* luts, int key0, int nk,
                         int i_lo, int ns, int lmax, int Tc, double x_last, double z_last,
                         double& za, double& zb, double& z0a, double& z0b, double& lz0a, double& lz0b, bool& bad) {
  const int ka = lut_key(xa, key0), kb = lut_key(xb, key0);
  const bool oka = (unsigned)ka < (unsigned)nk, okb = (unsigned)kb < (unsigned)nk;
  int pa = luts[oka ? ka : 0], pb = luts[okb ? kb : 0];
  const int i_hi = i_lo + ns - 1;
  // entries before lut[k] are <= x, entries from lut[k] + lmax on are > x: halving over [p, p + lmax), probes past the slice count as +inf
  for (int l = lmax; l > 1;) {
    const int half = l >> 1;
    const int ia = pa + half - 1, ib = pb + half - 1;
    const double va = rec[4 * ((ia < i_hi ? ia : i_hi) - i_lo)], vb = rec[4 * ((ib < i_hi ? ib : i_hi) - i_lo)];
    pa += (ia <= i_hi && va <= xa) ? half : 0;
    pb += (ib <= i_hi && vb <= xb) ? half : 0;
    l -= half;
  }
  if (lmax > 0) {
    const double va = rec[4 * ((pa < i_hi ? pa : i_hi) - i_lo)], vb = rec[4 * ((pb < i_hi ? pb : i_hi) - i_lo)];
    pa += (pa <= i_hi && va <= xa) ? 1 : 0;
    pb += (pb <= i_hi && vb <= xb) ? 1 : 0;
  }
  const int ja = (pa < 1 ? 1 : (pa > Tc - 1 ? Tc - 1 : pa)) - 1 - i_lo, jb = (pb < 1 ? 1 : (pb > Tc - 1 ? Tc - 1 : pb)) - 1 - i_lo;
  const double4 ra = *reinterpret_cast<const double4*>(rec + 4 * ja), rb = *reinterpret_cast<const double4*>(rec + 4 * jb);
  // jnp.interp clamps to fp[-1] beyond the last node (x < xp[0] = dL(z = 0) = 0 cannot occur for a valid key): the table is sorted here
  // (`fits`), so the interpolant of the last interval exceeds z_last exactly for x > x_last -- one v_min_f64 instead of a compare and two selects
  za = vmin_f64(__builtin_fma(xa - ra.x, ra.z, ra.y), z_last);
  zb = vmin_f64(__builtin_fma(xb - rb.x, rb.z, rb.y), z_last);
  z0a = ra.y; z0b = rb.y; lz0a = ra.w; lz0b = rb.w;
  // (ONE compare for "a key outside the table": the wave votes on it in the per-sample loops, and a vote on a disjunction makes the compiler rebuild the
  //  lane mask through a VGPR -- v_cndmask + v_cmp in front of every vote)
  const unsigned kmx = (unsigned)ka > (unsigned)kb ? (unsigned)ka : (unsigned)kb;
  bad = kmx >= (unsigned)nk;
}

// [r3] log(m1det), log(m2det) of every posterior sample, formed on the device at upload with chm_log -- the function the few-draw variant of
// k_samples_fast applies to the masses themselves instead of reading these two planes (a quarter of its HBM traffic): the two variants see
// the same bits, a batched call stays the scalar call bit for bit.
__global__ void __launch_bounds__(256) k_fill_logs(const double* m1, const double* m2, double* l1, double* l2, double* tiles, int E, int S, int NTl) {
  const size_t n = (size_t)E * S;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double a = chm_log(m1[i]), b = chm_log(m2[i]);
    l1[i] = a; l2[i] = b;
    const size_t e = i / S, k = i % S;
    double* o = tiles + ((e * NTl + k / SF_TILE) * 6) * SF_TILE + k % SF_TILE;
    o[4 * SF_TILE] = a; o[5 * SF_TILE] = b;
  }
}

// build-time knobs of the fast sample stage (A/B through scripts/build_variant.sh): waves per block (each chunk always has SAMPLE_WPB
// partial records: with fewer waves the surplus records are written neutral), waves per SIMD the register budget is cut for
#ifndef CHM_SF_WAVES
#define CHM_SF_WAVES 4
#endif
#ifndef CHM_SF_MINW
#define CHM_SF_MINW 4
#endif
static_assert(SAMPLE_WPB % CHM_SF_WAVES == 0, "records per chunk must be a multiple of the waves per block");
// (block bx of nbx of draw b)
template <int MASS, bool FULL, bool NT>
DEVFN void samples_fast_body(const LikeDev& L, const SampFast& F, const DevParams* params, const double* zt_all,
                             const double* dLt_all, const double* mg_all, const double* cdf_all,
                             const double* rec_all, int TcMax, int TmMax, const int b, const int bx, const int nbx, double* lds) {
#pragma clang fp contract(fast)                  // sums of products may fuse; z comes from z_from_lut_x2 / jnp_interp (contract off) untouched
  constexpr int NT_ = 64 * CHM_SF_WAVES;
  const int t = threadIdx.x, lane = t & 63;
  DevParams P = params[b];            // by value: uniform loads at kernel start, nothing re-read in the loops
#ifndef CHM_SF_NPV
#define CHM_SF_NPV 4
#endif
  mass_params_to_vgpr<MASS, FULL ? 0 : CHM_SF_NPV>(P);       // (mass-model parameters in vector registers: the scalar file cannot hold the whole draw)
  const double* g_zt = zt_all + (size_t)b * TcMax;
  const double* g_dLt = dLt_all + (size_t)b * TcMax;
  const int Tc = P.Tc, Tm = P.Tm;
  const int* info = F.lut.info + (size_t)b * 4;
  const int i_lo = info[0], ns = info[1], lmax = info[2];
  const bool fits = info[3] != 0;
  const int key0 = F.lut.key0, nk = F.lut.nk, cap = F.lut.cap;
  // LDS: node records of the table slice [cap x 4], m_grid [Tm], cdf_m2 [Tm], direct-index table [nk + 1] (u16)
  // [r3] + the 256-entry table of the mass model's exps first (chm_exp_tab: 13 instead of 17 VALU instructions per exp, four exps per sample)
  double* etab = lds;
  double* rec = lds + CHM_EXPTAB_N; double* mg = rec + 4 * (size_t)cap; double* cdf = mg + Tm;
  unsigned short* luts = reinterpret_cast<unsigned short*>(cdf + Tm);
  const ExpTab ex = { etab };
  {
    const double* gm = mg_all + (size_t)b * TmMax;
    const double* gc = cdf_all + (size_t)b * TmMax;
    const unsigned short* gl = F.lut.lut + (size_t)b * (nk + 1);
    for (int i = t; i < CHM_EXPTAB_N; i += NT_) etab[i] = exp_table_entry(i);
    for (int i = t; i < Tm; i += NT_) { mg[i] = gm[i]; cdf[i] = gc[i]; }
    if (fits) {
      const double2* gr = reinterpret_cast<const double2*>(rec_all + ((size_t)b * TcMax + i_lo) * 4);
      double2* lr = reinterpret_cast<double2*>(rec);
      for (int i = t; i < 2 * ns; i += NT_) lr[i] = gr[i];
      for (int i = t; i <= nk; i += NT_) luts[i] = gl[i];
    }
  }
  const double x_last = g_dLt[Tc - 1], z_last = g_zt[Tc - 1];
  __syncthreads();
  const int S = L.S;
  const int NCH = L.NC / SAMPLE_WPB;                        // chunks per event
  const int nchunk = L.E_cnt * NCH;
  for (int ch = bx; ch < nchunk; ch += nbx) {
    const int e = L.e_off + ch / NCH, c = ch % NCH;
    const size_t so = ((size_t)b * L.E + e) * S;
    const size_t eo = (size_t)e * S;
    double* wz = L.ws_z + so;
    double* ww = L.ws_w + so;
    // reference point of the shifted sums (any value near the event's z, the same for every chunk of the event): the table node
    // next to the event's smallest distance
    double z_ref;
    {
      const double xlo = L.dl_lo[e];
      const int k = lut_key(xlo, key0);
      if (fits) { const int q = ((unsigned)k < (unsigned)nk ? (int)luts[k] : i_lo) - i_lo; z_ref = rec[4 * (q < 0 ? 0 : (q > ns - 1 ? ns - 1 : q)) + 1]; }
      else { const int c_lo = (P.dl_sorted != 0. && xlo == xlo) ? searchsorted_right(g_dLt, Tc, xlo) : 0; z_ref = g_zt[c_lo < Tc ? c_lo : Tc - 1]; }
    }
    const double ra_ref = FULL ? L.ra[eo] : 0., dec_ref = FULL ? L.dec[eo] : 0.;
    double v[6] = { 0., 0., 0., 0., __builtin_inf(), -__builtin_inf() };     // sw, sw2, sd1, sd2, zmin, zmax
    double m[9] = { 0., 0., 0., 0., 0., 0., 0., 0., 0. };
    const int s_end = min(S, (c + 1) * SAMPLE_CHUNK);
    const double2* tbase = reinterpret_cast<const double2*>(F.tiles + (size_t)e * F.NT * 6 * SF_TILE) + lane;
    // the pass loop exists twice: the LUT form (the hot one) and the general searches (draws whose table slice is not in LDS)
    auto passes = [&](auto fits_tag) {
    constexpr bool FITS = decltype(fits_tag)::value;
    double2 a, bb, cc, dd, ee, ff;
    auto load_tile = [&](int s_, double2& a_, double2& b_, double2& c_, double2& d_, double2& e_, double2& f_) {
      const double2* tp = tbase + (size_t)(s_ / SF_TILE) * (6 * SF_TILE / 2);
      // NT (few draws per call): every tile is read once per call -- streamed past the caches, so that the z / w written below stay in the
      // memory-side cache for the GW kernel; with many draws per call the tiles are shared by the draws' blocks and stay cacheable
      auto ld = [&](const double2* q) { if (NT) { double2 v; v.x = __builtin_nontemporal_load(&q->x); v.y = __builtin_nontemporal_load(&q->y); return v; } return *q; };
      a_ = ld(tp); b_ = ld(tp + SF_TILE / 2); c_ = ld(tp + 2 * SF_TILE / 2); d_ = ld(tp + 3 * SF_TILE / 2);
      e_ = ld(tp + 4 * SF_TILE / 2); f_ = ld(tp + 5 * SF_TILE / 2);
    };
    const int s_first = c * SAMPLE_CHUNK + 2 * t;
#pragma unroll 1
    for (int s = s_first; s < s_end; s += 2 * NT_) {      // one tile of 128 samples per wave and pass
      load_tile(s, a, bb, cc, dd, ee, ff);
      const double dl[2] = { a.x, a.y }, md1[2] = { bb.x, bb.y }, md2[2] = { cc.x, cc.y }, ipr[2] = { dd.x, dd.y };
      const double l1[2] = { ee.x, ee.y }, l2[2] = { ff.x, ff.y };
      double zz[2], wv[2], z0[2] = { 0., 0. }, lz0[2] = { 0., 0. };
      bool bad = true, anybad = true;
      if (FITS) {                                     // z = z_from_dGW(dL) (cosmo.py:260-264)
        z_from_lut_x2(dl[0], dl[1], rec, luts, key0, nk, i_lo, ns, lmax, Tc, x_last, z_last, zz[0], zz[1], z0[0], z0[1], lz0[0], lz0[1], bad);
        anybad = wave_any(bad);
        if (anybad) {                                    // NaN / non-positive / infinite distances: the plain search on the global tables
          if (bad) { zz[0] = jnp_interp(dl[0], g_dLt, g_zt, Tc, false, 0., 0.); zz[1] = jnp_interp(dl[1], g_dLt, g_zt, Tc, false, 0., 0.); }
        }
      } else z_from_dGW_x2(P, dl[0], dl[1], g_dLt, g_zt, zz[0], zz[1]);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        // m_src = m_det/(1+z) (pop_wrapper.py:70); w = p_m1m2 / pe_prior (pop_wrapper.py:79; the tile holds 1/pe_prior)
        const double z = zz[h];
        const double zp1 = 1. + z;
        const double r = chm_rcp(zp1);
        const double m1 = md1[h] * r, m2 = md2[h] * r;
        // log(m_src) = log(m_det) - log(1+z): one log for both masses, from the node record when there is one
        double lz;
        if (FITS) {
          double v;
          lz = log1pz_from_node(z, z0[h], lz0[h], r, v);
          if (anybad || wave_any(!(v <= 0.02))) { if (bad || !(v <= 0.02)) lz = chm_log_pos(zp1); }      // (votes on single compares, see z_from_lut_x2)
        } else lz = chm_log_pos(zp1);
        const double w = p_m1m2_fused<MASS>(P, m1, m2, l1[h] - lz, l2[h] - lz, mg, cdf, ex) * ipr[h];
        wv[h] = w;
        if (s + h < s_end) {
          const double d = z - z_ref;
          v[0] += w; v[1] += w * w; v[2] += d; v[3] += d * d;
          v[4] = vmin_f64(v[4], z); v[5] = vmax_f64(v[5], z);                   // a NaN z is caught through sum(d) below
          if (FULL) {                             // un-normalised weighted moments of (z, ra, dec) about the reference
            const double d1 = L.ra[eo + s + h] - ra_ref, d2 = L.dec[eo + s + h] - dec_ref;
            m[0] += w * d; m[1] += w * d1; m[2] += w * d2;
            m[3] += w * d * d; m[4] += w * d * d1; m[5] += w * d * d2; m[6] += w * d1 * d1; m[7] += w * d1 * d2; m[8] += w * d2 * d2;
          }
        }
      }
      if ((S & 1) == 0) {                             // s even, S even: s + 1 < s_end, 16-byte aligned
        if (!NT && L.zw_stream) {
          // [r6] many draws per call: the (z, w) of the launch -- 8.4 GB at C3, read once by the GW kernel -- are stored with the streaming hint, so
          // that they do not displace the tiles and tables the draws' blocks share in L2 (stage -2.5 %, step -1.7 %; `sc1` / `sc0 sc1`
          // write-through: no gain; profiles/r06/ab_memory_path_r06.txt).  Few draws per call, or a launch whose (z, w) fit the memory-side
          // cache: plain stores, the GW kernel finds them there.
          typedef double d2_t __attribute__((ext_vector_type(2)));
          const d2_t vz = { zz[0], zz[1] }, vw = { wv[0], wv[1] };
          __builtin_nontemporal_store(vz, reinterpret_cast<d2_t*>(wz + s));
          __builtin_nontemporal_store(vw, reinterpret_cast<d2_t*>(ww + s));
        } else {
          *reinterpret_cast<double2*>(wz + s) = make_double2(zz[0], zz[1]);
          *reinterpret_cast<double2*>(ww + s) = make_double2(wv[0], wv[1]);
        }
      } else {
        wz[s] = zz[0]; ww[s] = wv[0];
        if (s + 1 < s_end) { wz[s + 1] = zz[1]; ww[s + 1] = wv[1]; }
      }
    }
    };
    if (fits) passes(std::true_type{}); else passes(std::false_type{});
    // one partial record per WAVE and chunk (as k_samples)
    double* q = L.part + (((size_t)b * L.E + e) * L.NC + (size_t)c * SAMPLE_WPB + (t >> 6)) * NPART;
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = wave_sum_dpp(v[i]);
    v[4] = wave_min_dpp(v[4]); v[5] = wave_max_dpp(v[5]);     // NaN-ignoring; NaN restored from the sums:
    if (v[2] != v[2]) { v[4] = v[2]; v[5] = v[2]; }         // jnp.min / jnp.max propagate NaN (any NaN z makes sum(d) NaN)
    if (lane == 0) { q[PT_SW] = v[0]; q[PT_SW2] = v[1]; q[PT_SD1] = v[2]; q[PT_SD2] = v[3]; q[PT_ZMIN] = v[4]; q[PT_ZMAX] = v[5]; q[PT_ZREF] = z_ref; }
    if (FULL) {
#pragma unroll
      for (int i = 0; i < 9; i++) m[i] = wave_sum_dpp(m[i]);
      if (lane == 0) { q[PT_WD0] = m[0]; q[PT_WD1] = m[1]; q[PT_WD2] = m[2]; q[PT_W00] = m[3]; q[PT_W01] = m[4]; q[PT_W02] = m[5];
                       q[PT_W11] = m[6]; q[PT_W12] = m[7]; q[PT_W22] = m[8]; }
    }
    if (CHM_SF_WAVES < SAMPLE_WPB && lane < NPART) {      // the chunk's surplus records: neutral for combine_stats (sums 0, min +inf, max -inf)
      for (int r = (t >> 6) + CHM_SF_WAVES; r < SAMPLE_WPB; r += CHM_SF_WAVES) {
        double* qn = L.part + (((size_t)b * L.E + e) * L.NC + (size_t)c * SAMPLE_WPB + r) * NPART;
        qn[lane] = lane == PT_ZMIN ? __builtin_inf() : (lane == PT_ZMAX ? -__builtin_inf() : (lane == PT_ZREF ? z_ref : 0.));
      }
    }
  }
}
template <int MASS, bool FULL, bool NT = false>
__global__ void __launch_bounds__(64 * CHM_SF_WAVES, CHM_SF_MINW) k_samples_fast(LikeDev L, SampFast F, const DevParams* params, const double* zt_all,
                                                                    const double* dLt_all, const double* mg_all, const double* cdf_all,
                                                                    const double* rec_all, int TcMax, int TmMax) {
  extern __shared__ double lds[];
  CLK_BEGIN;
  samples_fast_body<MASS, FULL, NT>(L, F, params, zt_all, dLt_all, mg_all, cdf_all, rec_all, TcMax, TmMax, blockIdx.x % L.nb, blockIdx.x / L.nb, gridDim.x / L.nb, lds);
  CLK_END(2);
}

// ------------------------------------------------------------------------------------------------------
// KDE building blocks shared by k_kde_marg (one wave) and k_kde1d (one block)
// ------------------------------------------------------------------------------------------------------
// bin index of binning1d (math.py:41): clip(floor((x - lo)/(hi - lo) * B), 0, B-1); a NaN index (hi == lo) -> 0,
// the density is NaN downstream either way (bandwidth 0).
DEVFN int bin_index(double z, double lo, double hi, int B) {
  double f = floor((z - lo) / (hi - lo) * (double)B);
  f = f < 0. ? 0. : (f > (double)(B - 1) ? (double)(B - 1) : f);
  return (f != f) ? 0 : (int)f;
}

// bin_index() with the pixel-constant divisor d = hi - lo and r = 1/d (one IEEE division per pixel): the quotient is formed as
// q0 = x r, q = q0 + r (x - q0 d) (fma residual), which IS the correctly rounded x/d for the operands met here (0 <= x <= d,
// Markstein), so the bin of every sample is the one bin_index() gives; the clamp runs on v_max/v_min (NaN -> bin 0).
// [r4] floor and the two-sided fp64 clamp became v_cvt_i32_f64 (truncation = floor for q >= 0, saturating, NaN -> 0) + two integer clamps on
// one v_med3_i32: the same bin for every input (negative products -- z below lo -- truncate towards 0 or clamp to 0, as max(floor, 0) did).
DEVFN int bin_index_r(double z, double lo, double d, double r, double dB) {
  double x = z - lo;
  double q = x * r;
  q = fma(fma(-q, d, x), r, q);
  return med3_i32(cvt_i32_sat(q * dB), 0, (int)dB - 1);
}
// the same bin with the upper clamp B - 1 held in a scalar register and the lower one as the inline constant 0 (see med3_i32_0s)
DEVFN int bin_index_rs(double z, double lo, double d, double r, double dB, int bm1_sgpr) {
  double x = z - lo;
  double q = x * r;
  q = fma(fma(-q, d, x), r, q);
  return med3_i32_0s(cvt_i32_sat(q * dB), bm1_sgpr);
}

DEVFN double kde_bandwidth_factor(int bw_method, double bw_scalar, double neff, int d) {
  // math.py:65-73 (d = 1), :178-183 (d = 3)
  // x^y as exp(y log x): |y log x| eps ~ 1e-16 away from pow()
  if (bw_method == 0) return chm_exp(chm_log(neff) * (-1. / (double)(d + 4)));
  if (bw_method == 1) return chm_exp(chm_log(neff * (double)(d + 2) / 4.0) * (-1. / (double)(d + 4)));
  return bw_scalar;
}

// x^(-1/5) for the bandwidth rules (math.py:65-73, d = 1) without fp64 log/exp: fp32 hardware log2/exp2 give a seed good to
// ~1e-7, two Newton steps y <- y (6 - x y^5)/5 (error -> 3 e^2) bring it to fp64 rounding (~20 instructions instead of ~70).
// x = 0, inf, NaN give inf / 0 / NaN as exp(-log(x)/5) does; used by the standard GW kernel, whose support test is tolerant
// of the last bits of the bandwidth.
DEVFN double pow_m1_5(double x) {
  double y = (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf((float)x));
#pragma unroll
  for (int it = 0; it < 2; it++) {
    double y2 = y * y, y5 = (y2 * y2) * y;
    y = y * ((6. - x * y5) * 0.2);
  }
  if (x == 0.) y = __builtin_inf();
  if (x == __builtin_inf()) y = 0.;
  return y;
}
DEVFN double kde_bandwidth_factor_fast(int bw_method, double bw_scalar, double neff) {
  if (bw_method == 0) return pow_m1_5(neff);
  if (bw_method == 1) return pow_m1_5(neff * 3. / 4.0);
  return bw_scalar;
}

// lower/upper ends of the effective grid (likelihood.py:119-120 for p_gw1d, :186-187 for p_gw3dmarg)
DEVFN void eff_bounds(bool marg, double zmin, double zmax, double sd, double cut, double& lb, double& ub) {
  lb = zmin - cut * sd;
  if (marg) lb = (lb != lb) ? lb : (lb > 1e-8 ? lb : 1e-8);     // jnp.maximum(., 1e-8)
  else lb = lb > 0. ? lb : 1.e-8;                                // jnp.where(. > 0, ., 1e-8)
  ub = zmax + cut * sd;
}

// Epanechnikov density at g from the prefix sums of the (sorted, uniform) binned dataset:
//   sum_j W_j 3/4 (1 - u_j^2) [|u_j| <= 1] / h,  u_j = (g - c_j)/h
// = 3/4 (S0 - (g'^2 S0 - 2 g' S1 + S2)/h^2) / h over the index range [ja, jb) of bins inside the support, with
// S_k = P_k[jb] - P_k[ja], P_k = prefix sums of W c'^k, c' = c - c_ref, g' = g - c_ref.  The range is found from the
// bin spacing and fixed up with the reference's own predicate |(g - c_j)/h| <= 1 (math.py:78,85).
// Differs from the dense left-to-right sum by rounding only (~(R/h)^2 eps, R = span of the bins).
// jl1 (wave_prefix3): one past the last lane chunk of bins that holds any weight.  The prefix values after it stem from
// different summation trees and agree only to an ulp, so a node whose support holds nothing but empty bins above the data would get
// 1e-16 of the peak instead of the dense sum's exact zero; the range is clipped to jl1, where the values are consistent.
// The prefix form carries an absolute error of ~1e-16 (R/h)^2 of the weight summed so far: where the bins in reach of a node hold
// less than 1e-4 of what lies below them (the far upper tail; weights spanning many decades), the reference's dense sum over those
// bins (math.py:77-81, `wgt` = the normalised bin weights) is evaluated instead -- exact, and rare.
DEVFN double epan_prefix_eval(double g, const double* cen, const double* wgt, const double* P0, const double* P1, const double* P2, int N,
                              double lo, double inv_dbin, double bw, double inv_bw, double c_ref, int jl1) {
  double fa = ceil((g - bw - lo) * inv_dbin - 0.5), fb = floor((g + bw - lo) * inv_dbin - 0.5) + 1.;
  int ja = fa > 0. ? (fa < (double)N ? (int)fa : N) : 0;
  int jb = fb > 0. ? (fb < (double)N ? (int)fb : N) : 0;
  // fix-ups with the exact predicate (each loop runs 0 or 1 times for a sane guess)
  while (ja > 0 && fabs((g - cen[ja - 1]) * inv_bw) <= 1.) ja--;
  while (ja < N && cen[ja] < g && !(fabs((g - cen[ja]) * inv_bw) <= 1.)) ja++;
  if (jb < ja) jb = ja;
  while (jb < N && fabs((g - cen[jb]) * inv_bw) <= 1.) jb++;
  while (jb > ja && !(fabs((g - cen[jb - 1]) * inv_bw) <= 1.)) jb--;
  if (jb > jl1) jb = jl1;
  if (ja > jl1) ja = jl1;
  double S0 = P0[jb] - P0[ja], S1 = P1[jb] - P1[ja], S2 = P2[jb] - P2[ja];
  if (jb > ja && !(S0 >= 1e-4 * P0[jb])) {                       // also taken for NaN sums
    double acc = 0.;
    for (int j = ja; j < jb; j++) {
      double u = (g - cen[j]) * inv_bw;
      acc += wgt[j] * (fabs(u) <= 1. ? 0.75 * (1. - u * u) : 0.);
    }
    return acc / bw;
  }
  double gp = g - c_ref;
  double qq = fma(gp, fma(gp, S0, -2. * S1), S2);                // sum W (g' - c')^2
  // a sum of non-negative kernel values: rounding of the prefix-sum form (~1e-14 of the peak) must not make it negative
  return __builtin_fmax(0.75 * (S0 - qq * (inv_bw * inv_bw)) * inv_bw, 0.);
}

// dense density at g (math.py:77-81), used for the Gaussian kernel, for binning=False and for degenerate bandwidths
DEVFN double kde_dense_eval(double g, const double* data, const double* wgt, int N, bool epan, double bw, double inv_bw) {
  double acc = 0.;
  if (epan) {
    for (int j = 0; j < N; j++) {
      double u = (g - data[j]) * inv_bw;
      double kv = fabs(u) <= 1. ? 0.75 * (1. - u * u) : 0.;
      acc += wgt[j] * kv;
    }
  } else {
    const double isq = 1. / sqrt(2. * CHM_PI);
    for (int j = 0; j < N; j++) {
      double u = (g - data[j]) * inv_bw;
      acc += wgt[j] * (chm_exp(-0.5 * (u * u)) * isq);
    }
  }
  return acc / bw;
}

// jnp.linspace(start, stop, num)[i] with the step fraction i/(num-1) read from a table (same value as the division)
DEVFN double linspace_tab(double start, double stop, int num, int i, const double* frac) {
  if (i >= num - 1) return stop;
  double step = frac[i];
  return start * (1. - step) + stop * step;
}

// searchsorted(eff, z, side='right') clipped to [1, G-1] (jnp.interp) starting from a guess
DEVFN int interp_index(const double* eff, int G, double z, int guess) {
  int i = guess < 1 ? 1 : (guess > G - 1 ? G - 1 : guess);
  while (i > 1 && eff[i - 1] > z) i--;
  while (i < G - 1 && eff[i] <= z) i++;
  return i;
}

// jnp.interp(z, eff, dens, left=0, right=0) with an index guess
DEVFN double interp_lr0(const double* eff, const double* dens, int G, double z, int guess) {
  int i = interp_index(eff, G, z, guess);
  double x0 = eff[i - 1], x1 = eff[i], f0 = dens[i - 1], f1 = dens[i];
  double dx = x1 - x0;
  double f = (fabs(dx) <= 4.930380657631324e-32) ? f0 : f0 + ((z - x0) / dx) * (f1 - f0);
  if (z < eff[0]) f = 0.;
  if (z > eff[G - 1]) f = 0.;
  return f;
}

DEVFN int eff_guess(double z, double lb, double ub, int G, bool has_cut, int k) {
  if (!has_cut) return k + 1;
  double tpos = (z - lb) / (ub - lb) * (double)(G - 1);
  if (!(tpos >= 0.)) return 1;
  if (tpos > (double)G) return G - 1;
  return (int)tpos + 1;
}

// One wave: integrand + trapezoid over the event grid for one (event, pixel)   (likelihood.py:274-278 / 291)
//   y_k = pgw_k * p_z / jac,  p_z = (fR p_cat + bkgA) prate  (pixelated)  |  bkgA prate  (no catalogue)
template <class PGW>
DEVFN double wave_integrate(PGW pgw_at, const double* zg, const double* jac, const double* prate, const double* bkgA,
                            const double* pc, double fR, int Z, double* dump) {
  const int lane = threadIdx.x & 63;
  double acc = 0.;
  for (int k0 = 0; k0 < Z - 1 || k0 == 0; k0 += 63) {
    int k = k0 + lane;
    double zk = 0., y = 0.;
    if (k < Z) {
      zk = zg[k];
      double pgw = pgw_at(k, zk);
      if (dump) dump[k] = pgw;
      if (pc) {
        double pcv = pc[k];
        if (pcv != -100.) {
          double p_gal = fR * pcv + bkgA[k];                 // catalog.py:202
          double p_z = p_gal * prate[k];                     // pop_wrapper.py:87
          y = (p_z != -100.) ? pgw * p_z / jac[k] : 0.;      // likelihood.py:274-277
        }
      } else {
        double p_z = bkgA[k] * prate[k];                     // pop_wrapper.py:89
        y = pgw * p_z / jac[k];                              // likelihood.py:291
      }
    }
    double y1 = __shfl_down(y, 1, 64), z1 = __shfl_down(zk, 1, 64);
    if (lane < 63 && k + 1 < Z) acc += (z1 - zk) * (y1 + y);
  }
  return 0.5 * wave_sum(acc);
}

// inclusive->exclusive prefix sums of W, W c', W c'^2 over N bins by one wave; P*[0..N].  Returns jl1, the end of the last lane
// chunk that holds a non-zero weight (see epan_prefix_eval).  Below the first weight every prefix is an exact zero.
DEVFN int wave_prefix3(const double* cen, const double* wgt, int N, double c_ref, double* P0, double* P1, double* P2) {
  const int lane = threadIdx.x & 63;
  const int per = (N + 63) / 64;
  const int j0 = lane * per, j1 = min(j0 + per, N);
  double s0 = 0., s1 = 0., s2 = 0.;
  for (int j = j0; j < j1; j++) { double W = wgt[j], cc = cen[j] - c_ref; s0 += W; s1 += W * cc; s2 += W * cc * cc; }
  // exclusive scan of the lane totals
  double x0 = s0, x1 = s1, x2 = s2;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double y0 = __shfl_up(x0, o, 64), y1 = __shfl_up(x1, o, 64), y2 = __shfl_up(x2, o, 64);
    if (lane >= o) { x0 += y0; x1 += y1; x2 += y2; }
  }
  double r0 = x0 - s0, r1 = x1 - s1, r2 = x2 - s2;
  if (lane == 0) { P0[0] = 0.; P1[0] = 0.; P2[0] = 0.; }
  for (int j = j0; j < j1; j++) {
    double W = wgt[j], cc = cen[j] - c_ref;
    r0 += W; r1 += W * cc; r2 += W * cc * cc;
    P0[j + 1] = r0; P1[j + 1] = r1; P2[j + 1] = r2;
  }
  const unsigned long long nz = __ballot(s0 != 0.);          // NaN counts as weight
  const int last = 63 - __clzll(nz);                          // -1: no weight at all
  return min((last + 1) * per, N);
}

// Per-(draw, event) statistics for the marginalized kernels: the chunk partials of k_samples combined once for all the event's
// pixels.  es[NEVSTAT]: zmin, zmax, std, norm, n_eff, sum w, lb, ub (effective-grid ends, likelihood.py:186-187), k_lo, k_hi
// (event-grid points inside [lb, ub]), de, 1/de (spacing of jnp.linspace(lb, ub, G), likelihood.py:188)
DEVFN void event_stats_from(const LikeDev& L, int e, const EvStats& st, double* es) {
  double lb = 0., ub = 0.;
  if (L.has_cut) eff_bounds(L.mode == 2, st.zmin, st.zmax, st.sd, L.cut_grid, lb, ub);
  else { lb = L.z_grids[(size_t)e * L.Z]; ub = L.z_grids[(size_t)e * L.Z + L.Z - 1]; }
  // k-range of the event grid that can see a non-zero KDE: z_k in [lb, ub].  Guess from the end points (the grid is a
  // linspace, pop_wrapper.py:207), verify against the stored grid, fall back to the whole grid otherwise.
  const int Z = L.Z;
  const double* zg = L.z_grids + (size_t)e * Z;
  int k_lo = 0, k_hi = Z - 1;
  if (!L.grid_unsorted) {
    const double z0 = zg[0], zl = zg[Z - 1];
    const double inv_dz = (double)(Z - 1) / (zl - z0);
    double fl = floor((lb - z0) * inv_dz) - 1., fh = ceil((ub - z0) * inv_dz) + 1.;
    int gl = fl > 0. ? (fl < (double)(Z - 1) ? (int)fl : Z - 1) : 0;
    int gh = fh < (double)(Z - 1) ? (fh > 0. ? (int)fh : 0) : Z - 1;
    if (fl == fl && fh == fh && gl <= gh && (gl == 0 || zg[gl] < lb) && (gh == Z - 1 || zg[gh] > ub)) { k_lo = gl; k_hi = gh; }
  }
  es[0] = st.zmin; es[1] = st.zmax; es[2] = st.sd; es[3] = st.norm; es[4] = st.n_eff; es[5] = st.sumw; es[6] = lb; es[7] = ub;
  // pairs (k, k + 1), k even: the range starts at an even and ends at an odd grid point (one more point on either side at most; the
  // standard GW kernel handles both points of a pair together, the per-z factors cover exactly [k_lo, k_hi])
  k_lo &= ~1; k_hi = min(k_hi | 1, Z - 1);
  es[8] = (double)k_lo; es[9] = (double)k_hi;
  es[10] = (ub - lb) / (double)(L.G - 1); es[11] = (double)(L.G - 1) / (ub - lb);
}
DEVFN void event_stats(const LikeDev& L, int b, int e, double* es) {
  event_stats_from(L, e, combine_stats(L.part + ((size_t)b * L.E + e) * L.NC * NPART, L.NC, L.S), es);
}

// k_event_prep: one wave per (event, draw), four waves per block: event_stats -> evstat (nb,E,NEVSTAT) and, for the general kernel (k_kde_marg),
// the effective grid effg (nb,E,G); the standard kernel (k_kde_marg_sub) forms its nodes arithmetically.
// k_event_stats: the same statistics with a THREAD per (event, draw) -- the standard kernel needs no effective grid, and a wave per
// event spent its 3 us of life on two dependent memory round trips with 63 lanes idle (107 us at C3 / 128 draws for 128 000 waves)
__global__ void __launch_bounds__(256) k_event_stats(LikeDev L) {
  const int ei = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (ei >= L.E_cnt) return;
  const int e = L.e_off + ei;
  double es[NEVSTAT];
  event_stats(L, b, e, es);
  double* o = L.evstat + ((size_t)b * L.E + e) * NEVSTAT;
#pragma unroll
  for (int i = 0; i < NEVSTAT; i++) o[i] = es[i];
}

__global__ void __launch_bounds__(256) k_event_prep(LikeDev L, int write_effg) {
  const int lane = threadIdx.x & 63, ei = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;     // four events (waves) per block
  if (ei >= L.E_cnt) return;
  const int e = L.e_off + ei;
  double es[NEVSTAT];
  event_stats(L, b, e, es);
  double* o = L.evstat + ((size_t)b * L.E + e) * NEVSTAT;
  if (lane < NEVSTAT) {
    double v = es[0];
#pragma unroll
    for (int i = 1; i < NEVSTAT; i++) if (lane == i) v = es[i];
    o[lane] = v;
  }
  if (!write_effg) return;
  const double lb = es[6], ub = es[7];
  double* eg = L.effg + ((size_t)b * L.E + e) * L.G;
  if (L.has_cut) { for (int i = lane; i < L.G; i += 64) eg[i] = linspace_tab(lb, ub, L.G, i, L.fracG); }   // likelihood.py:188
  else { for (int i = lane; i < L.G; i += 64) eg[i] = L.z_grids[(size_t)e * L.Z + i]; }                        // likelihood.py:190
}

// ------------------------------------------------------------------------------------------------------
// k_zfactors: per-z factors of the integrand on each event grid; one block per (event, draw)
//   jac = ddL/dz (1+z)^2 (likelihood.py:272);  prate = merger_rate/(1+z) (pop_wrapper.py:85);
//   bkgA = (1 - P_compl) p_bkg (catalog.py:202)  or  p_bkg for the empty catalogue (catalog.py:43)
// ------------------------------------------------------------------------------------------------------
#ifndef CHM_ZF_WPE
#define CHM_ZF_WPE 4                      // waves per SIMD the per-z-factor kernel is compiled for (128 VGPRs: 577 -> 536 us at C3 / 128 draws)
#endif
// jnp.interp(z, zt, fp) from a prepared bracket: fp[i-1] + t (fp[i] - fp[i-1]) with t = delta/dx formed once (the same two roundings
// as jnp_interp, in a body without fp contraction); t = -1: dx below jnp.interp's epsilon -> fp[i-1]; -2: z < zt[0] -> fp[0];
// -3: z > zt[n-1] -> fp[n-1]; NaN stays NaN
template <class A>
DEVFN double interp_pre(A fp, int n, int i, double t) {
  const double f0 = fp[i - 1], f1 = fp[i];
  double f = f0 + t * (f1 - f0);
  if (t == -1.) f = f0;
  if (t == -2.) f = fp[0];
  if (t == -3.) f = fp[n - 1];
  return f;
}

// k_grid_prep: bracket, weight and log(1 + z) of every event-grid point on the z table of a draw (all draws of equal z_max / z_grid_res
// share it bit for bit); a thread per grid point
__global__ void __launch_bounds__(256) k_grid_prep(int E, int Z, const double* z_grids, const double* zt, int Tc, int* out_i, double* out_t, double* out_lz) {
  const size_t n = (size_t)E * Z;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
    const double z = z_grids[q];
    int i = searchsorted_right(zt, Tc, z);
    i = i < 1 ? 1 : (i > Tc - 1 ? Tc - 1 : i);
    const double x0 = zt[i - 1], dx = zt[i] - x0;
    double t = (z - x0) / dx;
    if (fabs(dx) <= 4.930380657631324e-32) t = -1.;
    if (z < zt[0]) t = -2.;
    if (z > zt[Tc - 1]) t = -3.;
    out_i[q] = i; out_t[q] = t; out_lz[q] = chm_log_pos(1. + z);
  }
}

// per-z factors of ONE grid point k of event e (draw P): bkgA[k] = (1 - P_compl) p_bkg (catalog.py:202), Aw[k] = prate/jac * trapezoid weight
// (likelihood.py:270-278), and -- whole-grid launches only (!ranged) -- jac, prate.  Shared by zfactors_body and the fused event kernel.
template <class AZ, class AI, class EX>
DEVFN void zfactor_point(const LikeDev& L, const DevParams& P, const int e, const int k, const size_t zo, const double* zg, AZ zt, AI It,
                       const int ranged, const EX& ex) {
#pragma clang fp contract(fast)
  const int Z = L.Z;
  double z = zg[k];
  const bool own_cosmo = !(L.tab_jac && L.tab_bkg);          // plug-in cosmology: Jacobian and p_bkg both come from the caller
  double zp1 = 1. + z;
  double dCt = 0., lzp1;
  if (L.zg_i) {                                              // prepared bracket (k_grid_prep): no table search, no division, no log
    const size_t q = (size_t)e * Z + k;
    if (own_cosmo) dCt = dCt_from_dCr(P, P.dH * interp_pre(It, P.Tc, L.zg_i[q], L.zg_t[q]));      // cosmo.py:132-153
    lzp1 = L.zg_lz[q];
  } else {
    if (own_cosmo) dCt = dCt_at_z(P, z, zt, It);
    lzp1 = chm_log_pos(zp1);
  }
  // One reciprocal of E(z) serves the Jacobian (dH/E) and dVc/dz (.../E); the rate stays a quotient num/den until it meets the
  // Jacobian, so that A_k = prate/jac tw costs one division instead of three (five IEEE divisions per grid point before: 248 VALU
  // instructions per point, 129 M per launch at C3 / 128 draws).  Differences to the separate quotients: rounding (<= 4 ulp).
  const double Ez = own_cosmo ? E_at_z_l(P, z, lzp1) : 1.;
  const double rEz = chm_div(1., Ez);
  const double jac = L.tab_jac ? L.tab_jac[zo + k] : ddLdz_from_dCt_rE(P, dCt, zp1, rEz, lzp1) * (zp1 * zp1);
  double rnum, rden = 1.;
  if (L.tab_rate) rnum = L.tab_rate[zo + k];                                                       // plug-in rate model: tabulated
  else merger_rate_nd(P, z, lzp1, rnum, rden, ex);
  if (!ranged) { L.jac[zo + k] = jac; L.prate[zo + k] = rnum / (rden * zp1); }
  const double p_bkg = L.tab_bkg ? L.tab_bkg[zo + k] : (4. * CHM_PI * P.dH) * (dCt * dCt) * rEz;   // plug-in completeness: tabulated; cosmo.py:188-197
  // a 1-D handle built from a catalogue population (hyperlikelihood.p_gw1d on a pixelated object) carries no P_compl
  L.bkgA[zo + k] = (P.has_catalog && L.P_compl) ? (1. - L.P_compl[(size_t)e * Z + k]) * p_bkg : p_bkg;
  if (L.Aw) {
    // trapezoid weight of grid point k: y_k enters the two adjacent intervals (jnp.trapezoid, likelihood.py:278)
    double zl = k > 0 ? zg[k - 1] : z, zr = k < Z - 1 ? zg[k + 1] : z;
    double tw = 0.5 * ((z - zl) + (zr - z));
    L.Aw[zo + k] = (rnum * tw) / ((rden * zp1) * jac);
  }
}

// STATS (few draws per call, ranged == 1): the wave forms its event's statistics itself and writes them for the GW kernel -- no k_event_stats
// launch in front of this kernel
// (body shared by k_zfactors and k_zf_sel: block bx of nbx, draw b)
template <bool LDS_TAB, bool STATS>
DEVFN void zfactors_body(const LikeDev& L, const DevParams* params, const double* zt_all, const double* It_all, int TcMax, int ranged,
                         const int b, const int bx, const int nbx, double* lds) {
#pragma clang fp contract(fast)                  // smooth per-z factors: a*b+c may fuse (jnp_interp keeps the default, off)
  const int t = threadIdx.x, nt = blockDim.x;
  const DevParams P = params[b];      // by value: uniform loads at kernel start -> scalar registers, nothing re-read in the loops
  const double* zt = zt_all + (size_t)b * TcMax;
  const double* It = It_all + (size_t)b * TcMax;
  __shared__ double etab[CHM_EXPTAB_N];             // [r3] the two powers of the merger rate through the table exp (chm_exp_tab)
  for (int i = t; i < CHM_EXPTAB_N; i += nt) etab[i] = exp_table_entry(i);
  const ExpTab ex = { etab };
  if (LDS_TAB) {                                    // staged once per block; the block then walks over its events
    double* a = lds; double* c = lds + P.Tc;
    for (int i = t; i < P.Tc; i += nt) { a[i] = zt[i]; c[i] = It[i]; }
    zt = a; It = c;
  }
  __syncthreads();
  const int Z = L.Z;
  // ranged: a WAVE per event (the support of an event's KDE is ~Z/3 points: 64-lane passes waste less than 256-thread ones);
  // whole grids: the block walks over the events
  // [r3] STATS (few draws per call: the scalar call): CHM_ZF_WPE_FEW waves per event -- a wave per event left three quarters of the chip's
  // wave slots empty at 1000 events x 1 draw and walked ~9 dependent passes per event
#ifndef CHM_ZF_WPE_FEW
#define CHM_ZF_WPE_FEW 2
#endif
  const int wpe = (STATS && ranged) ? CHM_ZF_WPE_FEW : 1;  // waves per event
  const int epb = (nt >> 6) / wpe;                         // events per block pass
  const int lane0 = ranged ? ((t >> 6) % wpe) * 64 + (t & 63) : t, stride = ranged ? 64 * wpe : nt;
  const int ev_first = ranged ? bx * epb + (t >> 6) / wpe : bx, ev_step = ranged ? nbx * epb : nbx;
  for (int ei = ev_first; ei < L.E_cnt; ei += ev_step) {
    const int e = L.e_off + ei;
    const size_t zo = ((size_t)b * L.E + e) * Z;
    const double* zg = L.z_grids + (size_t)e * Z;
    // ranged: only the grid points the GW kernel reads -- 1 (marginalized, after k_event_prep): [k_lo & ~1, k_hi] of the event,
    // the support of its KDE (NEVSTAT slots 8-9), nothing for an event that fails the n_eff guard (likelihood.py:199);
    // 2 (1d / approximate, after k_kde1d): the range where p_gw1d is non-zero
    int k_first = 0, k_last = Z - 1;
    if (STATS) {                                    // every lane forms the same numbers (uniform loads); lane 0 keeps them for the GW kernel
      double es[NEVSTAT];
      event_stats(L, b, e, es);
      if (lane0 == 0) {
        double* o = L.evstat + ((size_t)b * L.E + e) * NEVSTAT;
#pragma unroll
        for (int i = 0; i < NEVSTAT; i++) o[i] = es[i];
      }
      if (!(es[4] >= L.pe_neff)) continue;
      k_first = ((int)es[8]) & ~1; k_last = (int)es[9];
    } else if (ranged == 1) {
      const double* es = L.evstat + ((size_t)b * L.E + e) * NEVSTAT;
      if (!(es[4] >= L.pe_neff)) continue;
      k_first = ((int)es[8]) & ~1; k_last = (int)es[9];
    } else if (ranged == 2) {                       // 1d / approximate: the grid points where p_gw1d != 0 (k_kde1d)
      const int* kr = L.krange + ((size_t)b * L.E + e) * 2;
      k_first = kr[0]; k_last = kr[1];
    }
    for (int k = k_first + lane0; k <= k_last; k += stride) zfactor_point(L, P, e, k, zo, zg, zt, It, ranged, ex);
  }
}

// [r5] k_rate_special: after a WHOLE-GRID k_zfactors launch of a call that carries a draw with an infinite rate parameter -- one block per (event, draw)
// of such a draw rewrites the draw's rate factors (prate, and the rate factor inside A_k) with merger_rate_special, the reference's own operations
// (rate.py:96-122: value classes of C99 pow), and flags the event when prate / jac is inf or NaN anywhere on its grid (event_poisoned).  Draws
// with finite rate parameters are left alone (their flag is cleared).
__global__ void __launch_bounds__(256) k_rate_special(LikeDev L, const DevParams* params) {
  const int b = blockIdx.y, e = L.e_off + blockIdx.x, t = threadIdx.x, Z = L.Z;
  const DevParams& P = params[b];
  if (!P.rate_special) { if (t == 0 && L.ev_rbad) L.ev_rbad[(size_t)b * L.E + e] = 0; return; }
  const size_t zo = ((size_t)b * L.E + e) * Z;
  const double* zg = L.z_grids + (size_t)e * Z;
  bool rbad = false;
  for (int k = t; k < Z; k += 256) {
    const double z = zg[k], zp1 = 1. + z;
    const double r = L.tab_rate ? L.tab_rate[zo + k] : merger_rate_special(P, z);
    const double pr = r / zp1, jac = L.jac[zo + k];                 // pop_wrapper.py:85
    L.prate[zo + k] = pr;
    if (L.Aw) {
      const double zl = k > 0 ? zg[k - 1] : z, zr = k < Z - 1 ? zg[k + 1] : z;
      L.Aw[zo + k] = (pr / jac) * (0.5 * ((z - zl) + (zr - z)));   // likelihood.py:275-278: p_z / jac times the trapezoid weight
    }
    const double f = pr / jac;
    rbad = rbad || !(fabs(f) < __builtin_inf());
  }
  const int any = __syncthreads_or(rbad ? 1 : 0);
  if (t == 0 && L.ev_rbad) L.ev_rbad[(size_t)b * L.E + e] = any ? 1 : 0;
}

template <bool LDS_TAB, bool STATS = false>
__global__ void __launch_bounds__(256, CHM_ZF_WPE) k_zfactors(LikeDev L, const DevParams* params, const double* zt_all, const double* It_all,
                                                   int TcMax, int ranged) {
  extern __shared__ double lds[];
  zfactors_body<LDS_TAB, STATS>(L, params, zt_all, It_all, TcMax, ranged, blockIdx.y, blockIdx.x, gridDim.x, lds);
}

// ------------------------------------------------------------------------------------------------------
// k_kde_marg: one wave per (event, pixel, draw): p_gw3dmarg (likelihood.py:160-205) fused with the integrand,
// the trapezoid and the -100 masking (likelihood.py:274-278)
// ------------------------------------------------------------------------------------------------------
// The integrand is non-zero only where the pixel's KDE is: inside [lb, ub] of the effective grid.  The wave therefore
// (1) finds the k-range of the event grid inside [lb, ub] from the event statistics and prefetches exactly that part of
// its p_cat row (HBM) into registers before doing anything else, (2) builds the pixel's histogram / prefix sums / KDE
// while those loads are in flight, (3) accumulates  sum_k p_gw[k] (fR p_cat[k] + bkgA[k]) A[k]  over that range, with
// A[k] = prate[k]/jac[k] * tw[k], tw = trapezoid weights of the event grid -- the same sum as
// trapz(p_gw3d * p_z / jac) up to rounding (terms outside the range are exact zeros for finite inputs).
// Dynamic LDS: cen[N], wgt[N], [P0,P1,P2 (N+1) when binning], eff[G], dens[G];  N = num_bins or S.
#define MARG_PF 8            // prefetch depth: 8 x (64 lanes x 2 doubles) = 1024 grid points per pass
// ordering point for LDS traffic inside ONE wave (its lanes exchange data through the wave's private LDS slice): LDS
// instructions of a wave execute in issue order, so only the compiler has to be kept from moving accesses across it
DEVFN void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one (event, pixel, draw) by one wave (a block of 64 threads); own_eff: form the effective grid here instead of reading k_event_prep's
// WSYNC: the wave is one of several in its block (the fused event kernel's dense redo): the ordering points are wave-level (the LDS slice is
// this wave's alone) instead of block barriers
template <bool WSYNC = false>
DEVFN void kde_marg_general(const LikeDev& L, const DevParams* params, const int b, const int e, const int p, double* lds, const bool own_eff) {
  auto bsync = [] { if (WSYNC) wave_sync(); else __syncthreads(); };
  const int lane = threadIdx.x & 63;
  const DevParams& P = params[b];
  const int S = L.S, Z = L.Z, B = L.num_bins, G = L.G;
  const int N = L.binning ? B : S;
  double* data = lds; double* wgt = data + N;
  double* P0 = wgt + N; double* P1 = P0 + (L.binning ? N + 1 : 0); double* P2 = P1 + (L.binning ? N + 1 : 0);
  double* eff = P2 + (L.binning ? N + 1 : 0); double* dens = eff + G;
  const size_t so = ((size_t)b * L.E + e) * S;
  const double* wz = L.ws_z + so;
  const double* ww = L.ws_w + so;
  double* out_like = L.like_pix + ((size_t)b * L.E + e) * L.P + p;
  double* dump = L.p_gw_dump ? L.p_gw_dump + (((size_t)b * L.E + e) * L.P + p) * Z : nullptr;

  if (p >= L.neff_pixels[e]) {        // padded pixel: p_cat == -100 there, integrand masked to 0 (likelihood.py:274-277)
    if (lane == 0) *out_like = 0.;
    if (dump) for (int k = lane; k < Z; k += 64) dump[k] = 0.;
    return;
  }
  const double* es = L.evstat + ((size_t)b * L.E + e) * NEVSTAT;
  const double zmin = es[0], norm = es[3], n_eff = es[4], lb = es[6], ub = es[7];
  const bool ok = n_eff >= L.pe_neff;                       // lax.cond(n_eff >= pe_neff, ...)   likelihood.py:199
  const bool poisoned = event_poisoned(L, P, b, e, L.z_grids + (size_t)e * Z, Z);
  const double* zg = L.z_grids + (size_t)e * Z;
  const double* pc = L.p_cat + ((size_t)e * L.P + p) * Z;

  // (1) k-range of the event grid that can see a non-zero KDE: z_k in [lb, ub].  Guess from the end points (the grid is
  //     a linspace, pop_wrapper.py:207), verify against the stored grid, fall back to the whole grid otherwise.
  int k_lo = 0, k_hi = Z - 1;
  if (ok && !L.grid_unsorted) {
    const double z0 = zg[0], zl = zg[Z - 1];
    const double inv_dz = (double)(Z - 1) / (zl - z0);
    double fl = floor((lb - z0) * inv_dz) - 1., fh = ceil((ub - z0) * inv_dz) + 1.;
    int gl = fl > 0. ? (fl < (double)(Z - 1) ? (int)fl : Z - 1) : 0;
    int gh = fh < (double)(Z - 1) ? (fh > 0. ? (int)fh : 0) : Z - 1;
    if (fl == fl && fh == fh && gl <= gh && (gl == 0 || zg[gl] < lb) && (gh == Z - 1 || zg[gh] > ub)) { k_lo = gl; k_hi = gh; }
  }
  k_lo &= ~1;                                               // 16-byte aligned pairs
  // prefetch the first pass of the p_cat segment: lane owns grid points k_lo + 128*i + 2*lane + {0,1}
  double pf0[MARG_PF], pf1[MARG_PF];
  if (ok) {
#pragma unroll
    for (int i = 0; i < MARG_PF; i++) {
      int k = k_lo + 128 * i + 2 * lane;
      pf0[i] = 0.; pf1[i] = 0.;
      if (k + 1 <= k_hi && ((Z & 1) == 0)) { double2 v = *reinterpret_cast<const double2*>(pc + k); pf0[i] = v.x; pf1[i] = v.y; }
      else { if (k <= k_hi) pf0[i] = pc[k]; if (k + 1 <= k_hi) pf1[i] = pc[k + 1]; }
    }
  }

  if (ok) {
    const int s0 = L.seg_off[(size_t)e * (L.P + 1) + p], s1 = L.seg_off[(size_t)e * (L.P + 1) + p + 1];
    const double lo = zmin;
    // hi = max(where(mask, z, min z))  (likelihood.py:180, math.py:36)
    double hi = lo;
    for (int s = s0 + lane; s < s1; s += 64) hi = nanmax2(hi, wz[s]);
    hi = wave_max(hi);
    if (lo != lo) hi = lo;
    if (L.binning) {
      for (int j = lane; j < B; j += 64) {                  // bin centres (math.py:37-39)
        double e0 = linspace_tab(lo, hi, B + 1, j, L.fracB), e1 = linspace_tab(lo, hi, B + 1, j + 1, L.fracB);
        data[j] = (e0 + e1) / 2.;
        wgt[j] = 0.;
      }
      bsync();
      for (int s = s0 + lane; s < s1; s += 64) atomicAdd(&wgt[bin_index(wz[s], lo, hi, B)], ww[s]);
      bsync();
    } else {
      for (int s = lane; s < S; s += 64) {
        bool in = s >= s0 && s < s1;
        data[s] = in ? wz[s] : zmin;                        // likelihood.py:180-181
        wgt[s] = in ? ww[s] : 0.;
      }
      bsync();
    }
    // kde1d prologue (math.py:58-75): normalise weights, neff, std(dataset), bandwidth
    double a = 0.;
    for (int j = lane; j < N; j += 64) a += wgt[j];
    const double tot = wave_sum(a);
    a = 0.;
    double c = 0.;
    for (int j = lane; j < N; j += 64) { double W = wgt[j] / tot; wgt[j] = W; a += W * W; c += data[j]; }
    const double neff_k = 1.0 / wave_sum(a);
    const double meanc = wave_sum(c) / (double)N;
    a = 0.;
    for (int j = lane; j < N; j += 64) { double d = data[j] - meanc; a += d * d; }
    const double stdc = sqrt(wave_sum(a) / (double)N);
    const double bw = kde_bandwidth_factor(L.bw_method, L.bw_scalar, neff_k, 1) * stdc;
    // effective grid (likelihood.py:185-190), built once per event by k_event_prep
    if (own_eff) { for (int i = lane; i < G; i += 64) eff[i] = L.has_cut ? linspace_tab(lb, ub, G, i, L.fracG) : L.z_grids[(size_t)e * Z + i]; }       // likelihood.py:188,190
    else { const double* eg = L.effg + ((size_t)b * L.E + e) * G; for (int i = lane; i < G; i += 64) eff[i] = eg[i]; }
    bsync();
    // density on the effective grid: always Epanechnikov here (kde1d is called without kernel=, likelihood.py:192)
    const double inv_bw = 1. / bw;
    const double dbin = (hi - lo) / (double)B;
    const bool fast = L.binning && dbin > 0. && bw > 0. && bw < 1e300 && tot == tot && (tot > 0. || tot < 0.) && !L.neg_w;
    if (fast) {
      const int jl1 = wave_prefix3(data, wgt, N, lo, P0, P1, P2);
      bsync();
      for (int i = lane; i < G; i += 64) dens[i] = epan_prefix_eval(eff[i], data, wgt, P0, P1, P2, N, lo, 1. / dbin, bw, inv_bw, lo, jl1);
    } else {
      for (int i = lane; i < G; i += 64) dens[i] = kde_dense_eval(eff[i], data, wgt, N, true, bw, inv_bw);
    }
    bsync();
  }

  // (3) integrand over [k_lo, k_hi]
  const size_t zo = ((size_t)b * L.E + e) * Z;
  const double* bkgA = L.bkgA + zo;
  const double* Aw = L.Aw + zo;
  const double gwp = L.gw_pdf[(size_t)e * L.P + p];
  const double fR = P.fR;
  const double inv_de = (double)(G - 1) / (ub - lb);
  const bool has_cut = L.has_cut;
  double acc = 0.;
  if (dump) { for (int k = lane; k < Z; k += 64) if (!ok || k < k_lo || k > k_hi) dump[k] = 0.; }
  if (ok) {
    for (int kb = k_lo; kb <= k_hi; kb += 128 * MARG_PF) {
#pragma unroll
      for (int i = 0; i < MARG_PF; i++) {
        const int k = kb + 128 * i + 2 * lane;
        if (k <= k_hi) {
          double pc0, pc1;
          if (kb == k_lo) { pc0 = pf0[i]; pc1 = pf1[i]; }
          else { pc0 = pc[k]; pc1 = (k + 1 <= k_hi) ? pc[k + 1] : 0.; }
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int kk = k + h;
            if (kk <= k_hi) {
              const double zk = zg[kk];
              int guess = has_cut ? ((zk - lb) * inv_de >= 0. ? (int)fmin((zk - lb) * inv_de, (double)G) + 1 : 1) : kk + 1;
              double f = interp_lr0(eff, dens, G, zk, guess);
              double pgw = f * norm * gwp;                  // kde_interp * norm * gw_pdf[i]    likelihood.py:194
              if (dump) dump[kk] = pgw;
              const double pcv = h == 0 ? pc0 : pc1;
              if (pcv != -100.) acc += pgw * (fR * pcv + bkgA[kk]) * Aw[kk];    // catalog.py:202, pop_wrapper.py:87, likelihood.py:275
            }
          }
        }
      }
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) *out_like = poisoned ? __builtin_nan("") : acc;     // NaN factors somewhere on the grid: 0 * NaN (see grid_is_poisoned)
}

__global__ void __launch_bounds__(64) k_kde_marg(LikeDev L, const DevParams* params) {
  extern __shared__ double lds[];
  kde_marg_general(L, params, blockIdx.y, L.e_off + blockIdx.x / L.P, blockIdx.x % L.P, lds, false);
}

// k_marg_fixup: one wave per (event, draw) after the standard GW kernel (kde_sub_item): sums the event's pixel integrals (L_i) and the
// bounds on what the prefix-sum form may have lost in them (err_pix).  If the summed bound exceeds tol L_i, every pixel whose bound
// exceeds its equal share tol L_i / P -- any pixel with a non-zero bound when L_i is not positive -- is evaluated again by the general
// kernel's body (dense kernel sums where the bins in reach are light, epan_prefix_eval), in pixel order.  For ordinary data the bound
// is ~1e-11 L_i: the wave reads 2 P doubles and exits.  After it, every L_i agrees with the dense form to ~tol + the general kernel's
// own 1e-10.
// L_i = sum of the event's pixel integrals IN PIXEL ORDER (jnp.sum over axis 1, likelihood.py:280; the order pixel_sum() keeps) by one wave:
// lane p holds pixel p, the running sum walks the lanes (v_readlane: 3 instructions per pixel, no memory round trips)
DEVFN double wave_pixel_sum_regs(double x, int n) {       // lane q < n holds pixel q (n <= 64): the running sum walks the lanes in pixel order
  const int xl = __double2loint(x), xh = __double2hiint(x);
  double Li = 0.;
#pragma unroll
  for (int q = 0; q < 64; q++)                            // compile-time lane: v_readlane_b32 x 2 + v_add_f64 per pixel (a run-time lane means an LDS round trip per pixel)
    if (q < n) Li += __hiloint2double(__builtin_amdgcn_readlane(xh, q), __builtin_amdgcn_readlane(xl, q));
  return Li;
}
DEVFN double wave_pixel_sum(const double* lp, int Pd) {
  const int lane = threadIdx.x & 63;
  double Li = 0.;
  for (int p0 = 0; p0 < Pd; p0 += 64) {
    const double x = p0 + lane < Pd ? lp[p0 + lane] : 0.;
    const int n = min(64, Pd - p0);
    const int xl = __double2loint(x), xh = __double2hiint(x);
#pragma unroll
    for (int q = 0; q < 64; q++)
      if (q < n) Li += __hiloint2double(__builtin_amdgcn_readlane(xh, q), __builtin_amdgcn_readlane(xl, q));
  }
  return Li;
}
// log L_i with jnp.nan_to_num(x, nan=-inf): NaN -> -inf, -inf -> -DBL_MAX, +inf -> DBL_MAX   (likelihood.py:296-297)
DEVFN double log_like_of(double Li) {
  double ll = log(Li);
  if (ll != ll) ll = -__builtin_inf();
  else if (ll == -__builtin_inf()) ll = -1.7976931348623157e308;
  else if (ll == __builtin_inf()) ll = 1.7976931348623157e308;
  return ll;
}

__global__ void __launch_bounds__(64) k_marg_fixup(LikeDev L, const DevParams* params, double tol) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x, e = L.e_off + blockIdx.x, b = blockIdx.y;
  const size_t po = ((size_t)b * L.E + e) * L.P;
  // [r3] few-draw calls (ev_publish): the event's L_i and log L_i leave this kernel (ev_li, ev_ll) and the reduction kernel sums E numbers per
  // draw instead of walking E x P pixel integrals and taking E logarithms in ONE block (11 -> 4 us on the scalar call's critical path).  With
  // many draws per call the reduction runs a block per draw side by side and this kernel's 128 000 single-wave blocks are better left short
  // (publishing there: 70 -> 150 us per 128 draws)
  auto publish = [&](const double Li) {
    if (lane == 0 && L.ev_publish) { L.ev_li[(size_t)b * L.E + e] = Li; L.ev_ll[(size_t)b * L.E + e] = log_like_of(Li); }
  };
  // L_i and the summed bound over the event's pixels (P <= 1024)
  double li = 0., es = 0., x0 = 0.;
  for (int p = lane; p < L.P; p += 64) { const double x = L.like_pix[po + p]; if (p < 64) x0 = x; li += x; es += L.err_pix[po + p]; }
  li = wave_sum(li); es = wave_sum(es);
  if (L.no_dense || !(es > tol * fabs(li))) {             // the event is within the tolerance as it stands (also: NaN anywhere -> stays NaN)
    if (L.ev_publish) publish(L.P <= 64 ? wave_pixel_sum_regs(x0, L.P) : wave_pixel_sum(L.like_pix + po, L.P));      // (the pixel integrals are in registers already)
    return;
  }
  const double share = tol * fabs(li) / (double)L.P;    // redo the pixels above their equal share: what is left sums to <= tol L_i
  for (int p0 = 0; p0 < L.P; p0 += 64) {
    const int p = p0 + lane;
    const double er = p < L.P ? L.err_pix[po + p] : 0.;
    unsigned long long need = __ballot(er > share);     // false for NaN (li or er): a NaN event stays NaN
    while (need) {
      const int q = __ffsll((long long)need) - 1;
      need &= need - 1ull;
      __syncthreads();
      kde_marg_general(L, params, b, e, p0 + q, lds, true);
    }
  }
  __syncthreads();                                      // the redone pixels were stored by lane 0 of this wave
  __threadfence_block();
  if (L.ev_publish) publish(wave_pixel_sum(L.like_pix + po, L.P));
}

// ------------------------------------------------------------------------------------------------------
// kde_sub_item<SW> / k_kde_marg_sub2: the marginalized GW kernel for the standard configuration (binning=True, cut_grid set), SW lanes per
// pixel (64/SW pixels per wave; SW = 32 in production).  Same quantity as k_kde_marg, organised for latency and occupancy:
//   * LDS holds only the three prefix-sum arrays P0 | P1 | P2 of (B + 1) doubles each (9.6 KB per wave at 200 bins -> 16 waves per CU);
//   * a node whose bins in reach hold < 1e-4 of the weight below them (rounding of the prefix differences: far tails, weights spanning
//     many decades) is evaluated by the reference's dense sum over the pixel's samples instead (dense_node), as k_kde_marg does;
//   * the KDE is evaluated on demand at the two effective-grid nodes that bracket each event-grid point (the values
//     jnp.interp combines, likelihood.py:193) instead of on the whole effective grid first;
//   * weights stay un-normalised in the prefix sums, 1/sum(w) is applied at the end; std of the bin centres is the
//     closed form for a uniform grid, (hi-lo) sqrt((B^2-1)/12)/B (math.py:67 evaluates it numerically: same to ~1e-16);
//   * the per-pixel set-up (histogram, prefix sums, bandwidth) is wave-instruction-bound with few active lanes: sharing a wave
//     between two pixels of the same event halves that cost per pixel while the grid loop keeps every lane busy.
// Degenerate pixels (no in-pixel weight, zero-width histogram, zero bandwidth) give NaN, as the reference's 0/0 does.
// ------------------------------------------------------------------------------------------------------
// inclusive prefix sum over each group of SW consecutive lanes (SW = 16, 32 or 64)
template <int SW> DEVFN double sg_scan_add(double x) {
  x += dpp_move<0x111, 0xf, true>(x);
  x += dpp_move<0x112, 0xf, true>(x);
  x += dpp_move<0x114, 0xf, true>(x);
  x += dpp_move<0x118, 0xf, true>(x);
  if (SW >= 32) x += dpp_move<0x142, 0xa, true>(x);       // row_bcast:15 into rows 1 and 3
  if (SW >= 64) x += dpp_move<0x143, 0xc, true>(x);       // row_bcast:31 into rows 2 and 3
  return x;
}
// value of the LAST lane of this lane's group, for every lane (the group total after sg_scan_add / sg_scan_max)
template <int SW> DEVFN double sg_last(double x, int sub) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  int rl = __builtin_amdgcn_readlane(lo, SW - 1), rh = __builtin_amdgcn_readlane(hi, SW - 1);
#pragma unroll
  for (int g = 1; g < 64 / SW; g++) {
    int l2 = __builtin_amdgcn_readlane(lo, (g + 1) * SW - 1), h2 = __builtin_amdgcn_readlane(hi, (g + 1) * SW - 1);
    if (sub == g) { rl = l2; rh = h2; }
  }
  return __hiloint2double(rh, rl);
}
// running maximum towards the last lane of the group (v_max_f64: NaN-ignoring; callers vote on NaNs separately)
template <int SW> DEVFN double sg_scan_max(double x) {
  x = vmax_f64(x, dpp_move<0x111, 0xf, false>(x));
  x = vmax_f64(x, dpp_move<0x112, 0xf, false>(x));
  x = vmax_f64(x, dpp_move<0x114, 0xf, false>(x));
  x = vmax_f64(x, dpp_move<0x118, 0xf, false>(x));
  if (SW >= 32) x = vmax_f64(x, dpp_move<0x142, 0xa, false>(x));
  if (SW >= 64) x = vmax_f64(x, dpp_move<0x143, 0xc, false>(x));
  return x;
}

template <int SW> DEVFN double sg_sum(double v) {
#pragma unroll
  for (int o = SW / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int SW> DEVFN double sg_max(double v) {
#pragma unroll
  for (int o = SW / 2; o > 0; o >>= 1) v = nanmax2(v, __shfl_xor(v, o, 64));
  return v;
}


// ------------------------------------------------------------------------------------------------------
// k_kde_marg_sub2<SW, IPW>: k_kde_marg_sub with IPW pixel groups of the same (event, draw) per wave, one after the other: the event
// statistics, the sample segments of all the wave's items and the grid-row addresses are fetched once (the first of the two memory
// round trips that make up 39 % of a wave's life in the one-item kernel -- phase timing, scripts/phase_prof.py).  Requesting the
// next item's samples ahead was measured too: the registers it holds across the grid loop cost more than the latency it hides.
// kde_sub_item is the one-item kernel's body from the p_cat prefetch on.
// ------------------------------------------------------------------------------------------------------
#ifndef CHM_NRS
#define CHM_NRS 256
#endif
#define CHM_WS_PAD 512           // doubles behind the (z, w) workspaces: the register rounds of the standard GW kernel read past a pixel's segment unconditionally
// Instruction-level helpers of the standard GW kernel.  On gfx950 every VALU instruction except the simplest 32-bit ones (v_mov_b32,
// v_add/sub_u32, v_and_b32, v_ashrrev_i32, v_fma/mul_f32) occupies the SIMD for 4 cycles per wave64 -- fp64 arithmetic, 64-bit moves,
// v_cndmask, v_med3, DPP moves and v_readlane alike (profiles/r03/issue_cost.txt) -- so the kernel is written for the fewest
// instructions, whatever their type:
//   * v_max_f64 / v_min_f64 as they are (llvm.maxnum/minnum put a canonicalising v_max_f64 x, x in front of every operand that was loaded);
//   * double -> bin index on v_cvt_i32_f64 (truncates, saturates at +-2^31, NaN -> 0) and ONE v_med3_i32 for the two-sided clamp
//   (vmax_f64, vmin_f64, cvt_i32_sat, med3_i32: chm_math.h).
// DPP move of a double with zeros shifted in (bound_ctrl: no separate v_mov of the fill value) -- row_shr steps of the scans
template <int CTRL>
DEVFN double dpp_shr0(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi2, lo2);
}
// inclusive prefix sum over each group of SW consecutive lanes, 3 instructions per level inside a row of 16 lanes
template <int SW> DEVFN double sg_scan_add0(double x) {
  x += dpp_shr0<0x111>(x);
  x += dpp_shr0<0x112>(x);
  x += dpp_shr0<0x114>(x);
  x += dpp_shr0<0x118>(x);
  if (SW >= 32) x += dpp_move<0x142, 0xa, true>(x);       // row_bcast:15 into rows 1 and 3
  if (SW >= 64) x += dpp_move<0x143, 0xc, true>(x);       // row_bcast:31 into rows 2 and 3
  return x;
}
// running maximum towards the last lane of the group for values >= 0 (redshifts): zeros shifted in are neutral, 3 instructions per level
template <int SW> DEVFN double sg_scan_max0(double x) {
  x = vmax_f64(x, dpp_shr0<0x111>(x));
  x = vmax_f64(x, dpp_shr0<0x112>(x));
  x = vmax_f64(x, dpp_shr0<0x114>(x));
  x = vmax_f64(x, dpp_shr0<0x118>(x));
  if (SW >= 32) x = vmax_f64(x, dpp_move<0x142, 0xa, true>(x));
  if (SW >= 64) x = vmax_f64(x, dpp_move<0x143, 0xc, true>(x));
  return x;
}
// value of the LAST lane of this lane's group on the LDS crossbar (ds_bpermute: no VALU slots; v_readlane + v_cndmask cost ten)
template <int SW> DEVFN double sg_last_perm(double x) {
  const int src = ((int)threadIdx.x | (SW - 1)) << 2;
  int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(x));
  return __hiloint2double(hi, lo);
}

// kde_sub_item<SW, NR, BINS, DUMP>: one pixel per group of SW lanes.  Preconditions checked by the host (chm_eval): binning with the
// effective grid cut (cut_grid set), an even number of grid points Z (16-byte pairs (k, k+1), k even, never leave the row).
// [r3] The grid loop was rewritten for the instruction count (191 -> ~110 VALU per pass of 2 x SW grid points):
//   * bin ranges of a node by truncation: ja = clamp(trunc(t - hb + 1), 0, jl1) equals ceil(t - hb) except when t - hb is an integer
//     (the bin with |u| = 1 exactly, kernel value 0), jb = clamp(trunc(t + hb + 1), ja, jl1) equals floor(t + hb) + 1 for t + hb >= 0 and
//     clamps to ja = 0 below (t + hb < 0 implies t - hb < 0): three instructions per index instead of seven;
//   * the prefix array of W c' is stored multiplied by -2 (exact), the node offsets of both nodes are per-pixel constants added to the
//     bin coordinate of the lower node, scale * norm * gw_pdf and the NaN of a degenerate pixel are one factor;
//   * the loads of a pass are unconditional (clamped pair index), both grid points of a lane sit in one exec region (k_hi is odd: event_stats),
//     NaN grid points propagate through the interpolation weight instead of a separate test.
// PRE (the fused event kernel, chm_fused.h): the pixel's histogram is already in Q[0, B) and its upper end `hi_pre` is known -- no sample is read
// here; the prefix arrays -2 P1 | P2 go to Q12 (2 (B + 1) doubles) instead of behind P0, the pixel's integral and rounding bound to out_like / out_err.
template <int SW, int NR, int BINS, bool DUMP, bool NT, bool PRE = false>
DEVFN void kde_sub_item(const LikeDev& L, const DevParams* params, double* Q, const double* es, const int b, const int e, const int p,
                        const int pp, const bool live, const bool poisoned, const int s0, const int s1, double (&zr)[NR], double (&wr)[NR],
                        const double hi_pre = 0., double* Q12 = nullptr, double* out_like_pre = nullptr, double* out_err_pre = nullptr, const int nit = NR,
                        unsigned long long* ph_prev_p = nullptr, const bool ph_on_ = false) {
#pragma clang fp contract(fast)                  // a*b+c may fuse in this body; the bin index lives in bin_index_r() (contract off)
  // [r5] FAST (the production instantiations: compile-time bin count, histogram formed here): the set-up sheds what the compiler had wrapped
  // round its arithmetic -- see the notes at each step (profiles/r05/ab_gw_setup_diet.txt)
  constexpr bool FAST = !PRE && BINS > 0;
  // [r6] SPLIT: THREE ds_read_b64 per bin index instead of ds_read2_b64 + ds_read_b64.  With the arrays of a pixel B + 1 = 201 doubles apart
  // the compiler fuses the reads of P0[i] and -2 P1[i] into one ds_read2_b64 (offset1:201) -- which the LDS serves at HALF the rate of two
  // ds_read_b64 (MI355X_MICROARCH.md, LDS: ds_read2_b64 = two accesses of 4 x 16 lanes, 8 cycles, 128 B/clk, banks mod 32; ds_read_b64 = 2 cycles,
  // 256 B/clk, banks mod 64): 10 LDS-array cycles per index where 6 do.  The kernel's LDS pipe was 0.67-0.70 busy beside a VALU at 0.73
  // (profiles/r05 PMC; the same on cache-resident data: profiles/r06/probe_pmc.txt).  Layout of the wave's slice with the three arrays of BOTH
  // pixels interleaved array-major, slots of QS doubles:  P0 a | P0 b | -2 P1 a | -2 P1 b | P2 a | P2 b  -- a pixel's arrays are 2 QS = 418 doubles
  // apart: beyond the 255-element reach of ds_read2_b64 and no multiple of 64 (ds_read2st64_b64), so the three reads stay three instructions at ONE
  // address register and immediate offsets.  QS = B + 1 + PERC + 1: the slot of P0 holds the zero padding the lanes beyond bin B read.
  // Same-box A/B with counters (profiles/r06/ab_lds_layouts_counters.txt): SQ_LDS_IDX_ACTIVE 1.471e9 -> 1.127e9 per launch, LDS pipe 0.68 -> 0.53 busy,
  // kernel 4.27 -> 4.17 ms (rocprofv3 average), 4.64 -> 4.48 ms (HIP events), bit-identical results.  (Packed rows {P0, -2 P1, P2} per bin -- the
  // layout the round-5 review asked for -- keep the ds_read2_b64 + ds_read_b64 pair and change nothing: same record; docs/history/ab_arms_r06.patch.)
  constexpr bool SPLIT = FAST;
  constexpr int QS = BINS > 0 ? BINS + 1 + (BINS + SW - 1) / SW + 1 : 0, QD = 2 * QS;
  static_assert(!SPLIT || (QD > 255 && QD % 64 != 0 && (2 * QD) % 64 != 0), "the three prefix arrays must be out of reach of the paired LDS reads");
  const int lane = threadIdx.x, sl = lane % SW;
  const int S = L.S, Z = L.Z, B = BINS > 0 ? BINS : L.num_bins, G = L.G;      // BINS > 0: the bin count is a compile-time constant (LDS offsets, loop bounds)
  const double zmin = es[0], norm = es[3], lb = es[6], ub = es[7];
  double* out_like = PRE ? out_like_pre : L.like_pix + ((size_t)b * L.E + e) * L.P + p;
  double* out_err = PRE ? out_err_pre : L.err_pix + ((size_t)b * L.E + e) * L.P + p;
  double* dump = (DUMP && p < L.P) ? L.p_gw_dump + (((size_t)b * L.E + e) * L.P + p) * Z : nullptr;
  const double* zg = L.z_grids + (size_t)e * Z;
  if (!live && p < L.P) { if (sl == 0) { *out_like = 0.; *out_err = 0.; } if (DUMP) for (int k = sl; k < Z; k += SW) dump[k] = 0.; }
  const double* pc = L.p_cat + ((size_t)e * L.P + pp) * Z;
  const size_t zo = ((size_t)b * L.E + e) * Z;
  const double* bkgA = L.bkgA + zo;
  const double* Aw = L.Aw + zo;
  const int k_lo = ((int)es[8]) & ~1, k_hi = (int)es[9];   // event_stats: k_hi is odd when Z is even -- both points of a pair are in range together
  // two consecutive grid points of every array for this lane: 16-byte loads at a clamped (always valid) pair index -- lanes beyond k_hi
  // load the row's last pair and never use it.  The three event-level rows are addressed as uniform base + 32-bit lane offset.
  const int kcap = Z - 2;
  struct Pass { double2 pc, z, bk, a; };                    // .x / .y: the lane's first / second grid point of the pass
  auto load_pass = [&](int k) {
    const unsigned off = (unsigned)(k < kcap ? k : kcap) * 8u;
    Pass q;
    // p_cat is read once per (event, pixel, call): few-draw calls (IPW = 2) stream it past the caches (non-temporal), so that the z / w the
    // sample stage has just written are still in the memory-side cache when this kernel asks for them; with many draws per call the
    // rows are shared by the draws' waves and stay cacheable
    const double2* pcp = reinterpret_cast<const double2*>(reinterpret_cast<const char*>(pc) + off);
    if (NT) { q.pc.x = __builtin_nontemporal_load(&pcp->x); q.pc.y = __builtin_nontemporal_load(&pcp->y); } else q.pc = *pcp;
    q.z = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(zg) + off);
    q.bk = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(bkgA) + off);
    q.a = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(Aw) + off);
    return q;
  };
  PHG(0);                                                   // (phase 0: the item's samples have arrived)
  // p_cat, grid, background and trapezoid factors of the first pass: in flight during the histogram phase
  const int k_first = k_lo + 2 * sl;
  Pass cur = load_pass(k_first);
  // histogram of the pixel's samples on [min z, max z in pixel] (math.py:32-46, likelihood.py:180-183)
  const size_t so = ((size_t)b * L.E + e) * S;
  const double* wz = L.ws_z + so;
  const double* ww = L.ws_w + so;
  const double lo = zmin;
  // hi = max(where(mask, z, min z)) (likelihood.py:180, math.py:36).  jnp.max propagates NaN: a NaN among the pixel's z is a NaN among
  // the event's z, and then lo -- jnp.min over all of them (combine_stats, NaN-propagating) -- is NaN already
  double hi = PRE ? hi_pre : lo;
  // FAST: the register rounds hold whatever stands at the lane's positions (k_kde_marg_sub2::run loads them unconditionally: the rows are padded);
  // `rem` = samples of the lane's pixel from its first position on -- round i holds one of them iff SW i < rem
  const int rem = s1 - (s0 + sl);
  if (!PRE) {
    // nit (uniform): register rounds that hold a sample of one of the wave's pixels -- the rounds beyond it were not loaded and are not touched
#pragma unroll
    for (int i = 0; i < NR; i++) if (i < nit) {
      if (FAST) { if (SW * i < rem) { asm volatile(""); hi = vmax_f64(hi, zr[i]); } }     // (the empty asm keeps this an exec-mask region: as selects it is two v_cndmask_b32 more)
      else hi = vmax_f64(hi, zr[i]);
    }
    if (nit >= NR) for (int s = s0 + sl + SW * NR; s < s1; s += SW) hi = vmax_f64(hi, wz[s]);
  }
  // P0 | -2 P1 | P2, (B + 1) doubles each behind one another; SPLIT: Q is the WAVE's slice, the pixel's slots are QS apart and its arrays QD
  double* const Q0 = SPLIT ? Q + (lane / SW) * QS : Q;
  double* const Q1 = SPLIT ? Q0 + QD : (PRE ? Q12 : Q + (B + 1));
  double* const Q2 = SPLIT ? Q0 + 2 * QD : (PRE ? Q12 + (B + 1) : Q + 2 * (B + 1));
  constexpr int PERC = BINS > 0 ? (BINS + SW - 1) / SW : 1;  // bins per lane (compile-time bin count)
  if (!PRE) {
    hi = sg_last_perm<SW>(sg_scan_max0<SW>(hi));       // z >= 0: z_from_dGW interpolates a table that starts at z = 0 (cosmo.py:43-46)
    if (lo != lo) hi = lo;
    if (FAST) {
      // PERC stores per lane at immediate offsets, no loop, no bound: [0, PERC SW) covers the B counts and runs into the first slots of the
      // -2 P1 array (B + 1 .. PERC SW - 1 of this pixel's slice), which the prefix pass below rewrites -- the lanes whose bins lie beyond B then
      // read zeros there without a test
      // (SPLIT: the stores beyond the slot of P0 -- indices QS .. PERC SW - 1 -- land in the first entries of the next slot: pixel b's own P0, which its
      //  lanes zero in the same instruction, or pixel a's -2 P1, rewritten by the prefix pass; the counts are read from at most index B + 1 + PERC - 1)
      static_assert(BINS <= 0 || PERC * SW <= 2 * (BINS + 1), "zero padding must stay inside the pixel's own prefix arrays");
      static_assert(!SPLIT || (PERC * SW - QS) <= QS, "zeroing must not run past the neighbouring slot");
#pragma unroll
      for (int i = 0; i < PERC; i++) Q0[sl + SW * i] = 0.;
    } else for (int j = sl; j < B; j += SW) Q0[j] = 0.;
    wave_sync();
  }
  const double dB = (double)B;
  const double dhl = hi - lo, rhl = 1. / dhl;
  const double dbin = dhl * L.inv_B;                        // c'_j = c_j - lo = (j + 1/2) dbin for the uniform edges of math.py:37-39
  if (!PRE) {
    if (FAST) {
      int bm1 = B - 1;
      asm volatile("" : "+s"(bm1));                       // the upper clamp of the bin index in ONE scalar register for all rounds
#pragma unroll
      for (int i = 0; i < NR; i++) if (i < nit) { if (SW * i < rem) { asm volatile(""); atomicAdd(&Q0[bin_index_rs(zr[i], lo, dhl, rhl, dB, bm1)], wr[i]); } }
    } else {
#pragma unroll
      for (int i = 0; i < NR; i++) if (i < nit) { int s = s0 + sl + SW * i; if (s < s1) atomicAdd(&Q0[bin_index_r(zr[i], lo, dhl, rhl, dB)], wr[i]); }
    }
    if (nit >= NR) for (int s = s0 + sl + SW * NR; s < s1; s += SW) atomicAdd(&Q0[bin_index_r(wz[s], lo, dhl, rhl, dB)], ww[s]);
    wave_sync();
  }
  PHG(1);                                                   // (phase 1: max z, zeroing, histogram -- and the first pass's loads, waited for here by the mark)
  // sums and prefix sums over the bins; every lane of the group owns `per` consecutive bins (a compile-time 7 for 200 bins on 32 lanes)
  const int per = (B + SW - 1) / SW;
  const int j0 = FAST ? sl * per : (sl * per < B ? sl * per : B), j1 = min(j0 + per, B);     // FAST: unclamped -- the lanes beyond B read the zero padding
  const int j0r = SPLIT ? min(j0, B + 1) : j0;              // SPLIT: ... which ends with the slot of P0: the lanes wholly beyond B read its last PERC entries
#ifndef CHM_MAXPER
#define CHM_MAXPER 8
#endif
  constexpr int MAXPER = BINS > 0 ? (BINS + SW - 1) / SW : CHM_MAXPER;      // bins per lane held in registers (8: up to 256 bins at 32 lanes per pixel)
  const bool small = per <= MAXPER;
  double wv[MAXPER];
  double s0w = 0., s1w = 0., s2w = 0., sq = 0.;
  // bin centres of the lane's bins by repeated addition, c'_{j+1} = c'_j + dbin (one instruction per bin instead of convert, add, multiply;
  // <= 8 roundings of 1e-16 relative on a centre, far below the bin width)
  const double c_first = ((double)j0 + 0.5) * dbin;
  if (small) {                                              // the lane's bin counts: all loads in flight at once, summed in bin order
#pragma unroll
    for (int i = 0; i < MAXPER; i++) wv[i] = FAST ? Q0[j0r + i] : ((j0 + i < j1) ? Q0[j0 + i] : 0.);      // FAST: zeros beyond B (padding above), no test
    double cc = c_first;
#pragma unroll
    for (int i = 0; i < MAXPER; i++) { const double w = wv[i], t = w * cc; s0w += w; s1w += t; s2w = fma(t, cc, s2w); sq = fma(w, w, sq); cc += dbin; }
  } else {
    for (int j = j0; j < j1; j++) { double w = Q0[j], cc = ((double)j + 0.5) * dbin; s0w += w; s1w += w * cc; s2w += w * cc * cc; sq += w * w; }
  }
  const double x0 = sg_scan_add0<SW>(s0w), x1 = sg_scan_add0<SW>(s1w), x2 = sg_scan_add0<SW>(s2w);
  const double tot = sg_last_perm<SW>(x0);
  const double sum2 = sg_last_perm<SW>(sg_scan_add0<SW>(sq));
  // End of the last lane chunk of bins that holds any weight.  The prefix values of the lanes after it come out of different
  // summation trees and agree only to an ulp: a node that sees nothing but the empty bins above the data would get 1e-16 of the
  // peak where the dense sum (math.py:80) has an exact zero -- which decides log L_i when the catalogue term is only non-zero out
  // there.  The bin ranges are clipped to it (below the first weight every prefix is an exact zero already).
  int jl1;
  {
    const int sub = lane / SW;
    const unsigned long long nz = __ballot(s0w != 0.);      // NaN counts as weight
    const unsigned long long mine = SW == 64 ? nz : ((nz >> (sub * (SW & 63))) & ((1ull << (SW & 63)) - 1ull));
    const int last = 63 - __clzll(mine);                    // -1: no weight at all (degenerate pixel, NaN below)
    jl1 = min((last + 1) * per, B);
  }
  {
    double r0 = x0 - s0w, r1 = x1 - s1w, r2 = x2 - s2w;
    wave_sync();
    if (small) {
      double cc = c_first;
#pragma unroll
      for (int i = 0; i < MAXPER; i++) {
        const double w = wv[i], t = w * cc;
        r0 += w; r1 += t; r2 = fma(t, cc, r2); cc += dbin;
        // FAST: bin j0 + i exists iff the lane lies below the one that holds bin B (all of its PERC bins) or is that lane and i < B mod PERC:
        // two lane predicates for the PERC stores instead of an add and a compare for each
        const bool st = FAST ? (sl < B / PERC || (sl == B / PERC && i < B % PERC)) : (j0 + i < j1);
        if (st) { Q0[j0 + i + 1] = r0; Q1[j0 + i + 1] = -2. * r1; Q2[j0 + i + 1] = r2; }
      }
    } else {                                                // many bins per lane: the count of bin j+1 shares the slot of P0[j+1]
      double a0 = r0, a1 = r1, a2 = r2;
      for (int j = j0; j < j1; j++) { double w = Q0[j], cc = ((double)j + 0.5) * dbin; a1 += w * cc; a2 += w * cc * cc; Q1[j + 1] = -2. * a1; Q2[j + 1] = a2; }
      a0 = r0; for (int j = j0; j < j1; j++) a0 += Q0[j];
      // the first store below lands in the NEXT lane's first slot (Q0[j1]), whose count that lane has summed in the loops above and will read once
      // more only for a value it never uses: every lane's reads of the counts are complete here, and stay in front of the stores
      wave_sync();
      for (int j = j1 - 1; j >= j0; j--) { double w = Q0[j]; Q0[j + 1] = a0; a0 -= w; }
    }
    if (sl == 0) { Q1[0] = 0.; Q2[0] = 0.; }
    wave_sync();
    if (sl == 0) Q0[0] = 0.;
    wave_sync();
  }
  PHG(2);                                                   // (phase 2: bin sums, four scans, prefix stores)
  // (quotients that feed smooth arithmetic only: reciprocal seed + two Newton steps instead of the IEEE division sequence)
  const double neff_k = chm_div(tot * tot, sum2);
  const double stdc = dhl * L.std_unit;
  const double bw = kde_bandwidth_factor_fast(L.bw_method, L.bw_scalar, neff_k) * stdc;
  const bool degenerate = !(dbin > 0.) || !(bw > 0.) || !(bw < 1e300) || !(tot > 0. || tot < 0.);
  // Per-pixel constants of the node evaluation.  A node g of the effective grid sees the bins [ja, jb) with |g - c_j| <= h,
  // c_j = lo + (j + 1/2) dbin:  ja = ceil(t - hb), jb = floor(t + hb) + 1 with t = (g - lo)/dbin - 1/2, hb = h/dbin; the next
  // node is t + dd, dd = de/dbin.  A bin whose |u| is within rounding of 1 may land on either side (kernel value < 1e-12).
  // one reciprocal serves 1/bw and the normalisation 3/4 / (bw sum w); 1/dbin = B / (hi - lo) re-uses the histogram's reciprocal
  const double rbt = chm_div(1., bw * tot);
  const double inv_dbin = dB * rhl, inv_bw = rbt * tot, m_inv_bw2 = -(inv_bw * inv_bw);
  const double hb = bw * inv_dbin;
  const double de = es[10], inv_de = es[11];                // spacing of jnp.linspace(lb, ub, G) and its inverse (k_event_prep)
  const double dd = de * inv_dbin;
  const double dG2 = (double)(G - 2);
  const double lbl = lb - lo;
  const double nan = __builtin_nan("");
  const double ng = norm * L.gw_pdf[(size_t)e * L.P + pp]; // kde_interp * norm * gw_pdf[i]    likelihood.py:194
  const double scale = 0.75 * rbt;
  const double sng = degenerate ? nan : scale * ng;         // degenerate pixels give NaN wherever the interpolant is evaluated
  const double fR = params[b].fR;
  // bin coordinate of node 0 and the offsets of the four truncations (lower node: t - hb + 1, t + hb + 1; upper node: + dd)
  const double t0 = fma(lbl, inv_dbin, -0.5);
  const double oa1 = 1. - hb, oa2 = 1. + hb, ob1 = dd + oa1, ob2 = dd + oa2;
  // Support of THIS pixel's interpolated KDE on the event grid: the bin centres lie in [lo + dbin/2, hi - dbin/2], a node sees
  // none of them beyond bw, and an event-grid point combines the two nodes within de of it -- so p_gw is an exact zero for
  // z outside (lo - bw - de, hi + bw + de), typically a third of the event's range [lb, ub] (every pixel's histogram starts
  // at the event's min z but ends at the pixel's own max z, likelihood.py:180).  Degenerate pixels keep [lb, ub] (NaN there).
  const double zlo = degenerate ? lb : __builtin_fmax(lb, lo - bw - de), zhi = degenerate ? ub : __builtin_fmin(ub, hi + bw + de);
  // density (without the common factor `scale`) at the node with g' = g - lo whose bin range comes from the truncations of xa, xb
  auto node = [&](double gp, double xa, double xb) {
    const int ia = med3_i32_0v(cvt_i32_sat(xa), jl1);
    const int ib = med3_i32(cvt_i32_sat(xb), ia, jl1);
    const double S0 = Q0[ib] - Q0[ia], S1 = Q1[ib] - Q1[ia], S2 = Q2[ib] - Q2[ia];   // S1 = -2 sum W c'
    const double qq = fma(gp, fma(gp, S0, S1), S2);         // sum W (g' - c')^2 over the support
    return __builtin_fmax(fma(m_inv_bw2, qq, S0), 0.);      // a sum of non-negative kernel values (rounding may leave -1e-14 of the peak)
  };
  // A-posteriori bound on what the prefix-sum form can have lost.  Every prefix value carries <= 12 roundings (7 additions in the lane,
  // 5 scan levels) relative to |P0| <= T, |P1| <= T R, |P2| <= T R^2 (T = sum w, R = hi - lo >= c'); a node's value
  // S0 - (g'^2 S0 - 2 g' S1 + S2)/h^2 with |g'| <= R + h is therefore within  errD = 24 eps T (1 + (2 R/h + 1)^2)  of the dense sum
  // (math.py:77-81), and so is the interpolant; the pixel's integral sum_k p_gw[k] C_k (C_k = (fR p_cat + bkgA) A_k) within
  // errD sum_k |C_k| over the grid points inside the pixel's support (outside, p_gw is an exact zero in either form).  The bound goes
  // to err_pix; k_marg_fixup compares it with the event's L_i and redoes the pixels that matter with the general kernel's dense sums
  // (weights spanning many decades, a gap inside the data: the catalogue term sits where the KDE is far below its peak).
  const double rh = dhl * inv_bw;
  const double errD = (24. * 1.1102230246251565e-16) * fabs(tot) * (1. + (2. * rh + 1.) * (2. * rh + 1.));
  double acc = 0., accC = 0.;
  if (DUMP && live) { for (int k = sl; k < Z; k += SW) if (k < k_lo || k > k_hi) dump[k] = 0.; }
  // p_gw at a grid point = interpolant of the two effective-grid nodes that bracket it (0 outside the pixel's support); each point takes its
  // own two nodes.
  // Bracket of z: nodes x_i = lb + i de, i = floor((z - lb)/de) in [0, G - 1] inside the support (lb <= zlo, zhi <= ub); z = ub lands on the
  // last node with weight 0 (jnp.interp's fp[-1]); a z within rounding of a node may pick either neighbouring segment -- the interpolant is
  // continuous there.  A NaN grid point counts as inside and comes out NaN through the interpolation weight.
  auto interp = [&](const double zrel, const double tp, const double da, const double db) {
    const double wgt = fma(-tp, de, zrel) * inv_de;         // (z - x_a)/dx
    return fma(wgt, db - da, da) * sng;
  };
  auto single = [&](const double zk) {                      // one point on its own two nodes
    const double zrel = zk - lb;
    const double tp = floor(zrel * inv_de);
    const double ga = fma(tp, de, lbl);                     // x_a - lo
    const double ta = fma(tp, dd, t0);                      // its bin coordinate (x_a - lo)/dbin - 1/2
    return interp(zrel, tp, node(ga, ta + oa1, ta + oa2), node(ga + de, ta + ob1, ta + ob2));
  };
  // integrand and bound.  The bound sums |C_k| over every unmasked grid point of [k_lo, k_hi], not only those inside the pixel's support:
  // looser by the share of the event's range the pixel does not cover (a factor ~1.5), one addition instead of a comparison-dependent select.
  auto integrand = [&](const int kk, const double pgw, const double pcv, const double bk, const double ak, const bool inrange = true) {
    if (DUMP) { if (inrange) dump[kk] = pgw; }
    // (the empty volatile asm keeps this a branch on the exec mask: as selects it is four v_cndmask_b32 per point)
    if (inrange && pcv != -100.) { asm volatile(""); const double cz = fma(fR, pcv, bk) * ak; acc = fma(pgw, cz, acc); accC += fabs(cz); }   // catalog.py:202, likelihood.py:275
  };
  // [r5] a pass in which no lane's FIRST point lies at or below its pixel's zhi ends the loop: the grid ascends, so every later point of either
  // pixel is beyond the support too (p_gw = 0 exactly: nothing to add to the integral; the rounding bound then sums |C_k| over [k_lo, here),
  // which still covers the support).  A NaN grid point or a NaN zhi keeps the loop going.  Not in the instantiation that stores p_gw.
  constexpr bool EXIT_EARLY = FAST && !DUMP;
  auto do_pass = [&](const int k, const Pass& q) -> bool {
    // (a lane beyond k_hi holds the row's last pair and may keep the loop going, which ends at k_hi anyway.  Folding `live` into zhi -- ONE compare under
    //  the vote instead of the mask rebuilt through a VGPR, two instructions fewer per pass -- costs two spilled registers at the 128-register cap.)
    if (EXIT_EARLY && !wave_any(live && !(q.z.x > zhi))) return false;
    if (k <= k_hi && live) {
      const double z0 = q.z.x, z1 = q.z.y;
      const bool in0 = !(z0 < zlo) && !(z0 > zhi), in1 = !(z1 < zlo) && !(z1 > zhi);
      double pg0 = 0., pg1 = 0.;
      if (in0 || in1) {
        if (in0) pg0 = single(z0);
        if (in1) pg1 = single(z1);
      }
      integrand(k, pg0, q.pc.x, q.bk.x, q.a.x);
      integrand(k + 1, pg1, q.pc.y, q.bk.y, q.a.y);
    }
    return true;
  };
  PHG(3);                                                   // (phase 3: bandwidth and the pixel's constants)
  // one pass = SW lanes x 2 consecutive grid points per pixel.  Software pipeline in two alternating register sets: the loads of the next
  // pass are issued before the arithmetic of this one and waited for where that pass begins (no register rotation, no wait at the loop end)
  for (int kb = k_lo; kb <= k_hi; kb += 4 * SW) {
    const int k = kb + 2 * sl;
    const Pass nxt = load_pass(k + 2 * SW);
    if (!do_pass(k, cur)) break;                            // uniform
    if (kb + 2 * SW > k_hi) break;                          // uniform
    cur = load_pass(k + 4 * SW);
    if (!do_pass(k + 2 * SW, nxt)) break;
  }
  // (same-address LDS atomics for these reductions -- ds_max_f64 / ds_add_f64 on one cell per pixel, no VALU slots -- were measured:
  //  5.53 instead of 4.43 ms for the kernel, the LDS pipe serialises the 32 lanes of every such instruction)
  PHG(4);                                                   // (phase 4: the grid loop)
  acc = sg_scan_add0<SW>(acc);                              // the group's last lane holds the pixel's integral
  accC = sg_scan_add0<SW>(accC);
  if (sl == SW - 1 && live) {
    *out_like = poisoned ? nan : acc;
    *out_err = (degenerate || poisoned) ? 0. : errD * accC * fabs(scale * ng);     // NaN results stay NaN: nothing to redo
  }
  PHG(5);                                                   // (phase 5: the two final scans and the stores)
}

// (the kernel's body with the block coordinates as arguments: scripts/gw_loop_probe.hip replays it on a cache-resident workload)
template <int SW, int IPW, int BINS, bool DUMP>
DEVFN void kde_marg_sub2_body(const LikeDev& L, const DevParams* params, const int bx, const int by, const int bz, double* lds_all) {
  constexpr int NPW = 64 / SW;
  constexpr int NR = CHM_NRS / SW;
  const int lane = threadIdx.x, sub = lane / SW, sl = lane % SW;
  const int PG = (L.P + NPW - 1) / NPW, H = (PG + IPW - 1) / IPW;   // pixel groups of an event; the wave's items: by + i H
  const int b = bx, e = L.e_off + bz;
  const int Z = L.Z;
  double* Q = BINS > 0 ? lds_all : lds_all + (size_t)sub * (3 * L.num_bins + 3);      // (compile-time bin count: the wave's slice, laid out by kde_sub_item -- SPLIT)
  const double* es = L.evstat + ((size_t)b * L.E + e) * NEVSTAT;
  const bool ok = es[4] >= L.pe_neff;                       // likelihood.py:199 (same for every pixel of the event)
  const int* so_ = L.seg_off + (size_t)e * (L.P + 1);
  // every item's sample segment is requested up front, together with the event statistics (one memory round trip)
  const int pA = by * NPW + sub, pB = (by + H) * NPW + sub;
  const int pC = (by + 2 * H) * NPW + sub, pD = (by + 3 * H) * NPW + sub;
  const int ppA = pA < L.P ? pA : L.P - 1, ppB = pB < L.P ? pB : L.P - 1, ppC = pC < L.P ? pC : L.P - 1, ppD = pD < L.P ? pD : L.P - 1;
  const int a0 = so_[ppA], a1 = so_[ppA + 1], b0 = so_[ppB], b1 = so_[ppB + 1];
  int c0 = 0, c1 = 0, d0 = 0, d1 = 0;
  if (IPW > 2) { c0 = so_[ppC]; c1 = so_[ppC + 1]; d0 = so_[ppD]; d1 = so_[ppD + 1]; }
  const int npx = L.neff_pixels[e];
  const double* zg = L.z_grids + (size_t)e * Z;
  const bool poisoned = grid_is_poisoned(params[b].z_bad, zg, Z);      // (a call with an infinite rate parameter -- event_poisoned -- never takes this kernel)
  const size_t so = ((size_t)b * L.E + e) * L.S;
  const double* wz = L.ws_z + so;
  const double* ww = L.ws_w + so;
  const double lo = es[0];
#ifdef CHM_PHASE_PROF
  unsigned long long ph_clock = clock64();
  const bool ph_on_k = threadIdx.x == 0 && (bx & 63) == 0 && (bz & 15) == 0;
  if (ph_on_k) atomicAdd(&g_phase[7], 1ull);
  unsigned long long* const ph_ptr = &ph_clock;
#else
  unsigned long long* const ph_ptr = nullptr; const bool ph_on_k = false;
#endif
  auto run = [&](const int pgi, const int p, const int pp, const int q0, const int q1, const bool first) {
    if (pgi >= PG) return;                                  // uniform
    const bool live = p < L.P && p < npx;
    if (!ok || !wave_any(live)) {                              // uniform: every pixel of the event (of this item: padded pixels) is 0 (or 0 * NaN)
      if (p < L.P) {
        if (sl == 0) { L.like_pix[((size_t)b * L.E + e) * L.P + p] = (live && poisoned) ? __builtin_nan("") : 0.; L.err_pix[((size_t)b * L.E + e) * L.P + p] = 0.; }
        if (DUMP) { double* d = L.p_gw_dump + (((size_t)b * L.E + e) * L.P + p) * Z; for (int k = sl; k < Z; k += SW) d[k] = 0.; }
      }
      return;
    }
    if (!first) wave_sync();                                // the next item reuses the wave's LDS slice
    const int s0 = q0, s1 = live ? q1 : q0;
    // [r4] rounds of SW samples per pixel actually needed by this wave's pixels (uniform): the mean segment holds 125 of the 256 samples the NR
    // register rounds cover -- rounds beyond the longer of the wave's segments are neither loaded nor binned (5.1 of 8 on average at C3)
    int nmax = 0;
#pragma unroll
    for (int g = 0; g < NPW; g++) nmax = max(nmax, __builtin_amdgcn_readlane(s1 - s0, g * SW));
    const int nit = min((nmax + SW - 1) / SW, NR);
    double zr[NR], wr[NR];                                   // the register rounds of this item: dead after its histogram
    if (BINS > 0) {
      // [r5] unconditional loads at ONE 32-bit lane offset + an immediate per round (uniform row base in scalar registers): what stands behind the
      // pixel's segment is the next pixel's samples or the CHM_WS_PAD doubles behind the workspace, and kde_sub_item uses round i only where
      // SW i < s1 - (s0 + sl).  Before: per round a compare, two 64-bit address computations and two moves of the neutral values.
      const size_t boff = (size_t)((unsigned)(s0 + sl) * 8u);
#pragma unroll
      for (int j = 0; j < NR; j++) if (j < nit) {
        zr[j] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(wz) + boff + (size_t)(8 * SW * j));
        wr[j] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(ww) + boff + (size_t)(8 * SW * j));
      }
    } else {
#pragma unroll
      for (int j = 0; j < NR; j++) if (j < nit) { int s = s0 + sl + SW * j; zr[j] = s < s1 ? wz[s] : lo; wr[j] = s < s1 ? ww[s] : 0.; }
    }
    kde_sub_item<SW, NR, BINS, DUMP, (IPW <= 2)>(L, params, Q, es, b, e, p, pp, live, poisoned, s0, s1, zr, wr, 0., nullptr, nullptr, nullptr, nit, ph_ptr, ph_on_k);
  };
  run(by, pA, ppA, a0, a1, true);
  run(by + H, pB, ppB, b0, b1, false);
  if (IPW > 2) { run(by + 2 * H, pC, ppC, c0, c1, false); run(by + 3 * H, pD, ppD, d0, d1, false); }
}
template <int SW, int IPW, int BINS, bool DUMP>
__global__ void __launch_bounds__(64, 4) k_kde_marg_sub2(LikeDev L, const DevParams* params) {
  extern __shared__ double lds_all[];
  CLK_BEGIN;
  kde_marg_sub2_body<SW, IPW, BINS, DUMP>(L, params, blockIdx.x, blockIdx.y, blockIdx.z, lds_all);
  CLK_END(0);
}

// ------------------------------------------------------------------------------------------------------
// k_kde1d: p_gw1d (likelihood.py:105-144): one block (256 threads) per (event, draw) -> pgw1d (nb,E,Z)
// ------------------------------------------------------------------------------------------------------
// Dynamic LDS: data[N], wgt[N], hw[3*N] (per-wave private histograms / prefix arrays), eff[G], dens[G]
__global__ void __launch_bounds__(256) k_kde1d(LikeDev L, const DevParams* params) {
  extern __shared__ double lds[];
  __shared__ double red[16];
  __shared__ int s_jl1;
  const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wid = t >> 6;
  const int e = L.e_off + blockIdx.x, b = blockIdx.y;
  const int S = L.S, Z = L.Z, B = L.num_bins, G = L.G;
  const int N = L.binning ? B : S;
  double* data = lds; double* wgt = data + N;
  double* hw = wgt + N;                                     // binning: 3*(N+1) doubles
  double* eff = hw + (L.binning ? 3 * (N + 1) : 0); double* dens = eff + G;
  const size_t so = ((size_t)b * L.E + e) * S;
  const double* wz = L.ws_z + so;
  const double* ww = L.ws_w + so;
  double* out = L.pgw1d + ((size_t)b * L.E + e) * Z;
  const EvStats st = combine_stats(L.part + ((size_t)b * L.E + e) * L.NC * NPART, L.NC, S);
  const bool ok = st.n_eff >= L.pe_neff;                    // likelihood.py:133
  int* kr = L.krange + ((size_t)b * L.E + e) * 2;
  if (!ok) { for (int k = t; k < Z; k += nt) out[k] = 0.; if (t == 0) { kr[0] = 0; kr[1] = -1; } return; }
  const double lo = st.zmin, hi = st.zmax;
  if (L.binning) {
    // per-wave private histograms (deterministic: each wave owns every 4th chunk of 64 samples), then a fixed-order sum
    for (int j = t; j < 3 * (N + 1); j += nt) hw[j] = 0.;
    for (int j = t; j < B; j += nt) {
      double e0 = jnp_linspace_at(lo, hi, B + 1, j), e1 = jnp_linspace_at(lo, hi, B + 1, j + 1);
      data[j] = (e0 + e1) / 2.;
      wgt[j] = 0.;
    }
    __syncthreads();
    double* mine = wid == 0 ? wgt : hw + (size_t)(wid - 1) * N;
    const double dhl = hi - lo, rhl = 1. / dhl, dB = (double)B;       // same bins as bin_index(), one division per event (see bin_index_r)
    for (int s = t; s < S; s += nt) atomicAdd(&mine[bin_index_r(wz[s], lo, dhl, rhl, dB)], ww[s]);
    __syncthreads();
    for (int j = t; j < B; j += nt) wgt[j] = ((wgt[j] + hw[j]) + hw[N + j]) + hw[2 * N + j];
    __syncthreads();
  } else {
    for (int s = t; s < S; s += nt) { data[s] = wz[s]; wgt[s] = ww[s]; }
    __syncthreads();
  }
  double a = 0.;
  for (int j = t; j < N; j += nt) a += wgt[j];
  const double tot = block_reduce<RED_SUM>(a, red);
  a = 0.;
  double c = 0.;
  for (int j = t; j < N; j += nt) { double W = wgt[j] / tot; wgt[j] = W; a += W * W; c += data[j]; }
  const double neff_k = 1.0 / block_reduce<RED_SUM>(a, red);
  const double meanc = block_reduce<RED_SUM>(c, red) / (double)N;
  a = 0.;
  for (int j = t; j < N; j += nt) { double d = data[j] - meanc; a += d * d; }
  const double stdc = sqrt(block_reduce<RED_SUM>(a, red) / (double)N);
  const double bw = kde_bandwidth_factor(L.bw_method, L.bw_scalar, neff_k, 1) * stdc;
  double lb = 0., ub = 0.;
  if (L.has_cut) {
    eff_bounds(false, st.zmin, st.zmax, st.sd, L.cut_grid, lb, ub);
    for (int i = t; i < G; i += nt) eff[i] = jnp_linspace_at(lb, ub, G, i);
  } else {
    for (int i = t; i < G; i += nt) eff[i] = L.z_grids[(size_t)e * Z + i];
  }
  __syncthreads();
  const bool epan = L.kernel == 0;
  const double inv_bw = 1. / bw;
  const double dbin = (hi - lo) / (double)B;
  const bool fast = epan && L.binning && dbin > 0. && bw > 0. && bw < 1e300 && tot == tot && (tot > 0. || tot < 0.) && !L.neg_w;
  if (fast) {
    double* P0 = hw; double* P1 = hw + (N + 1); double* P2 = hw + 2 * (N + 1);
    if (wid == 0) { int j = wave_prefix3(data, wgt, N, lo, P0, P1, P2); if (lane == 0) s_jl1 = j; }
    __syncthreads();
    const int jl1 = s_jl1;
    for (int i = t; i < G; i += nt) dens[i] = epan_prefix_eval(eff[i], data, wgt, P0, P1, P2, N, lo, 1. / dbin, bw, inv_bw, lo, jl1) * st.norm;
  } else {
    for (int i = t; i < G; i += nt) dens[i] = kde_dense_eval(eff[i], data, wgt, N, epan, bw, inv_bw) * st.norm;
  }
  __syncthreads();
  // kde*norm interpolated to the event grid, left = right = 0 (likelihood.py:137)
  double kmn = 1e300, kmx = -1.;
  for (int k = t; k < Z; k += nt) {
    double zk = L.z_grids[(size_t)e * Z + k];
    // left = right = 0 outside the effective grid: most of the event grid, decided before any bracket search
    double v = (zk < eff[0] || zk > eff[G - 1]) ? 0. : interp_lr0(eff, dens, G, zk, eff_guess(zk, lb, ub, G, L.has_cut, k));
    out[k] = v;
    if (v != 0.) { kmn = fmin(kmn, (double)k); kmx = fmax(kmx, (double)k); }     // NaN counts as non-zero
  }
  kmn = block_reduce<RED_MIN>(kmn, red); kmx = block_reduce<RED_MAX>(kmx, red);
  if (t == 0) { kr[0] = kmx >= 0. ? (int)kmn : 0; kr[1] = (int)kmx; }
}

// k_integrate_1d: one block (four waves) per (event, draw); a wave walks over the event's pixels p = wave, wave + 4, ...:
// p_gw3dapprox (likelihood.py:150-154) or the 1-D case, integrand and trapezoid (likelihood.py:266-292).  (One wave per
// (event, pixel, draw) was dispatch-bound: 2 M waves of five short passes each took 1.46 ms at C3 / 64 draws.)
__global__ void __launch_bounds__(256) k_integrate_1d(LikeDev L, const DevParams* params) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int Pd = L.P > 0 ? L.P : 1;
  const int b = blockIdx.x % L.nb, e = L.e_off + blockIdx.x / L.nb;    // draw fastest: the draws of an event share its p_cat rows in L2
  const DevParams& P = params[b];
  const int Z = L.Z;
  const bool pixelated = L.mode != 0;
  const size_t zo = ((size_t)b * L.E + e) * Z;
  // [r5] a draw with an infinite rate parameter (P.rate_special: its per-z factors hold inf / NaN, k_rate_special): the products are formed over the
  // WHOLE grid as the reference forms them (likelihood.py:274-278, 288-291) -- 0 * inf = NaN where p_gw1d vanishes, +inf where every infinite
  // factor meets a positive p_gw1d (a Gaussian kernel without cut_grid: found by scripts/fuzz_parity.py, seed 6001464) -- instead of the blanket
  // NaN of event_poisoned()
  const bool special = P.rate_special != 0;
  const bool poisoned = special ? grid_is_poisoned(P.z_bad, L.z_grids + (size_t)e * Z, Z) : event_poisoned(L, P, b, e, L.z_grids + (size_t)e * Z, Z);
  const double* g1 = L.pgw1d + zo;
  // p_gw1d vanishes outside [k_lo, k_hi] (found by k_kde1d): those terms of the trapezoid are exact zeros and are skipped;
  // inside, sum_k p_gw3d[k] p_z[k]/jac[k] tw[k] with the per-z factors folded into A[k] (see k_kde_marg)
  const int* kr = L.krange + ((size_t)b * L.E + e) * 2;
  const int k_lo = special ? 0 : kr[0], k_hi = special ? Z - 1 : kr[1];
  const double* bkgA = L.bkgA + zo;
  const double* Aw = L.Aw + zo;
  const double fR = P.fR;
  const int npix = pixelated ? L.neff_pixels[e] : 1;
  // The event's own factors on [k_lo, k_hi] are the same for all its pixels: when the range fits IG_NP passes of 64 lanes
  // (512 grid points) they are loaded once into registers, and a pixel costs one batch of independent p_cat loads.
  constexpr int IG_NP = 8;
  const bool short_range = k_hi - k_lo < 64 * IG_NP;
  double gv[IG_NP], bv[IG_NP], av[IG_NP];
#pragma unroll
  for (int i = 0; i < IG_NP; i++) {
    const int k = k_lo + 64 * i + lane;
    const bool in = short_range && k <= k_hi;
    gv[i] = in ? g1[k] : 0.; bv[i] = in ? bkgA[k] : 0.; av[i] = in ? Aw[k] : 0.;
  }
  for (int p = wid; p < Pd; p += nw) {
    double* out_like = L.like_pix + ((size_t)b * L.E + e) * Pd + p;
    double* dump = (L.p_gw_dump && pixelated) ? L.p_gw_dump + (((size_t)b * L.E + e) * Pd + p) * Z : nullptr;
    if (pixelated && p >= npix) {                           // padded pixel: p_cat == -100, masked to 0 (likelihood.py:274-277)
      if (lane == 0) *out_like = 0.;
      if (dump) for (int k = lane; k < Z; k += 64) dump[k] = 0.;
      continue;
    }
    const double gwp = pixelated ? L.gw_pdf[(size_t)e * L.P + p] : 1.;
    const double* pc = pixelated ? L.p_cat + ((size_t)e * L.P + p) * Z : nullptr;
    if (dump) for (int k = lane; k < Z; k += 64) dump[k] = (k >= k_lo && k <= k_hi) ? g1[k] * gwp : 0.;
    double acc = 0.;
    if (short_range) {
      double pv[IG_NP];
#pragma unroll
      for (int i = 0; i < IG_NP; i++) { const int k = k_lo + 64 * i + lane; pv[i] = (pixelated && k <= k_hi) ? pc[k] : 0.; }
#pragma unroll
      for (int i = 0; i < IG_NP; i++) {
        double pgw = pixelated ? gv[i] * gwp : gv[i];        // p_gw1d[:,None,:] * gw_loc2d_pdf[:,:,None]   likelihood.py:153
        if (pixelated) { if (pv[i] != -100.) acc += pgw * (fR * pv[i] + bv[i]) * av[i]; }   // catalog.py:202, pop_wrapper.py:87, likelihood.py:275
        else acc += pgw * bv[i] * av[i];                     // pop_wrapper.py:89, likelihood.py:291
      }
    } else {
      for (int k = k_lo + lane; k <= k_hi; k += 64) {
        double pgw = pixelated ? g1[k] * gwp : g1[k];
        if (pixelated) {
          double pcv = pc[k];
          if (pcv != -100.) acc += pgw * (fR * pcv + bkgA[k]) * Aw[k];
        } else {
          acc += pgw * bkgA[k] * Aw[k];
        }
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) *out_like = poisoned ? __builtin_nan("") : acc;
  }
}

// sum(W^2), W = w / sum(w), of one event (math.py:173-176 normalise the weights first).  The partial sums carry sum(w) and sum(w^2) un-normalised:
// below sum(w) ~ 1e-140 (a mass model that puts its whole weight many widths away from every sample: found by scripts/fuzz_parity.py with
// lambda_peak = 1, sigma_g = 0.5) w^2 underflows and sum(w)^2 with it, while the reference's normalised weights are fine -- the block then sums
// (w / sum w)^2 over the event's samples.  Uniform over the block (every thread combines the same partials).
DEVFN double full_mode_sW2(const double* part, int NC, const double* ww, int S, double* red) {
  double sw = 0., sw2 = 0.;
  for (int c = 0; c < NC; c++) { sw += part[(size_t)c * NPART + PT_SW]; sw2 += part[(size_t)c * NPART + PT_SW2]; }
  if (sw >= 1e-140 || !(sw > 0.)) return sw2 / (sw * sw);
  double a = 0.;
  for (int s = threadIdx.x; s < S; s += blockDim.x) { const double r = ww[s] / sw; a += r * r; }
  return block_reduce<RED_SUM>(a, red);
}

// ------------------------------------------------------------------------------------------------------
// k_full_kde: 3-D Gaussian KDE, one block (256 threads) per (event, pixel, draw)   likelihood.py:211-260, math.py:154-229
// ------------------------------------------------------------------------------------------------------
// val(q_k) = sum_j W_j exp(log_norm - 1/2 |x_j - q_k|^2) in whitened coordinates.  The queries of one pixel differ only in z,
// and whitening with the lower Cholesky factor maps z to the FIRST whitened coordinate only:
//   |x_j - q_k|^2 = (a_j - t_k)^2 + b_j,   a_j = (x_j L)_0,  t_k = z_k l00 + ra_p l10 + dec_p l20,  b_j = k-independent,
// so  val_k = sum_j c_j g_jk,  c_j = W_j exp(log_norm - b_j/2)  (one exp per sample and pixel),  g_jk = exp(-(a_j - t_k)^2/2).
// On a uniform stretch of the event grid (t_i = t_0 + i D, d = a - t_0) the Gaussian factorises,
//   g_ji = exp(-d^2/2) u^i exp(-i^2 D^2/2),   u = exp(d D),
// and the last factor is common to all samples: a thread accumulates the power sums sum_j c_j g_j0 u_j^i of FULL_LK = 32 grid points
// (two exps with bounded arguments per sample and chunk: chm_exp_nb; then 4 fma + 1 multiply per 4 pairs against u, u^2, u^3, u^4) and
// multiplies grid point i by exp(-i^2 D^2/2) once per chunk.  Relative error <= ~12 eps from the powers + 3e-14 from the exps; the
// sums stay in range: |d D| <= 18 and i^2 D^2/2 <= 113 for a chunk that spans <= 15 kernel widths.
// Round 2 before this form: the recurrence g <- g r, r <- r rho (3 instructions per pair, 192 evaluations/s at C3); an LDS table of
// exp(-D^2 i (i-1)/2) is hoisted into 64 registers by the compiler and spills (175/s); 16 points per chunk 140/s; 48: spills, 69/s.
// Staging U_j = exp((a_j - t_first) D) per sample so that u = U_j exp(-32 c D^2) needs no exp per chunk: 240 vs 247/s (8 spilled registers).
// A chunk whose grid is not uniform to 1e-11 falls back to one exp per pair.
#define FULL_TILE 1024
#ifndef FULL_LK
#define FULL_LK 32            // grid points a thread marches per sample from one pair of exps (16: the exps cost as much as the march)
#endif
#ifndef FULL_NSMAX
#define FULL_NSMAX 128
#endif
#define FULL_RH 8             // grid points of a chunk reduced through LDS at a time
#ifndef FULL_MINW
#define FULL_MINW 4           // [r3] 128 VGPRs: the 16 spilled registers sit outside the march (272 against 260 evaluations/s at C3 / 4 draws)
#endif
__global__ void __launch_bounds__(256, FULL_MINW) k_full_kde(LikeDev L, const DevParams* params, const int* todo) {
  if (todo && !todo[((size_t)blockIdx.y * L.E + L.e_off + blockIdx.x / L.P) * L.P + blockIdx.x % L.P]) return;      // done by k_full_kde_chain
  __shared__ double sa[FULL_TILE], sc[FULL_TILE];
  __shared__ double racc[256 * FULL_RH];                   // per-thread partial sums of a pass, FULL_RH grid points at a time (blockDim.x = 256)
  __shared__ double chd[256];                              // per chunk of a pass: D^2 / 2 of its uniform grid (the common factors of the march), < 0: none
  __shared__ double red[16];
  __shared__ double wh[12];
  const int t = threadIdx.x, nt = blockDim.x;
  const int p = blockIdx.x % L.P, e = L.e_off + blockIdx.x / L.P, b = blockIdx.y;
  const DevParams& P = params[b];
  const int S = L.S, Z = L.Z;
  const size_t so = ((size_t)b * L.E + e) * S;
  const size_t eo = (size_t)e * S;
  const double* wz = L.ws_z + so;
  const double* ww = L.ws_w + so;
  double* out_like = L.like_pix + ((size_t)b * L.E + e) * L.P + p;
  double* dump = L.p_gw_dump ? L.p_gw_dump + (((size_t)b * L.E + e) * L.P + p) * Z : nullptr;
  if (p >= L.neff_pixels[e]) {                            // result[ev, :npix] only (likelihood.py:253)
    if (t == 0) *out_like = 0.;
    if (dump) for (int k = t; k < Z; k += nt) dump[k] = 0.;
    return;
  }
  const double* part = L.part + ((size_t)b * L.E + e) * L.NC * NPART;
  const EvStats st = combine_stats(part, L.NC, S);
  const bool ok = !(st.n_eff < L.pe_neff);                // `if n_effs[ev] < pe_neff: continue`   likelihood.py:234
  const double sW2_ev = full_mode_sW2(part, L.NC, L.ws_w + ((size_t)b * L.E + e) * S, S, red);
  if (t == 0) {
    // weighted mean / covariance of (z, ra, dec) (math.py:173-190) from the shifted un-normalised moments, then
    // inv_cov / factor^2 and its lower Cholesky factor (math.py:191-195)
    double sw = 0., sw2 = 0., a[3] = {0., 0., 0.}, m[6] = {0., 0., 0., 0., 0., 0.};
    for (int c = 0; c < L.NC; c++) {
      const double* q = part + (size_t)c * NPART;
      sw += q[PT_SW]; sw2 += q[PT_SW2];
      a[0] += q[PT_WD0]; a[1] += q[PT_WD1]; a[2] += q[PT_WD2];
      m[0] += q[PT_W00]; m[1] += q[PT_W01]; m[2] += q[PT_W02]; m[3] += q[PT_W11]; m[4] += q[PT_W12]; m[5] += q[PT_W22];
    }
    (void)sw2;
    double sW2 = sW2_ev;                                  // sum(W^2), W = w / sum(w)  (full_mode_sW2)
    double m0 = a[0] / sw, m1 = a[1] / sw, m2 = a[2] / sw;
    double den = 1. - sW2;
    double c00 = (m[0] / sw - m0 * m0) / den, c01 = (m[1] / sw - m0 * m1) / den, c02 = (m[2] / sw - m0 * m2) / den;
    double c11 = (m[3] / sw - m1 * m1) / den, c12 = (m[4] / sw - m1 * m2) / den, c22 = (m[5] / sw - m2 * m2) / den;
    double neff = 1. / sW2;
    double factor = kde_bandwidth_factor(L.bw_method, L.bw_scalar, neff, 3);
    double a00 = c11 * c22 - c12 * c12, a01 = c02 * c12 - c01 * c22, a02 = c01 * c12 - c02 * c11;
    double a11 = c00 * c22 - c02 * c02, a12 = c01 * c02 - c00 * c12, a22 = c00 * c11 - c01 * c01;
    double det = c00 * a00 + c01 * a01 + c02 * a02;
    double f2 = factor * factor;
    double i00 = a00 / det / f2, i01 = a01 / det / f2, i02 = a02 / det / f2;
    double i11 = a11 / det / f2, i12 = a12 / det / f2, i22 = a22 / det / f2;
    double l00 = sqrt(i00), l10 = i01 / l00, l20 = i02 / l00;
    double l11 = sqrt(i11 - l10 * l10), l21 = (i12 - l20 * l10) / l11;
    double l22 = sqrt(i22 - l20 * l20 - l21 * l21);
    wh[0] = l00; wh[1] = l10; wh[2] = l11; wh[3] = l20; wh[4] = l21; wh[5] = l22;
    wh[6] = (chm_log(l00) + chm_log(l11) + chm_log(l22)) - 0.5 * 3. * chm_log(2. * CHM_PI);      // log_norm  math.py:215
  }
  __syncthreads();
  const double l00 = wh[0], l10 = wh[1], l11 = wh[2], l20 = wh[3], l21 = wh[4], l22 = wh[5], log_norm = wh[6];
  const double zhi = st.zmax + L.cut_grid * st.sd, zlo = st.zmin - L.cut_grid * st.sd;      // z mask, likelihood.py:225
  const double* zg = L.z_grids + (size_t)e * Z;
  const double rp = L.ra_pix[(size_t)e * L.P + p], dp = L.dec_pix[(size_t)e * L.P + p];
  // whitened query (math.py:196): q = (z, ra_p, dec_p) . L
  const double q1 = rp * l11 + dp * l21, q2 = dp * l22, t_base = rp * l10 + dp * l20;
  const size_t zo = ((size_t)b * L.E + e) * Z;
  const double* pc = L.p_cat + ((size_t)e * L.P + p) * Z;

  // Only the grid points inside the mask [zlo, zhi] carry a KDE value (likelihood.py:225-226, 248-250); they form one
  // contiguous stretch [k_first, k_last] of the event grid, and only that stretch is chunked -- with the whole grid half the
  // threads held chunks outside the mask and idled while the others worked.
  double kf = 1e300, kl = -1.;
  for (int k = t; k < Z; k += nt) { double z = zg[k]; if (z <= zhi && z >= zlo) { kf = fmin(kf, (double)k); kl = fmax(kl, (double)k); } }
  kf = block_reduce<RED_MIN>(kf, red); kl = block_reduce<RED_MAX>(kl, red);
  const int k_first = kl >= 0. ? (int)kf : 0, k_last = (int)kl;
  // chunks of FULL_LK grid points; NS threads share a chunk and split the samples (any NS: their partial sums meet in LDS)
  const int nch = k_last >= k_first ? (k_last - k_first + FULL_LK) / FULL_LK : 0;
  int NS = nch > 0 ? nt / nch : 1;
  NS = NS < 1 ? 1 : (NS > FULL_NSMAX ? FULL_NSMAX : NS);
  const int cpp = nt / NS;                                // chunks per pass
  double accl = 0.;
  // outside the mask the reference's row is kde_vals (zeros) * norm (likelihood.py:250-253): 0 * norm -- NaN for an event whose mean weight is NaN or
  // inf (a NaN mass, prior or distance among its samples), from end to end; an event skipped by the n_eff guard keeps its zeros
  const double zero_n = ok ? 0. * st.norm : 0.;
  if (dump) for (int k = t; k < Z; k += nt) if (!ok || k < k_first || k > k_last) dump[k] = zero_n;
  const double inv_sumw = 1. / st.sumw;
  for (int cb = 0; cb < nch && ok; cb += cpp) {
    const int c = cb + t / NS, sl = t % NS;
    const bool has = c < nch && t < cpp * NS;
    const int k0 = has ? k_first + c * FULL_LK : 0;
    const int nk = has ? min(FULL_LK, k_last + 1 - k0) : 0;
    // the chunk's grid in the first whitened coordinate; uniform?
    double t0 = 0., D = 0.;
    bool uni = false, any = false;
    if (has) {
      double z0 = zg[k0], z1 = zg[k0 + nk - 1];
      double dz = nk > 1 ? (z1 - z0) / (double)(nk - 1) : 0.;
      uni = nk > 1;
      for (int i = 0; i < nk; i++) {
        double z = zg[k0 + i];
        if (fabs(z - (z0 + (double)i * dz)) > 1e-11 * fabs(dz)) uni = false;
        if (z <= zhi && z >= zlo) any = true;
      }
      t0 = z0 * l00 + t_base; D = dz * l00;
      if (!(fabs(D) * (double)(FULL_LK - 1) <= 15.)) uni = false;      // the chunk spans more than 15 kernel widths: one exp per pair (no recurrence)
    }
    const double hD2 = 0.5 * D * D;
    double acc[FULL_LK];
#pragma unroll
    for (int i = 0; i < FULL_LK; i++) acc[i] = 0.;
    for (int s0 = 0; s0 < S; s0 += FULL_TILE) {
      __syncthreads();
      const int ns = min(FULL_TILE, S - s0);
      for (int s = t; s < ns; s += nt) {                  // stage a_j and c_j = W_j exp(log_norm - b_j/2) of this tile
        double x0 = wz[s0 + s], x1 = L.ra[eo + s0 + s], x2 = L.dec[eo + s0 + s];
        double d1 = (x1 * l11 + x2 * l21) - q1, d2 = x2 * l22 - q2;
        sa[s] = x0 * l00 + x1 * l10 + x2 * l20;
        sc[s] = (ww[s0 + s] * inv_sumw) * chm_exp_nb(log_norm - 0.5 * (d1 * d1 + d2 * d2));      // (no range checks: a huge negative argument ends in v_ldexp_f64's 0)
      }
      __syncthreads();
      if (has && any) {
        if (uni) {
          for (int s = sl; s < ns; s += NS) {
            double d = sa[s] - t0;
            const double e1 = -0.5 * (d * d);
            // a sample more than 37 kernel widths from the chunk's first point is left out: exp(e1) underflows (and the step
            // factor may overflow: 0 * inf); the chunk spans <= 15 widths, so it stays >= 22 widths from every point of the chunk,
            // i.e. it would add < 1e-105 of its own weight
            // (both arguments are bounded: -700 < e1 <= 0, |d D - D^2/2| <= 37.5 x 15/31 + 1: the exps need no range checks)
            const bool in = e1 > -700.;
            double pw = in ? sc[s] * chm_exp_nb(e1) : 0.;
            const double u = in ? chm_exp_nb(d * D) : 0.;
            const double u2 = u * u, u3 = u2 * u, u4 = u2 * u2;
#pragma unroll
            for (int i = 0; i < FULL_LK; i += 4) {
              acc[i] += pw;
              acc[i + 1] = __builtin_fma(pw, u, acc[i + 1]);
              acc[i + 2] = __builtin_fma(pw, u2, acc[i + 2]);
              acc[i + 3] = __builtin_fma(pw, u3, acc[i + 3]);
              pw *= u4;
            }
          }

        } else {
          // one exp per pair (a chunk that is not uniform, or spans many kernel widths): the grid points one after the other -- a rolled
          // loop, so that the 32 exps do not all live in registers at once; the sum of point i reaches acc[i] through a compile-time
          // indexed select (acc[] must stay in registers)
#pragma unroll 1
          for (int i = 0; i < nk; i++) {
            const double ti = zg[k0 + i] * l00 + t_base;
            double a = 0.;
            for (int s = sl; s < ns; s += NS) { const double d = sa[s] - ti; a += sc[s] * chm_exp(-0.5 * (d * d)); }
#pragma unroll
            for (int j = 0; j < FULL_LK; j++) acc[j] += (j == i) ? a : 0.;
          }
        }
      }
    }
    // a marched chunk holds sum_j c_j g_j0 u_j^i in acc[i]: the common factor exp(-i^2 D^2 / 2) of grid point i completes the Gaussians.
    // [r3] It is applied once per grid point AFTER the slices' partial sums have met (below), not by each of the NS threads of the chunk to
    // its 32 partial sums (32 exps per thread: 6 % of the kernel's instructions); chd[chunk] = D^2 / 2, or < 0 for a chunk without the factor.
    __syncthreads();
    if (has && sl == 0) chd[t / NS] = (any && uni) ? hD2 : -1.;
    // the NS partial sums of every grid point meet in LDS (FULL_RH points of each chunk at a time); then one thread per grid point
    // forms p_gw and the integrand
#pragma unroll
    for (int h0 = 0; h0 < FULL_LK; h0 += FULL_RH) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < FULL_RH; i++) racc[i * 256 + t] = acc[h0 + i];
      __syncthreads();
      const int npts = min(cpp, nch - cb) * FULL_RH;
      for (int idx = t; idx < npts; idx += nt) {
        const int cl = idx / FULL_RH, i = idx % FULL_RH;
        const int k = k_first + (cb + cl) * FULL_LK + h0 + i;
        if (k > k_last) continue;
        double v = 0.;
        for (int q = 0; q < NS; q++) v += racc[i * 256 + cl * NS + q];
        { const double hq = chd[cl]; const int ii = h0 + i; if (hq >= 0.) v *= chm_exp(-hq * (double)(ii * ii)); }
        const double z = zg[k];
        const bool inm = (z <= zhi) && (z >= zlo);
        double pgw = inm ? v * st.norm : zero_n;          // kde_vals[eff_mask] ... * norm   likelihood.py:252-253
        if (dump) dump[k] = pgw;
        double pcv = pc[k];
        double y = 0.;
        if (pcv != -100.) {
          double p_gal = P.fR * pcv + L.bkgA[zo + k];
          double p_z = p_gal * L.prate[zo + k];
          y = (p_z != -100.) ? pgw * p_z / L.jac[zo + k] : 0.;
        }
        // trapezoid: y_k enters the two adjacent intervals
        double zl = k > 0 ? zg[k - 1] : z, zr = k < Z - 1 ? zg[k + 1] : z;
        accl += y * ((z - zl) + (zr - z));
      }
    }
  }
  accl = block_reduce<RED_SUM>(accl, red);
  // (+ zero_n: the grid points outside the mask enter the reference's trapezoid as 0 * norm * p_z / jac -- NaN when the event's mean weight is)
  if (t == 0) *out_like = event_poisoned(L, P, b, e, zg, Z) ? __builtin_nan("") : 0.5 * accl + zero_n;   // NaN factors on the grid: 0 * NaN
}

// ------------------------------------------------------------------------------------------------------
// k_full_kde_chain [r3]: the same 3-D KDE + integrand, SAMPLE-stationary   likelihood.py:211-260, math.py:154-229
// ------------------------------------------------------------------------------------------------------
// k_full_kde hands a chunk of 32 grid points to a group of threads and pays two exps per (sample, chunk) to start the march of the power sums
// (3.7 VALU instructions per (sample, grid point) pair, 1.25 of them the march).  On a stretch of the grid that is uniform from end to end the
// starting values of chunk c+1 follow from those of chunk c without an exp: with t_c = t_0 + c LK D, d_c = a_j - t_c,
//   c_j exp(-d_{c+1}^2 / 2) = [c_j exp(-d_c^2 / 2) u_c^LK] K1,   u_{c+1} = exp(d_{c+1} D) = u_c K2,   K1 = exp(-(LK D)^2 / 2), K2 = exp(-LK D^2),
// and the bracket is what the march leaves in its running product.  So here a thread OWNS up to FULLC_SPT samples (their (pw, u) pairs stay in
// registers), walks the chunks of the stretch one after the other, and the block adds the 256 threads' power sums of a chunk through LDS: one
// wave's 64 x LK sums per round, every thread adding eight of them to its running partial sum of (grid point t/8, slot t%8), three DPP steps at
// the end -- a fixed order, the result does not depend on the schedule.  Two exps per sample and block instead of two per sample and chunk.
// A sample further than 37 kernel widths from the first point of the stretch (its Gaussian underflows there) starts with (0, 0) and is
// looked at again before every chunk (a bit per owned sample; re-read and started with two exps once it is inside 37 widths of the chunk's
// first point: it was >= 22 widths from every point of the chunks it missed, which span <= 15 widths each).
// Blocks this form does not cover (a stretch that is not uniform to 1e-11 of its step, a chunk spanning > 15 kernel widths, more than
// FULLC_NPT grid points inside the mask) are flagged in todo[] and done by k_full_kde, which skips the others.  Events of more than
// 256 x FULLC_SPT samples are walked in sets of that many.
// Round-off: the running product takes LK/4 + 1 roundings per chunk and 4 x the step factor's (one more per chunk): <~ 1e-13 relative at the
// far end of a 250-point stretch (test tolerance on p_gw and L_i: 1e-9).
#ifndef FULLC_LK
#define FULLC_LK 32
#endif
// [r5] block shape (build knobs, A/B in profiles/r05/ab_full_mode_r05.txt): FULLC_NW waves per block, a thread keeps FULLC_SPT samples in registers
// (4096 per block and walk over the stretch either way).  4 x 16: 128 + 64 accumulators fit three waves per SIMD; 2 x 32: two waves per SIMD, the
// cross-lane exchange and the accumulators' reset once per 1024 instead of once per 512 (sample, grid point) pairs of a lane.
#ifndef FULLC_NW
#define FULLC_NW 4
#endif
#ifndef FULLC_SPT
#define FULLC_SPT (64 / FULLC_NW)
#endif
#ifndef FULLC_MINW
#define FULLC_MINW (FULLC_SPT > 16 ? 2 : 3)
#endif
#ifndef FULLC_NPT
#define FULLC_NPT 1024        // grid points of the stretch (longer: general kernel)
#endif
#ifndef FULLC_PB
#define FULLC_PB 4            // samples of a thread whose values are requested together at the block's start, two such sets at a time (divides FULLC_SPT)
#endif
// (measured in round 5 and dropped, profiles/r05/ab_full_mode_r05.txt; text: docs/history/ab_arms_r06.patch -- the running product one group ahead (equal), sixteen
//  points x one half of the wave per exchange (44 instead of 68 VALU per chunk: 8.41-8.53 against 8.26-8.30 ms), the march without its exchange (a timing diagnostic))
#define FULLC_ROW 72          // doubles per grid point in the exchange buffer: 64 lanes + 8 (the eight points a wave reads fall in distinct banks)
// per (draw, event) record of k_full_prep (doubles): what the pixels of an event share
enum { FE_L00 = 0, FE_L10, FE_L11, FE_L20, FE_L21, FE_L22, FE_LOGNORM, FE_ZLO, FE_ZHI, FE_NORM, FE_OK, FE_KFIRST, FE_KLAST, FE_CHAIN, FE_D, FE_K1, FE_K2,
       FE_AREF, FE_ZF, FE_CF /* FULLC_LK factors */, FULLEV = FE_CF + FULLC_LK };

// k_full_prep [r3]: once per (draw, event) what k_full_kde_chain's blocks (one per pixel) would each derive again -- the whitening of the event's
// samples (math.py:173-196), the masked stretch of its grid (likelihood.py:222-226) and whether it is uniform, the constants of the march --
// and per sample the whitened coordinates (a, y1, y2), the normalised weight and U = exp((a - a_ref) D): the step factor of a sample in a pixel
// is exp((a - t_0p) D) = U V_p with one exp per PIXEL, V_p = exp((a_ref - t_0p) D)  (a_ref: the stretch's first point at the mean ra, dec).
__global__ void __launch_bounds__(256) k_full_prep(LikeDev L) {
  constexpr int LK = FULLC_LK;
  __shared__ double red[16];
  __shared__ double wh[12];
  const int t = threadIdx.x, nt = 256;
  const int e = L.e_off + blockIdx.x, b = blockIdx.y;
  const int S = L.S, Z = L.Z;
  const size_t so = ((size_t)b * L.E + e) * S;
  const size_t eo = (size_t)e * S;
  const double* wz = L.ws_z + so;
  const double* ww = L.ws_w + so;
  double* fe = L.full_ev + ((size_t)b * L.E + e) * FULLEV;
  const double* part = L.part + ((size_t)b * L.E + e) * L.NC * NPART;
  const EvStats st = combine_stats(part, L.NC, S);
  const bool ok = !(st.n_eff < L.pe_neff);                // `if n_effs[ev] < pe_neff: continue`   likelihood.py:234
  const double sW2_ev = full_mode_sW2(part, L.NC, ww, S, red);
  if (t == 0) {
    // weighted covariance of (z, ra, dec), inv_cov / factor^2, its lower Cholesky factor and log_norm (math.py:173-195, 215): as in k_full_kde
    double sw = 0., sw2 = 0., a[3] = {0., 0., 0.}, m[6] = {0., 0., 0., 0., 0., 0.};
    for (int c = 0; c < L.NC; c++) {
      const double* q = part + (size_t)c * NPART;
      sw += q[PT_SW]; sw2 += q[PT_SW2];
      a[0] += q[PT_WD0]; a[1] += q[PT_WD1]; a[2] += q[PT_WD2];
      m[0] += q[PT_W00]; m[1] += q[PT_W01]; m[2] += q[PT_W02]; m[3] += q[PT_W11]; m[4] += q[PT_W12]; m[5] += q[PT_W22];
    }
    (void)sw2;
    double sW2 = sW2_ev;
    double m0 = a[0] / sw, m1 = a[1] / sw, m2 = a[2] / sw;
    double den = 1. - sW2;
    double c00 = (m[0] / sw - m0 * m0) / den, c01 = (m[1] / sw - m0 * m1) / den, c02 = (m[2] / sw - m0 * m2) / den;
    double c11 = (m[3] / sw - m1 * m1) / den, c12 = (m[4] / sw - m1 * m2) / den, c22 = (m[5] / sw - m2 * m2) / den;
    double neff = 1. / sW2;
    double factor = kde_bandwidth_factor(L.bw_method, L.bw_scalar, neff, 3);
    double a00 = c11 * c22 - c12 * c12, a01 = c02 * c12 - c01 * c22, a02 = c01 * c12 - c02 * c11;
    double a11 = c00 * c22 - c02 * c02, a12 = c01 * c02 - c00 * c12, a22 = c00 * c11 - c01 * c01;
    double det = c00 * a00 + c01 * a01 + c02 * a02;
    double f2 = factor * factor;
    double i00 = a00 / det / f2, i01 = a01 / det / f2, i02 = a02 / det / f2;
    double i11 = a11 / det / f2, i12 = a12 / det / f2, i22 = a22 / det / f2;
    double l00 = sqrt(i00), l10 = i01 / l00, l20 = i02 / l00;
    double l11 = sqrt(i11 - l10 * l10), l21 = (i12 - l20 * l10) / l11;
    double l22 = sqrt(i22 - l20 * l20 - l21 * l21);
    wh[0] = l00; wh[1] = l10; wh[2] = l11; wh[3] = l20; wh[4] = l21; wh[5] = l22;
    wh[6] = (chm_log(l00) + chm_log(l11) + chm_log(l22)) - 0.5 * 3. * chm_log(2. * CHM_PI);
    wh[7] = L.ra[eo] + m1; wh[8] = L.dec[eo] + m2;        // weighted mean ra, dec (the moments are taken about the event's first sample)
  }
  __syncthreads();
  const double l00 = wh[0], l10 = wh[1], l11 = wh[2], l20 = wh[3], l21 = wh[4], l22 = wh[5];
  const double zhi = st.zmax + L.cut_grid * st.sd, zlo = st.zmin - L.cut_grid * st.sd;      // z mask, likelihood.py:225
  const double* zg = L.z_grids + (size_t)e * Z;
  double kf = 1e300, kl = -1.;
  for (int k = t; k < Z; k += nt) { double z = zg[k]; if (z <= zhi && z >= zlo) { kf = fmin(kf, (double)k); kl = fmax(kl, (double)k); } }
  kf = block_reduce<RED_MIN>(kf, red); kl = block_reduce<RED_MAX>(kl, red);
  const int k_first = kl >= 0. ? (int)kf : 0, k_last = (int)kl;
  const int npt = k_last - k_first + 1;
  // is the stretch uniform, and a chunk of it no wider than 15 kernel widths?
  const double z_f = zg[k_first];
  const double dz = npt > 1 ? (zg[k_last] - z_f) / (double)(npt - 1) : 0.;
  double dev = 0.;
  for (int k = k_first + t; k <= k_last; k += nt) dev = fmax(dev, fabs(zg[k] - (z_f + (double)(k - k_first) * dz)));
  dev = block_reduce<RED_MAX>(dev, red);
  const double D = dz * l00;
  const bool chain = ok && npt > 1 && dev <= 1e-11 * fabs(dz) && fabs(D) * (double)LK <= 15. && npt <= FULLC_NPT;
  const double a_ref = z_f * l00 + wh[7] * l10 + wh[8] * l20;
  if (t < LK) fe[FE_CF + t] = chm_exp(-0.5 * D * D * (double)(t * t));      // the factor of grid point i of a chunk that is common to all samples
  if (t == 0) {
    for (int i = 0; i < 7; i++) fe[i] = wh[i];
    fe[FE_ZLO] = zlo; fe[FE_ZHI] = zhi; fe[FE_NORM] = st.norm; fe[FE_OK] = ok ? 1. : 0.;
    fe[FE_KFIRST] = (double)k_first; fe[FE_KLAST] = (double)k_last; fe[FE_CHAIN] = chain ? 1. : 0.;
    fe[FE_D] = D; fe[FE_K1] = chm_exp(-0.5 * (D * (double)LK) * (D * (double)LK)); fe[FE_K2] = chm_exp(-(double)LK * D * D);
    fe[FE_AREF] = a_ref; fe[FE_ZF] = z_f * l00;
  }
  if (!chain) return;
  const double inv_sumw = 1. / st.sumw;
  double* fa = L.full_s + so; double* fy1 = fa + (size_t)L.nb_alloc * L.E * S; double* fy2 = fy1 + (size_t)L.nb_alloc * L.E * S;
  double* fw = fy2 + (size_t)L.nb_alloc * L.E * S; double* fu = fw + (size_t)L.nb_alloc * L.E * S;
  for (int s = t; s < S; s += nt) {
    const double x0 = wz[s], x1 = L.ra[eo + s], x2 = L.dec[eo + s];
    const double a = x0 * l00 + x1 * l10 + x2 * l20;
    fa[s] = a; fy1[s] = x1 * l11 + x2 * l21; fy2[s] = x2 * l22; fw[s] = ww[s] * inv_sumw;
    fu[s] = chm_exp((a - a_ref) * D);                      // (NaN / inf coordinates propagate as they do through the direct form)
  }
}

__global__ void __launch_bounds__(64 * FULLC_NW, FULLC_MINW) k_full_kde_chain(LikeDev L, const DevParams* params, int* todo) {
  constexpr int LK = FULLC_LK;
  __shared__ double xw[FULLC_NW][8 * FULLC_ROW];                  // per wave: eight grid points x 64 lanes of power sums on their way across the lanes
  __shared__ double vw[FULLC_NW][FULLC_NPT];                      // per wave: its samples' sums at every grid point of the stretch
  __shared__ double red[16];
  __shared__ double wh[8];
  constexpr int nt = 64 * FULLC_NW;
  const int t = threadIdx.x;
  const int p = blockIdx.x % L.P, e = L.e_off + blockIdx.x / L.P, b = blockIdx.y;
  const DevParams& P = params[b];
  const int S = L.S, Z = L.Z;
  const size_t so = ((size_t)b * L.E + e) * S;
  PH_INIT;                                                  // (diagnostic builds: scripts/phase_full.py)
  int* my_todo = todo + ((size_t)b * L.E + e) * L.P + p;
  double* out_like = L.like_pix + ((size_t)b * L.E + e) * L.P + p;
  double* dump = L.p_gw_dump ? L.p_gw_dump + (((size_t)b * L.E + e) * L.P + p) * Z : nullptr;
  // [r5] everything the block's start waits for is requested here, before the first branch -- the pixel count, the pixel's direction, the event's
  // record and the first FULLC_PB owned samples' values: one memory latency, where the branches below had put four one after the other
  // (a wave spent half its life starting up: scripts/phase_full.py, profiles/r05/phase_full_kernel.txt)
  const double* fa = L.full_s + so; const double* fy1 = fa + (size_t)L.nb_alloc * L.E * S; const double* fy2 = fy1 + (size_t)L.nb_alloc * L.E * S;
  const double* fw = fy2 + (size_t)L.nb_alloc * L.E * S; const double* fu = fw + (size_t)L.nb_alloc * L.E * S;
  constexpr int PB = FULLC_PB, NG = FULLC_SPT / PB;
  static_assert(FULLC_SPT % PB == 0, "FULLC_PB divides FULLC_SPT");
  double bA[5][PB], bB[5][PB];                              // two sets of PB samples' (a, y1, y2, weight, U): one in use, one in flight
#define FULLC_LOADB(dst, sb_, g_) _Pragma("unroll") for (int jj = 0; jj < PB; jj++) { \
      const int s_ = min((sb_) + t + ((g_) * PB + jj) * nt, S - 1);      /* (beyond the event's last sample: a valid address, the values unused) */ \
      dst[0][jj] = fa[s_]; dst[1][jj] = fy1[s_]; dst[2][jj] = fy2[s_]; dst[3][jj] = fw[s_]; dst[4][jj] = fu[s_]; }
  FULLC_LOADB(bA, 0, 0)
  const int npix_e = L.neff_pixels[e];
  const double rp = L.ra_pix[(size_t)e * L.P + p], dp = L.dec_pix[(size_t)e * L.P + p];
  const double* fe = L.full_ev + ((size_t)b * L.E + e) * FULLEV;      // k_full_prep's record of the event (uniform address: scalar loads)
  double fr[FE_CF];                                         // its scalars (the FULLC_LK factors behind them are read per grid point at the end)
#pragma unroll
  for (int i = 0; i < FE_CF; i++) fr[i] = fe[i];
  if (p >= npix_e) {                                       // result[ev, :npix] only (likelihood.py:253)
    if (t == 0) { *out_like = 0.; *my_todo = 0; }
    if (dump) for (int k = t; k < Z; k += nt) dump[k] = 0.;
    return;
  }
  const double* zg = L.z_grids + (size_t)e * Z;
  const int k_first = (int)fr[FE_KFIRST], k_last = (int)fr[FE_KLAST];
  const int npt = k_last - k_first + 1;
  const double zero_n = fr[FE_OK] != 0. ? 0. * fr[FE_NORM] : 0.;      // the reference's 0 * norm outside the mask (see k_full_kde)
  if (fr[FE_OK] == 0. || npt <= 0) {                       // nothing to integrate: the general kernel's answer for these, without it
    if (t == 0) { *out_like = grid_is_poisoned(P.z_bad, zg, Z) ? __builtin_nan("") : zero_n; *my_todo = 0; }
    if (dump) for (int k = t; k < Z; k += nt) dump[k] = zero_n;
    return;
  }
  if (fr[FE_CHAIN] == 0.) { if (t == 0) *my_todo = 1; return; }        // (k_full_kde writes the whole pixel, dump included)
  if (t == 0) *my_todo = 0;
  if (dump) for (int k = t; k < Z; k += nt) if (k < k_first || k > k_last) dump[k] = zero_n;
  const double q1 = rp * fr[FE_L11] + dp * fr[FE_L21], q2 = dp * fr[FE_L22], t_base = rp * fr[FE_L10] + dp * fr[FE_L20];      // whitened query (math.py:196)
  const size_t zo = ((size_t)b * L.E + e) * Z;
  const double* pc = L.p_cat + ((size_t)e * L.P + p) * Z;
  const double D = fr[FE_D], K1 = fr[FE_K1], K2 = fr[FE_K2];
  const double t0 = fr[FE_ZF] + t_base;
  // the pixel's factor of the step factors, u = U V; a pixel further than 30 / D widths from the mean direction forms them directly
  const double varg = (fr[FE_AREF] - t0) * D;
  const bool direct0 = !(fabs(varg) <= 30.);
  const double V = direct0 ? 0. : chm_exp(varg);
  // what a sample's start needs sits in LDS: after the first chunk it is a rare path whose constants would occupy registers of the march
  if (t == 0) { wh[0] = q1; wh[1] = q2; wh[2] = fr[FE_LOGNORM]; wh[3] = D; wh[4] = V; }
  __syncthreads();
  PH(0);                                                    // (phase 0: the event's record, the pixel's constants, the first samples' values)

  // one sample's starting values at a chunk whose first point sits at tc.  1: started; 0: further than 37 widths ahead (stays (0, 0), looked at
  // again at the next chunk); -1: further than 37 widths behind -- the chunks move away from it, it never enters
  auto start_v = [&](double a, double y1, double y2, double w, double us, double tc, bool direct, double& pw, double& u) -> int {
    const double d1 = y1 - wh[0], d2 = y2 - wh[1];
    const double d = a - tc;
    const double e1 = -0.5 * (d * d), Dl = wh[3];
    const bool in = e1 > -700.;
    // one exp for W_j exp(log_norm - b_j / 2) exp(-d^2 / 2) (a huge negative argument ends in v_ldexp_f64's 0); |d D| <= 37.5 x 15 / LK
    pw = in ? w * chm_exp_nb(fmax(wh[2] + e1 - 0.5 * (d1 * d1 + d2 * d2), -800.)) : 0.;
    if (direct) u = in ? chm_exp_nb(d * Dl) : 0.;
    else u = in ? us * wh[4] : 0.;
    return in ? 1 : (d * Dl > 0. ? 0 : -1);
  };
  auto start = [&](int s, double tc, bool direct, double& pw, double& u) -> int {
    return start_v(fa[s], fy1[s], fy2[s], fw[s], fu[s], tc, direct, pw, u);
  };
  const int lane = t & 63, wv = t >> 6;
  double* xb = xw[wv];
  double* vrow = vw[wv];
  const int nch = (npt + LK - 1) / LK;
  // events of more than 256 x FULLC_SPT samples: one set of 4096 samples after the other, a wave's sums of the later sets added to its row
  for (int sb = 0; sb < S; sb += nt * FULLC_SPT) {
  const int spt = (min(S - sb, nt * FULLC_SPT) + nt - 1) / nt;
  double pw[FULLC_SPT], uu[FULLC_SPT];
  unsigned waiting = 0;                                    // bit j: owned sample j has not started yet
  // [r5] the values of the next FULLC_PB owned samples are requested before the starting values of the present ones are formed
  // (bA holds the set's first FULLC_PB samples: requested at the top of the kernel, or at the end of the set before)
#define FULLC_STARTB(src, g_) _Pragma("unroll") for (int jj = 0; jj < PB; jj++) { \
      const int j = (g_) * PB + jj; \
      double np, nu; \
      const int r = start_v(src[0][jj], src[1][jj], src[2][jj], src[3][jj], src[4][jj], t0, direct0, np, nu); \
      const bool own = sb + t + j * nt < S; \
      pw[j] = own ? np : 0.; uu[j] = own ? nu : 0.; \
      if (own && r == 0) waiting |= 1u << j; }
#pragma unroll
  for (int g = 0; g < NG; g += 2) {
    if (g + 1 < NG) { FULLC_LOADB(bB, sb, g + 1) }
    FULLC_STARTB(bA, g)
    if (g + 1 < NG) {
      if (g + 2 < NG) { FULLC_LOADB(bA, sb, g + 2) }
      FULLC_STARTB(bB, g + 1)
    }
  }
  PH(6);                                                    // (phase 6: the owned samples' starting values -- one exp each -- and the wait for their values)


  // The four waves walk the chunks on their own.  After a chunk a wave adds its 64 lanes' power sums through its private exchange buffer, eight
  // grid points at a time: every lane writes eight sums, lane l adds lanes l%8, l%8 + 8, ... of point l/8 and three DPP steps complete the
  // point in lane 8 (l/8) + 7, which files it in the wave's row of vw.  No barrier before the end of the stretch; a fixed order of additions.
  for (int c = 0; c < nch; c++) {
    const double tc = t0 + (double)(c * LK) * D;
    if (c > 0 && waiting) {                                // rare: one copy of the start code, the sample's registers picked by compile-time selects
      unsigned wm = waiting;
      while (wm) {
        const int j = __builtin_ctz(wm);
        wm &= wm - 1;
        double np, nu;
        const int r = start(sb + t + j * nt, tc, true, np, nu);
        if (r != 0) waiting &= ~(1u << j);
        if (r == 1) {
#pragma unroll
          for (int jj = 0; jj < FULLC_SPT; jj++) { pw[jj] = jj == j ? np : pw[jj]; uu[jj] = jj == j ? nu : uu[jj]; }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int ng = min(LK, npt - c * LK);                  // grid points of this chunk (the last one may be short: whole groups of four beyond it are skipped)
    double acc[LK];
    if (ng == LK) {
#pragma unroll
      for (int j = 0; j < FULLC_SPT; j++) {
        if (j == 0 || j < spt) {                           // (spt >= 1)
          double q = pw[j];
          const double u = uu[j];
          const double u2 = u * u, u3 = u2 * u, u4 = u2 * u2;
#pragma unroll
          for (int i = 0; i < LK; i += 4) {
            if (j == 0) {                                  // [r5] the first sample sets the accumulators (0 + q, fma(q, u, 0): the same roundings) -- no reset
              acc[i] = q; acc[i + 1] = q * u; acc[i + 2] = q * u2; acc[i + 3] = q * u3;
            } else {
              acc[i] += q;
              acc[i + 1] = __builtin_fma(q, u, acc[i + 1]);
              acc[i + 2] = __builtin_fma(q, u2, acc[i + 2]);
              acc[i + 3] = __builtin_fma(q, u3, acc[i + 3]);
            }
            q *= u4;
          }
          pw[j] = q * K1; uu[j] = u * K2;
        }
        __builtin_amdgcn_sched_barrier(0);                 // one sample after the other: hoisting the u^2, u^3, u^4 of all sixteen costs 96 registers
      }
    } else {                                               // the last chunk of the stretch: the groups of four beyond its end are left out (no state to carry on)
#pragma unroll
      for (int i = 0; i < LK; i++) acc[i] = 0.;
#pragma unroll
      for (int j = 0; j < FULLC_SPT; j++) {
        if (j < spt) {
          double q = pw[j];
          const double u = uu[j];
          const double u2 = u * u, u3 = u2 * u, u4 = u2 * u2;
#pragma unroll
          for (int i = 0; i < LK; i += 4) {
            if (i < ng) {
              acc[i] += q;
              acc[i + 1] = __builtin_fma(q, u, acc[i + 1]);
              acc[i + 2] = __builtin_fma(q, u2, acc[i + 2]);
              acc[i + 3] = __builtin_fma(q, u3, acc[i + 3]);
              q *= u4;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    PH(1);                                                  // (phase 1: the march of a chunk)
#pragma unroll
    for (int h = 0; h < LK; h += 8) {
      if (h < ng) {
#pragma unroll
        for (int i = 0; i < 8; i++) xb[i * FULLC_ROW + lane] = acc[h + i];
        wave_sync();
        const double* src = xb + (lane >> 3) * FULLC_ROW + (lane & 7);
        double v = 0.;
#pragma unroll
        for (int q = 0; q < 8; q++) v += src[8 * q];
        wave_sync();
        v += dpp_move<0x111, 0xf, true>(v); v += dpp_move<0x112, 0xf, true>(v); v += dpp_move<0x114, 0xf, true>(v);
        if ((lane & 7) == 7) { double* o = vrow + c * LK + h + (lane >> 3); *o = sb == 0 ? v : *o + v; }
      }
    }
    PH(2);                                                  // (phase 2: the chunk's sums across the lanes)
  }
  if (sb + nt * FULLC_SPT < S) { FULLC_LOADB(bA, sb + nt * FULLC_SPT, 0) }
  }
#undef FULLC_LOADB
#undef FULLC_STARTB
  // [r5] what the integrand of this thread's first grid point needs is requested before the barrier
  int k = k_first + t;
  double e_pc = -100., e_bk = 0., e_pr = 0., e_jc = 1., e_z = 0., e_zl = 0., e_zr = 0., e_cf = 0.;
  auto eload = [&](int kk) {
    e_cf = fe[FE_CF + (kk - k_first) % LK];
    e_pc = pc[kk]; e_bk = L.bkgA[zo + kk]; e_pr = L.prate[zo + kk]; e_jc = L.jac[zo + kk];
    e_z = zg[kk]; e_zl = zg[kk > 0 ? kk - 1 : kk]; e_zr = zg[kk < Z - 1 ? kk + 1 : kk];
  };
  if (k <= k_last) eload(k);
  const double e_zhi = fe[FE_ZHI], e_zlo = fe[FE_ZLO], e_norm = fe[FE_NORM];      // (read again here: held from the top they would sit in registers of the march)
  __syncthreads();
  PH(3);                                                    // (phase 3: waiting for the block's other waves)
  // p_gw and the integrand of every grid point of the stretch   catalog.py:202, pop_wrapper.py:87, likelihood.py:252-275
  double accl = 0.;
  for (; k <= k_last; k += nt) {
    const int r = k - k_first;
#if FULLC_NW == 4
    const double val = ((vw[0][r] + vw[1][r]) + (vw[2][r] + vw[3][r])) * e_cf;
#else
    double vs = vw[0][r];
#pragma unroll
    for (int w = 1; w < FULLC_NW; w++) vs += vw[w][r];
    const double val = vs * e_cf;
#endif
    const double z = e_z;
    const bool inm = (z <= e_zhi) && (z >= e_zlo);
    const double pgw = inm ? val * e_norm : zero_n;        // kde_vals[eff_mask] ... * norm   likelihood.py:252-253
    if (dump) dump[k] = pgw;
    const double pcv = e_pc;
    double y = 0.;
    if (pcv != -100.) {
      const double p_gal = P.fR * pcv + e_bk;
      const double p_z = p_gal * e_pr;
      y = (p_z != -100.) ? pgw * p_z / e_jc : 0.;
    }
    accl += y * ((z - e_zl) + (e_zr - z));                 // trapezoid: y_k enters the two adjacent intervals
    if (k + nt <= k_last) eload(k + nt);
  }
  accl = block_reduce<RED_SUM>(accl, red);
  if (t == 0) *out_like = grid_is_poisoned(P.z_bad, zg, Z) ? __builtin_nan("") : 0.5 * accl + zero_n;
  PH(4);                                                    // (phase 4: integrand and trapezoid of the stretch, the block's sum and the store)
}

// ------------------------------------------------------------------------------------------------------
// k_selection: dN/dtheta_det per injection (pop_wrapper.py:102-111) / p_draw, block partial sums
// ------------------------------------------------------------------------------------------------------
struct SelDev {
  long long I;
  const double *dL, *m1det, *m2det, *p_draw;
  const double *lm1det, *lm2det;  // log(m1det), log(m2det), formed once at upload (as for the posterior samples)
  const double *tab_pm, *tab_rate, *tab_bkg;   // (nb,I) plug-in models evaluated by the caller (chm_tab); NULL = built-in
  const double *tab_jac;          // (nb,I) plug-in cosmology: |ddL/dz| (1+z)^2 per injection
  double N_inj, N_eff; int has_neff, pad;
  double* partial;                // (nb, nblocks, 2)
  int nblocks;                    // records per draw: ceil(I / 512), at most 2048
  int tile;                       // injections per block pass of k_selection_fast (512: one pass of the block; the scalar call gets ~200 selection blocks of one pass each)
};

// one injection: dN/dtheta_det / p_draw                                     pop_wrapper.py:102-111, selection_function.py:38
template <bool SPECIAL = false, class A1, class A2>
DEVFN double sel_term(const DevParams& P, double dl, double m1d, double m2d, double l1d, double l2d, double ipd, double z,
                      A1 mg, A2 cdf, const double* tpm, const double* trate, const double* tbkg, const double* tjac) {
#pragma clang fp contract(fast)                  // smooth arithmetic only: a*b+c may fuse (the translation unit default is off)
  double zp1 = 1. + z;
  double rz = 1. / zp1;
  double m1 = m1d * rz, m2 = m2d * rz;
  double lzp1 = chm_log_pos(zp1);
  const bool own_cosmo = !(tjac && tbkg);                          // plug-in cosmology (chm_tab): both cosmological factors tabulated
  double Ez = own_cosmo ? E_at_z_l(P, z, lzp1) : 1.;
  double dCt = own_cosmo ? dL2dCt_l(P, dl, z, lzp1) : 0.;          // original distances: cosmo.py:191-192,215-216
  double p_z = tbkg ? *tbkg : dVcdz_from_dCt_E(P, dCt, Ez);        // gal_cat.p_bkg              pop_wrapper.py:106
  p_z = p_z * ((trate ? *trate : ((SPECIAL && P.rate_special) ? merger_rate_special(P, z) : merger_rate_l(P, z, lzp1))) / (1. + z));      //         pop_wrapper.py:107
  double dN = P.R0 * (tpm ? *tpm : p_m1m2_fused(P, m1, m2, l1d - lzp1, l2d - lzp1, mg, cdf)) * p_z;   // pop_wrapper.py:108
  double jacobian = tjac ? *tjac : fabs(ddLdz_from_dCt_E(P, dCt, z, Ez, lzp1)) * (zp1 * zp1);      //           pop_wrapper.py:109
  dN = dN / jacobian;
  return dN * ipd;                                                 // selection_function.py:38 (array holds 1/p_draw)
}

// Blocks of 256 threads stage the draw's tables in LDS once and walk over tiles of SEL_TILE injections (tile = blockIdx.x,
// += gridDim.x); a thread takes two consecutive injections per pass (16 B loads, lock-step table searches).
#define SEL_TILE 1024
// SPECIAL: the instantiation chm_eval launches for a call that carries a draw with an infinite rate parameter (merger_rate_special)
template <bool LDS_TAB, bool SPECIAL = false>
__global__ void __launch_bounds__(256) k_selection(SelDev Sd, const DevParams* params, const double* zt_all, const double* It_all,
                                                    const double* dLt_all, const double* mg_all, const double* cdf_all,
                                                    int TcMax, int TmMax) {
  extern __shared__ double lds[];
  __shared__ double red[16];
  const int b = blockIdx.y, t = threadIdx.x;
  const DevParams P = params[b];      // by value: uniform loads at kernel start -> scalar registers, nothing re-read in the loops
  TablePtrs g = { zt_all + (size_t)b * TcMax, It_all + (size_t)b * TcMax, dLt_all + (size_t)b * TcMax,
                  mg_all + (size_t)b * TmMax, cdf_all + (size_t)b * TmMax };
  TabView T = stage_tables(P, g, LDS_TAB, lds, false);
  double s1 = 0., s2 = 0.;
  const long long I = Sd.I;
  for (long long base = (long long)blockIdx.x * SEL_TILE; base < I; base += (long long)gridDim.x * SEL_TILE) {
#pragma unroll 1
    for (int off = 2 * t; off < SEL_TILE; off += 512) {
      const long long i = base + off;
      if (i >= I) break;
      const bool two = i + 1 < I;
      double dl[2], md1[2], md2[2], ipd[2], l1[2], l2[2];
      if (two) {                                    // i even: 16-byte aligned pairs
        double2 a = *reinterpret_cast<const double2*>(Sd.dL + i), bb = *reinterpret_cast<const double2*>(Sd.m1det + i);
        double2 cc = *reinterpret_cast<const double2*>(Sd.m2det + i), dd = *reinterpret_cast<const double2*>(Sd.p_draw + i);
        double2 ee = *reinterpret_cast<const double2*>(Sd.lm1det + i), ff = *reinterpret_cast<const double2*>(Sd.lm2det + i);
        dl[0] = a.x; dl[1] = a.y; md1[0] = bb.x; md1[1] = bb.y; md2[0] = cc.x; md2[1] = cc.y; ipd[0] = dd.x; ipd[1] = dd.y;
        l1[0] = ee.x; l1[1] = ee.y; l2[0] = ff.x; l2[1] = ff.y;
      } else {
        dl[0] = dl[1] = Sd.dL[i]; md1[0] = md1[1] = Sd.m1det[i]; md2[0] = md2[1] = Sd.m2det[i]; ipd[0] = ipd[1] = Sd.p_draw[i];
        l1[0] = l1[1] = Sd.lm1det[i]; l2[0] = l2[1] = Sd.lm2det[i];
      }
      double zz[2];
      z_from_dGW_x2(P, dl[0], dl[1], T.dLt, T.zt, zz[0], zz[1]);      // z = z_from_dGW(dL)   cosmo.py:260-264
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const size_t ti = (size_t)b * (size_t)I + (size_t)(i + (two ? h : 0));      // plug-in models: (nb,I) tables of the caller
        double dN = sel_term<SPECIAL>(P, dl[h], md1[h], md2[h], l1[h], l2[h], ipd[h], zz[h], T.mg, T.cdf, Sd.tab_pm ? Sd.tab_pm + ti : nullptr,
                             Sd.tab_rate ? Sd.tab_rate + ti : nullptr, Sd.tab_bkg ? Sd.tab_bkg + ti : nullptr, Sd.tab_jac ? Sd.tab_jac + ti : nullptr);
        if (h == 0 || two) {
          if (dN == dN) s1 += dN;                                    // nansum                     selection_function.py:39
          s2 += dN * dN;                                             // plain sum     selection_function.py:44
        }
      }
    }
  }
  s1 = block_reduce<RED_SUM>(s1, red);
  s2 = block_reduce<RED_SUM>(s2, red);
  if (threadIdx.x == 0) {
    double* o = Sd.partial + ((size_t)b * Sd.nblocks + blockIdx.x) * 2;
    o[0] = s1; o[1] = s2;
  }
}

// k_selection_fast<MASS>: the selection sums for the built-in models of a flat-or-curved FLRW draw (cosmo_model 0) with the front
// end of k_samples_fast -- z from the draw's direct-index table and node records (lutB of k_tables), log(1 + z) from the record,
// the mass model a template parameter -- and ONE division per injection: with dCt = dL/(1+z), X = dCt E + dH (1+z),
//   dN/p_draw = R0 p_m1m2 (4 pi dH dCt^2 / E) (rate/(1+z)) / (|dCt + dH (1+z)/E| (1+z)^2) / p_draw            pop_wrapper.py:102-111
//             = R0 p_m1m2 4 pi dH dCt^2 rate_num / (rate_den |X| (1+z)^3) / p_draw                              (E(z) > 0 cancels)
// (k_selection: 880 VALU instructions per injection, of which 6 IEEE divisions, a binary search and a log).
#ifndef CHM_SELF_MINW
#define CHM_SELF_MINW 3
#endif
// [r5] MG (cosmo_model 1, modified GW propagation: cosmo.py:225-257): with q = (1+z)^-n (one table exp), Xi = Xi0 + (1 - Xi0) q,
//   dCt = dL / Xi / (1+z),   |ddL_gw/dz| E = |Xi (dCt E + dH (1+z)) + dCt E n (Xi0 - 1) q|   (dL_flrw dXi/dz = dCt n (Xi0 - 1) q)
// in place of X; everything else is the FLRW line.  BASELINE config 5 ran the general kernel (880 instructions per injection) before.
// (body shared by k_selection_fast and k_zf_sel: block bx of nbx, draw b; red: 16 doubles of LDS)
template <int MASS, bool MG = false>
DEVFN void selection_fast_body(const SelDev& Sd, const LutDesc& lut, const DevParams* params, const double* zt_all, const double* dLt_all,
                               const double* mg_all, const double* cdf_all, const double* rec_all, int TcMax, int TmMax,
                               const int b, const int bx, const int nbx, double* lds, double* red) {
#pragma clang fp contract(fast)
  const int t = threadIdx.x;
  DevParams P = params[b];
#ifndef CHM_SELF_NPV
#define CHM_SELF_NPV 11       // 127 VGPRs: four waves per SIMD (one more: three waves; C4 step 2.47 -> 2.42 ms on one box)
#endif
  mass_params_to_vgpr<MASS, CHM_SELF_NPV>(P);
  const double* g_zt = zt_all + (size_t)b * TcMax;
  const double* g_dLt = dLt_all + (size_t)b * TcMax;
  const int Tc = P.Tc, Tm = P.Tm;
  const int* info = lut.info + (size_t)b * 4;
  const int i_lo = info[0], ns = info[1], lmax = info[2];
  const bool fits = info[3] != 0;
  const int key0 = lut.key0, nk = lut.nk, cap = lut.cap;
  double* etab = lds;                                        // [r3] the table of the mass model's exps (chm_exp_tab), as in k_samples_fast
  double* rec = lds + CHM_EXPTAB_N; double* mg = rec + 4 * (size_t)cap; double* cdf = mg + Tm;
  unsigned short* luts = reinterpret_cast<unsigned short*>(cdf + Tm);
  const ExpTab ex = { etab };
  {
    const double* gm = mg_all + (size_t)b * TmMax;
    const double* gc = cdf_all + (size_t)b * TmMax;
    const unsigned short* gl = lut.lut + (size_t)b * (nk + 1);
    for (int i = t; i < CHM_EXPTAB_N; i += 256) etab[i] = exp_table_entry(i);
    for (int i = t; i < Tm; i += 256) { mg[i] = gm[i]; cdf[i] = gc[i]; }
    if (fits) {
      const double2* gr = reinterpret_cast<const double2*>(rec_all + ((size_t)b * TcMax + i_lo) * 4);
      double2* lr = reinterpret_cast<double2*>(rec);
      for (int i = t; i < 2 * ns; i += 256) lr[i] = gr[i];
      for (int i = t; i <= nk; i += 256) luts[i] = gl[i];
    }
  }
  const double x_last = g_dLt[Tc - 1], z_last = g_zt[Tc - 1];
  const double c0 = P.R0 * (4. * CHM_PI * P.dH);
  __syncthreads();
  double s1 = 0., s2 = 0.;
  const long long I = Sd.I;
  const int tile = Sd.tile > 0 ? Sd.tile : SEL_TILE;
  for (long long base = (long long)bx * tile; base < I; base += (long long)nbx * tile) {
#pragma unroll 1
    for (int off = 2 * t; off < tile; off += 512) {
      const long long i = base + off;
      if (i >= I) break;
      const bool two = i + 1 < I;
      double dl[2], md1[2], md2[2], ipd[2], l1[2], l2[2];
      if (two) {                                    // i even: 16-byte aligned pairs
        double2 a = *reinterpret_cast<const double2*>(Sd.dL + i), bb = *reinterpret_cast<const double2*>(Sd.m1det + i);
        double2 cc = *reinterpret_cast<const double2*>(Sd.m2det + i), dd = *reinterpret_cast<const double2*>(Sd.p_draw + i);
        double2 ee = *reinterpret_cast<const double2*>(Sd.lm1det + i), ff = *reinterpret_cast<const double2*>(Sd.lm2det + i);
        dl[0] = a.x; dl[1] = a.y; md1[0] = bb.x; md1[1] = bb.y; md2[0] = cc.x; md2[1] = cc.y; ipd[0] = dd.x; ipd[1] = dd.y;
        l1[0] = ee.x; l1[1] = ee.y; l2[0] = ff.x; l2[1] = ff.y;
      } else {
        dl[0] = dl[1] = Sd.dL[i]; md1[0] = md1[1] = Sd.m1det[i]; md2[0] = md2[1] = Sd.m2det[i]; ipd[0] = ipd[1] = Sd.p_draw[i];
        l1[0] = l1[1] = Sd.lm1det[i]; l2[0] = l2[1] = Sd.lm2det[i];
      }
      double zz[2], z0[2] = { 0., 0. }, lz0[2] = { 0., 0. };
      bool bad = true, anybad = true;
      if (fits) {                                   // z = z_from_dGW(dL)   cosmo.py:260-264
        z_from_lut_x2(dl[0], dl[1], rec, luts, key0, nk, i_lo, ns, lmax, Tc, x_last, z_last, zz[0], zz[1], z0[0], z0[1], lz0[0], lz0[1], bad);
        anybad = wave_any(bad);
        if (anybad) {
          if (bad) { zz[0] = jnp_interp(dl[0], g_dLt, g_zt, Tc, false, 0., 0.); zz[1] = jnp_interp(dl[1], g_dLt, g_zt, Tc, false, 0., 0.); }
        }
      } else z_from_dGW_x2(P, dl[0], dl[1], g_dLt, g_zt, zz[0], zz[1]);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const double z = zz[h];
        const double zp1 = 1. + z;
        const double r = chm_rcp(zp1);
        const double m1 = md1[h] * r, m2 = md2[h] * r;
        double vv;
        double lz = log1pz_from_node(z, z0[h], lz0[h], r, vv);
        if (!fits || anybad || wave_any(!(vv <= 0.02))) { if (!fits || bad || !(vv <= 0.02)) lz = chm_log_pos(zp1); }      // (votes on single compares, see z_from_lut_x2)
        const double pm = p_m1m2_fused<MASS>(P, m1, m2, l1[h] - lz, l2[h] - lz, mg, cdf, ex);
        const double Ez = E_at_z_lr(P, z, zp1, r, lz);
        double dCt = dl[h] * r;                                          // original distances: cosmo.py:191-192,215-216
        double X;
        if (MG) {
          const double q = ex.pw(lz, -P.n_mg);                           // (1+z)^-n
          const double Xi = __builtin_fma(1. - P.Xi0, q, P.Xi0);         // cosmo.py:225-228
          dCt = chm_div(dCt, Xi);                                        // cosmo.py:230-235
          const double cE = dCt * Ez;
          X = __builtin_fma(Xi, __builtin_fma(P.dH, zp1, cE), (cE * (P.n_mg * (P.Xi0 - 1.))) * q);      // cosmo.py:245-257, times E(z)
        } else X = __builtin_fma(dCt, Ez, P.dH * zp1);
        double rnum, rden;
        merger_rate_nd(P, z, lz, rnum, rden, ex);
        const double num = ((c0 * pm) * (dCt * dCt)) * (rnum * ipd[h]);
        const double den = (rden * fabs(X)) * ((zp1 * zp1) * zp1);
        const double dN = num / den;
        if (h == 0 || two) {
          if (dN == dN) s1 += dN;                                        // nansum                     selection_function.py:39
          s2 += dN * dN;                                                 // plain sum     selection_function.py:44
        }
      }
    }
  }
  s1 = block_reduce<RED_SUM>(s1, red);
  s2 = block_reduce<RED_SUM>(s2, red);
  if (threadIdx.x == 0) {                           // the grid may hold fewer blocks than the partial array has records: the rest are zeros
    double* o = Sd.partial + (size_t)b * Sd.nblocks * 2;
    o[2 * bx] = s1; o[2 * bx + 1] = s2;
    for (int x = bx + nbx; x < Sd.nblocks; x += nbx) { o[2 * x] = 0.; o[2 * x + 1] = 0.; }
  }
}

template <int MASS, bool MG = false>
__global__ void __launch_bounds__(256, CHM_SELF_MINW) k_selection_fast(SelDev Sd, LutDesc lut, const DevParams* params, const double* zt_all,
                                                             const double* dLt_all, const double* mg_all, const double* cdf_all,
                                                             const double* rec_all, int TcMax, int TmMax) {
  extern __shared__ double lds[];
  __shared__ double red[16];
  selection_fast_body<MASS, MG>(Sd, lut, params, zt_all, dLt_all, mg_all, cdf_all, rec_all, TcMax, TmMax, blockIdx.y, blockIdx.x, gridDim.x, lds, red);
}

// k_zf_sel<MASS>: the per-z factors (with the event statistics: zfactors_body<true, true>) and the selection sums in ONE launch -- blocks
// [0, zf_blocks) do the former, the rest the latter.  For calls of few draws (the scalar call): the selection function then needs no stream
// of its own, the captured graph is a plain chain (hipGraphLaunch 10 instead of 39 us) and the selection kernel's 16 us hide behind the
// per-z factors instead of competing with the sample stage.
template <int MASS>
__global__ void __launch_bounds__(256, CHM_SELF_MINW) k_zf_sel(LikeDev L, SelDev Sd, LutDesc lut, const DevParams* params, const double* zt_all,
                                                     const double* It_all, const double* dLt_all, const double* mg_all, const double* cdf_all,
                                                     const double* rec_all, int TcMax, int TmMax, int zf_blocks) {
  extern __shared__ double lds[];
  __shared__ double red[16];
  // the selection blocks come FIRST in the grid: blocks are dispatched in index order, and the ~100 selection blocks (16 us each) must
  // start with the kernel, not after a thousand per-z-factor blocks have gone through
  const int sel_blocks = (int)gridDim.x - zf_blocks;
  if ((int)blockIdx.x >= sel_blocks) zfactors_body<true, true>(L, params, zt_all, It_all, TcMax, 1, blockIdx.y, blockIdx.x - sel_blocks, zf_blocks, lds);
  else selection_fast_body<MASS>(Sd, lut, params, zt_all, dLt_all, mg_all, cdf_all, rec_all, TcMax, TmMax, blockIdx.y, blockIdx.x, sel_blocks, lds, red);
}


// ------------------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------------------
// L_i = sum over the pixels of one event, in pixel order (jnp.sum over axis 1, likelihood.py:280); the loads of 8 pixels are
// issued together so that a thread waits for 4 memory round trips per 32 pixels instead of 32
DEVFN double pixel_sum(const double* lp, int Pd) {
  double Li = 0.;
  int p = 0;
  for (; p + 8 <= Pd; p += 8) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = lp[p + i];
#pragma unroll
    for (int i = 0; i < 8; i++) Li += v[i];
  }
  for (; p < Pd; p++) Li += lp[p];
  return Li;
}

// k_reduce_events: one thread per event: L_i = sum_p like_pix (likelihood.py:280), log, nan_to_num (:296-297); block sums
__global__ void __launch_bounds__(256) k_reduce_events(int E, int Pd, const double* like_pix, double* ev_partial /* (nb, nblk) */,
                                                        double* log_like_evs, double* numlike_evs, const double* ev_li, const double* ev_ll,
                                                        const unsigned char* ev_bad) {
  __shared__ double red[16];
  const int b = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
  double ll = 0.;
  if (e < E) {
    double Li;
    if (ev_ll) { Li = ev_li[(size_t)b * E + e]; ll = ev_ll[(size_t)b * E + e]; }      // formed by k_marg_fixup (same sums, same order)
    else {
      const double* lp = like_pix + ((size_t)b * E + e) * Pd;
      Li = pixel_sum(lp, Pd);
      ll = log_like_of(Li);
    }
    if (ev_bad && ev_bad[e]) { Li = __builtin_nan(""); ll = -__builtin_inf(); }      // a NaN among the event's catalogue / grid inputs: 0 * NaN on every grid point (chm_like::d_ev_bad)
    if (numlike_evs) numlike_evs[(size_t)b * E + e] = Li;
    if (log_like_evs) log_like_evs[(size_t)b * E + e] = ll;
  }
  double s = block_reduce<RED_SUM>(ll, red);
  if (threadIdx.x == 0) ev_partial[(size_t)b * gridDim.x + blockIdx.x] = s;
}

// N_exp with the N_eff guard (selection_function.py:38-47) and the final combination (likelihood.py:298-300, 313-316)
// [r5] Completion of a few-draw call seen through MEMORY: the kernel that stores a draw's three results in pinned host memory (zero-copy) then
// stores the call's sequence number -- which the host wrote into the same pinned block before the launch -- into the draw's flag, behind a
// system-scope fence; the host spins on the flags instead of on hipStreamQuery (the end-of-kernel release, the completion signal of the graph and the
// runtime's look at it are then off the call's critical path).
DEVFN void completion_flag(long long* flag, long long seq) {
  __threadfence_system();
  __builtin_nontemporal_store(seq, flag);
}
DEVFN void combine_one(const DevParams& P, double log_num, double s1, double s2, double E_total, double N_inj, double N_eff,
                       int has_neff, int has_like, int has_sel, double* out) {
  double Nexp = __builtin_nan("");
  if (has_sel) {
    double xi = s1 / N_inj;
    Nexp = P.Tobs * xi;
    if (has_neff) {
      double variance2 = s2 / (N_inj * N_inj) - (xi * xi) / N_inj;
      double neff = (xi * xi) / variance2;
      if (neff < N_eff) Nexp = 0.0;
    }
  }
  double log_hyper = __builtin_nan("");
  if (has_like) {
    if (!P.scale_free) log_num += E_total * log(P.R0 * P.Tobs);
    if (has_sel) log_hyper = P.scale_free ? log_num - E_total * log(Nexp) : log_num - Nexp;
  } else log_num = __builtin_nan("");
  // (NaNs leave with the canonical bit pattern: the host recognises a result that has not arrived yet by a NaN payload of its own, chm_eval: CHM_PENDING)
  const double cn = __builtin_nan("");
  out[0] = log_hyper != log_hyper ? cn : log_hyper; out[1] = log_num != log_num ? cn : log_num; out[2] = Nexp != Nexp ? cn : Nexp;
}

// k_final: one block per draw: shard partials [sum_i log L_i, nansum dN, sum dN^2] in fixed order; when do_combine, also
// the final combination (single GPU); otherwise k_combine runs after the all-reduce.
__global__ void __launch_bounds__(256) k_final(int nblk_ev, const double* ev_partial, int nblk_sel, const double* sel_partial,
                                                double* partials /* (nb,3) */, const DevParams* params, double E_total, double N_inj,
                                                double N_eff, int has_neff, int has_like, int has_sel, int do_combine, double* out3,
                                                const long long* seq_in, long long* seq_out) {
  __shared__ double red[16];
  const int b = blockIdx.x, t = threadIdx.x;
  const long long seq = (t == 0 && seq_out) ? __builtin_nontemporal_load(seq_in) : 0;      // (completion_flag: requested early, used at the end)
  double acc = 0., s1 = 0., s2 = 0.;
  for (int i = t; i < nblk_ev; i += blockDim.x) acc += ev_partial[(size_t)b * nblk_ev + i];
  for (int i = t; i < nblk_sel; i += blockDim.x) {
    s1 += sel_partial[((size_t)b * nblk_sel + i) * 2];
    s2 += sel_partial[((size_t)b * nblk_sel + i) * 2 + 1];
  }
  acc = block_reduce<RED_SUM>(acc, red);
  s1 = block_reduce<RED_SUM>(s1, red);
  s2 = block_reduce<RED_SUM>(s2, red);
  if (t == 0) {
    partials[b * 3] = acc; partials[b * 3 + 1] = s1; partials[b * 3 + 2] = s2;
    if (do_combine) combine_one(params[b], acc, s1, s2, E_total, N_inj, N_eff, has_neff, has_like, has_sel, out3 + b * 3);
    if (do_combine && seq_out) completion_flag(seq_out + b, seq);
  }
}

// k_reduce_final: k_reduce_events + k_final in one launch (one block of 1024 threads per draw) for shards of up to a few
// thousand events: per-event pixel sums, log, nan_to_num, the three partial sums and (single GPU) the combination.
__global__ void __launch_bounds__(1024) k_reduce_final(int E, int Pd, const double* like_pix, int nblk_sel, const double* sel_partial,
                                                        double* partials, const DevParams* params, double E_total, double N_inj,
                                                        double N_eff, int has_neff, int has_like, int has_sel, int do_combine,
                                                        double* out3, double* log_like_evs, double* numlike_evs,
                                                        const double* ev_li, const double* ev_ll, const unsigned char* ev_bad,
                                                        const long long* seq_in, long long* seq_out) {
  __shared__ double red[16];
  const int b = blockIdx.x, t = threadIdx.x;
  const long long seq = (t == 0 && seq_out) ? __builtin_nontemporal_load(seq_in) : 0;      // (completion_flag: requested early, used at the end)
  double acc = 0., s1 = 0., s2 = 0.;
  for (int e = t; e < E; e += blockDim.x) {
    double Li, ll;
    if (ev_ll) { Li = ev_li[(size_t)b * E + e]; ll = ev_ll[(size_t)b * E + e]; }      // formed by k_marg_fixup (same sums, same order)
    else {
      const double* lp = like_pix + ((size_t)b * E + e) * Pd;
      Li = pixel_sum(lp, Pd);                                        // jnp.sum over pixels          likelihood.py:280
      ll = log_like_of(Li);                                          // likelihood.py:296,329; nan_to_num(nan=-inf)
    }
    if (ev_bad && ev_bad[e]) { Li = __builtin_nan(""); ll = -__builtin_inf(); }      // (chm_like::d_ev_bad)
    if (numlike_evs) numlike_evs[(size_t)b * E + e] = Li;
    if (log_like_evs) log_like_evs[(size_t)b * E + e] = ll;
    acc += ll;
  }
  for (int i = t; i < nblk_sel; i += blockDim.x) {
    s1 += sel_partial[((size_t)b * nblk_sel + i) * 2];
    s2 += sel_partial[((size_t)b * nblk_sel + i) * 2 + 1];
  }
  acc = block_reduce<RED_SUM>(acc, red);
  s1 = block_reduce<RED_SUM>(s1, red);
  s2 = block_reduce<RED_SUM>(s2, red);
  if (t == 0) {
    partials[b * 3] = acc; partials[b * 3 + 1] = s1; partials[b * 3 + 2] = s2;
    if (do_combine) combine_one(params[b], acc, s1, s2, E_total, N_inj, N_eff, has_neff, has_like, has_sel, out3 + b * 3);
    if (do_combine && seq_out) completion_flag(seq_out + b, seq);
  }
}

__global__ void k_combine(int nb, const DevParams* params, const double* partials, double E_total, double N_inj, double N_eff,
                          int has_neff, int has_like, int has_sel, double* out3, const long long* seq_in, long long* seq_out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const long long seq = seq_out ? __builtin_nontemporal_load(seq_in) : 0;
  combine_one(params[b], partials[b * 3], partials[b * 3 + 1], partials[b * 3 + 2], E_total, N_inj, N_eff, has_neff, has_like,
              has_sel, out3 + b * 3);
  if (seq_out) completion_flag(seq_out + b, seq);
}

// ------------------------------------------------------------------------------------------------------
// k_pcat: pixelated_catalog.precompute_p_cat (catalog.py:152-195, 212-221); one block per (event, pixel)
// ------------------------------------------------------------------------------------------------------
struct PcatDev {
  int E, P, Z, pad;
  const double* z_grids;
  const long long* offsets;
  const double *gal_z, *gal_sig, *gal_w;
  const double* weight_grid;      // (E,Z) or NULL: p_bkg of a plug-in completeness on the event grids (sumgauss='pbkg', catalog.py:223-231)
  double* p_cat;
};

__global__ void __launch_bounds__(256) k_pcat(PcatDev D, const DevParams* params, TablePtrs g) {
  extern __shared__ double lds[];                   // dv[Z], zz[Z], acc[Z]
  __shared__ double red[16];
  const DevParams& P = params[0];
  const int Z = D.Z, t = threadIdx.x, nt = blockDim.x;
  const int e = blockIdx.x / D.P;
  double* dv = lds; double* zz = dv + Z; double* acc = zz + Z;
  const double* zg = D.z_grids + (size_t)e * Z;
  for (int k = t; k < Z; k += nt) {
    double z = zg[k];
    zz[k] = z;
    dv[k] = D.weight_grid ? D.weight_grid[(size_t)e * Z + k]         // p_bkg(cosmo, zgrid), caller-evaluated   catalog.py:229
                          : dVcdz_from_dCt(P, dCt_at_z(P, z, g.zt, g.It), z);      // dVcdz_at_z(cosmo, zgrid)   catalog.py:219
    acc[k] = 0.;
  }
  __syncthreads();
  const long long g0 = D.offsets[blockIdx.x], g1 = D.offsets[blockIdx.x + 1];
  double sw = 0.;
  for (long long gi = g0; gi < g1; gi++) {
    const double mu = D.gal_z[gi], sg = D.gal_sig[gi], w = D.gal_w[gi];
    const double pref = 1. / sqrt(2. * CHM_PI * (sg * sg));          // np.power(2 pi sigma^2, -0.5)   catalog.py:210
    // norm = trapz(gauss * dVdz, zgrid) = 0.5 sum dx (y1 + y0)                                         catalog.py:220
    double part = 0.;
    for (int k = t; k < Z - 1; k += nt) {
      double u0 = (zz[k] - mu) / sg, u1 = (zz[k + 1] - mu) / sg;
      double y0 = pref * chm_exp(-0.5 * (u0 * u0)) * dv[k], y1 = pref * chm_exp(-0.5 * (u1 * u1)) * dv[k + 1];
      part += (zz[k + 1] - zz[k]) * (y1 + y0);
    }
    const double norm = 0.5 * block_reduce<RED_SUM>(part, red);
    for (int k = t; k < Z; k += nt) {
      double u = (zz[k] - mu) / sg;
      double y = pref * chm_exp(-0.5 * (u * u)) * dv[k];
      acc[k] += w * y / norm;                                          // np.sum(weights * gauss / norm, axis=1)   :221
    }
    sw += w;
  }
  __syncthreads();
  double* out = D.p_cat + (size_t)blockIdx.x * Z;
  for (int k = t; k < Z; k += nt) {
    double v = g1 > g0 ? acc[k] / sw : 0.;                             // len(mu) == 0 -> zeros           :213-214
    if (!(fabs(v) <= 1.7976931348623157e308)) v = 0.;                  // p_cat[~isfinite(p_cat)] = 0     catalog.py:172
    out[k] = v;
  }
}

// ------------------------------------------------------------------------------------------------------
// k_kde2d: jax_gkde_nd for d = 2 (math.py:95-148) at the pixel centres of each event; one block per event
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_kde2d(int S, int Pmax, const double* ra, const double* dec, const double* ra_pix,
                                                const double* dec_pix, const int* npix, double* out) {
  __shared__ double red[16];
  __shared__ double wh[8];
  const int e = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  const double* x = ra + (size_t)e * S;
  const double* y = dec + (size_t)e * S;
  // unweighted: W = 1/S, neff = S, factor = S^(-1/6) (scott, d = 2)                      math.py:115-118
  const double W = 1. / (double)S;
  double sx = 0., sy = 0.;
  for (int s = t; s < S; s += nt) { sx += W * x[s]; sy += W * y[s]; }
  sx = block_reduce<RED_SUM>(sx, red); sy = block_reduce<RED_SUM>(sy, red);
  double cxx = 0., cxy = 0., cyy = 0.;
  for (int s = t; s < S; s += nt) { double rx = x[s] - sx, ry = y[s] - sy; cxx += rx * W * rx; cxy += rx * W * ry; cyy += ry * W * ry; }
  cxx = block_reduce<RED_SUM>(cxx, red); cxy = block_reduce<RED_SUM>(cxy, red); cyy = block_reduce<RED_SUM>(cyy, red);
  if (t == 0) {
    double den = 1. - (double)S * (W * W);                 // 1 - sum(W^2)                      math.py:128
    cxx /= den; cxy /= den; cyy /= den;
    double factor = chm_exp(chm_log((double)S) * (-1. / 6.));
    double det = cxx * cyy - cxy * cxy, f2 = factor * factor;
    double ixx = cyy / det / f2, ixy = -cxy / det / f2, iyy = cxx / det / f2;      // inv_cov        math.py:129-131
    double l00 = sqrt(ixx), l10 = ixy / l00, l11 = sqrt(iyy - l10 * l10);           // cholesky       math.py:132
    wh[0] = l00; wh[1] = l10; wh[2] = l11;
    wh[3] = (chm_log(l00) + chm_log(l11)) - 0.5 * 2. * chm_log(2. * CHM_PI);                    // log_norm       math.py:135
  }
  __syncthreads();
  const double l00 = wh[0], l10 = wh[1], l11 = wh[2], log_norm = wh[3];
  const int np_ = npix[e];
  for (int p = 0; p < np_ && p < Pmax; p++) {
    const double qx = ra_pix[(size_t)e * Pmax + p], qy = dec_pix[(size_t)e * Pmax + p];
    const double q0 = qx * l00 + qy * l10, q1 = qy * l11;  // points . L                          math.py:133
    double acc = 0.;
    for (int s = t; s < S; s += nt) {
      double d0 = (x[s] * l00 + y[s] * l10) - q0, d1 = y[s] * l11 - q1;
      acc += W * chm_exp(log_norm - 0.5 * (d0 * d0 + d1 * d1));                          // math.py:141-146
    }
    acc = block_reduce<RED_SUM>(acc, red);
    if (t == 0) out[(size_t)e * Pmax + p] = acc;
  }
}

// ------------------------------------------------------------------------------------------------------
// k_model_eval: elementwise model functions for the Python free functions (cosmo.py / mass.py / rate.py)
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_model_eval(const DevParams* params, TablePtrs g, int func, const double* a, const double* bb,
                                                     long long n, double* out) {
  const DevParams& P = params[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double x = a[i];
    double r = 0.;
    switch (func) {
      case 0: r = E_at_z(P, x); break;
      case 1: r = jnp_interp(x, g.zt, g.It, P.Tc, false, 0., 0.); break;
      case 2: r = P.dH * jnp_interp(x, g.zt, g.It, P.Tc, false, 0., 0.); break;
      case 3: r = dCt_at_z(P, x, g.zt, g.It); break;
      case 4: r = dL_from_dCt(P, dCt_at_z(P, x, g.zt, g.It), x); break;
      case 5: r = ddLdz_from_dCt(P, bb ? dL2dCt(P, bb[i], x) : dCt_at_z(P, x, g.zt, g.It), x); break;
      case 6: r = dVcdz_from_dCt(P, bb ? dL2dCt(P, bb[i], x) : dCt_at_z(P, x, g.zt, g.It), x); break;
      case 7: r = Vc_from_dCt(P, bb ? dL2dCt(P, bb[i], x) : dCt_at_z(P, x, g.zt, g.It)); break;
      case 8: r = Xi_at_z(P, x); break;
      case 9: r = jnp_interp(x, g.dLt, g.zt, P.Tc, false, 0., 0., P.dl_sorted == 0.); break;     // z_from_dGW: jax's scan search on an unsorted table
      case 10: r = merger_rate(P, x); break;
      case 11: r = p_m1m2(P, x, bb[i], g.mg, g.cdf); break;
      case 12: r = primary_notnorm(P, x); break;
      case 13: r = secondary_notnorm(P, x, bb[i]); break;
      case 14: r = smoothing(x, mass_delta_m(P), P.m[0]); break;
      case 15: r = p_m1m2_fused(P, x, bb[i], chm_log(x), chm_log(bb[i]), g.mg, g.cdf); break;   // the hot loops' form of case 11
      case 16: r = tpl_cdf(-P.m[2], P.m[0], x); break;                                          // tpl_cdf(alpha, m_low, m): a tpl struct carries (-alpha, m_low)
      case 17: { double mu = P.m[6], sg = P.m[7];                                                 // gaussian(x, mu, sigma): a plp struct carries (mu_g, sigma_g)   mass.py:267-269
                 r = exp((-0.5 * log(2. * CHM_PI) - log(sg)) - (x - mu) * (x - mu) / (2. * sg * sg)); } break;
      case 18: { double mu = P.m[6], sg = P.m[7], x_min = P.m[0], x_max = P.m[1];                 // truncated_gaussian(x, mu, sigma, x_min = m_low, x_max = m_high)   mass.py:271-279
                 double norm = 0.5 * erf((x_max - mu) / (sg * sqrt(2.))) - 0.5 * erf((x_min - mu) / (sg * sqrt(2.)));
                 double G = exp((-0.5 * log(2. * CHM_PI) - log(sg)) - (x - mu) * (x - mu) / (2. * sg * sg));
                 r = (x_min <= x && x <= x_max) ? G / norm : 0.; } break;
    }
    out[i] = r;
  }
}

// ------------------------------------------------------------------------------------------------------
// Stand-alone forms of CHIMERA/utils/math.py (the building blocks the kernels above fuse), one call = one array:
// chm_kde1d, chm_binning1d, chm_gkde_nd, chm_trapz, chm_cumtrapz.  Dense sums in the reference's order of operations.
// ------------------------------------------------------------------------------------------------------
// kde1d set-up (math.py:58-73), one block: st[0] = sum(w) (1 when weights are absent: W = 1/N), st[1] = bandwidth
__global__ void __launch_bounds__(1024) k_math_kde1d_setup(const double* data, const double* wgt, long long N, int bw_method, double bw_scalar,
                                                            double* st) {
  __shared__ double red[16];
  const int t = threadIdx.x, nt = blockDim.x;
  double a = 0.;
  for (long long j = t; j < N; j += nt) a += wgt ? wgt[j] : 1.;
  const double tot = wgt ? block_reduce<RED_SUM>(a, red) : (double)N;            // weights / sum(weights)  |  ones / size
  a = 0.;
  double c = 0.;
  for (long long j = t; j < N; j += nt) { double W = (wgt ? wgt[j] : 1.) / tot; a += W * W; c += data[j]; }
  const double neff = 1.0 / block_reduce<RED_SUM>(a, red);
  const double mean = block_reduce<RED_SUM>(c, red) / (double)N;
  a = 0.;
  for (long long j = t; j < N; j += nt) { double d = data[j] - mean; a += d * d; }
  const double sd = sqrt(block_reduce<RED_SUM>(a, red) / (double)N);            // jnp.std: two-pass, ddof = 0
  if (t == 0) { st[0] = tot; st[1] = kde_bandwidth_factor(bw_method, bw_scalar, neff, 1) * sd; }
}

// density[i] = sum_j W_j K((grid_i - x_j)/h) / h (math.py:77-81); one thread per grid point, the dataset through LDS tiles
__global__ void __launch_bounds__(256) k_math_kde1d_eval(const double* data, const double* wgt, long long N, const double* grid, long long G,
                                                          int epan, const double* st, double* out) {
  __shared__ double xs[512], ws[512];
  const int t = threadIdx.x;
  const long long i = (long long)blockIdx.x * blockDim.x + t;
  const double tot = st[0], bw = st[1];
  const double g = i < G ? grid[i] : 0.;
  const double isq = 1. / sqrt(2. * CHM_PI);
  double acc = 0.;
  for (long long j0 = 0; j0 < N; j0 += 512) {
    const int m = (int)((N - j0) < 512 ? (N - j0) : 512);
    __syncthreads();
    for (int j = t; j < m; j += blockDim.x) { xs[j] = data[j0 + j]; ws[j] = (wgt ? wgt[j0 + j] : 1.) / tot; }
    __syncthreads();
    for (int j = 0; j < m; j++) {
      double u = (g - xs[j]) / bw;
      double kv = epan ? (fabs(u) <= 1. ? 0.75 * (1. - u * u) : 0.) : exp(-0.5 * (u * u)) * isq;      // math.py:83-89
      acc += ws[j] * kv;
    }
  }
  if (i < G) out[i] = acc / bw;
}

// binning1d (math.py:32-46), ONE wave (deterministic accumulation order): centres[B], counts[B]; the histogram lives in `counts`
__global__ void __launch_bounds__(64) k_math_binning1d(const double* data, const double* wgt, long long N, int B, double* centres, double* counts) {
  const int lane = threadIdx.x;
  double mn = __builtin_inf(), mx = -__builtin_inf();
  bool sawnan = false;
  for (long long j = lane; j < N; j += 64) { double v = data[j]; mn = __builtin_fmin(mn, v); mx = __builtin_fmax(mx, v); sawnan = sawnan || (v != v); }
  mn = wave_min_dpp(mn); mx = wave_max_dpp(mx);
  if (__ballot(sawnan)) { mn = __builtin_nan(""); mx = mn; }                                            // jnp.min / jnp.max propagate NaN
  for (int j = lane; j < B; j += 64) {
    double e0 = jnp_linspace_at(mn, mx, B + 1, j), e1 = jnp_linspace_at(mn, mx, B + 1, j + 1);
    centres[j] = (e0 + e1) / 2.;
    counts[j] = 0.;
  }
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
  // lanes take turns on colliding bins in lane order: one sample per lane per pass, lane by lane within a pass
  for (long long j0 = 0; j0 < N; j0 += 64) {
    const long long j = j0 + lane;
    const int idx = j < N ? bin_index(data[j], mn, mx, B) : -1;
    const double w = j < N ? wgt[j] : 0.;
    for (int l = 0; l < 64; l++) {
      if (l == lane && idx >= 0) counts[idx] += w;
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// gkde_nd set-up (math.py:111-135), one block, d <= CHM_GKDE_MAXD: st = [sum w, log_norm, L (d x d, row-major lower Cholesky factor of inv_cov)]
#define CHM_GKDE_MAXD 4
__global__ void __launch_bounds__(1024) k_math_gkde_setup(const double* data /* (d,N) */, const double* wgt, int d, long long N, int bw_method,
                                                           double bw_scalar, double* st) {
  __shared__ double red[16];
  __shared__ double mean[CHM_GKDE_MAXD];
  __shared__ double cov[CHM_GKDE_MAXD * CHM_GKDE_MAXD];
  const int t = threadIdx.x, nt = blockDim.x;
  double a = 0.;
  for (long long j = t; j < N; j += nt) a += wgt ? wgt[j] : 1.;
  const double tot = wgt ? block_reduce<RED_SUM>(a, red) : (double)N;
  a = 0.;
  for (long long j = t; j < N; j += nt) { double W = (wgt ? wgt[j] : 1.) / tot; a += W * W; }
  const double sw2 = block_reduce<RED_SUM>(a, red);
  for (int k = 0; k < d; k++) {
    a = 0.;
    for (long long j = t; j < N; j += nt) a += ((wgt ? wgt[j] : 1.) / tot) * data[(size_t)k * N + j];
    a = block_reduce<RED_SUM>(a, red);
    if (t == 0) mean[k] = a;
  }
  __syncthreads();
  for (int k = 0; k < d; k++) for (int l = 0; l <= k; l++) {
    a = 0.;
    for (long long j = t; j < N; j += nt) {
      double W = (wgt ? wgt[j] : 1.) / tot;
      a += ((data[(size_t)k * N + j] - mean[k]) * W) * (data[(size_t)l * N + j] - mean[l]);
    }
    a = block_reduce<RED_SUM>(a, red);
    if (t == 0) { cov[k * d + l] = a / (1. - sw2); cov[l * d + k] = cov[k * d + l]; }                // math.py:127-128
  }
  __syncthreads();
  if (t == 0) {
    const double neff = 1. / sw2;
    const double factor = kde_bandwidth_factor(bw_method, bw_scalar, neff, d);                         // math.py:116-123
    // inverse by Gauss-Jordan with partial pivoting (np.linalg.inv), then inv_cov = inv / factor^2
    double A[CHM_GKDE_MAXD][2 * CHM_GKDE_MAXD];
    for (int r = 0; r < d; r++) for (int c2 = 0; c2 < d; c2++) { A[r][c2] = cov[r * d + c2]; A[r][d + c2] = r == c2 ? 1. : 0.; }
    for (int c2 = 0; c2 < d; c2++) {
      int piv = c2;
      for (int r = c2 + 1; r < d; r++) if (fabs(A[r][c2]) > fabs(A[piv][c2])) piv = r;
      if (piv != c2) for (int k = 0; k < 2 * d; k++) { double tmp = A[c2][k]; A[c2][k] = A[piv][k]; A[piv][k] = tmp; }
      const double pv = A[c2][c2];
      for (int k = 0; k < 2 * d; k++) A[c2][k] /= pv;
      for (int r = 0; r < d; r++) if (r != c2) { const double f = A[r][c2]; for (int k = 0; k < 2 * d; k++) A[r][k] -= f * A[c2][k]; }
    }
    double ic[CHM_GKDE_MAXD][CHM_GKDE_MAXD], Lm[CHM_GKDE_MAXD][CHM_GKDE_MAXD];
    for (int r = 0; r < d; r++) for (int c2 = 0; c2 < d; c2++) { ic[r][c2] = A[r][d + c2] / (factor * factor); Lm[r][c2] = 0.; }
    for (int r = 0; r < d; r++) for (int c2 = 0; c2 <= r; c2++) {                                       // lower Cholesky factor (math.py:132)
      double s = ic[r][c2];
      for (int k = 0; k < c2; k++) s -= Lm[r][k] * Lm[c2][k];
      Lm[r][c2] = r == c2 ? sqrt(s) : s / Lm[c2][c2];
    }
    double ln = 0.;
    for (int r = 0; r < d; r++) ln += log(Lm[r][r]);
    st[0] = tot; st[1] = ln - 0.5 * (double)d * log(2. * CHM_PI);                                     // math.py:135
    for (int r = 0; r < d; r++) for (int c2 = 0; c2 < d; c2++) st[2 + r * d + c2] = Lm[r][c2];
  }
}

// out[i] = sum_j W_j exp(log_norm - 1/2 |x_j L - p_i L|^2) (math.py:133-147); one thread per point, the dataset through LDS tiles.
// LOG (in_log=True, math.py:223-226): out[i] = logsumexp_j(log W_j + log_norm - 1/2 |.|^2), accumulated in the reference's own order by
// np.logaddexp -- max(a, b) + log1p(exp(-|a - b|)) -- from log_sum = -inf
template <bool LOG>
__global__ void __launch_bounds__(256) k_math_gkde_eval(const double* data, const double* wgt, int d, long long N, const double* pts /* (d,M) */,
                                                         long long M, const double* st, double* out) {
  __shared__ double xs[CHM_GKDE_MAXD][256], ws[256];
  __shared__ double Ls[CHM_GKDE_MAXD * CHM_GKDE_MAXD];
  const int t = threadIdx.x;
  if (t < d * d) Ls[t] = st[2 + t];
  __syncthreads();
  const long long i = (long long)blockIdx.x * blockDim.x + t;
  const double tot = st[0], log_norm = st[1];
  double q[CHM_GKDE_MAXD];
  for (int c = 0; c < d; c++) { double s = 0.; for (int k = 0; k < d; k++) s += (i < M ? pts[(size_t)k * M + i] : 0.) * Ls[k * d + c]; q[c] = s; }      // points.T @ L
  double acc = LOG ? -__builtin_inf() : 0.;
  for (long long j0 = 0; j0 < N; j0 += 256) {
    const int m = (int)((N - j0) < 256 ? (N - j0) : 256);
    __syncthreads();
    if (t < m) {
      for (int c = 0; c < d; c++) { double s = 0.; for (int k = 0; k < d; k++) s += data[(size_t)k * N + j0 + t] * Ls[k * d + c]; xs[c][t] = s; }     // dataset.T @ L
      const double W = (wgt ? wgt[j0 + t] : 1.) / tot;
      ws[t] = LOG ? log(W) : W;
    }
    __syncthreads();
    for (int j = 0; j < m; j++) {
      double r2 = 0.;
      for (int c = 0; c < d; c++) { double dd = xs[c][j] - q[c]; r2 += dd * dd; }
      if (LOG) {
        const double a = acc, bq = ws[j] + (log_norm - 0.5 * r2);
        if (a == bq) acc = a + 0.6931471805599453;            // np.logaddexp: equal arguments (also -inf, -inf -> -inf)
        else { const double mx = a > bq ? a : bq, df = a > bq ? bq - a : a - bq; acc = (df != df) ? a + bq : mx + log1p(exp(df)); }
      } else acc += ws[j] * exp(log_norm - 0.5 * r2);
    }
  }
  if (i < M) out[i] = acc;
}

// jnp.trapezoid(y, x, axis=-1) of `rows` rows of n points (x: one shared row or one per row), a wave per row; cumtrapz (math.py:22-26) of
// one row by one block
__global__ void __launch_bounds__(256) k_math_trapz(const double* y, const double* x, long long rows, int n, int x_per_row, double* out) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const double* yy = y + (size_t)r * n;
  const double* xx = x + (x_per_row ? (size_t)r * n : 0);
  double acc = 0.;
  for (int k = lane; k < n - 1; k += 64) acc += (xx[k + 1] - xx[k]) * (yy[k + 1] + yy[k]);
  acc = wave_sum(acc);
  if (lane == 0) out[r] = 0.5 * acc;
}
__global__ void __launch_bounds__(1024) k_math_cumtrapz(const double* y, const double* x, int n, double* out) {
  __shared__ double sh[32];
  block_cumtrapz(y, x, out, n, sh);
}
