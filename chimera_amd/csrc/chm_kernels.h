// chm_kernels.h -- HIP kernels of the hyper-likelihood path (gfx950 / CDNA4, wave64, fp64).
//
// Pipeline of one evaluation (all on one HIP stream; nb = draws in the batch, blockIdx.y = draw):
//   k_tables        1 block / draw         per-draw tables + constants        cosmo.py:43-46, mass.py:45-52
//   k_samples       1 block / event        det->src, weights, event stats,    pop_wrapper.py:67-80,
//                                          per-z factors of the integrand      likelihood.py:111-121,272
//   k_kde_integrate 1 wave  / (event,pix)  histogram, KDE, interp, integrand,  math.py:32-89,
//                                          trapz                               likelihood.py:105-205,266-292
//   k_full_kde      1 block / (event,pix)  3-D Gaussian KDE                    math.py:154-229, likelihood.py:211-260
//   k_selection     grid-stride            dN/dtheta per injection + 2 sums    pop_wrapper.py:102-111
//   k_reduce/k_combine                     log, nan_to_num, sums, N_exp guard  likelihood.py:294-338, selection_function.py:38-48
#pragma once
#include "chm_models.h"

#define NSTAT 16
// per-(draw,event) statistics written by k_samples
enum { ST_ZMIN = 0, ST_ZMAX, ST_STD, ST_NORM, ST_NEFF, ST_SUMW, ST_LOGNORM, ST_L00, ST_L10, ST_L11, ST_L20, ST_L21, ST_L22,
       ST_FACTOR3, ST_X0, ST_X1 };

struct LikeDev {                  // device-resident shard of events (see chm_like_desc)
  int E, S, Z, P;
  int mode, kernel, bw_method, binning, num_bins, G, has_cut, pad;
  double bw_scalar, cut_grid, pe_neff;
  const double *dL, *m1det, *m2det, *pe_prior, *ra, *dec;
  const int* pix;
  const double *z_grids, *p_cat, *P_compl, *gw_pdf, *ra_pix, *dec_pix;
  const int* neff_pixels;
  // workspaces (nb-major)
  double *ws_z, *ws_w;            // (nb,E,S)
  double *stats;                  // (nb,E,NSTAT)
  double *pixmax;                 // (nb,E,P)
  double *jac, *prate, *bkgA;     // (nb,E,Z)
  double *like_pix;               // (nb,E,max(P,1))
  double *p_gw_dump;              // optional (nb,E,P,Z) or NULL
};

// ------------------------------------------------------------------------------------------------------
// k_tables
// ------------------------------------------------------------------------------------------------------
// cumtrapz(y, x) (math.py:22-26) of n points held in global memory, by one block: thread t owns a contiguous
// chunk; chunk totals are combined by an exclusive scan in LDS.  out[0] = 0.
DEVFN void block_cumtrapz(const double* y, const double* x, double* out, int n, double* sh /* blockDim+1 */) {
  int nt = blockDim.x, t = threadIdx.x;
  int nterm = n - 1;
  int per = (nterm + nt - 1) / nt;
  int k0 = t * per, k1 = min(k0 + per, nterm);
  double acc = 0.;
  for (int k = k0; k < k1; k++) acc += 0.5 * (y[k] + y[k + 1]) * (x[k + 1] - x[k]);
  sh[t] = acc;
  __syncthreads();
  if (t == 0) { double run = 0.; for (int i = 0; i < nt; i++) { double v = sh[i]; sh[i] = run; run += v; } }
  __syncthreads();
  acc = sh[t];
  if (t == 0) out[0] = 0.;
  for (int k = k0; k < k1; k++) { acc += 0.5 * (y[k] + y[k + 1]) * (x[k + 1] - x[k]); out[k + 1] = acc; }
  __syncthreads();
}

__global__ void __launch_bounds__(256) k_tables(DevParams* params, double* zt_all, double* It_all, double* dLt_all,
                                                 double* mg_all, double* cdf_all, double* tmp_all, int TcMax, int TmMax) {
  __shared__ double sh[260];
  int b = blockIdx.x, t = threadIdx.x;
  DevParams& P = params[b];
  int Tc = P.Tc, Tm = P.Tm;
  double* zt = zt_all + (size_t)b * TcMax;
  double* It = It_all + (size_t)b * TcMax;
  double* dLt = dLt_all + (size_t)b * TcMax;
  double* mg = mg_all + (size_t)b * TmMax;
  double* cdf = cdf_all + (size_t)b * TmMax;
  int Tmax = TcMax > TmMax ? TcMax : TmMax;
  double* tmp = tmp_all + (size_t)b * Tmax;

  if (t == 0) {                                   // per-draw constants
    double m_low = P.m[0], m_high = P.m[1];
    if (P.mass_model == 2) {
      double mu = P.m[6], sg = P.m[7];
      P.plp_plnorm = tpl_cdf(-P.m[3], m_low, m_high);
      P.tg_hi = mu + 5. * sg;
      double max_point = (P.tg_hi - mu) / (sg * sqrt(2.));
      double min_point = (m_low - mu) / (sg * sqrt(2.));
      P.tg_norm = 0.5 * erf(max_point) - 0.5 * erf(min_point);
      P.g_c0 = -0.5 * log(2. * CHM_PI) - log(sg);
    } else if (P.mass_model == 1) {
      double mb = m_low + P.m[6] * (m_high - m_low);
      P.bpl_mbreak = mb;
      P.bpl_pl1 = tpl_notnorm(mb, -P.m[2], m_low, mb);
      P.bpl_pl2 = tpl_notnorm(mb, -P.m[3], mb, m_high);
    }
    double g = P.r[0], k = P.r[1], zp = P.r[2], zmax = P.r[3];
    P.md_norm = 1. + pow(1. + zp, -g - k);
    P.tpl_rate_norm = (pow(1. + zmax, g + 1.) - 1.) / (g + 1.);
  }
  __syncthreads();

  // cosmology: zt = [0] U logspace(-10, log10 z_max, Tc-1); It = cumtrapz(1/E, zt)     cosmo.py:43-46
  double lzmax = log10(P.z_max);
  for (int i = t; i < Tc; i += blockDim.x) {
    double z = i == 0 ? 0. : pow(10., jnp_linspace_at(-10., lzmax, Tc - 1, i - 1));
    zt[i] = z;
    tmp[i] = 1. / E_at_z(P, z);
  }
  __syncthreads();
  block_cumtrapz(tmp, zt, It, Tc, sh);
  // dL table of z_from_dGW: dL_at_z(cosmo, z_grid_interp)                               cosmo.py:263
  for (int i = t; i < Tc; i += blockDim.x) {
    double z = zt[i];
    dLt[i] = dL_from_dCt(P, dCt_at_z(P, z, zt, It), z);
  }
  // fR = Vc(z1) - Vc(z0)                                                                completeness.py:54-58
  if (t == 0) {
    double v0 = Vc_from_dCt(P, dCt_at_z(P, P.zc0, zt, It));
    double v1 = Vc_from_dCt(P, dCt_at_z(P, P.zc1, zt, It));
    P.fR = v1 - v0;
  }
  // mass: m_grid = logspace(log10 m_low, log10 m_high, Tm); cdf = cumtrapz(secondary(m_grid; m_high))   mass.py:45-49
  double l0 = log10(P.m[0]), l1 = log10(P.m[1]);
  for (int i = t; i < Tm; i += blockDim.x) {
    double m = pow(10., jnp_linspace_at(l0, l1, Tm, i));
    mg[i] = m;
    tmp[i] = secondary_notnorm(P, m, P.m[1]);
  }
  __syncthreads();
  block_cumtrapz(tmp, mg, cdf, Tm, sh);
  // norm_p_m1 = trapz(primary(m_grid), m_grid) = 0.5 * sum(dx * (y1 + y0))             mass.py:50-52
  double acc = 0.;
  for (int k = t; k < Tm - 1; k += blockDim.x) {
    double y0 = primary_notnorm(P, mg[k]), y1 = primary_notnorm(P, mg[k + 1]);
    acc += (mg[k + 1] - mg[k]) * (y1 + y0);
  }
  acc = block_reduce<RED_SUM>(acc, sh);
  if (t == 0) P.norm_p_m1 = 0.5 * acc;
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
  for (int i = threadIdx.x; i < Tc; i += blockDim.x) { zt[i] = g.zt[i]; dLt[i] = g.dLt[i]; if (need_It) It[i] = g.It[i]; }
  for (int i = threadIdx.x; i < Tm; i += blockDim.x) { mg[i] = g.mg[i]; cdf[i] = g.cdf[i]; }
  __syncthreads();
  v.zt = zt; v.It = It; v.dLt = dLt; v.mg = mg; v.cdf = cdf;
  return v;
}

// ------------------------------------------------------------------------------------------------------
// k_samples: one block per (event, draw)
// ------------------------------------------------------------------------------------------------------
template <bool LDS_TAB>
__global__ void __launch_bounds__(1024) k_samples(LikeDev L, const DevParams* params, const double* zt_all, const double* It_all,
                                                   const double* dLt_all, const double* mg_all, const double* cdf_all,
                                                   int TcMax, int TmMax) {
  extern __shared__ double lds[];
  __shared__ double red[16];
  __shared__ unsigned long long pmax_bits[1024];
  const int e = blockIdx.x, b = blockIdx.y, t = threadIdx.x, nt = blockDim.x;
  const DevParams& P = params[b];
  TablePtrs g = { zt_all + (size_t)b * TcMax, It_all + (size_t)b * TcMax, dLt_all + (size_t)b * TcMax,
                  mg_all + (size_t)b * TmMax, cdf_all + (size_t)b * TmMax };
  TabView T = stage_tables(P, g, LDS_TAB, lds, true);
  const int S = L.S, Z = L.Z, Pn = L.P;
  const size_t so = ((size_t)b * L.E + e) * S;
  const size_t eo = (size_t)e * S;
  double* wz = L.ws_z + so;
  double* ww = L.ws_w + so;
  for (int i = t; i < Pn && i < 1024; i += nt) pmax_bits[i] = 0ull;
  __syncthreads();

  // pass 1: z = z_from_dGW(dL) (cosmo.py:260-264); m_src = m_det/(1+z) (pop_wrapper.py:70);
  //         w = p_m1m2 / pe_prior (pop_wrapper.py:79)
  double sw = 0., sw2 = 0., sz = 0.;
  double zmn = __builtin_inf(), zmx = -__builtin_inf();
  for (int s = t; s < S; s += nt) {
    double dl = L.dL[eo + s];
    double z = jnp_interp(dl, T.dLt, T.zt, P.Tc, false, 0., 0.);
    double m1 = L.m1det[eo + s] / (1. + z);
    double m2 = L.m2det[eo + s] / (1. + z);
    double w = p_m1m2(P, m1, m2, T.mg, T.cdf) / L.pe_prior[eo + s];
    wz[s] = z; ww[s] = w;
    sw += w; sw2 += w * w; sz += z;
    zmn = nanmin2(zmn, z); zmx = nanmax2(zmx, z);
    if (L.pix) {
      int px = L.pix[eo + s];
      if (px >= 0 && px < Pn && px < 1024) atomicMax(&pmax_bits[px], (unsigned long long)__double_as_longlong(z));
    }
  }
  sw = block_reduce<RED_SUM>(sw, red);
  sw2 = block_reduce<RED_SUM>(sw2, red);
  sz = block_reduce<RED_SUM>(sz, red);
  zmn = block_reduce<RED_MIN>(zmn, red);
  zmx = block_reduce<RED_MAX>(zmx, red);
  // pass 2: jnp.std = sqrt(mean(|z - mean|^2))   (likelihood.py:118,186,222)
  double mean = sz / (double)S;
  double sv = 0.;
  for (int s = t; s < S; s += nt) { double d = wz[s] - mean; sv += d * d; }
  sv = block_reduce<RED_SUM>(sv, red);
  double sd = sqrt(sv / (double)S);
  double* st = L.stats + ((size_t)b * L.E + e) * NSTAT;
  if (t == 0) {
    st[ST_ZMIN] = zmn; st[ST_ZMAX] = zmx; st[ST_STD] = sd;
    st[ST_NORM] = sw / (double)S;                  // jnp.mean(weights)          likelihood.py:111
    st[ST_NEFF] = (sw * sw) / sw2;                 // sum(w)^2 / sum(w^2)        likelihood.py:112
    st[ST_SUMW] = sw;
  }
  // per-pixel upper histogram edge: max(where(mask, z, min z))                   likelihood.py:180, math.py:36
  for (int i = t; i < Pn && i < 1024; i += nt) {
    double pm = __longlong_as_double((long long)pmax_bits[i]);
    L.pixmax[((size_t)b * L.E + e) * Pn + i] = (zmn != zmn) ? zmn : (pm > zmn ? pm : zmn);
  }

  // per-z factors of the integrand on the event grid                             likelihood.py:270-272, pop_wrapper.py:82-90
  const size_t zo = ((size_t)b * L.E + e) * Z;
  for (int k = t; k < Z; k += nt) {
    double z = L.z_grids[(size_t)e * Z + k];
    double dCt = dCt_at_z(P, z, T.zt, T.It);
    double zp1 = 1. + z;
    L.jac[zo + k] = ddLdz_from_dCt(P, dCt, z) * (zp1 * zp1);
    L.prate[zo + k] = merger_rate(P, z) / (1. + z);
    double p_bkg = dVcdz_from_dCt(P, dCt, z);
    L.bkgA[zo + k] = P.has_catalog ? (1. - L.P_compl[(size_t)e * Z + k]) * p_bkg : p_bkg;   // catalog.py:202 / :43
  }

  // full mode: weighted mean / covariance / whitening of (z, ra, dec)            math.py:173-197
  if (L.mode == 3) {
    double m0 = 0., m1 = 0., m2 = 0., sW2 = 0.;
    for (int s = t; s < S; s += nt) {
      double W = ww[s] / sw;
      m0 += W * wz[s]; m1 += W * L.ra[eo + s]; m2 += W * L.dec[eo + s]; sW2 += W * W;
    }
    m0 = block_reduce<RED_SUM>(m0, red); m1 = block_reduce<RED_SUM>(m1, red);
    m2 = block_reduce<RED_SUM>(m2, red); sW2 = block_reduce<RED_SUM>(sW2, red);
    double c00 = 0., c01 = 0., c02 = 0., c11 = 0., c12 = 0., c22 = 0.;
    for (int s = t; s < S; s += nt) {
      double W = ww[s] / sw;
      double r0 = wz[s] - m0, r1 = L.ra[eo + s] - m1, r2 = L.dec[eo + s] - m2;
      c00 += r0 * W * r0; c01 += r0 * W * r1; c02 += r0 * W * r2;
      c11 += r1 * W * r1; c12 += r1 * W * r2; c22 += r2 * W * r2;
    }
    c00 = block_reduce<RED_SUM>(c00, red); c01 = block_reduce<RED_SUM>(c01, red); c02 = block_reduce<RED_SUM>(c02, red);
    c11 = block_reduce<RED_SUM>(c11, red); c12 = block_reduce<RED_SUM>(c12, red); c22 = block_reduce<RED_SUM>(c22, red);
    if (t == 0) {
      double den = 1. - sW2;
      c00 /= den; c01 /= den; c02 /= den; c11 /= den; c12 /= den; c22 /= den;
      double neff = 1. / sW2, factor;
      if (L.bw_method == 0) factor = pow(neff, -1. / 7.);
      else if (L.bw_method == 1) factor = pow(neff * 5. / 4.0, -1. / 7.);
      else factor = L.bw_scalar;
      // inverse of the symmetric 3x3 covariance (adjugate / determinant)
      double a00 = c11 * c22 - c12 * c12, a01 = c02 * c12 - c01 * c22, a02 = c01 * c12 - c02 * c11;
      double a11 = c00 * c22 - c02 * c02, a12 = c01 * c02 - c00 * c12, a22 = c00 * c11 - c01 * c01;
      double det = c00 * a00 + c01 * a01 + c02 * a02;
      double f2 = factor * factor;
      double i00 = a00 / det / f2, i01 = a01 / det / f2, i02 = a02 / det / f2;
      double i11 = a11 / det / f2, i12 = a12 / det / f2, i22 = a22 / det / f2;
      // lower Cholesky factor of inv_cov
      double l00 = sqrt(i00), l10 = i01 / l00, l20 = i02 / l00;
      double l11 = sqrt(i11 - l10 * l10), l21 = (i12 - l20 * l10) / l11;
      double l22 = sqrt(i22 - l20 * l20 - l21 * l21);
      st[ST_L00] = l00; st[ST_L10] = l10; st[ST_L11] = l11; st[ST_L20] = l20; st[ST_L21] = l21; st[ST_L22] = l22;
      st[ST_LOGNORM] = (log(l00) + log(l11) + log(l22)) - 0.5 * 3. * log(2. * CHM_PI);
      st[ST_FACTOR3] = factor;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// k_kde_integrate: one wave per (event, pixel, draw)   [modes 1d / approximate / marginalized]
// ------------------------------------------------------------------------------------------------------
// Dynamic LDS: data[N] (bin centres or raw z), wgt[N] (bin counts -> normalised weights), eff[G], dens[G];
// N = num_bins when binning else S.
__global__ void __launch_bounds__(64) k_kde_integrate(LikeDev L, const DevParams* params) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int Pd = L.P > 0 ? L.P : 1;
  const int p = blockIdx.x % Pd, e = blockIdx.x / Pd, b = blockIdx.y;
  const DevParams& P = params[b];
  const int S = L.S, Z = L.Z, B = L.num_bins, G = L.G;
  const int N = L.binning ? B : S;
  double* data = lds; double* wgt = data + N; double* eff = wgt + N; double* dens = eff + G;
  const double* st = L.stats + ((size_t)b * L.E + e) * NSTAT;
  const size_t so = ((size_t)b * L.E + e) * S;
  const double* wz = L.ws_z + so;
  const double* ww = L.ws_w + so;
  const int* pix = L.pix ? L.pix + (size_t)e * S : nullptr;
  const bool marg = L.mode == 2;
  const bool pixelated = L.mode != 0;
  double* out_like = L.like_pix + ((size_t)b * L.E + e) * Pd + p;
  double* dump = L.p_gw_dump ? L.p_gw_dump + (((size_t)b * L.E + e) * Pd + p) * Z : nullptr;

  const int npix = pixelated ? L.neff_pixels[e] : 1;
  if (p >= npix) {                    // padded pixel: p_cat == -100 there, integrand masked to 0 (likelihood.py:274-277)
    if (lane == 0) *out_like = 0.;
    if (dump) for (int k = lane; k < Z; k += 64) dump[k] = 0.;
    return;
  }
  const double zmin = st[ST_ZMIN], zmax = st[ST_ZMAX], sd = st[ST_STD], norm = st[ST_NORM], n_eff = st[ST_NEFF];
  const bool ok = n_eff >= L.pe_neff;                       // lax.cond(n_eff >= pe_neff, ...)   likelihood.py:133,199
  const double gwp = pixelated ? L.gw_pdf[(size_t)e * L.P + p] : 1.;

  if (ok) {
    // ---- dataset for the KDE: binned (math.py:32-46) or raw samples
    const double lo = zmin;
    const double hi = marg ? L.pixmax[((size_t)b * L.E + e) * L.P + p] : zmax;
    if (L.binning) {
      for (int j = lane; j < B; j += 64) {
        double e0 = jnp_linspace_at(lo, hi, B + 1, j), e1 = jnp_linspace_at(lo, hi, B + 1, j + 1);
        data[j] = (e0 + e1) / 2.;
        wgt[j] = 0.;
      }
      __syncthreads();
      for (int s0 = 0; s0 < S; s0 += 64) {
        int s = s0 + lane;
        if (s < S && (!marg || pix[s] == p)) {
          double z = wz[s], w = ww[s];
          double f = floor((z - lo) / (hi - lo) * (double)B);
          f = f < 0. ? 0. : (f > (double)(B - 1) ? (double)(B - 1) : f);
          int idx = (f != f) ? 0 : (int)f;
          atomicAdd(&wgt[idx], w);
        }
      }
      __syncthreads();
    } else {
      for (int s = lane; s < S; s += 64) {
        bool in = !marg || pix[s] == p;
        data[s] = in ? wz[s] : zmin;                        // likelihood.py:180-181
        wgt[s] = in ? ww[s] : 0.;
      }
      __syncthreads();
    }
    // ---- kde1d prologue (math.py:58-75): normalise weights, neff, std(dataset), bandwidth
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
    double bw;
    if (L.bw_method == 0) bw = pow(neff_k, -1. / 5.);
    else if (L.bw_method == 1) bw = pow(neff_k * 3. / 4.0, -1. / 5.);
    else bw = L.bw_scalar;
    bw *= stdc;
    // ---- effective grid (likelihood.py:115-123, 185-190)
    if (L.has_cut) {
      double lb = zmin - L.cut_grid * sd;
      if (marg) lb = (lb != lb) ? lb : (lb > 1e-8 ? lb : 1e-8);      // jnp.maximum(., 1e-8)         :186
      else lb = lb > 0. ? lb : 1.e-8;                                 // jnp.where(. > 0, ., 1e-8)    :119
      double ub = zmax + L.cut_grid * sd;
      for (int i = lane; i < G; i += 64) eff[i] = jnp_linspace_at(lb, ub, G, i);
    } else {
      for (int i = lane; i < G; i += 64) eff[i] = L.z_grids[(size_t)e * Z + i];
    }
    __syncthreads();
    // ---- density on the effective grid (math.py:77-81).  u = (g - x) * (1/bw): one rounding away from the
    //      reference's (g - x)/bw.  The Epanechnikov sum runs over the bins that can have |u| <= 1 only; the
    //      skipped terms are exact zeros, so the sum is the one the dense product gives.
    const bool epan = marg || L.kernel == 0;                 // p_gw3dmarg never passes kernel= (likelihood.py:192)
    const double inv_bw = 1. / bw;
    const double dbin = (hi - lo) / (double)B;
    const bool window = epan && L.binning && dbin > 0. && bw > 0. && bw < 1e300;
    for (int i = lane; i < G; i += 64) {
      double g = eff[i];
      int j0 = 0, j1 = N - 1;
      if (window) {
        double f0 = floor((g - bw - lo) / dbin - 0.5) - 1., f1 = ceil((g + bw - lo) / dbin - 0.5) + 1.;
        j0 = f0 > 0. ? (f0 < (double)N ? (int)f0 : N) : 0;
        j1 = f1 < (double)(N - 1) ? (f1 >= 0. ? (int)f1 : -1) : N - 1;
      }
      double acc = 0.;
      if (epan) {
        for (int j = j0; j <= j1; j++) {
          double u = (g - data[j]) * inv_bw;
          double kv = fabs(u) <= 1. ? 0.75 * (1. - u * u) : 0.;
          acc += wgt[j] * kv;
        }
      } else {
        const double isq = 1. / sqrt(2. * CHM_PI);
        for (int j = j0; j <= j1; j++) {
          double u = (g - data[j]) * inv_bw;
          acc += wgt[j] * (exp(-0.5 * (u * u)) * isq);
        }
      }
      double d = acc / bw;
      dens[i] = marg ? d : d * norm;                         // 1-D: kde*norms before interp (likelihood.py:137)
    }
    __syncthreads();
  }

  // ---- interp to the event grid, integrand, trapezoid (likelihood.py:137/193, 274-278 / 291)
  const double* zg = L.z_grids + (size_t)e * Z;
  const size_t zo = ((size_t)b * L.E + e) * Z;
  const double* jac = L.jac + zo;
  const double* prate = L.prate + zo;
  const double* bkgA = L.bkgA + zo;
  const double* pc = pixelated ? L.p_cat + ((size_t)e * L.P + p) * Z : nullptr;
  const double fR = P.fR;
  double acc = 0.;
  for (int k0 = 0; k0 < Z - 1 || k0 == 0; k0 += 63) {
    int k = k0 + lane;
    double zk = 0., y = 0.;
    if (k < Z) {
      zk = zg[k];
      double pgw = 0.;
      if (ok) {
        double f = jnp_interp(zk, eff, dens, G, true, 0., 0.);
        pgw = marg ? f * norm * gwp : (pixelated ? f * gwp : f);
      }
      if (dump) dump[k] = pgw;
      if (pixelated) {
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
  acc = wave_sum(acc);
  if (lane == 0) *out_like = 0.5 * acc;
}

// ------------------------------------------------------------------------------------------------------
// k_full_kde: 3-D Gaussian KDE, one block (256 threads) per (event, pixel, draw)   likelihood.py:211-260
// ------------------------------------------------------------------------------------------------------
#define FULL_TILE 1024
__global__ void __launch_bounds__(256) k_full_kde(LikeDev L, const DevParams* params) {
  __shared__ double xs0[FULL_TILE], xs1[FULL_TILE], xs2[FULL_TILE], xw[FULL_TILE];
  __shared__ double red[16];
  const int t = threadIdx.x, nt = blockDim.x;
  const int p = blockIdx.x % L.P, e = blockIdx.x / L.P, b = blockIdx.y;
  const DevParams& P = params[b];
  const int S = L.S, Z = L.Z;
  const double* st = L.stats + ((size_t)b * L.E + e) * NSTAT;
  const size_t so = ((size_t)b * L.E + e) * S;
  const size_t eo = (size_t)e * S;
  const double* wz = L.ws_z + so;
  const double* ww = L.ws_w + so;
  double* out_like = L.like_pix + ((size_t)b * L.E + e) * L.P + p;
  double* dump = L.p_gw_dump ? L.p_gw_dump + (((size_t)b * L.E + e) * L.P + p) * Z : nullptr;
  const int npix = L.neff_pixels[e];
  if (p >= npix) {                                        // result[ev, :npix] only (likelihood.py:253)
    if (t == 0) *out_like = 0.;
    if (dump) for (int k = t; k < Z; k += nt) dump[k] = 0.;
    return;
  }
  const double zmin = st[ST_ZMIN], zmax = st[ST_ZMAX], sd = st[ST_STD], norm = st[ST_NORM], n_eff = st[ST_NEFF], sumw = st[ST_SUMW];
  const bool ok = !(n_eff < L.pe_neff);                   // `if n_effs[ev] < pe_neff: continue`   likelihood.py:234
  const double l00 = st[ST_L00], l10 = st[ST_L10], l11 = st[ST_L11], l20 = st[ST_L20], l21 = st[ST_L21], l22 = st[ST_L22];
  const double log_norm = st[ST_LOGNORM];
  const double zhi = zmax + L.cut_grid * sd, zlo = zmin - L.cut_grid * sd;      // likelihood.py:225
  const double* zg = L.z_grids + (size_t)e * Z;
  const double rp = L.ra_pix[(size_t)e * L.P + p], dp = L.dec_pix[(size_t)e * L.P + p];
  // whitened query: q = (z, ra_p, dec_p) . L   (math.py:196): q0 = z*l00 + ra*l10 + dec*l20; q1 = ra*l11 + dec*l21; q2 = dec*l22
  const double q1 = rp * l11 + dp * l21, q2 = dp * l22;
  const int KPT = (Z + nt - 1) / nt;                      // grid points per thread (<= 8 supported per pass)
  const size_t zo = ((size_t)b * L.E + e) * Z;
  double accl = 0.;
  for (int kb = 0; kb < Z; kb += nt * 4) {
    double q0[4], val[4]; bool inm[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int k = kb + r * nt + t;
      double z = k < Z ? zg[k] : 0.;
      inm[r] = ok && k < Z && (z <= zhi) && (z >= zlo);
      q0[r] = z * l00 + rp * l10 + dp * l20;
      val[r] = 0.;
    }
    if (ok) {
      for (int s0 = 0; s0 < S; s0 += FULL_TILE) {
        __syncthreads();
        for (int s = t; s < FULL_TILE && s0 + s < S; s += nt) {
          double x0 = wz[s0 + s], x1 = L.ra[eo + s0 + s], x2 = L.dec[eo + s0 + s];
          xs0[s] = x0 * l00 + x1 * l10 + x2 * l20;
          xs1[s] = x1 * l11 + x2 * l21;
          xs2[s] = x2 * l22;
          xw[s] = ww[s0 + s] / sumw;
        }
        __syncthreads();
        int ns = min(FULL_TILE, S - s0);
        for (int s = 0; s < ns; s++) {
          double d1 = xs1[s] - q1, d2 = xs2[s] - q2;
          double base = d1 * d1 + d2 * d2;
          double w = xw[s], a0 = xs0[s];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            double d0 = a0 - q0[r];
            val[r] += w * exp(log_norm - 0.5 * (d0 * d0 + base));
          }
        }
      }
    }
    // integrand + trapezoid for these grid points: y_k needs y_{k+1}; store y in the dump-free way via LDS ring
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int k = kb + r * nt + t;
      if (k < Z) {
        double pgw = inm[r] ? val[r] * norm : 0.;
        if (dump) dump[k] = pgw;
        double pcv = L.p_cat[((size_t)e * L.P + p) * Z + k];
        double y = 0.;
        if (pcv != -100.) {
          double p_gal = P.fR * pcv + L.bkgA[zo + k];
          double p_z = p_gal * L.prate[zo + k];
          y = (p_z != -100.) ? pgw * p_z / L.jac[zo + k] : 0.;
        }
        // trapezoid weights: y_k * (z_{k+1} - z_{k-1}) / 2 at interior points -- written as the sum of the two
        // adjacent half-intervals so that no neighbour exchange is needed
        double zl = k > 0 ? zg[k - 1] : zg[k], zr = k < Z - 1 ? zg[k + 1] : zg[k];
        accl += y * ((zg[k] - zl) + (zr - zg[k]));
      }
    }
  }
  (void)KPT;
  accl = block_reduce<RED_SUM>(accl, red);
  if (t == 0) *out_like = 0.5 * accl;
}

// ------------------------------------------------------------------------------------------------------
// k_selection: dN/dtheta_det per injection (pop_wrapper.py:102-111) / p_draw, block partial sums
// ------------------------------------------------------------------------------------------------------
struct SelDev {
  long long I;
  const double *dL, *m1det, *m2det, *p_draw;
  double N_inj, N_eff; int has_neff, pad;
  double* partial;                // (nb, nblocks, 2)
  int nblocks;
};

template <bool LDS_TAB>
__global__ void __launch_bounds__(256) k_selection(SelDev Sd, const DevParams* params, const double* zt_all, const double* It_all,
                                                    const double* dLt_all, const double* mg_all, const double* cdf_all,
                                                    int TcMax, int TmMax) {
  extern __shared__ double lds[];
  __shared__ double red[16];
  const int b = blockIdx.y;
  const DevParams& P = params[b];
  TablePtrs g = { zt_all + (size_t)b * TcMax, It_all + (size_t)b * TcMax, dLt_all + (size_t)b * TcMax,
                  mg_all + (size_t)b * TmMax, cdf_all + (size_t)b * TmMax };
  TabView T = stage_tables(P, g, LDS_TAB, lds, false);
  double s1 = 0., s2 = 0.;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < Sd.I; i += (long long)gridDim.x * blockDim.x) {
    double dl = Sd.dL[i];
    double z = jnp_interp(dl, T.dLt, T.zt, P.Tc, false, 0., 0.);
    double m1 = Sd.m1det[i] / (1. + z), m2 = Sd.m2det[i] / (1. + z);
    double dCt = dL2dCt(P, dl, z);                                   // original distances: cosmo.py:191-192,215-216
    double p_z = dVcdz_from_dCt(P, dCt, z);                          // gal_cat.p_bkg              pop_wrapper.py:106
    p_z = p_z * (merger_rate(P, z) / (1. + z));                      //                            pop_wrapper.py:107
    double dN = P.R0 * p_m1m2(P, m1, m2, T.mg, T.cdf) * p_z;         //                            pop_wrapper.py:108
    double zp1 = 1. + z;
    double jacobian = fabs(ddLdz_from_dCt(P, dCt, z)) * (zp1 * zp1); //                            pop_wrapper.py:109
    dN = dN / jacobian;
    dN = dN / Sd.p_draw[i];                                          // selection_function.py:38
    if (dN == dN) s1 += dN;                                          // nansum                     selection_function.py:39
    s2 += dN * dN;                                                   // plain sum     selection_function.py:44
  }
  s1 = block_reduce<RED_SUM>(s1, red);
  s2 = block_reduce<RED_SUM>(s2, red);
  if (threadIdx.x == 0) {
    double* o = Sd.partial + ((size_t)b * Sd.nblocks + blockIdx.x) * 2;
    o[0] = s1; o[1] = s2;
  }
}

// ------------------------------------------------------------------------------------------------------
// k_reduce: shard partials [sum_i nan_to_num(log L_i), nansum dN, sum dN^2] per draw; optional per-event outputs
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_reduce(int E, int Pd, const double* like_pix, int nblocks_sel, const double* sel_partial,
                                                 double* partials /* (nb,3) */, double* log_like_evs, double* numlike_evs) {
  __shared__ double red[16];
  const int b = blockIdx.x, t = threadIdx.x;
  double acc = 0.;
  bool any_neginf = false;
  for (int e = t; e < E; e += blockDim.x) {
    const double* lp = like_pix + ((size_t)b * E + e) * Pd;
    double Li = 0.;
    for (int p = 0; p < Pd; p++) Li += lp[p];                        // jnp.sum over pixels          likelihood.py:280
    double ll = log(Li);                                             // likelihood.py:296,329
    // jnp.nan_to_num(x, nan=-inf): NaN -> -inf, -inf -> -DBL_MAX, +inf -> DBL_MAX   
    if (ll != ll) ll = -__builtin_inf();
    else if (ll == -__builtin_inf()) ll = -1.7976931348623157e308;
    else if (ll == __builtin_inf()) ll = 1.7976931348623157e308;
    if (numlike_evs) numlike_evs[(size_t)b * E + e] = Li;
    if (log_like_evs) log_like_evs[(size_t)b * E + e] = ll;
    if (ll == -__builtin_inf()) any_neginf = true; else acc += ll;
  }
  acc = block_reduce<RED_SUM>(acc, red);
  double inf_flag = block_reduce<RED_SUM>(any_neginf ? 1. : 0., red);
  if (inf_flag > 0.) acc = -__builtin_inf();
  double s1 = 0., s2 = 0.;
  for (int i = t; i < nblocks_sel; i += blockDim.x) {
    s1 += sel_partial[((size_t)b * nblocks_sel + i) * 2];
    s2 += sel_partial[((size_t)b * nblocks_sel + i) * 2 + 1];
  }
  s1 = block_reduce<RED_SUM>(s1, red);
  s2 = block_reduce<RED_SUM>(s2, red);
  if (t == 0) { partials[b * 3] = acc; partials[b * 3 + 1] = s1; partials[b * 3 + 2] = s2; }
}

// k_combine: N_exp with the N_eff guard (selection_function.py:38-47) and the final combination
// (likelihood.py:298-300, 313-316, 331-337).  out: (nb,3) = [log_hyper, log_num, N_exp]
__global__ void k_combine(int nb, const DevParams* params, const double* partials, double E_total, double N_inj, double N_eff,
                          int has_neff, int has_like, int has_sel, double* out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const DevParams& P = params[b];
  double log_num = partials[b * 3];
  double Nexp = __builtin_nan("");
  if (has_sel) {
    double xi = partials[b * 3 + 1] / N_inj;
    Nexp = P.Tobs * xi;
    if (has_neff) {
      double variance2 = partials[b * 3 + 2] / (N_inj * N_inj) - (xi * xi) / N_inj;
      double neff = (xi * xi) / variance2;
      if (neff < N_eff) Nexp = 0.0;
    }
  }
  double log_hyper = __builtin_nan("");
  if (has_like) {
    if (!P.scale_free) log_num += E_total * log(P.R0 * P.Tobs);
    if (has_sel) log_hyper = P.scale_free ? log_num - E_total * log(Nexp) : log_num - Nexp;
  } else log_num = __builtin_nan("");
  out[b * 3] = log_hyper; out[b * 3 + 1] = log_num; out[b * 3 + 2] = Nexp;
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
      case 9: r = jnp_interp(x, g.dLt, g.zt, P.Tc, false, 0., 0.); break;
      case 10: r = merger_rate(P, x); break;
      case 11: r = p_m1m2(P, x, bb[i], g.mg, g.cdf); break;
      case 12: r = primary_notnorm(P, x); break;
      case 13: r = secondary_notnorm(P, x, bb[i]); break;
      case 14: r = smoothing(x, mass_delta_m(P), P.m[0]); break;
    }
    out[i] = r;
  }
}
