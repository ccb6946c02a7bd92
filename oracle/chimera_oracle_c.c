/*
 * oracle/chimera_oracle_c.c -- plain-C (OpenMP) restatement of CHIMERA's marginalized hyper-likelihood path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under chimera_amd/ links, loads or calls this file.  It has two uses:
 *   1. an independent second restatement that cross-checks oracle/chimera_oracle.py (tests/test_oracle_c.py:
 *      both must agree to rounding on the same inputs), and
 *   2. the all-cores CPU baseline of bench.py (`cpu_baseline.kind = "port"`, `cores` = OpenMP threads used),
 *      so that the GPU/CPU ratio is not quoted against a single NumPy thread only.
 *
 * PARITY UNPINNED (as oracle/chimera_oracle.py): the reference ships no tests or golden vectors for this path and
 * cannot be imported here; every function below cites the reference lines it follows (paths relative to the
 * reference root, CHIMERA/...), and the operation order is the reference's (dense G x B kernel sums, two-pass
 * standard deviations, jnp.interp / jnp.linspace / jnp.trapezoid forms).
 *
 * Scope: flrw / mg_flrw cosmology, tpl / bpl / plp mass models, the four rate models, the pixelated catalogue with
 * step completeness, kind_p_gw3d = 'marginalized' (likelihood.py:160-205, 266-281), the 1-D and 'approximate' modes
 * (likelihood.py:105-154, 283-292), kind_p_gw3d = 'full' (likelihood.py:211-260 with utils/math.py:154-229) and the injection selection
 * function (selection_function.py:34-48, pop_wrapper.py:102-111).  Parameter block: `chm_params` of
 * include/chimera_hip.h (plain data; the same struct the product's C ABI takes).
 *
 * Build: make -C oracle   ->  oracle/libchimera_oracle_c.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/chimera_hip.h"

#define ORC_PI 3.141592653589793238462643383279502884

/* ------------------------------------------------------------------------------------------------------
 * jax.numpy semantics
 * ---------------------------------------------------------------------------------------------------- */
/* jnp.linspace(start, stop, num)[i]: start*(1 - i/div) + stop*(i/div), last point = stop */
static double lin_at(double start, double stop, int num, int i) {
  int div = num - 1;
  if (i >= div) return stop;
  double step = (double)i / (double)div;
  return start * (1. - step) + stop * step;
}

/* searchsorted(xp, x, side='right') */
static int ss_right(const double* xp, int n, double x) {
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid + 1; else hi = mid; }
  return lo;
}

/* jnp.searchsorted(..., side='right', method='scan') step by step (jax/_src/numpy/lax_numpy.py:_searchsorted_via_scan): the same
 * answer as ss_right on a sorted table; on a non-monotonic one (dL table of an unphysical cosmology) the probe sequence matters */
static int ss_right_scan(const double* xp, int n, double x) {
  int lo = 0, hi = n, levels = 0;
  while ((1LL << levels) < (long long)n + 1) levels++;
  for (int l = 0; l < levels; l++) {
    int mid = (lo + hi) / 2;
    double v = xp[mid];
    int go_left = (x < v) || ((v != v) && (x == x));      /* NaN sorts last */
    if (go_left) hi = mid; else lo = mid;
  }
  return hi;
}

/* jnp.interp(x, xp, fp, left, right); has_lr == 0 -> clamp; scan != 0: the table may be non-monotonic */
static double interp1s(double x, const double* xp, const double* fp, int n, int has_lr, double left, double right, int scan) {
  int i = scan ? ss_right_scan(xp, n, x) : ss_right(xp, n, x);
  if (i < 1) i = 1;
  if (i > n - 1) i = n - 1;
  double df = fp[i] - fp[i - 1], dx = xp[i] - xp[i - 1], delta = x - xp[i - 1];
  double f = (fabs(dx) <= 4.930380657631324e-32) ? fp[i - 1] : fp[i - 1] + (delta / dx) * df;
  if (x < xp[0]) f = has_lr ? left : fp[0];
  if (x > xp[n - 1]) f = has_lr ? right : fp[n - 1];
  return f;
}
static double interp1(double x, const double* xp, const double* fp, int n, int has_lr, double left, double right) {
  return interp1s(x, xp, fp, n, has_lr, left, right, 0);
}

/* ------------------------------------------------------------------------------------------------------
 * models
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
  const chm_params* p;
  int Tc, Tm;
  double *zt, *It, *dLt, *mg, *cdf;
  double H0, Om0, Ok0, Or0, w0, wa, Xi0, n_mg, Ode0, dH;
  double norm_p_m1, fR;
  int dl_unsorted;                 /* the dL table is non-monotonic or holds NaNs: z_from_dGW follows jax's scan search */
} orc_model;

/* cosmo.py:122-130 */
static double E_at_z(const orc_model* m, double z) {
  double w_z = m->w0 + m->wa * z / (1. + z);
  return sqrt(m->Om0 * pow(1. + z, 3.) + m->Or0 * pow(1. + z, 4.) + m->Ok0 * pow(1. + z, 2.) +
              m->Ode0 * pow(1. + z, 3. * (1. + w_z)));
}
/* cosmo.py:141-153 */
static double dCt_from_dCr(const orc_model* m, double dCr) {
  double sq = sqrt(fabs(m->Ok0 + 1.e-10));
  if (m->Ok0 == 0.0) return dCr;
  if (m->Ok0 > 0.0) return (m->dH / sq) * sinh(sq * dCr / m->dH);
  return (m->dH / sq) * sin(sq * dCr / m->dH);
}
static double dCt_at_z(const orc_model* m, double z) {
  return dCt_from_dCr(m, m->dH * interp1(z, m->zt, m->It, m->Tc, 0, 0., 0.));          /* cosmo.py:132-139 */
}
/* cosmo.py:225-228 */
static double Xi_at_z(const orc_model* m, double z) { return m->Xi0 + (1. - m->Xi0) / pow(1. + z, m->n_mg); }
/* cosmo.py:201-203, 230-235 */
static double dL2dCt(const orc_model* m, double dist, double z) {
  if (m->p->cosmo_model == 1) return (dist / Xi_at_z(m, z)) / (1. + z);
  return dist / (1. + z);
}
/* cosmo.py:166-186 */
static double Vc_from_dCt(const orc_model* m, double dCt) {
  double regOk0 = m->Ok0 + 1e-10, sq = sqrt(fabs(regOk0)), dH = m->dH;
  if (m->Ok0 == 0.0) return 4. * ORC_PI * (dCt * dCt * dCt) / 3.;
  double pre = 4. * ORC_PI * (dH * dH * dH) / (2. * regOk0);
  double a = (dCt / dH) * sqrt(1. + regOk0 * (dCt * dCt) / (dH * dH));
  if (m->Ok0 > 0.0) return pre * (a - asinh(sq * dCt / dH) / sq);
  return pre * (a - asin(sq * dCt / dH) / sq);
}
/* cosmo.py:188-197 */
static double dVcdz(const orc_model* m, double dCt, double z) { return 4. * ORC_PI * m->dH * (dCt * dCt) / E_at_z(m, z); }
/* cosmo.py:205-210, 237-243 */
static double dL_at_z(const orc_model* m, double z) {
  double dL = dCt_at_z(m, z) * (1. + z);
  return m->p->cosmo_model == 1 ? dL * Xi_at_z(m, z) : dL;
}
/* cosmo.py:212-221, 245-257 */
static double ddLdz(const orc_model* m, double dCt, double z) {
  double Ez = E_at_z(m, z);
  double ddLflrw = dCt + (m->dH / Ez) * (1. + z);
  if (m->p->cosmo_model == 1) {
    double dLflrw = dCt * (1. + z);
    double Xiz = Xi_at_z(m, z);
    double dXiz = m->n_mg * (m->Xi0 - 1.) / pow(1. + z, m->n_mg + 1.);
    return ddLflrw * Xiz + dLflrw * dXiz;
  }
  return ddLflrw;
}

/* mass.py:240-245 */
static double tpl_notnorm(double x, double alpha, double m_low, double m_high) {
  return (m_low <= x && x <= m_high) ? pow(x, alpha) : 0.;
}
/* mass.py:247-252 */
static double tpl_cdf(double alpha, double m_low, double x) {
  if (alpha == -1.) return log(m_low) - log(x);
  return (pow(x, 1. + alpha) - pow(m_low, 1. + alpha)) / (1. + alpha);
}
/* jnp.logaddexp(0, x) */
static double logaddexp0(double x) {
  if (x != x) return x;
  return (x > 0. ? x : 0.) + log1p(exp(-fabs(x)));
}
/* mass.py:255-264 */
static double smoothing(double x, double delta_m, double m_low) {
  const double eps = 1.e-99;
  if (x < m_low) return 0.;                    /* exp(-inf) */
  if (x > m_low + delta_m) return 1.;          /* exp(0)    */
  return exp(-logaddexp0(delta_m / (x - m_low + eps) + delta_m / (x - m_low - delta_m + eps)));
}
/* mass.py:267-279 */
static double trunc_gauss(double x, double mu, double sg, double x_min, double x_max) {
  double max_point = (x_max - mu) / (sg * sqrt(2.)), min_point = (x_min - mu) / (sg * sqrt(2.));
  double norm = 0.5 * erf(max_point) - 0.5 * erf(min_point);
  if (!(x_min <= x && x <= x_max)) return 0.;
  double log_G = -0.5 * log(2. * ORC_PI) - log(sg) - ((x - mu) * (x - mu)) / (2. * (sg * sg));
  return exp(log_G) / norm;
}
/* mass.py:285-305 */
static double primary_notnorm(const chm_params* p, double x) {
  const double* q = p->mass;
  double m_low = q[0], m_high = q[1];
  if (p->mass_model == 0) return tpl_notnorm(x, -q[2], m_low, m_high);
  if (p->mass_model == 1) {
    double mb = m_low + q[6] * (m_high - m_low);
    double pl1 = tpl_notnorm(mb, -q[2], m_low, mb), pl2 = tpl_notnorm(mb, -q[3], mb, m_high);
    double pdf = tpl_notnorm(x, -q[2], m_low, mb);
    pdf = pdf + tpl_notnorm(x, -q[3], mb, m_high) * pl1 / pl2;
    return pdf * smoothing(x, q[5], m_low);
  }
  double lam = q[2], mu = q[6], sg = q[7];
  double P = tpl_notnorm(x, -q[3], m_low, m_high) / tpl_cdf(-q[3], m_low, m_high);
  double G = trunc_gauss(x, mu, sg, m_low, mu + 5. * sg);
  double pdf = (1. - lam) * P + lam * G;
  return pdf * smoothing(x, q[5], m_low);
}
static double mass_beta(const chm_params* p) { return p->mass_model == 0 ? p->mass[3] : p->mass[4]; }
/* mass.py:320-328 */
static double secondary_notnorm(const chm_params* p, double m2, double m1) {
  double pdf = tpl_notnorm(m2, mass_beta(p), p->mass[0], m1);
  if (p->mass_model == 0) return pdf;
  return pdf * smoothing(m2, p->mass[5], p->mass[0]);
}
/* mass.py:334-341 */
static double p_m1m2(const orc_model* m, double m1, double m2) {
  double p_m1 = primary_notnorm(m->p, m1) / m->norm_p_m1;
  double p_m2m1 = secondary_notnorm(m->p, m2, m1) / interp1(m1, m->mg, m->cdf, m->Tm, 0, 0., 0.);
  if (p_m2m1 != p_m2m1) p_m2m1 = 0.;
  return p_m1 * p_m2m1;
}

/* rate.py:96-122 */
static double merger_rate(const chm_params* p, double z) {
  const double* r = p->rate;
  double g = r[0];
  if (p->rate_model == 0) return pow(1. + z, g);
  if (p->rate_model == 2) {
    double norm = (pow(1. + r[3], g + 1.) - 1.) / (g + 1.);
    return z < r[3] ? pow(1. + z, g) / norm : 0.;
  }
  double k = r[1], zp = r[2];
  double md = pow(1. + z, g) / (1. + pow((1. + z) / (1. + zp), g + k));
  double one_over_norm = 1. + pow(1. + zp, -g - k);
  if (p->rate_model == 1) return one_over_norm * md;
  return z < r[3] ? one_over_norm * md : 0.;
}

/* cumtrapz(y, x), math.py:22-26 */
static void cumtrapz(const double* y, const double* x, double* out, int n) {
  double acc = 0.;
  out[0] = 0.;
  for (int k = 0; k < n - 1; k++) { acc += 0.5 * (y[k] + y[k + 1]) * (x[k + 1] - x[k]); out[k + 1] = acc; }
}

static void model_free(orc_model* m) { free(m->zt); free(m->It); free(m->dLt); free(m->mg); free(m->cdf); }

/* tables: cosmo.py:43-46 (setup_interp), :263 (dL table); mass.py:45-52 (get_normalizations); completeness.py:54-58 (fR) */
static int model_init(orc_model* m, const chm_params* p) {
  memset(m, 0, sizeof(*m));
  m->p = p; m->Tc = p->z_grid_res; m->Tm = p->mass_grid_res;
  m->H0 = p->cosmo[CHM_C_H0]; m->Om0 = p->cosmo[CHM_C_OM0]; m->Ok0 = p->cosmo[CHM_C_OK0]; m->Or0 = p->cosmo[CHM_C_OR0];
  m->w0 = p->cosmo[CHM_C_W0]; m->wa = p->cosmo[CHM_C_WA]; m->Xi0 = p->cosmo[CHM_C_XI0]; m->n_mg = p->cosmo[CHM_C_N];
  m->Ode0 = 1.0 - m->Om0 - m->Or0 - m->Ok0;                        /* cosmo.py:79-81 */
  m->dH = 299792.458e-3 / m->H0;                                   /* cosmo.py:82-84 */
  int Tc = m->Tc, Tm = m->Tm;
  m->zt = calloc(Tc, sizeof(double)); m->It = calloc(Tc, sizeof(double)); m->dLt = calloc(Tc, sizeof(double));
  m->mg = calloc(Tm, sizeof(double)); m->cdf = calloc(Tm, sizeof(double));
  double* tmp = calloc((size_t)(Tc > Tm ? Tc : Tm), sizeof(double));
  if (!m->zt || !m->It || !m->dLt || !m->mg || !m->cdf || !tmp) { free(tmp); model_free(m); return -1; }
  double lzmax = log10(p->z_max);
  for (int i = 0; i < Tc; i++) {
    m->zt[i] = i == 0 ? 0. : pow(10., lin_at(-10., lzmax, Tc - 1, i - 1));
    tmp[i] = 1. / E_at_z(m, m->zt[i]);
  }
  cumtrapz(tmp, m->zt, m->It, Tc);
  for (int i = 0; i < Tc; i++) m->dLt[i] = dL_at_z(m, m->zt[i]);
  for (int i = 0; i < Tc; i++) if (m->dLt[i] != m->dLt[i] || (i > 0 && m->dLt[i] < m->dLt[i - 1])) m->dl_unsorted = 1;
  double l0 = log10(p->mass[0]), l1 = log10(p->mass[1]);
  for (int i = 0; i < Tm; i++) { m->mg[i] = pow(10., lin_at(l0, l1, Tm, i)); tmp[i] = secondary_notnorm(p, m->mg[i], p->mass[1]); }
  cumtrapz(tmp, m->mg, m->cdf, Tm);
  double acc = 0.;
  for (int i = 0; i < Tm; i++) tmp[i] = primary_notnorm(p, m->mg[i]);
  for (int k = 0; k < Tm - 1; k++) acc += (m->mg[k + 1] - m->mg[k]) * (tmp[k + 1] + tmp[k]);
  m->norm_p_m1 = 0.5 * acc;                                        /* jnp.trapezoid */
  m->fR = Vc_from_dCt(m, dCt_at_z(m, p->compl_z1)) - Vc_from_dCt(m, dCt_at_z(m, p->compl_z0));
  free(tmp);
  return 0;
}

int orc_tables(const chm_params* p, double* zt, double* It, double* dLt, double* mg, double* cdf, double* scalars) {
  orc_model m;
  if (model_init(&m, p)) return -1;
  if (zt) memcpy(zt, m.zt, sizeof(double) * m.Tc);
  if (It) memcpy(It, m.It, sizeof(double) * m.Tc);
  if (dLt) memcpy(dLt, m.dLt, sizeof(double) * m.Tc);
  if (mg) memcpy(mg, m.mg, sizeof(double) * m.Tm);
  if (cdf) memcpy(cdf, m.cdf, sizeof(double) * m.Tm);
  if (scalars) { scalars[0] = m.norm_p_m1; scalars[1] = m.fR; }
  model_free(&m);
  return 0;
}

/* ------------------------------------------------------------------------------------------------------
 * p_gw3dmarg + the pixelated numerator, one event          likelihood.py:160-205, 266-281
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
  int E, S, P, Z, num_bins, binning, has_cut, bw_method;
  double cut_grid, pe_neff, bw_scalar;
  const double *dL, *m1det, *m2det, *pe_prior;        /* (E,S) */
  const int64_t *pe_pix;                              /* (E,S)  pixels_pe_opt_nside */
  const int64_t *pixels;                              /* (E,P)  pixels_opt_nsides   */
  const double *z_grids;                              /* (E,Z) */
  const double *p_cat;                                /* (E,P,Z), -100 padded */
  const double *gw_pdf;                               /* (E,P) */
} orc_like;

static double two_pass_std(const double* x, int n) {   /* jnp.std, ddof = 0 */
  double mean = 0., var = 0.;
  for (int i = 0; i < n; i++) mean += x[i];
  mean /= (double)n;
  for (int i = 0; i < n; i++) { double d = x[i] - mean; var += d * d; }
  return sqrt(var / (double)n);
}

/* kde1d (math.py:52-81); the marginalized path never passes `kernel` and so always gets Epanechnikov */
static void kde1d(const double* data, const double* wgt_in, int N, const double* grid, int G, int bw_method, double bw_scalar,
                  int gauss, double* Wn, double* out) {
  double tot = 0., s2 = 0.;
  for (int j = 0; j < N; j++) tot += wgt_in[j];
  for (int j = 0; j < N; j++) { Wn[j] = wgt_in[j] / tot; s2 += Wn[j] * Wn[j]; }
  double neff = 1.0 / s2;
  double sd = two_pass_std(data, N);
  double bw;
  if (bw_method == 0) bw = pow(neff, -1. / 5.) * sd;
  else if (bw_method == 1) bw = pow(neff * 3. / 4.0, -1. / 5.) * sd;
  else bw = bw_scalar * sd;
  for (int g = 0; g < G; g++) {
    double acc = 0.;
    for (int j = 0; j < N; j++) {
      double u = (grid[g] - data[j]) / bw;
      double kv = gauss ? exp(-0.5 * (u * u)) / sqrt(2. * ORC_PI)                  /* _gaussian_kernel, math.py:87-89 */
                        : (fabs(u) <= 1. ? 3. / 4. * (1. - u * u) : 0.);           /* _epan_kernel,     math.py:83-85 */
      acc += Wn[j] * kv;
    }
    out[g] = acc / bw;
  }
}

static double event_numlike(const orc_model* m, const orc_like* L, int ev, double* scratch) {
  const int S = L->S, P = L->P, Z = L->Z, B = L->num_bins;
  const int G = L->has_cut ? Z / 2 : Z;
  const int N = L->binning ? B : S;
  double* z = scratch;           double* w = z + S;
  double* zm = w + S;            double* wm = zm + S;
  double* cen = wm + S;          double* cnt = cen + N;      double* Wn = cnt + N;
  double* eff = Wn + N;          double* dens = eff + G;
  double* pz = dens + G;         double* jac = pz + Z;       double* pgw = jac + Z;
  const double* zg = L->z_grids + (size_t)ev * Z;
  const size_t eo = (size_t)ev * S;
  /* get_theta_src_and_weights, pop_wrapper.py:67-80 */
  double sw = 0., sw2 = 0.;
  for (int s = 0; s < S; s++) {
    double zz = interp1s(L->dL[eo + s], m->dLt, m->zt, m->Tc, 0, 0., 0., m->dl_unsorted);           /* z_from_dGW, cosmo.py:260-264 */
    double m1 = L->m1det[eo + s] / (1. + zz), m2 = L->m2det[eo + s] / (1. + zz);
    z[s] = zz;
    w[s] = p_m1m2(m, m1, m2) / L->pe_prior[eo + s];
    sw += w[s]; sw2 += w[s] * w[s];
  }
  double norm = sw / (double)S;                                                     /* likelihood.py:169 */
  double n_eff = (sw * sw) / sw2;                                                   /* likelihood.py:170 */
  if (!(n_eff >= L->pe_neff)) return 0.;                                            /* lax.cond -> zeros, :199-203 */
  double zmin = z[0], zmax = z[0];
  for (int s = 1; s < S; s++) { if (z[s] < zmin || z[s] != z[s]) zmin = z[s]; if (z[s] > zmax || z[s] != z[s]) zmax = z[s]; }
  if (L->has_cut) {
    double sd = two_pass_std(z, S);
    double lb = zmin - L->cut_grid * sd; lb = lb > 1e-8 ? lb : (lb != lb ? lb : 1e-8);  /* jnp.maximum(., 1e-8), :186 */
    double ub = zmax + L->cut_grid * sd;                                                /* :187 */
    for (int i = 0; i < G; i++) eff[i] = lin_at(lb, ub, G, i);                          /* :188 */
  } else {
    for (int i = 0; i < G; i++) eff[i] = zg[i];                                         /* :190 */
  }
  /* per-z factors: p_cbc (pop_wrapper.py:82-90), jacobian (likelihood.py:272) */
  for (int k = 0; k < Z; k++) {
    double dCt = dCt_at_z(m, zg[k]);
    pz[k] = merger_rate(m->p, zg[k]) / (1. + zg[k]);                 /* p_rate */
    jac[k] = ddLdz(m, dCt, zg[k]) * ((1. + zg[k]) * (1. + zg[k]));
    pgw[k] = dVcdz(m, dCt, zg[k]);                                   /* p_bkg, reused below */
  }
  double Li = 0.;
  for (int i = 0; i < P; i++) {
    const int64_t pid = L->pixels[(size_t)ev * P + i];
    const double* pc = L->p_cat + ((size_t)ev * P + i) * Z;
    /* mask = pe_pix == pixels[i]; z_m = where(mask, z, min z); w_m = where(mask, w, 0)            :179-181 */
    double mx = zmin;
    for (int s = 0; s < S; s++) {
      int in = L->pe_pix[eo + s] == pid;
      zm[s] = in ? z[s] : zmin; wm[s] = in ? w[s] : 0.0;
      if (zm[s] > mx || zm[s] != zm[s]) mx = zm[s];
    }
    const double* data = zm; const double* wgt = wm;
    if (L->binning) {                                                /* binning1d, math.py:32-46 */
      double mn = zmin;
      for (int j = 0; j < B; j++) { cen[j] = (lin_at(mn, mx, B + 1, j) + lin_at(mn, mx, B + 1, j + 1)) / 2.; cnt[j] = 0.; }
      for (int s = 0; s < S; s++) {
        double f = floor((zm[s] - mn) / (mx - mn) * (double)B);
        f = f < 0. ? 0. : (f > (double)(B - 1) ? (double)(B - 1) : f);
        int idx = (f != f) ? 0 : (int)f;
        cnt[idx] += wm[s];
      }
      data = cen; wgt = cnt;
    }
    kde1d(data, wgt, N, eff, G, L->bw_method, L->bw_scalar, 0, Wn, dens);          /* :192 */
    const double gwp = L->gw_pdf[(size_t)ev * P + i];
    /* integrand and trapezoid                                                      :193-194, 270-278 */
    double acc = 0., yprev = 0.;
    for (int k = 0; k < Z; k++) {
      double pg = interp1(zg[k], eff, dens, G, 1, 0., 0.) * norm * gwp;
      double P_compl = (zg[k] > m->p->compl_z0 && zg[k] < m->p->compl_z1) ? 1. : 0.;        /* completeness.py:43-47 */
      double p_gal = (pc[k] != -100.) ? m->fR * pc[k] + (1. - P_compl) * pgw[k] : -100.;    /* catalog.py:202-203 */
      double p_z = (p_gal != -100.) ? p_gal * pz[k] : -100.;                                /* pop_wrapper.py:87 */
      double y = (p_z != -100.) ? pg * p_z / jac[k] : 0.;                                   /* likelihood.py:274-277 */
      if (k > 0) acc += (zg[k] - zg[k - 1]) * (y + yprev);
      yprev = y;
    }
    Li += 0.5 * acc;
  }
  return Li;
}

/* like_evs[ev] = L_i for every event (likelihood.py:266-281); returns 0, or -1 on allocation failure */
int orc_numlike_marg(const chm_params* p, int E, int S, int P, int Z, const double* dL, const double* m1det, const double* m2det,
                     const double* pe_prior, const int64_t* pe_pix, const int64_t* pixels, const double* z_grids,
                     const double* p_cat, const double* gw_pdf, double cut_grid /* NaN = None */, int binning, int num_bins,
                     double pe_neff, int bw_method, double bw_scalar, int nthreads, double* like_evs) {
  orc_model m;
  if (model_init(&m, p)) return -1;
  orc_like L;
  L.E = E; L.S = S; L.P = P; L.Z = Z; L.num_bins = num_bins; L.binning = binning; L.has_cut = !(cut_grid != cut_grid);
  L.bw_method = bw_method; L.cut_grid = cut_grid; L.pe_neff = pe_neff; L.bw_scalar = bw_scalar;
  L.dL = dL; L.m1det = m1det; L.m2det = m2det; L.pe_prior = pe_prior; L.pe_pix = pe_pix; L.pixels = pixels;
  L.z_grids = z_grids; L.p_cat = p_cat; L.gw_pdf = gw_pdf;
  const int N = binning ? num_bins : S;
  const size_t nscr = (size_t)4 * S + 3 * N + 2 * Z + 3 * Z;
  int fail = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    double* scratch = malloc(sizeof(double) * nscr);
    if (!scratch) {
#pragma omp atomic write
      fail = 1;
    }
#pragma omp for schedule(dynamic, 1)
    for (int ev = 0; ev < E; ev++) if (scratch) like_evs[ev] = event_numlike(&m, &L, ev, scratch);
    free(scratch);
  }
  model_free(&m);
  return fail ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------------------
 * p_gw1d (likelihood.py:105-144) and the 1-D / approximate numerators (likelihood.py:150-154, 266-292), one event.
 * P = 0: no catalogue (p_z = p_bkg * rate / (1+z), likelihood.py:283-292); P > 0: p_gw3dapprox = p_gw1d x gw_loc2d_pdf.
 * ---------------------------------------------------------------------------------------------------- */
static double event_numlike_1d(const orc_model* m, const orc_like* L, int gauss, const int32_t* neff_pixels, int ev, double* scratch) {
  const int S = L->S, P = L->P, Z = L->Z, B = L->num_bins;
  const int G = L->has_cut ? Z / 2 : Z;
  const int N = L->binning ? B : S;
  double* z = scratch;           double* w = z + S;
  double* cen = w + S;           double* cnt = cen + N;      double* Wn = cnt + N;
  double* eff = Wn + N;          double* dens = eff + G;
  double* pz = dens + G;         double* jac = pz + Z;       double* bkg = jac + Z;     double* pg = bkg + Z;
  const double* zg = L->z_grids + (size_t)ev * Z;
  const size_t eo = (size_t)ev * S;
  double sw = 0., sw2 = 0.;
  for (int s = 0; s < S; s++) {                                                     /* pop_wrapper.py:67-80 */
    double zz = interp1s(L->dL[eo + s], m->dLt, m->zt, m->Tc, 0, 0., 0., m->dl_unsorted);
    z[s] = zz;
    w[s] = p_m1m2(m, L->m1det[eo + s] / (1. + zz), L->m2det[eo + s] / (1. + zz)) / L->pe_prior[eo + s];
    sw += w[s]; sw2 += w[s] * w[s];
  }
  double norm = sw / (double)S, n_eff = (sw * sw) / sw2;                             /* likelihood.py:111-112 */
  for (int k = 0; k < Z; k++) pg[k] = 0.;
  if (n_eff >= L->pe_neff) {                                                        /* lax.cond, :133-139 */
    double zmin = z[0], zmax = z[0];
    for (int s = 1; s < S; s++) { if (z[s] < zmin || z[s] != z[s]) zmin = z[s]; if (z[s] > zmax || z[s] != z[s]) zmax = z[s]; }
    if (L->has_cut) {
      double sd = two_pass_std(z, S);
      double lb = zmin - L->cut_grid * sd; lb = lb > 0. ? lb : 1.e-8;               /* jnp.where(. > 0, ., 1e-8), :119 */
      double ub = zmax + L->cut_grid * sd;
      for (int i = 0; i < G; i++) eff[i] = lin_at(lb, ub, G, i);                    /* :121 */
    } else {
      for (int i = 0; i < G; i++) eff[i] = zg[i];
    }
    const double* data = z; const double* wgt = w;
    if (L->binning) {                                                               /* binning1d, math.py:32-46 */
      for (int j = 0; j < B; j++) { cen[j] = (lin_at(zmin, zmax, B + 1, j) + lin_at(zmin, zmax, B + 1, j + 1)) / 2.; cnt[j] = 0.; }
      for (int s = 0; s < S; s++) {
        double f = floor((z[s] - zmin) / (zmax - zmin) * (double)B);
        f = f < 0. ? 0. : (f > (double)(B - 1) ? (double)(B - 1) : f);
        cnt[(f != f) ? 0 : (int)f] += w[s];
      }
      data = cen; wgt = cnt;
    }
    kde1d(data, wgt, N, eff, G, L->bw_method, L->bw_scalar, gauss, Wn, dens);
    for (int i = 0; i < G; i++) dens[i] *= norm;                                    /* kde * norm, :136 */
    for (int k = 0; k < Z; k++) pg[k] = interp1(zg[k], eff, dens, G, 1, 0., 0.);    /* :137 */
  }
  for (int k = 0; k < Z; k++) {                                                     /* p_cbc / jacobian on the event grid */
    double dCt = dCt_at_z(m, zg[k]);
    pz[k] = merger_rate(m->p, zg[k]) / (1. + zg[k]);
    jac[k] = ddLdz(m, dCt, zg[k]) * ((1. + zg[k]) * (1. + zg[k]));
    bkg[k] = dVcdz(m, dCt, zg[k]);
  }
  if (P == 0) {                                                                     /* likelihood.py:283-292 */
    double acc = 0., yprev = 0.;
    for (int k = 0; k < Z; k++) {
      double y = pg[k] * (bkg[k] * pz[k]) / jac[k];
      if (k > 0) acc += (zg[k] - zg[k - 1]) * (y + yprev);
      yprev = y;
    }
    return 0.5 * acc;
  }
  double Li = 0.;
  for (int i = 0; i < P; i++) {                                                     /* likelihood.py:150-154, 266-281 */
    const double* pc = L->p_cat + ((size_t)ev * P + i) * Z;
    const double gwp = L->gw_pdf[(size_t)ev * P + i];
    double acc = 0., yprev = 0.;
    for (int k = 0; k < Z; k++) {
      double pg3 = pg[k] * gwp;
      double P_compl = (zg[k] > m->p->compl_z0 && zg[k] < m->p->compl_z1) ? 1. : 0.;
      double p_gal = (pc[k] != -100.) ? m->fR * pc[k] + (1. - P_compl) * bkg[k] : -100.;
      double p_z = (p_gal != -100.) ? p_gal * pz[k] : -100.;
      double y = (p_z != -100.) ? pg3 * p_z / jac[k] : 0.;
      if (k > 0) acc += (zg[k] - zg[k - 1]) * (y + yprev);
      yprev = y;
    }
    Li += 0.5 * acc;
  }
  (void)neff_pixels;
  return Li;
}

/* like_evs[ev] = L_i in the 1-D (P = 0, p_cat / gw_pdf NULL) or approximate (P > 0) mode; kernel: 0 = 'epan', 1 = 'gauss' */
int orc_numlike_1d(const chm_params* p, int E, int S, int P, int Z, const double* dL, const double* m1det, const double* m2det,
                   const double* pe_prior, const double* z_grids, const double* p_cat, const double* gw_pdf, double cut_grid,
                   int binning, int num_bins, double pe_neff, int bw_method, double bw_scalar, int kernel, int nthreads,
                   double* like_evs) {
  orc_model m;
  if (model_init(&m, p)) return -1;
  orc_like L;
  memset(&L, 0, sizeof(L));
  L.E = E; L.S = S; L.P = P; L.Z = Z; L.num_bins = num_bins; L.binning = binning; L.has_cut = !(cut_grid != cut_grid);
  L.bw_method = bw_method; L.cut_grid = cut_grid; L.pe_neff = pe_neff; L.bw_scalar = bw_scalar;
  L.dL = dL; L.m1det = m1det; L.m2det = m2det; L.pe_prior = pe_prior; L.z_grids = z_grids; L.p_cat = p_cat; L.gw_pdf = gw_pdf;
  const int N = binning ? num_bins : S;
  const size_t nscr = (size_t)2 * S + 3 * N + 2 * Z + 4 * Z;
  int fail = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    double* scratch = malloc(sizeof(double) * nscr);
    if (!scratch) {
#pragma omp atomic write
      fail = 1;
    }
#pragma omp for schedule(dynamic, 1)
    for (int ev = 0; ev < E; ev++) if (scratch) like_evs[ev] = event_numlike_1d(&m, &L, kernel, NULL, ev, scratch);
    free(scratch);
  }
  model_free(&m);
  return fail ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------------------
 * p_gw3dfull (likelihood.py:211-260) with numba_gkde_nd / numba_gaussian_kernel (utils/math.py:154-229, CPU branch,
 * in_log=False) and the pixelated numerator (likelihood.py:266-281), one event.  The pair sum is the reference's
 * plain double loop: one exp per (grid point, sample).
 * ---------------------------------------------------------------------------------------------------- */
/* np.linalg.inv of a 3x3 matrix: LU with partial pivoting (LAPACK getrf), then the columns of the identity are solved for (getri) */
static void inv3_lu(const double A[3][3], double inv[3][3]) {
  double lu[3][3];
  int piv[3] = {0, 1, 2};
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) lu[r][c] = A[r][c];
  for (int c = 0; c < 3; c++) {
    int best = c;
    for (int r = c + 1; r < 3; r++) if (fabs(lu[r][c]) > fabs(lu[best][c])) best = r;
    if (best != c) {
      for (int k = 0; k < 3; k++) { double t = lu[c][k]; lu[c][k] = lu[best][k]; lu[best][k] = t; }
      int t = piv[c]; piv[c] = piv[best]; piv[best] = t;
    }
    for (int r = c + 1; r < 3; r++) {
      lu[r][c] = lu[r][c] / lu[c][c];
      for (int k = c + 1; k < 3; k++) lu[r][k] -= lu[r][c] * lu[c][k];
    }
  }
  for (int col = 0; col < 3; col++) {
    double b[3], y[3], x[3];
    for (int r = 0; r < 3; r++) b[r] = piv[r] == col ? 1. : 0.;
    for (int r = 0; r < 3; r++) { double s = b[r]; for (int k = 0; k < r; k++) s -= lu[r][k] * y[k]; y[r] = s; }
    for (int r = 2; r >= 0; r--) { double s = y[r]; for (int k = r + 1; k < 3; k++) s -= lu[r][k] * x[k]; x[r] = s / lu[r][r]; }
    for (int r = 0; r < 3; r++) inv[r][col] = x[r];
  }
}

typedef struct {
  const double *ra, *dec;            /* (E,S) */
  const double *ra_pix, *dec_pix;    /* (E,P) */
  const int32_t* neff_pixels;        /* (E,)  */
} orc_full;

static double event_numlike_full(const orc_model* m, const orc_like* L, const orc_full* F, int ev, double* scratch) {
  const int S = L->S, P = L->P, Z = L->Z;
  double* z = scratch;           double* w = z + S;          double* Wn = w + S;
  double* dw = Wn + S;           /* (S,3) whitened dataset */
  double* pz = dw + 3 * (size_t)S; double* jac = pz + Z;     double* bkg = jac + Z;     double* pg = bkg + Z;
  double* msk = pg + Z;
  const double* zg = L->z_grids + (size_t)ev * Z;
  const size_t eo = (size_t)ev * S;
  double sw = 0., sw2 = 0.;
  for (int s = 0; s < S; s++) {                                                     /* pop_wrapper.py:67-80 */
    double zz = interp1s(L->dL[eo + s], m->dLt, m->zt, m->Tc, 0, 0., 0., m->dl_unsorted);
    z[s] = zz;
    w[s] = p_m1m2(m, L->m1det[eo + s] / (1. + zz), L->m2det[eo + s] / (1. + zz)) / L->pe_prior[eo + s];
    sw += w[s]; sw2 += w[s] * w[s];
  }
  const double norm = sw / (double)S, n_eff = (sw * sw) / sw2;                       /* likelihood.py:214-215 */
  for (int k = 0; k < Z; k++) {                                                     /* p_cbc / jacobian on the event grid */
    double dCt = dCt_at_z(m, zg[k]);
    pz[k] = merger_rate(m->p, zg[k]) / (1. + zg[k]);
    jac[k] = ddLdz(m, dCt, zg[k]) * ((1. + zg[k]) * (1. + zg[k]));
    bkg[k] = dVcdz(m, dCt, zg[k]);
  }
  const int skip = n_eff < L->pe_neff;                                              /* `if n_effs[ev] < pe_neff: continue`  :234 */
  int npix = F->neff_pixels[ev];
  if (npix > P) npix = P;
  double Lw[3][3] = {{0}}, log_norm = 0.;
  int nmask = 0;
  if (!skip) {
    /* z mask                                                                        likelihood.py:222-225 */
    double zmin = z[0], zmax = z[0];
    for (int s = 1; s < S; s++) { if (z[s] < zmin || z[s] != z[s]) zmin = z[s]; if (z[s] > zmax || z[s] != z[s]) zmax = z[s]; }
    const double sd = two_pass_std(z, S);
    for (int k = 0; k < Z; k++) { msk[k] = (zg[k] <= zmax + L->cut_grid * sd) && (zg[k] >= zmin - L->cut_grid * sd) ? 1. : 0.; nmask += msk[k] != 0.; }
    /* numba_gkde_nd set-up                                                          math.py:171-196 */
    double s2 = 0.;
    for (int s = 0; s < S; s++) { Wn[s] = w[s] / sw; s2 += Wn[s] * Wn[s]; }
    const double neff = 1.0 / s2;
    double factor;
    if (L->bw_method == 0) factor = pow(neff, -1. / 7.);
    else if (L->bw_method == 1) factor = pow(neff * 5. / 4.0, -1. / 7.);
    else factor = L->bw_scalar;
    const double* X[3] = { z, F->ra + eo, F->dec + eo };
    double mean[3], cov[3][3], icov[3][3];
    for (int c = 0; c < 3; c++) { double a = 0.; for (int s = 0; s < S; s++) a += Wn[s] * X[c][s]; mean[c] = a; }
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
      double acc = 0.;
      for (int s = 0; s < S; s++) acc += ((X[a][s] - mean[a]) * Wn[s]) * (X[b][s] - mean[b]);
      cov[a][b] = acc / (1. - s2);
    }
    inv3_lu(cov, icov);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) icov[a][b] = icov[a][b] / (factor * factor);
    for (int r = 0; r < 3; r++) for (int c = 0; c <= r; c++) {                      /* np.linalg.cholesky (lower) */
      double s = icov[r][c];
      for (int k = 0; k < c; k++) s -= Lw[r][k] * Lw[c][k];
      Lw[r][c] = r == c ? sqrt(s) : s / Lw[c][c];
    }
    for (int s = 0; s < S; s++) for (int c = 0; c < 3; c++) {                       /* dataset.T @ whitening */
      double a = 0.;
      for (int k = 0; k < 3; k++) a += X[k][s] * Lw[k][c];
      dw[3 * (size_t)s + c] = a;
    }
    log_norm = (log(Lw[0][0]) + log(Lw[1][1]) + log(Lw[2][2])) - 0.5 * 3. * log(2. * ORC_PI);     /* math.py:215 */
  }
  double Li = 0.;
  for (int i = 0; i < P; i++) {
    const double* pc = L->p_cat + ((size_t)ev * P + i) * Z;
    for (int k = 0; k < Z; k++) pg[k] = 0.;                                         /* np.zeros result, kde_vals           :230,250 */
    if (!skip && i < npix && nmask > 0) {
      const double q[3] = { 0., F->ra_pix[(size_t)ev * P + i], F->dec_pix[(size_t)ev * P + i] };
      for (int k = 0; k < Z; k++) {
        if (msk[k] == 0.) continue;
        double pw[3];
        for (int c = 0; c < 3; c++) pw[c] = (zg[k] * Lw[0][c] + q[1] * Lw[1][c]) + q[2] * Lw[2][c];       /* points.T @ whitening */
        double acc = 0.;
        for (int s = 0; s < S; s++) {                                               /* numba_gaussian_kernel, math.py:217-227 */
          double d0 = dw[3 * (size_t)s] - pw[0], d1 = dw[3 * (size_t)s + 1] - pw[1], d2 = dw[3 * (size_t)s + 2] - pw[2];
          acc += Wn[s] * exp(log_norm - 0.5 * ((d0 * d0 + d1 * d1) + d2 * d2));
        }
        pg[k] = acc * norm;                                                         /* kde_vals ... * norm                 :252-253 */
      }
    }
    double acc = 0., yprev = 0.;
    for (int k = 0; k < Z; k++) {                                                   /* likelihood.py:270-278 */
      double P_compl = (zg[k] > m->p->compl_z0 && zg[k] < m->p->compl_z1) ? 1. : 0.;
      double p_gal = (pc[k] != -100.) ? m->fR * pc[k] + (1. - P_compl) * bkg[k] : -100.;
      double p_z = (p_gal != -100.) ? p_gal * pz[k] : -100.;
      double y = (p_z != -100.) ? pg[k] * p_z / jac[k] : 0.;
      if (k > 0) acc += (zg[k] - zg[k - 1]) * (y + yprev);
      yprev = y;
    }
    Li += 0.5 * acc;
  }
  return Li;
}

/* like_evs[ev] = L_i for kind_p_gw3d = 'full'; cut_grid must be set (the reference multiplies it into the mask) */
int orc_numlike_full(const chm_params* p, int E, int S, int P, int Z, const double* dL, const double* m1det, const double* m2det,
                     const double* pe_prior, const double* ra, const double* dec, const double* ra_pix, const double* dec_pix,
                     const int32_t* neff_pixels, const double* z_grids, const double* p_cat, double cut_grid, double pe_neff,
                     int bw_method, double bw_scalar, int nthreads, double* like_evs) {
  orc_model m;
  if (model_init(&m, p)) return -1;
  orc_like L;
  memset(&L, 0, sizeof(L));
  L.E = E; L.S = S; L.P = P; L.Z = Z; L.has_cut = 1; L.bw_method = bw_method; L.cut_grid = cut_grid; L.pe_neff = pe_neff; L.bw_scalar = bw_scalar;
  L.dL = dL; L.m1det = m1det; L.m2det = m2det; L.pe_prior = pe_prior; L.z_grids = z_grids; L.p_cat = p_cat;
  orc_full F = { ra, dec, ra_pix, dec_pix, neff_pixels };
  const size_t nscr = (size_t)6 * S + 5 * (size_t)Z;
  int fail = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    double* scratch = malloc(sizeof(double) * nscr);
    if (!scratch) {
#pragma omp atomic write
      fail = 1;
    }
#pragma omp for schedule(dynamic, 1)
    for (int ev = 0; ev < E; ev++) if (scratch) like_evs[ev] = event_numlike_full(&m, &L, &F, ev, scratch);
    free(scratch);
  }
  model_free(&m);
  return fail ? -1 : 0;
}

/* selection_function.N_exp (selection_function.py:34-48) with pop_rate_det (pop_wrapper.py:102-111).
 * out[0] = N_exp, out[1] = xi, out[2] = n_eff */
int orc_nexp(const chm_params* p, long long I, const double* dL, const double* m1det, const double* m2det, const double* p_draw,
             double N_inj, double N_eff /* NaN = None */, int nthreads, double* out) {
  orc_model m;
  if (model_init(&m, p)) return -1;
  double s1 = 0., s2 = 0.;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for reduction(+ : s1, s2) schedule(static)
  for (long long i = 0; i < I; i++) {
    double z = interp1s(dL[i], m.dLt, m.zt, m.Tc, 0, 0., 0., m.dl_unsorted);
    double m1 = m1det[i] / (1. + z), m2 = m2det[i] / (1. + z);
    double dCt = dL2dCt(&m, dL[i], z);                               /* original distances */
    double p_z = dVcdz(&m, dCt, z);                                  /* gal_cat.p_bkg       pop_wrapper.py:106 */
    p_z = p_z * (merger_rate(p, z) / (1. + z));                      /*                     :107 */
    double dN = p->R0 * p_m1m2(&m, m1, m2) * p_z;                    /*                     :108 */
    double jac = fabs(ddLdz(&m, dCt, z)) * ((1. + z) * (1. + z));    /*                     :109 */
    dN = dN / jac;
    dN = dN / p_draw[i];                                             /* selection_function.py:38 */
    if (dN == dN) s1 += dN;                                          /* nansum              :39 */
    s2 += dN * dN;                                                   /* plain sum (Q10)     :44 */
  }
  double xi = s1 / N_inj;
  double Nexp = p->Tobs * xi;
  double neff = NAN;
  if (N_eff == N_eff) {
    double variance2 = s2 / (N_inj * N_inj) - (xi * xi) / N_inj;
    neff = (xi * xi) / variance2;
    if (neff < N_eff) Nexp = 0.0;
  }
  out[0] = Nexp; out[1] = xi; out[2] = neff;
  model_free(&m);
  return 0;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
