// chm_models.h -- device-side population models and jnp-semantics helpers (gfx950, fp64).
//
// Every function restates one function of the reference (file:line cited, paths relative to CHIMERA/),
// with the reference's operation order; the translation unit is compiled with -ffp-contract=off so that
// a*b+c is never fused behind the author's back (fused forms are written explicitly as fma() where used).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CHM_PI 3.141592653589793238462643383279502884

struct DevParams {
  int cosmo_model, mass_model, rate_model, Tc, Tm, scale_free, has_catalog;
  int rate_special;                      // 1: a rate parameter the model reads is infinite -- merger_rate takes the reference's own operations (pow_c99), see merger_rate_special
  double z_max;
  double H0, Om0, Ok0, Or0, w0, wa, Xi0, n_mg;
  double Ode0, dH;                       // cosmo.py:79-84
  double m[8];                           // chm_params.mass
  double r[4];                           // chm_params.rate
  double R0, Tobs, zc0, zc1;
  double mg_first, mg_last;              // 10**log10(m_low), 10**log10(m_high) formed by the HOST's libm (see k_tables, mass grid)
  // constants derived once per draw by k_tables (same expressions the reference re-evaluates per element)
  double plp_plnorm;                     // tpl_cdf(-alpha, m_low, m_high)           mass.py:301
  double tg_norm;                        // truncated_gaussian norm                   mass.py:272-274
  double tg_hi;                          // mu_g + 5 sigma_g                          mass.py:302
  double g_c0;                           // -0.5 log(2 pi) - log(sigma)               mass.py:268
  double bpl_mbreak, bpl_pl1, bpl_pl2;   // mass.py:291-293
  double md_norm;                        // 1 + (1+zp)^(-gamma-kappa)                 rate.py:114
  double tpl_rate_norm;                  // rate.py:105
  double l1pzp;                          // log(1 + zp)
  double lmg0, inv_dlmg;                 // log(m_low), (Tm-1)/(log(m_high) - log(m_low)): position of log(m) on the mass grid
  double inv_plnorm, inv_tg_norm, inv_2s2;   // reciprocals of plp_plnorm, tg_norm, 2 sigma_g^2
  double norm_p_m1, inv_norm_p_m1;       // mass.py:51
  double cdf_last;                       // cdf_m2[Tm - 1] (k_tables): the value jnp.interp clamps to above the mass grid
  double fR;                             // completeness.py:54-58
  double fR_given;                       // != 0: fR was supplied by the caller (plug-in completeness), k_tables keeps it
  double z_bad;                          // first z at which the comoving-distance table is non-finite (+inf: nowhere), see grid_is_poisoned
  double dl_sorted;                      // 1 if the dL table of z_from_dGW is non-decreasing (k_tables), else 0: see z_from_dGW_x2
};

struct TablePtrs {                       // per-draw tables (global memory)
  const double* zt;  const double* It;  const double* dLt;   // (Tc)
  const double* mg;  const double* cdf;                      // (Tm)
};

#define DEVFN __device__ __forceinline__
#include "chm_math.h"

// x^y for x > 0 as exp(y log x).  The reference's jnp.power is correctly rounded to ~1 ulp; this form is within
// (|y ln x| + 2) ulp of it -- <= 2e-15 relative for every use below (|y ln x| <= 16) -- and costs a third of ocml's pow().
// Callers that already hold log(x) pass it in and share it between several powers.
DEVFN double pow_l(double lx, double y) { return chm_exp(y * lx); }
DEVFN double pow_el(double x, double y) { return chm_exp(y * chm_log(x)); }

// ------------------------------------------------------------------------------------------------------
// jax.numpy semantics
// ------------------------------------------------------------------------------------------------------

// jnp.linspace(start, stop, num)[i]  (endpoint=True): start*(1-i/div) + stop*(i/div); last point == stop.
DEVFN double jnp_linspace_at(double start, double stop, int num, int i) {
  int div = num - 1;
  if (i >= div) return stop;
  double step = (double)i / (double)div;
  return start * (1. - step) + stop * step;
}

// searchsorted(xp, x, side='right'): number of elements <= x (xp ascending).
template <class Acc>
DEVFN int searchsorted_right(Acc xp, int n, double x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (xp[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// jnp.searchsorted(xp, x, side='right') exactly as jax's default method does it ('scan': jax/_src/numpy/lax_numpy.py,
// _searchsorted_via_scan): ceil(log2(n+1)) steps of  mid = (low+high)/2; go_left = x < xp[mid] (NaN sorts last); high = mid or
// low = mid; result high.  Same answer as searchsorted_right on a sorted table; on a NON-monotonic one (the dL table of an
// unphysical draw) the probe sequence decides, and this is the reference's.
template <class Acc>
DEVFN int searchsorted_right_scan(Acc xp, int n, double x) {
  int lo = 0, hi = n, levels = 0;
  while ((1LL << levels) < (long long)n + 1) levels++;
  for (int l = 0; l < levels; l++) {
    int mid = (lo + hi) >> 1;
    double v = xp[mid];
    bool go_left = (x < v) || ((v != v) && (x == x));
    if (go_left) hi = mid; else lo = mid;
  }
  return hi;
}

// jnp.interp(x, xp, fp, left, right); has_lr == false -> clamp to fp[0] / fp[n-1].
template <class AccX, class AccF>
DEVFN double jnp_interp(double x, AccX xp, AccF fp, int n, bool has_lr, double left, double right, bool scan = false) {
  int i = scan ? searchsorted_right_scan(xp, n, x) : searchsorted_right(xp, n, x);
  i = i < 1 ? 1 : (i > n - 1 ? n - 1 : i);
  double x0 = xp[i - 1], x1 = xp[i];
  double f0 = fp[i - 1], f1 = fp[i];
  double df = f1 - f0, dx = x1 - x0, delta = x - x0;
  const double epsilon = 4.930380657631324e-32;          // np.spacing(np.finfo(float64).eps)
  bool dx0 = fabs(dx) <= epsilon;
  double f = dx0 ? f0 : f0 + (delta / dx) * df;
  double xfirst = xp[0], xlast = xp[n - 1];
  if (x < xfirst) f = has_lr ? left : (double)fp[0];
  if (x > xlast) f = has_lr ? right : (double)fp[n - 1];
  return f;
}

// Two independent jnp.interp evaluations on the same table with the binary searches run in lock step (fixed trip count,
// no data-dependent branches), so that the LDS latencies of the two searches overlap.  Same result as jnp_interp().
template <class AccX, class AccF>
DEVFN void jnp_interp_x2(double xa, double xb, AccX xp, AccF fp, int n, double& fa, double& fb) {
  // number of elements <= x (searchsorted side='right') by halving [base, base + len): the probe base + half - 1 is always in
  // bounds and the sequence of lengths is the same for every lane, so a step is a read, a compare and a conditional add
  int pa = 0, pb = 0;
  for (int len = n; len > 1;) {
    const int half = len >> 1;
    double va = xp[pa + half - 1], vb = xp[pb + half - 1];
    pa += (va <= xa) ? half : 0;
    pb += (vb <= xb) ? half : 0;
    len -= half;
  }
  pa += (xp[pa] <= xa) ? 1 : 0;
  pb += (xp[pb] <= xb) ? 1 : 0;
  int ia = pa < 1 ? 1 : (pa > n - 1 ? n - 1 : pa), ib = pb < 1 ? 1 : (pb > n - 1 ? n - 1 : pb);
  double x0a = xp[ia - 1], x1a = xp[ia], f0a = fp[ia - 1], f1a = fp[ia];
  double x0b = xp[ib - 1], x1b = xp[ib], f0b = fp[ib - 1], f1b = fp[ib];
  const double epsilon = 4.930380657631324e-32;
  double dxa = x1a - x0a, dxb = x1b - x0b;
  fa = (fabs(dxa) <= epsilon) ? f0a : f0a + ((xa - x0a) / dxa) * (f1a - f0a);
  fb = (fabs(dxb) <= epsilon) ? f0b : f0b + ((xb - x0b) / dxb) * (f1b - f0b);
  double xfirst = xp[0], xlast = xp[n - 1];
  if (xa < xfirst) fa = fp[0];
  if (xa > xlast) fa = fp[n - 1];
  if (xb < xfirst) fb = fp[0];
  if (xb > xlast) fb = fp[n - 1];
}

// jnp_interp_x2 restricted to the table entries [base, base + len): for callers that know  #entries <= x  lies in
// [base, base + len] for both arguments (all entries before `base` are <= x, all from base + len on are > x).
template <class AccX, class AccF>
DEVFN void jnp_interp_x2_range(double xa, double xb, AccX xp, AccF fp, int n, int base, int len, double& fa, double& fb) {
  int pa = base, pb = base;
  if (len > 0) {
    for (int l = len; l > 1;) {
      const int half = l >> 1;
      double va = xp[pa + half - 1], vb = xp[pb + half - 1];
      pa += (va <= xa) ? half : 0;
      pb += (vb <= xb) ? half : 0;
      l -= half;
    }
    pa += (xp[pa] <= xa) ? 1 : 0;
    pb += (xp[pb] <= xb) ? 1 : 0;
  }
  int ia = pa < 1 ? 1 : (pa > n - 1 ? n - 1 : pa), ib = pb < 1 ? 1 : (pb > n - 1 ? n - 1 : pb);
  double x0a = xp[ia - 1], x1a = xp[ia], f0a = fp[ia - 1], f1a = fp[ia];
  double x0b = xp[ib - 1], x1b = xp[ib], f0b = fp[ib - 1], f1b = fp[ib];
  const double epsilon = 4.930380657631324e-32;
  double dxa = x1a - x0a, dxb = x1b - x0b;
  fa = (fabs(dxa) <= epsilon) ? f0a : f0a + ((xa - x0a) / dxa) * (f1a - f0a);
  fb = (fabs(dxb) <= epsilon) ? f0b : f0b + ((xb - x0b) / dxb) * (f1b - f0b);
  double xfirst = xp[0], xlast = xp[n - 1];
  if (xa < xfirst) fa = fp[0];
  if (xa > xlast) fa = fp[n - 1];
  if (xb < xfirst) fb = fp[0];
  if (xb > xlast) fb = fp[n - 1];
}

// z = z_from_dGW(dL) (cosmo.py:260-264) for two samples.  jnp_interp_x2's halving search returns searchsorted's answer on a sorted
// table; on a NON-monotonic dL table (modified propagation with Xi(z) falling fast, or a closed universe past the antipode) the
// result of a binary search depends on its probe sequence, so the reference's own search (jnp.searchsorted, method 'scan':
// searchsorted_right_scan) is followed step by step there -- flagged per draw by k_tables, never taken for sensible parameters.
template <class AccX, class AccF>
DEVFN void z_from_dGW_x2(const DevParams& p, double xa, double xb, AccX dLt, AccF zt, double& za, double& zb) {
  if (p.dl_sorted != 0.) { jnp_interp_x2(xa, xb, dLt, zt, p.Tc, za, zb); return; }
  za = jnp_interp(xa, dLt, zt, p.Tc, false, 0., 0., true);
  zb = jnp_interp(xb, dLt, zt, p.Tc, false, 0., 0., true);
}

// An unphysical draw (E(z)^2 < 0 somewhere, e.g. a strongly closed universe) leaves NaNs in the cumulative table of 1/E from
// some node on; every event grid reaching that far has NaN Jacobian / background factors there, and the reference's integrand
// 0 * NaN = NaN makes L_i NaN -> log L_i = -inf for EVERY live pixel, also where p_gw is zero (likelihood.py:274-278, 296-297).
// The kernels skip grid points outside the KDE's support, so they test this condition explicitly.
DEVFN bool grid_is_poisoned(double z_bad, const double* zg, int Z) {
  double a = zg[0], b = zg[Z - 1];
  return (a > b ? a : b) >= z_bad;
}

// jnp.logaddexp(0, x) = max(0,x) + log1p(exp(-|x|))
DEVFN double logaddexp0(double x) {
  double amax = x > 0. ? x : 0.;
  if (x != x) return x;
  return amax + log1p(chm_exp(-fabs(x)));
}

// ------------------------------------------------------------------------------------------------------
// cosmology  (population/cosmo.py)
// ------------------------------------------------------------------------------------------------------

// cosmo.py:122-130; lzp1 = log(1+z)
DEVFN double E_at_z_l(const DevParams& p, double z, double lzp1) {
  double zp1 = 1. + z;
  double w_z = p.w0 + p.wa * z / (1. + z);
  double z2 = zp1 * zp1;
  double z3 = z2 * zp1;
  double z4 = z2 * z2;
  double ex = 3. * (1. + w_z);
  double de = ex == 0. ? 1. : pow_l(lzp1, ex);               // (1+z)^(3(1+w(z))); exponent 0 for a cosmological constant
  return sqrt(p.Om0 * z3 + p.Or0 * z4 + p.Ok0 * z2 + p.Ode0 * de);
}
// the same with 1 + z and its reciprocal given: z/(1+z) as a product (1 ulp of w(z))
DEVFN double E_at_z_lr(const DevParams& p, double z, double zp1, double rzp1, double lzp1) {
  const double z2 = zp1 * zp1, z3 = z2 * zp1, z4 = z2 * z2;
  double de = 1.;
  if (!(p.wa == 0. && p.w0 == -1.)) {
    const double ex = 3. * (1. + (p.w0 + p.wa * (z * rzp1)));
    de = ex == 0. ? 1. : pow_l(lzp1, ex);
  }
  return sqrt(p.Om0 * z3 + p.Or0 * z4 + p.Ok0 * z2 + p.Ode0 * de);
}
DEVFN bool de_needs_log(const DevParams& p) { return !(p.wa == 0. && p.w0 == -1.); }
DEVFN double E_at_z(const DevParams& p, double z) {
  return E_at_z_l(p, z, de_needs_log(p) ? chm_log(1. + z) : 0.);
}

// cosmo.py:225-228
DEVFN double Xi_at_z_l(const DevParams& p, double lzp1) {
  return p.Xi0 + (1. - p.Xi0) / pow_l(lzp1, p.n_mg);
}
DEVFN double Xi_at_z(const DevParams& p, double z) { return Xi_at_z_l(p, chm_log(1. + z)); }

// cosmo.py:141-153 given dCr
DEVFN double dCt_from_dCr(const DevParams& p, double dCr) {
  if (p.Ok0 == 0.0) return dCr;
  double sqrtOk0 = sqrt(fabs(p.Ok0 + 1.e-10));
  if (p.Ok0 > 0.0) return (p.dH / sqrtOk0) * sinh(sqrtOk0 * dCr / p.dH);
  return (p.dH / sqrtOk0) * sin(sqrtOk0 * dCr / p.dH);
}

// cosmo.py:132-153
template <class A1, class A2>
DEVFN double dCt_at_z(const DevParams& p, double z, A1 zt, A2 It) {
  double dCr = p.dH * jnp_interp(z, zt, It, p.Tc, false, 0., 0.);
  return dCt_from_dCr(p, dCr);
}

// cosmo.py:201-203, 230-235
DEVFN double dL2dCt_l(const DevParams& p, double dist, double z, double lzp1) {
  if (p.cosmo_model == 1) return (dist / Xi_at_z_l(p, lzp1)) / (1. + z);
  return dist / (1. + z);
}
DEVFN double dL2dCt(const DevParams& p, double dist, double z) {
  return dL2dCt_l(p, dist, z, p.cosmo_model == 1 ? chm_log(1. + z) : 0.);
}

// cosmo.py:205-210, 237-243 given dCt
DEVFN double dL_from_dCt(const DevParams& p, double dCt, double z) {
  double dL = dCt * (1. + z);
  if (p.cosmo_model == 1) return dL * Xi_at_z(p, z);
  return dL;
}

// cosmo.py:212-221, 245-257 given dCt, E(z) and log(1+z)
DEVFN double ddLdz_from_dCt_E(const DevParams& p, double dCt, double z, double Ez, double lzp1) {
  double ddLflrw = dCt + (p.dH / Ez) * (1. + z);
  if (p.cosmo_model == 1) {
    double dLflrw = dCt * (1. + z);
    double Xiz = Xi_at_z_l(p, lzp1);
    double dXiz = p.n_mg * (p.Xi0 - 1.) / pow_l(lzp1, p.n_mg + 1.);
    return ddLflrw * Xiz + dLflrw * dXiz;
  }
  return ddLflrw;
}
// the same with 1/E(z) given and the modified-propagation factors from ONE exp: Xi = Xi0 + (1 - Xi0) q, q = (1+z)^-n, and
// dL_flrw dXi/dz = dCt (1+z) n (Xi0 - 1) q / (1+z) = dCt n (Xi0 - 1) q  (no division; k_zfactors)
DEVFN double ddLdz_from_dCt_rE(const DevParams& p, double dCt, double zp1, double rEz, double lzp1) {
  const double ddLflrw = dCt + (p.dH * rEz) * zp1;
  if (p.cosmo_model == 1) {
    const double q = chm_exp(-p.n_mg * lzp1);
    const double Xiz = p.Xi0 + (1. - p.Xi0) * q;
    return ddLflrw * Xiz + dCt * (p.n_mg * (p.Xi0 - 1.) * q);
  }
  return ddLflrw;
}
DEVFN double ddLdz_from_dCt(const DevParams& p, double dCt, double z) {
  double l = (p.cosmo_model == 1 || de_needs_log(p)) ? chm_log(1. + z) : 0.;
  return ddLdz_from_dCt_E(p, dCt, z, E_at_z_l(p, z, l), l);
}

// cosmo.py:188-197 given dCt and E(z)
DEVFN double dVcdz_from_dCt_E(const DevParams& p, double dCt, double Ez) {
  return 4. * CHM_PI * p.dH * (dCt * dCt) / Ez;
}
DEVFN double dVcdz_from_dCt(const DevParams& p, double dCt, double z) {
  return dVcdz_from_dCt_E(p, dCt, E_at_z(p, z));
}

// cosmo.py:166-186 given dCt
DEVFN double Vc_from_dCt(const DevParams& p, double dCt) {
  double dH = p.dH;
  if (p.Ok0 == 0.0) return 4. * CHM_PI * (dCt * dCt * dCt) / 3.;
  double regOk0 = p.Ok0 + 1e-10;
  double sqrtOk0 = sqrt(fabs(regOk0));
  double dH3 = dH * dH * dH;
  double pre = 4. * CHM_PI * dH3 / (2. * regOk0);
  double a = (dCt / dH) * sqrt(1. + regOk0 * (dCt * dCt) / (dH * dH));
  if (p.Ok0 > 0.0) return pre * (a - asinh(sqrtOk0 * dCt / dH) / sqrtOk0);
  return pre * (a - asin(sqrtOk0 * dCt / dH) / sqrtOk0);
}

// ------------------------------------------------------------------------------------------------------
// mass  (population/mass.py)
// ------------------------------------------------------------------------------------------------------

// mass.py:255-264
DEVFN double smoothing(double m, double delta_m, double m_low) {
  if (m < m_low) return 0.;
  if (m > m_low + delta_m) return 1.;
  const double eps = 1.e-99;
  double a = m - m_low + eps, b = m - m_low - delta_m + eps;
  double x = delta_m * ((a + b) / (a * b));                  // = delta_m/a + delta_m/b with one division
  // exp(-logaddexp(0, x)) = 1/(1 + e^x), evaluated without overflow; equal to the reference's form to ~2 ulp
  if (x != x) return x;
  if (x > 0.) { double t = chm_exp(-x); return t / (1. + t); }
  return 1. / (1. + chm_exp(x));
}

// mass.py:240-245; lm = log(m)
DEVFN double tpl_notnorm_l(double m, double lm, double alpha, double m_low, double m_high) {
  return (m_low <= m && m <= m_high) ? pow_l(lm, alpha) : 0.;
}
DEVFN double tpl_notnorm(double m, double alpha, double m_low, double m_high) {
  return (m_low <= m && m <= m_high) ? pow_el(m, alpha) : 0.;
}

// mass.py:247-252
DEVFN double tpl_cdf(double alpha, double m_low, double m) {
  if (alpha == -1.) return chm_log(m_low) - chm_log(m);
  return (pow(m, 1. + alpha) - pow(m_low, 1. + alpha)) / (1. + alpha);
}

// mass.py:285-305; lm = log(m)
DEVFN double primary_notnorm_l(const DevParams& p, double m, double lm) {
  double m_low = p.m[0], m_high = p.m[1];
  if (p.mass_model == 0) {                       // tpl: alpha=m[2]
    return tpl_notnorm_l(m, lm, -p.m[2], m_low, m_high);
  } else if (p.mass_model == 1) {                // bpl: alpha_1, alpha_2, beta, delta_m, break_fraction
    double pdf = tpl_notnorm_l(m, lm, -p.m[2], m_low, p.bpl_mbreak);
    pdf = pdf + tpl_notnorm_l(m, lm, -p.m[3], p.bpl_mbreak, m_high) * p.bpl_pl1 / p.bpl_pl2;
    return pdf * smoothing(m, p.m[5], m_low);
  } else {                                       // plp: lambda_peak, alpha, beta, delta_m, mu_g, sigma_g
    double lam = p.m[2], mu = p.m[6], sg = p.m[7];
    double P = tpl_notnorm_l(m, lm, -p.m[3], m_low, m_high) * p.inv_plnorm;
    double G = 0.;
    if (m_low <= m && m <= p.tg_hi) {
      double d = m - mu;
      G = chm_exp(p.g_c0 - (d * d) * p.inv_2s2) * p.inv_tg_norm;       // mass.py:267-279 (divisions by constants as reciprocals)
    }
    double pdf = (1. - lam) * P + lam * G;
    return pdf * smoothing(m, p.m[5], m_low);
  }
}

DEVFN double primary_notnorm(const DevParams& p, double m) { return primary_notnorm_l(p, m, chm_log(m)); }

DEVFN double mass_beta(const DevParams& p) { return p.mass_model == 0 ? p.m[3] : (p.mass_model == 1 ? p.m[4] : p.m[4]); }
DEVFN double mass_delta_m(const DevParams& p) { return p.m[5]; }

// mass.py:320-328; lm2 = log(m2)
DEVFN double secondary_notnorm_l(const DevParams& p, double m2, double lm2, double m1) {
  double pdf = tpl_notnorm_l(m2, lm2, mass_beta(p), p.m[0], m1);
  if (p.mass_model == 0) return pdf;
  return pdf * smoothing(m2, mass_delta_m(p), p.m[0]);
}
DEVFN double secondary_notnorm(const DevParams& p, double m2, double m1) { return secondary_notnorm_l(p, m2, chm_log(m2), m1); }

// jnp.interp(m1, m_grid, cdf) (mass.py:339) with the bracket found from log(m1): m_grid is a logspace (mass.py:46), so
// the index is floor((log10 m1 - l0)/(l1 - l0) (Tm-1)) up to rounding; the fix-up loops restore searchsorted exactly.
template <class A1, class A2>
DEVFN double interp_mgrid(const DevParams& p, double m1, double lm1, A1 mg, A2 cdf) {
  const int n = p.Tm;
  double t = (lm1 - p.lmg0) * p.inv_dlmg;                  // position in units of grid steps
  int i = (t >= 0.) ? (t < (double)n ? (int)t + 1 : n - 1) : 1;
  i = i < 1 ? 1 : (i > n - 1 ? n - 1 : i);
  while (i > 1 && mg[i - 1] > m1) i--;
  while (i < n - 1 && mg[i] <= m1) i++;
  double x0 = mg[i - 1], x1 = mg[i], f0 = cdf[i - 1], f1 = cdf[i];
  double dx = x1 - x0;
  double f = (fabs(dx) <= 4.930380657631324e-32) ? f0 : f0 + ((m1 - x0) / dx) * (f1 - f0);
  if (m1 < (double)mg[0]) f = cdf[0];
  if (m1 > (double)mg[n - 1]) f = cdf[n - 1];
  return f;
}

// mass.py:334-341
template <class A1, class A2>
DEVFN double p_m1m2_l(const DevParams& p, double m1, double m2, double lm1, double lm2, A1 mg, A2 cdf) {
  double p_m1 = primary_notnorm_l(p, m1, lm1) * p.inv_norm_p_m1;
  double p_m2m1 = secondary_notnorm_l(p, m2, lm2, m1) / interp_mgrid(p, m1, lm1, mg, cdf);
  if (p_m2m1 != p_m2m1) p_m2m1 = 0.;
  return p_m1 * p_m2m1;
}
template <class A1, class A2>
DEVFN double p_m1m2(const DevParams& p, double m1, double m2, A1 mg, A2 cdf) {
  return p_m1m2_l(p, m1, m2, chm_log(m1), chm_log(m2), mg, cdf);
}

// p_m1m2 (mass.py:334-341) for the per-sample hot loops, same quantity as p_m1m2_l() with the operation count cut:
//   p_m1m2 = [Pn(m1) S(m1) / norm] [m2^beta S(m2) 1(m_low <= m2 <= m1)] / interp(m1; m_grid, cdf_m2)
// * the two smoothing arguments x = dm (a+b)/(a b) (mass.py:261-263) share one reciprocal, and S = 1/(1 + e^x) is kept as its
//   denominator D = 1 + e^x  (e^x = inf -> S = 0, as the reference's exp(-logaddexp(0, x)) underflows);
// * m2^beta = e^(beta lm2) is folded into the exponents of the primary's power laws / Gaussian (one exp fewer);
// * the cdf interpolant f0 + (dm1/dx) df is kept as (f0 dx + dm1 df)/dx, so that S(m1), S(m2), the interpolant and the
//   final quotient need ONE division:  w = Pn norm^-1 dx / (D1 D2 (f0 dx + dm1 df)).
// Relative difference to p_m1m2_l(): rounding only, <= ~(|beta lm2| + 8) ulp ~ 3e-15; the `p_m2m1 = NaN -> 0` rule
// (mass.py:340) is applied to the same cases (0/0 at m1 = m2 = m_low).
// MASS >= 0: the mass model is a compile-time constant (k_samples_fast / k_selection instantiate the hot loops per model, so the
// other models' parameters never occupy scalar registers); MASS = -1: read from the draw (generic kernels).
// The mass-model parameters the per-sample / per-injection loops read, moved from scalar into VECTOR registers (a uniform value in every
// lane; the empty asm hides the uniformity from the compiler).  The kernels that call p_m1m2_fused hold ~35 doubles of the draw, a dozen
// pointers and the exec-mask stack in 102 SGPRs: the allocator spilled half of them to VGPR lanes and re-read them with v_readlane inside
// the loops (84 of 860 VALU instructions per pass of k_selection_fast), while -- with the polynomial coefficients out of the VGPRs -- a
// quarter of the vector registers stood empty.
#define CHM_TO_VGPR(x) asm volatile("" : "+v"(x))
// N: how many of them (in the order of their use count in p_m1m2_fused<MASS>) -- as many as the kernel's vector registers take without spilling
template <int MASS, int N>
DEVFN void mass_params_to_vgpr(DevParams& p) {
  int n = 0;
#define CHM_TV(x) do { if (n++ < N) CHM_TO_VGPR(x); } while (0)
  CHM_TV(p.m[0]); CHM_TV(p.m[1]); CHM_TV(p.lmg0); CHM_TV(p.inv_dlmg); CHM_TV(p.inv_norm_p_m1); CHM_TV(p.m[3]);
  if (MASS != 0) { CHM_TV(p.m[4]); CHM_TV(p.m[5]); }
  CHM_TV(p.m[2]);
  if (MASS == 1) { CHM_TV(p.bpl_mbreak); }
  if (MASS == 2) { CHM_TV(p.m[6]); CHM_TV(p.inv_2s2); CHM_TV(p.g_c0); CHM_TV(p.inv_plnorm); CHM_TV(p.inv_tg_norm); CHM_TV(p.tg_hi); }
#undef CHM_TV
}

// A window [off, off + len) of a per-draw table held in LDS, addressed with the table's own indices (p = window start - off); the end values
// of the whole table travel with it (k_marg_fused stages only the stretch of m_grid / cdf_m2 an event's source-frame masses can reach)
struct TabSlice { const double* p; double first, last; DEVFN double operator[](int i) const { return p[i]; } };
template <class A> DEVFN double tab_first(A a) { return (double)a[0]; }
template <class A> DEVFN double tab_last(A a, int n) { return (double)a[n - 1]; }
DEVFN double tab_first(const TabSlice& a) { return a.first; }
DEVFN double tab_last(const TabSlice& a, int) { return a.last; }
// [r5] the end nodes of the mass grid and the last value of cdf_m2 travel with the draw (DevParams: k_tables stores exactly these values in mg[0], mg[Tm - 1],
// cdf[Tm - 1]): the per-sample loops compare against scalars instead of reading the table ends from LDS for every sample
template <class A> DEVFN double mgrid_first(const DevParams& p, A) { return p.mg_first; }
template <class A> DEVFN double mgrid_last(const DevParams& p, A, int) { return p.mg_last; }
template <class A> DEVFN double cdf_last_of(const DevParams& p, A, int) { return p.cdf_last; }
DEVFN double mgrid_first(const DevParams&, const TabSlice& a) { return a.first; }
DEVFN double mgrid_last(const DevParams&, const TabSlice& a, int) { return a.last; }
DEVFN double cdf_last_of(const DevParams&, const TabSlice& a, int) { return a.last; }

template <int MASS = -1, class A1, class A2, class EX = ExpPoly>
DEVFN double p_m1m2_fused(const DevParams& p, double m1, double m2, double lm1, double lm2, A1 mg, A2 cdf, const EX ex = EX()) {
#pragma clang fp contract(fast)                  // smooth arithmetic only (no rounding-sensitive predicate): a*b+c may fuse
  const int mass_model = MASS >= 0 ? MASS : p.mass_model;
  const double m_low = p.m[0], m_high = p.m[1];
  const bool in2 = (m_low <= m2 && m2 <= m1);               // tpl_notnorm(m2, beta, m_low, m1)   mass.py:240-245,322
  const double e5 = in2 ? (mass_model == 0 ? p.m[3] : p.m[4]) * lm2 : 0.;
  // primary numerator (without smoothing), times m2^beta                                         mass.py:285-305
  // [r3] chm_exp_nb: every exponential below is selected by a window predicate on finite masses, so an argument outside the range of exp
  // only ever produces a value that is discarded (or the inf / 0 that v_ldexp_f64 saturates to) -- no range checks (6 instructions each)
  double Pn;
  if (mass_model == 0) {
    Pn = (m_low <= m1 && m1 <= m_high) ? ex.nb(-p.m[2] * lm1 + e5) : 0.;
  } else if (mass_model == 1) {
    double a = (m_low <= m1 && m1 <= p.bpl_mbreak) ? ex.nb(-p.m[2] * lm1 + e5) : 0.;
    double b = (p.bpl_mbreak <= m1 && m1 <= m_high) ? ex.nb(-p.m[3] * lm1 + e5) : 0.;
    Pn = a + b * p.bpl_pl1 / p.bpl_pl2;
  } else {
    double Pw = (m_low <= m1 && m1 <= m_high) ? ex.nb(-p.m[3] * lm1 + e5) * p.inv_plnorm : 0.;
    double G = 0.;
    if (m_low <= m1 && m1 <= p.tg_hi) { double d = m1 - p.m[6]; G = ex.nb((p.g_c0 - (d * d) * p.inv_2s2) + e5) * p.inv_tg_norm; }
    Pn = (1. - p.m[2]) * Pw + p.m[2] * G;
  }
  // smoothing denominators                                                                       mass.py:255-264
  double D1 = 1., D2 = 1.;
  bool zero = !in2;
  if (mass_model != 0) {
    const double dm = p.m[5], eps = 1.e-99;
    const bool w1 = !(m1 < m_low) && !(m1 > m_low + dm), w2 = !(m2 < m_low) && !(m2 > m_low + dm);
    zero = zero || (m1 < m_low) || (m2 < m_low);
    double ab1 = 1., ab2 = 1., s1 = 0., s2 = 0.;
    if (w1) { double a = m1 - m_low + eps, b = m1 - m_low - dm + eps; ab1 = a * b; s1 = a + b; }
    if (w2) { double a = m2 - m_low + eps, b = m2 - m_low - dm + eps; ab2 = a * b; s2 = a + b; }
    if (w1 || w2) {
      double r = chm_div(dm, ab1 * ab2);
      const double x1 = (s1 * ab2) * r, x2 = (s2 * ab1) * r;
      if (w1) D1 = 1. + ex.clamped(x1);
      if (w2) D2 = 1. + ex.clamped(x2);
      zero = zero || (w1 && x1 > 745.14) || (w2 && x2 > 745.14);      // exp(-logaddexp(0, x)) underflows to an exact 0 there   mass.py:264
    }
  }
  // interp(m1; m_grid, cdf_m2) as (f0 dx + (m1 - x0) df) / dx                                     mass.py:339
  // [r3] the bracket comes from the position of log(m1) on the (logspace) grid; one look at the two nodes it names decides whether the
  // neighbouring interval is the right one instead (rounding of the position, a grid that is not exactly logspace) -- the stepping loops
  // run only then
  const int n = p.Tm;
  double t = (lm1 - p.lmg0) * p.inv_dlmg;
  int i = (t >= 0.) ? (t < (double)n ? (int)t + 1 : n - 1) : 1;
  i = i < 1 ? 1 : (i > n - 1 ? n - 1 : i);
  double x0 = mg[i - 1], x1 = mg[i];
  if (wave_any((i > 1 && x0 > m1) || (i < n - 1 && x1 <= m1))) {
    while (i > 1 && mg[i - 1] > m1) i--;
    while (i < n - 1 && mg[i] <= m1) i++;
    x0 = mg[i - 1]; x1 = mg[i];
  }
  const double f0 = cdf[i - 1], f1 = cdf[i];
  double dx = x1 - x0;
  double cn = f0 * dx + (m1 - x0) * (f1 - f0);
  // (jnp.interp's epsilon rule for a degenerate interval: under a vote -- four selects per sample otherwise, for a case a logspace grid never meets)
  if (wave_any(fabs(dx) <= 4.930380657631324e-32)) { asm volatile(""); if (fabs(dx) <= 4.930380657631324e-32) { cn = f0; dx = 1.; } }
  if (m1 < mgrid_first(p, mg)) cn = tab_first(cdf) * dx;
  if (m1 > mgrid_last(p, mg, n)) cn = cdf_last_of(p, cdf, n) * dx;
  // one quotient without IEEE special cases: every factor is finite (chm_exp_clamped), the product of two saturated denominators is
  // capped at 1e300 (a factor 1e-300 on the weight); a zero interpolant (m1 at the lowest node) gives NaN here and 0 below
  const double num = (Pn * p.inv_norm_p_m1) * dx, den = vmin_f64((D1 * D2) * cn, 1e300);
  double w = chm_div(num, den);
  // m1 at the lowest node of the grid: the IEEE quotient (x/0 = inf, 0/0 = NaN) as the reference forms it.  [r5] The empty asm keeps the IEEE sequence INSIDE
  // the voted branch: the compiler had hoisted it (a division is speculatable) and every sample paid both quotients -- 11 instructions and a second
  // v_rcp_f64 -- for a select that is false in all but pathological inputs
  if (wave_any(den == 0.)) { asm volatile(""); if (den == 0.) w = num / den; }
  // sec = 0 -> p_m2m1 = 0 (or 0/0 = NaN -> 0): w = p_m1 * 0
  if (zero || (w != w && cn == 0.)) w = Pn * 0.;
  // a NaN primary mass: the smoothing window of bpl / plp turns p_m1 NaN (NaN * 0 = NaN); the truncated power law has no window, every factor
  // is a masked 0 there (mass.py:240-245, 334-341: p_m1 = 0, p_m2m1 = NaN -> 0) -- found by scripts/fuzz_parity.py in round 4
  if (mass_model != 0 && m1 != m1) w = m1;
  return w;
}

// ------------------------------------------------------------------------------------------------------
// rate  (population/rate.py:96-122)
// ------------------------------------------------------------------------------------------------------
// pow() with the value classes of C99 / IEEE 754 (what jnp.power and NumPy's ** return for infinite or zero operands); ordinary arguments go
// through exp(y log x) like every other power on the device
DEVFN double pow_c99(double x, double y) {
  const double inf = __builtin_inf(), nan = __builtin_nan("");
  if (y == 0. || x == 1.) return 1.;
  if (x != x || y != y) return nan;
  const double ax = fabs(x);
  if (fabs(y) == inf) return ax == 1. ? 1. : (((ax > 1.) == (y > 0.)) ? inf : 0.);
  const bool yint = floor(y) == y, yodd = yint && fabs(y) < 9007199254740992. && (((long long)y) & 1ll);
  if (ax == inf || ax == 0.) {
    const bool big = (ax == inf) == (y > 0.), neg = __builtin_signbit(x) && yodd;
    return big ? (neg ? -inf : inf) : (neg ? -0. : 0.);
  }
  if (x < 0.) { if (!yint) return nan; const double v = chm_exp(y * chm_log_pos(ax)); return yodd ? -v : v; }
  return chm_exp(y * chm_log_pos(x));
}
// merger_rate (rate.py:96-122) for a draw with an infinite gamma, kappa or z_p, in the reference's own operations: two powers, one quotient, one
// product -- NumPy / XLA reach a finite limit, 0, inf or NaN there (e.g. kappa = +inf: (1+z)^gamma below z_p, 0 above; gamma = +inf: inf / NaN),
// where exp(y log x) on y log x = inf - inf or 0 * inf gives NaN throughout.  Kept OUT of the per-point / per-injection loops of the production kernels
// (inlined there it cost k_zfactors 40 scalar spills and a private segment): a call with such a draw runs the whole-grid per-z factors, then
// k_rate_special rewrites the draw's rate factors with this function and flags the events whose grid holds a non-finite one (event_poisoned), and
// takes k_selection<., true> for the selection sums.
DEVFN double merger_rate_special(const DevParams& p, double z) {
  const double g = p.r[0], zp1 = 1. + z;
  if (p.rate_model == 0) return pow_c99(zp1, g);
  if (p.rate_model == 2) { const double pdf = pow_c99(zp1, g); return z < p.r[3] ? pdf / p.tpl_rate_norm : 0.; }
  const double md = pow_c99(zp1, g) / (1. + pow_c99(zp1 / (1. + p.r[2]), g + p.r[1]));
  const double v = p.md_norm * md;
  return (p.rate_model == 1 || z < p.r[3]) ? v : 0.;
}
DEVFN double merger_rate_l(const DevParams& p, double z, double lzp1) {
  double g = p.r[0];
  if (p.rate_model == 0) return pow_l(lzp1, g);
  if (p.rate_model == 2) {
    double pdf = pow_l(lzp1, g);
    return z < p.r[3] ? pdf / p.tpl_rate_norm : 0.;
  }
  double k = p.r[1];
  // ((1+z)/(1+zp))^(g+k) with log((1+z)/(1+zp)) = log(1+z) - log(1+zp)
  double md = pow_l(lzp1, g) / (1. + pow_l(lzp1 - p.l1pzp, g + k));
  if (p.rate_model == 1) return p.md_norm * md;
  return z < p.r[3] ? p.md_norm * md : 0.;
}
DEVFN double merger_rate(const DevParams& p, double z) { return p.rate_special ? merger_rate_special(p, z) : merger_rate_l(p, z, chm_log(1. + z)); }
// merger_rate_l as a quotient num/den (k_selection_fast folds den into its one division); the same value classes: a cut model is
// 0 / den above its z cut, (1 + ...) overflowing to inf gives 0 as the division does.
// (EX: the exp the powers go through -- ExpTab in the fast selection kernel: v_ldexp_f64 saturates where chm_exp tests its range)
template <class EX = ExpPoly>
DEVFN void merger_rate_nd(const DevParams& p, double z, double lzp1, double& num, double& den, const EX ex = EX()) {
  const double g = p.r[0];
  const double a = ex.pw(lzp1, g);
  if (p.rate_model == 0) { num = a; den = 1.; return; }
  if (p.rate_model == 2) { num = z < p.r[3] ? a : 0.; den = p.tpl_rate_norm; return; }
  den = 1. + ex.pw(lzp1 - p.l1pzp, g + p.r[1]);
  num = p.md_norm * a;
  if (p.rate_model != 1 && !(z < p.r[3])) num = 0.;
}

// ------------------------------------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes)
// ------------------------------------------------------------------------------------------------------
DEVFN double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// NaN-propagating min / max (jnp.min / jnp.max semantics)
DEVFN double nanmin2(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
DEVFN double nanmax2(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b > a ? b : a)); }
DEVFN double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = nanmin2(v, __shfl_xor(v, o, 64));
  return v;
}
DEVFN double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = nanmax2(v, __shfl_xor(v, o, 64));
  return v;
}

// Cross-lane steps of the scans / reductions inside a lane group as DPP moves (two v_mov_b32_dpp per double) instead of
// ds_bpermute shuffles: row_shr:n inside rows of 16 lanes, row_bcast:15 / row_bcast:31 across rows (gfx9 DPP controls).
// FILL0: lanes without a source lane receive 0 (neutral for a sum); otherwise they keep their own value (neutral for a max).
template <int CTRL, int ROW_MASK, bool FILL0>
DEVFN double dpp_move(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  int lo2 = __builtin_amdgcn_update_dpp(FILL0 ? 0 : lo, lo, CTRL, ROW_MASK, 0xf, false);
  int hi2 = __builtin_amdgcn_update_dpp(FILL0 ? 0 : hi, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi2, lo2);
}
// 64-lane reductions on DPP moves; the result is read from lane 63 and is uniform.  ALL 64 lanes must be active at the call.
DEVFN double lane63(double x) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
  return __hiloint2double(hi, lo);
}
DEVFN double wave_sum_dpp(double v) {
  v += dpp_move<0x111, 0xf, true>(v); v += dpp_move<0x112, 0xf, true>(v); v += dpp_move<0x114, 0xf, true>(v);
  v += dpp_move<0x118, 0xf, true>(v); v += dpp_move<0x142, 0xa, true>(v); v += dpp_move<0x143, 0xc, true>(v);
  return lane63(v);
}
DEVFN double wave_max_dpp(double v) {                      // v_max_f64: NaN-ignoring
  v = __builtin_fmax(v, dpp_move<0x111, 0xf, false>(v)); v = __builtin_fmax(v, dpp_move<0x112, 0xf, false>(v));
  v = __builtin_fmax(v, dpp_move<0x114, 0xf, false>(v)); v = __builtin_fmax(v, dpp_move<0x118, 0xf, false>(v));
  v = __builtin_fmax(v, dpp_move<0x142, 0xa, false>(v)); v = __builtin_fmax(v, dpp_move<0x143, 0xc, false>(v));
  return lane63(v);
}
DEVFN double wave_min_dpp(double v) {
  v = __builtin_fmin(v, dpp_move<0x111, 0xf, false>(v)); v = __builtin_fmin(v, dpp_move<0x112, 0xf, false>(v));
  v = __builtin_fmin(v, dpp_move<0x114, 0xf, false>(v)); v = __builtin_fmin(v, dpp_move<0x118, 0xf, false>(v));
  v = __builtin_fmin(v, dpp_move<0x142, 0xa, false>(v)); v = __builtin_fmin(v, dpp_move<0x143, 0xc, false>(v));
  return lane63(v);
}

// block-wide reductions for blockDim.x <= 1024; scratch: >= 16 doubles of LDS; result broadcast to all threads.
enum { RED_SUM = 0, RED_MIN = 1, RED_MAX = 2 };
template <int OP>
DEVFN double block_reduce(double v, double* scratch) {
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = OP == RED_SUM ? wave_sum_dpp(v) : (OP == RED_MIN ? wave_min(v) : wave_max(v));     // every thread of the block reaches this call
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  double r;
  if (OP == RED_SUM) { r = 0.; for (int i = 0; i < nw; i++) r += scratch[i]; }
  else if (OP == RED_MIN) { r = scratch[0]; for (int i = 1; i < nw; i++) r = nanmin2(r, scratch[i]); }
  else { r = scratch[0]; for (int i = 1; i < nw; i++) r = nanmax2(r, scratch[i]); }
  return r;
}
