#include <cstdio>
#include <cmath>
#include <random>
// host build of chimera_amd/csrc/chm_math.h's algorithms (rcp emulated in fp32):  g++ -O2 -ffp-contract=off scripts/check_fastmath.cpp && ./a.out
#define DEVFN static inline
static inline double FM_RCP(double x) { return (double)(1.0f / (float)x); }
static inline double FM_FREXP_M(double x) { int e; return std::frexp(x, &e); }
static inline int FM_FREXP_E(double x) { int e; std::frexp(x, &e); return e; }
DEVFN double chm_exp(double x) {
  const double L2E = 1.44269504088896338700e+00, LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
  double n = __builtin_rint(x * L2E);
  double r = __builtin_fma(-n, LN2HI, x);
  r = __builtin_fma(-n, LN2LO, r);
  double p = 1.6059043836821613e-10;                 // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);     // 1/12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);    // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);    // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06);   // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);     // 1/8!
  p = __builtin_fma(p, r, 0.0001984126984126984);    // 1/7!
  p = __builtin_fma(p, r, 0.001388888888888889);     // 1/6!
  p = __builtin_fma(p, r, 0.008333333333333333);     // 1/5!
  p = __builtin_fma(p, r, 0.041666666666666664);     // 1/4!
  p = __builtin_fma(p, r, 0.16666666666666666);      // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  int k = (int)n;
  double v = __builtin_ldexp(p, k);
  if (x > 709.782712893384) v = __builtin_inf();
  if (x < -745.1332191019412) v = 0.;
  return v;
}

// degree-11 form: exp(r) = 1 + r + r^2 q(r), q the degree-9 Chebyshev fit of (e^r - 1 - r)/r^2 on |r| <= ln2/2 (mpmath.chebyfit, error 1e-17)
DEVFN double chm_exp11(double x) {
  const double L2E = 1.44269504088896338700e+00, LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
  double n = __builtin_rint(x * L2E);
  double r = __builtin_fma(-n, LN2HI, x);
  r = __builtin_fma(-n, LN2LO, r);
  double p = 2.510038549551032e-08;
  p = __builtin_fma(p, r, 2.7620088445409746e-07);
  p = __builtin_fma(p, r, 2.7557268459997064e-06);
  p = __builtin_fma(p, r, 2.4801521295954376e-05);
  p = __builtin_fma(p, r, 0.00019841269863053618);
  p = __builtin_fma(p, r, 0.0013888888917213717);
  p = __builtin_fma(p, r, 0.008333333333330062);
  p = __builtin_fma(p, r, 0.04166666666662413);
  p = __builtin_fma(p, r, 0.16666666666666669);
  p = __builtin_fma(p, r, 0.5000000000000001);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  int k = (int)n;
  double v = __builtin_ldexp(p, k);
  if (x > 709.782712893384) v = __builtin_inf();
  if (x < -745.1332191019412) v = 0.;
  return v;
}
// log(x) for finite x > 0 (NaN propagates); fdlibm e_log.c scheme, < 1 ulp
DEVFN double chm_log_pos(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  double m = FM_FREXP_M(x);
  int e = FM_FREXP_E(x);
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  double f = m - 1.0;
  double d = 2.0 + f;
  double r = FM_RCP(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  double s = f * r;
  s = __builtin_fma(__builtin_fma(-d, s, f), r, s);
  double z = s * s, w = z * z;
  double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
  double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  double dk = (double)e;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}
DEVFN double chm_log(double x) {
  double v = chm_log_pos(x);
  if (x == 0.) v = -__builtin_inf();
  if (x < 0.) v = __builtin_nan("");
  if (x == __builtin_inf()) v = x;
  return v;
}

// [r3] table form of the exp of the mass model (chm_exp_tab / exp_table_entry of chm_math.h): T[j] = 2^(j/256) from chm_exp11 on a two-piece
// argument, exp(x) = 2^k T[j] (1 + r + r^2/2 + r^3/6 + r^4/24)
static double TAB[256];
DEVFN double exp_table_entry(int j) {
  const double C_HI = 6.93147180369123816490e-01 / 256., C_LO = 1.90821492927058770002e-10 / 256.;
  const double a = (double)j * C_HI;
  const double v = chm_exp11(a);
  return __builtin_fma(v, (double)j * C_LO, v);
}
DEVFN double chm_exp_tab(double x) {
  const double SC = 3.69329930467574632e+02;
  const double L_HI = 6.93147180369123816490e-01 / 256., L_LO = 1.90821492927058770002e-10 / 256.;
  double n = __builtin_rint(x * SC);
  double r = __builtin_fma(-n, L_HI, x);
  r = __builtin_fma(-n, L_LO, r);
  const int ni = (int)n;
  const double tj = TAB[ni & 255];
  double p = __builtin_fma(r, 4.16666666666666644e-02, 1.66666666666666657e-01);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(tj * p, ni >> 8);
}

int main() {
  for (int j = 0; j < 256; j++) TAB[j] = exp_table_entry(j);
  {
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> ue(-700, 700), us(-1, 1);
    double mt = 0, wt = 0, tabmax = 0;
    for (int j = 0; j < 256; j++) { long double ref = exp2l((long double)j / 256.0L); double e = fabs((double)(((long double)TAB[j] - ref) / ref)); if (e > tabmax) tabmax = e; }
    for (int i = 0; i < 20000000; i++) {
      double x = i % 3 == 0 ? ue(g) : (i % 3 == 1 ? us(g) * 40 : us(g));
      long double ref = expl((long double)x);
      double e = fabs((double)(((long double)chm_exp_tab(x) - ref) / ref));
      if (e > mt) { mt = e; wt = x; }
    }
    printf("table exp (chm_exp_tab) max rel err %.3e (at %g) = %.2f half-ulp; table entries max rel err %.3e = %.2f half-ulp; specials: exp_tab(-800)=%g exp_tab(800)=%g exp_tab(nan)=%g\n",
           mt, wt, mt / 1.11e-16, tabmax, tabmax / 1.11e-16, chm_exp_tab(-800.), chm_exp_tab(800.), chm_exp_tab(NAN));
  }
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> ue(-745, 709), ul(-700, 700), us(-1, 1);
  double maxe = 0, maxl = 0, maxe11 = 0; double we=0, wl=0;
  for (int i = 0; i < 20000000; i++) {
    double x = i % 3 == 0 ? ue(g) : (i % 3 == 1 ? us(g) * 40 : us(g));
    long double ref = expl((long double)x);
    double v = chm_exp(x);
    double err = fabs((double)(((long double)v - ref) / ref));
    if (x > -708 && err > maxe) { maxe = err; we = x; }
    { double v2 = chm_exp11(x); double e2 = fabs((double)(((long double)v2 - ref) / ref)); if (x > -708 && e2 > maxe11) maxe11 = e2; }
    double y = i % 2 ? exp(ul(g)) : 1.0 + fabs(us(g)) * 6;
    long double rl = logl((long double)y);
    double vl = chm_log(y);
    double el = rl != 0 ? fabs((double)(((long double)vl - rl) / rl)) : fabs(vl);
    if (el > maxl) { maxl = el; wl = y; }
  }
  printf("exp max rel err %.3e (at %g) = %.2f ulp; log max rel err %.3e (at %g) = %.2f ulp\n", maxe, we, maxe / 1.11e-16, maxl, wl, maxl / 1.11e-16);
  printf("degree-11 exp (chm_exp since round 2) max rel err %.3e = %.2f half-ulp\n", maxe11, maxe11 / 1.11e-16);
  printf("specials: exp(-inf)=%g exp(inf)=%g exp(nan)=%g exp(-746)=%g exp(-740)=%g (ref %g) exp(710)=%g log(0)=%g log(-1)=%g log(inf)=%g log(nan)=%g log(1)=%g log(5e-324)=%g (ref %g)\n",
    chm_exp(-INFINITY), chm_exp(INFINITY), chm_exp(NAN), chm_exp(-746), chm_exp(-740), exp(-740.), chm_exp(710), chm_log(0.), chm_log(-1.), chm_log(INFINITY), chm_log(NAN), chm_log(1.), chm_log(5e-324), log(5e-324));
  // near 1
  double mx = 0;
  for (int i = -2000; i <= 2000; i++) { double y = 1.0 + i * 1e-7; long double rl = logl((long double)y); double vl = chm_log(y); if (rl != 0) { double el = fabs((double)(((long double)vl - rl) / rl)); if (el > mx) mx = el; } }
  printf("log near 1 max rel err %.3e\n", mx);
}
