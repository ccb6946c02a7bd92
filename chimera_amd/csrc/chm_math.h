// chm_math.h -- fp64 exp / log for the per-sample and per-grid-point loops (gfx950).
//
// ocml's log() is a ~95-instruction double-double routine and exp() ~36 instructions; the hot kernels are fp64-VALU bound and
// evaluate 1 log + 4-5 exp per posterior sample, so these are written out here at ~36 and ~24 instructions:
//   chm_exp      n = rint(x log2 e), r = x - n ln2 (two-term), 1 + r + r^2 q(r) with a degree-9 fit of q on |r| <= 0.347, ldexp;  max error 0.65 ulp
//   chm_log_pos  fdlibm's scheme: x = 2^k m, m in [sqrt(1/2), sqrt 2), s = f/(2+f), degree-7 minimax in s^2;  max error 0.69 ulp
//                for finite x > 0 (NaN propagates); chm_log adds log(0) = -inf, log(<0) = NaN, log(inf) = inf
// (errors measured against long double on 2e7 random arguments, scripts/check_fastmath.cpp).  Subnormal results of exp and
// subnormal arguments of log are handled by v_ldexp_f64 / v_frexp_*_f64.
#pragma once
#include <hip/hip_runtime.h>
#ifndef DEVFN
#define DEVFN __device__ __forceinline__
#endif
#define FM_RCP(x) __builtin_amdgcn_rcp(x)
#define FM_FREXP_M(x) __builtin_amdgcn_frexp_mant(x)
#define FM_FREXP_E(x) __builtin_amdgcn_frexp_exp(x)
// a*b + c as ONE three-address v_fma_f64 with the coefficient c held in a VGPR pair.  In the large kernels the compiler turns a
// Horner step with a register-resident coefficient into v_mov_b64 + v_fmac_f64 (two issue slots, seen in the gfx950 ISA of
// k_samples: 260 of 1134 VALU instructions in the loop were such copies); spelling the instruction out halves the polynomial cost.
// [r3] The coefficient c is an immediate: two s_mov_b32 into a scratch SGPR pair (s[98:99], declared clobbered; s[100:101] are reserved by the compiler on gfx950) right in front of the
// v_fma_f64 that reads it as its one scalar operand.  The ~20 coefficients of exp / log then occupy no registers at all -- in VGPRs they
// took 40 registers or two v_mov_b32 (2 cycles each) per use, in allocated SGPRs the register allocator spilled them to VGPR lanes
// (v_readlane: 4 cycles) -- and the s_mov_b32 issue on the scalar unit beside the VALU stream (v_fma_f64 + SALU pairs: 4.8 against
// 4.4 cycles, profiles/r03/issue_cost.txt).
template <unsigned LO, unsigned HI>
DEVFN double fm_fma_k(double a, double b) {
  double d;
  asm("s_mov_b32 s98, %3\n\ts_mov_b32 s99, %4\n\tv_fma_f64 %0, %1, %2, s[98:99]" : "=v"(d) : "v"(a), "v"(b), "n"(LO), "n"(HI) : "s98", "s99");
  return d;
}
#define FM_BITS(c) __builtin_bit_cast(unsigned long long, (double)(c))
#define FM_FMA(a, b, c) fm_fma_k<(unsigned)(FM_BITS(c) & 0xffffffffull), (unsigned)(FM_BITS(c) >> 32)>((a), (b))

// Single instructions the compiler does not emit on its own: v_max_f64 / v_min_f64 without the canonicalising v_max_f64 x, x that
// llvm.maxnum / minnum put in front of every loaded operand (IEEE maxNum / minNum: a quiet NaN operand yields the other one);
// v_cvt_i32_f64 (truncation toward zero, saturation at +-2^31, NaN -> 0: a C++ cast of an out-of-range double is undefined);
// v_med3_i32 as the two-sided clamp min(max(x, lo), hi) for lo <= hi.
DEVFN double vmax_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEVFN double vmin_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEVFN int cvt_i32_sat(double x) { int r; asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x)); return r; }
DEVFN int med3_i32(int x, int lo, int hi) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi)); return r; }
// wave votes straight on the ballot (a v_cmp into a scalar pair + s_cmp): HIP's __any / __all take an int, and the bool -> int -> "!= 0" round trip
// survives as v_cndmask_b32 + v_cmp_ne_u32 in front of every vote -- two VALU instructions per vote in the per-sample / per-pass loops
DEVFN bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
DEVFN bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }
// clamp to [0, hi]: the lower bound is the inline constant 0 (the three-register form above made the compiler materialise a VGPR 0, and a VGPR
// copy of a literal upper bound, in front of EVERY use: two v_mov_b32 per binned sample); `hi` in a vector / in a scalar register
DEVFN int med3_i32_0v(int x, int hi) { int r; asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(hi)); return r; }
DEVFN int med3_i32_0s(int x, int hi) { int r; asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi)); return r; }

// a / b without the IEEE special-case scaffolding (v_div_scale / v_div_fmas / v_div_fixup): reciprocal seed, two Newton steps,
// quotient and one residual correction -- 8 instructions instead of 11, the correctly rounded quotient except for rare last-bit
// cases.  For operands well inside the normal range (no denormals, no overflow in 1/b); inf and NaN still propagate.  Used only
// where the quotient feeds smooth arithmetic (never the z interpolation or a bin index).
DEVFN double chm_div(double a, double b) {
  double r = FM_RCP(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}

DEVFN double chm_exp(double x) {
  const double L2E = 1.44269504088896338700e+00, LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
  double n = __builtin_rint(x * L2E);
  double r = __builtin_fma(-n, LN2HI, x);
  r = __builtin_fma(-n, LN2LO, r);
  // e^r = 1 + r + r^2 q(r): q = degree-9 Chebyshev fit of (e^r - 1 - r)/r^2 on |r| <= ln2/2 (mpmath.chebyfit; |error| < 2e-17 of e^r) -- two
  // Horner steps fewer than the degree-13 Taylor sum at the same measured accuracy (0.65 against 0.63 ulp, scripts/check_fastmath.cpp)
  double p = 2.510038549551032e-08;
  p = FM_FMA(p, r, 2.7620088445409746e-07);
  p = FM_FMA(p, r, 2.7557268459997064e-06);
  p = FM_FMA(p, r, 2.4801521295954376e-05);
  p = FM_FMA(p, r, 0.00019841269863053618);
  p = FM_FMA(p, r, 0.0013888888917213717);
  p = FM_FMA(p, r, 0.008333333333330062);
  p = FM_FMA(p, r, 0.04166666666662413);
  p = FM_FMA(p, r, 0.16666666666666669);
  p = FM_FMA(p, r, 0.5000000000000001);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  int k = (int)n;
  double v = __builtin_ldexp(p, k);
  if (x > 709.782712893384) v = __builtin_inf();
  if (x < -745.1332191019412) v = 0.;
  return v;
}

// 10^y as exp(y ln 10) with the product formed in two pieces (ln 10 = hi + lo, the rounding of y * hi recovered by an fma): the argument
// of the exp carries < 1e-17 relative error, so the result is exp's own (0.63 ulp) -- ~35 instructions against ~300 of ocml's pow().
// For the logspace nodes of the z and mass tables (cosmo.py:43-46, mass.py:45): k_tables runs two blocks per draw, and the two pow()
// per thread were 5 us of its 30.
DEVFN double chm_pow10(double y) {
  const double LN10_HI = 2.302585092994046, LN10_LO = -2.1707562233822494e-16;
  const double a = y * LN10_HI;
  const double e = __builtin_fma(y, LN10_HI, -a) + y * LN10_LO;
  const double v = chm_exp(a);
  return __builtin_fma(v, e, v);
}

// exp(x) for |x| <= 708: chm_exp without the overflow / underflow handling (NaN propagates), the same polynomial (its coefficients then
// occupy one set of registers in a kernel that uses both: a shorter degree-9 fit spilled 9 registers in k_full_kde).  For arguments that
// are bounded by construction (k_full_kde).
DEVFN double chm_exp_nb(double x) {
  const double L2E = 1.44269504088896338700e+00, LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
  double n = __builtin_rint(x * L2E);
  double r = __builtin_fma(-n, LN2HI, x);
  r = __builtin_fma(-n, LN2LO, r);
  double p = 2.510038549551032e-08;
  p = FM_FMA(p, r, 2.7620088445409746e-07);
  p = FM_FMA(p, r, 2.7557268459997064e-06);
  p = FM_FMA(p, r, 2.4801521295954376e-05);
  p = FM_FMA(p, r, 0.00019841269863053618);
  p = FM_FMA(p, r, 0.0013888888917213717);
  p = FM_FMA(p, r, 0.008333333333330062);
  p = FM_FMA(p, r, 0.04166666666662413);
  p = FM_FMA(p, r, 0.16666666666666669);
  p = FM_FMA(p, r, 0.5000000000000001);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)n);
}

// exp(x) for the smoothing denominators 1 + e^x of the mass models (mass.py:255-264), x = delta_m/a + delta_m/b unbounded in both
// directions: the argument is clamped to [-745.2, 700] (v_max / v_min: a NaN x comes out as a bound, the callers test their windows
// separately) and goes through chm_exp_nb -- every value is finite.  The caller restores the reference's exact zero of the smoothing
// factor for x > 745.14 (where its exp(-x) underflows); in between, 1/(1 + e^700) = 1e-304 stands for e^-x in (1e-324, 1e-304).
DEVFN double chm_exp_clamped(double x) {
  double lo, hi;
  asm("v_max_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(-745.2));
  asm("v_min_f64 %0, %1, %2" : "=v"(hi) : "v"(lo), "v"(700.));
  return chm_exp_nb(hi);
}

// [r3] exp(x) through a table of 2^(j/256) (256 doubles in LDS, exp_table_fill): x = (256 k + j) ln2/256 + r, |r| <= ln2/512, e^r by its degree-4
// Taylor sum (remainder < 4e-17) -- 13 VALU instructions against the 17 of chm_exp_nb (the table read is an LDS instruction).  Same contract
// as chm_exp_nb: no range checks (a huge |x| ends in v_ldexp_f64's 0 / inf, NaN propagates through the polynomial); <= 1.8 ulp (4.0e-16 on 2e7 arguments, scripts/check_fastmath.cpp).
#define CHM_EXPTAB_N 256
DEVFN double chm_exp_tab(double x, const double* T) {
  const double SC = 3.69329930467574632e+02;               // 256 / ln 2
  const double L_HI = 6.93147180369123816490e-01 / 256., L_LO = 1.90821492927058770002e-10 / 256.;      // ln2/256 in two pieces (exact scalings of fdlibm's)
  double n = __builtin_rint(x * SC);
  double r = __builtin_fma(-n, L_HI, x);
  r = __builtin_fma(-n, L_LO, r);
  const int ni = (int)n;
  const double tj = T[ni & (CHM_EXPTAB_N - 1)];
  double p = FM_FMA(r, 4.16666666666666644e-02, 1.66666666666666657e-01);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(tj * p, ni >> 8);
}
DEVFN double chm_exp_tab_clamped(double x, const double* T) {      // chm_exp_clamped with the table
  double lo, hi;
  asm("v_max_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(-745.2));
  asm("v_min_f64 %0, %1, %2" : "=v"(hi) : "v"(lo), "v"(700.));
  return chm_exp_tab(hi, T);
}
// entry j of the table: 2^(j/256) = exp(j ln2/256) with the argument formed in two pieces (as chm_pow10): chm_exp's own 0.63 ulp
DEVFN double exp_table_entry(int j) {
  const double C_HI = 6.93147180369123816490e-01 / 256., C_LO = 1.90821492927058770002e-10 / 256.;
  const double a = (double)j * C_HI;                       // exact: j < 2^8, C_HI ends in 20 zero bits
  const double v = chm_exp(a);
  return __builtin_fma(v, (double)j * C_LO, v);
}
// the two forms of exp the mass model is instantiated with
// (pw: x^y from log x, the pow_l of chm_models.h -- with the range tests of chm_exp in the polynomial form)
struct ExpPoly { DEVFN double nb(double x) const { return chm_exp_nb(x); } DEVFN double clamped(double x) const { return chm_exp_clamped(x); }
                 DEVFN double pw(double lx, double y) const { return chm_exp(y * lx); } };
struct ExpTab { const double* T; DEVFN double nb(double x) const { return chm_exp_tab(x, T); } DEVFN double clamped(double x) const { return chm_exp_tab_clamped(x, T); }
                DEVFN double pw(double lx, double y) const { return chm_exp_tab(y * lx, T); } };

// 1 / b for b well inside the normal range: reciprocal seed + two Newton steps (~1 ulp; chm_div(1, b) spends three more instructions on
// the correctly rounded quotient).  For factors of smooth arithmetic: 1/(1 + z) of the det -> src conversion.
DEVFN double chm_rcp(double b) {
  double r = FM_RCP(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  return __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
}

// log(x) for finite x > 0 (NaN propagates); fdlibm e_log.c scheme, < 1 ulp
DEVFN double chm_log_pos(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
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
  double t1 = w * FM_FMA(w, FM_FMA(w, Lg6, Lg4), Lg2);
  double t2 = z * FM_FMA(w, FM_FMA(w, FM_FMA(w, Lg7, Lg5), Lg3), Lg1);
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
