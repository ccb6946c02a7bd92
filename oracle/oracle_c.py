"""
oracle/oracle_c.py -- ctypes front end of oracle/chimera_oracle_c.c (the plain-C / OpenMP restatement of the path).

TEST INFRASTRUCTURE ONLY (tests/, bench.py's ``cpu_baseline`` leg).  It takes the model objects and data containers of
``oracle/chimera_oracle.py`` and hands plain arrays to the C library; nothing here imports ``chimera_amd``.
"""
import ctypes as C
import os
import subprocess
import numpy as np
from . import chimera_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, 'libchimera_oracle_c.so')
_lib = None

c_dp = C.POINTER(C.c_double)
c_lp = C.POINTER(C.c_int64)


class chm_params(C.Structure):
  """Mirror of ``chm_params`` in include/chimera_hip.h (plain data)."""
  _fields_ = [('cosmo_model', C.c_int32), ('mass_model', C.c_int32), ('rate_model', C.c_int32),
              ('z_grid_res', C.c_int32), ('mass_grid_res', C.c_int32), ('scale_free', C.c_int32),
              ('has_catalog', C.c_int32), ('_pad', C.c_int32),
              ('z_max', C.c_double), ('cosmo', C.c_double * 8), ('mass', C.c_double * 8), ('rate', C.c_double * 4),
              ('R0', C.c_double), ('Tobs', C.c_double), ('compl_z0', C.c_double), ('compl_z1', C.c_double)]


def build(force=False):
  src = os.path.join(HERE, 'chimera_oracle_c.c')
  if force or not os.path.exists(LIBPATH) or os.path.getmtime(LIBPATH) < os.path.getmtime(src):
    subprocess.check_call(['make', '-s', '-C', HERE])
  return LIBPATH


def lib():
  global _lib
  if _lib is None:
    build()
    L = C.CDLL(LIBPATH)
    L.orc_tables.argtypes = [C.POINTER(chm_params)] + [c_dp] * 6
    L.orc_tables.restype = C.c_int
    L.orc_numlike_marg.argtypes = [C.POINTER(chm_params), C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_lp, c_lp,
                                   c_dp, c_dp, c_dp, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, c_dp]
    L.orc_numlike_marg.restype = C.c_int
    L.orc_numlike_1d.argtypes = [C.POINTER(chm_params), C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp,
                                 C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, c_dp]
    L.orc_numlike_1d.restype = C.c_int
    L.orc_numlike_full.argtypes = [C.POINTER(chm_params), C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp,
                                   C.POINTER(C.c_int32), c_dp, c_dp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, c_dp]
    L.orc_numlike_full.restype = C.c_int
    L.orc_nexp.argtypes = [C.POINTER(chm_params), C.c_longlong, c_dp, c_dp, c_dp, c_dp, C.c_double, C.c_double, C.c_int, c_dp]
    L.orc_nexp.restype = C.c_int
    L.orc_max_threads.restype = C.c_int
    _lib = L
  return _lib


def _f64(a):
  return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
  return a.ctypes.data_as(c_dp)


def pack_params(pop):
  """One hyper-parameter draw (an oracle ``population``) -> chm_params."""
  p = chm_params()
  co, ma, ra = pop.cosmo, pop.mass, pop.rate
  p.cosmo_model = 1 if isinstance(co, O.mg_flrw) else 0
  p.mass_model = {O.tpl: 0, O.bpl: 1, O.plp: 2}[type(ma)]
  p.rate_model = {O.power_law: 0, O.madau_dickinson: 1, O.trunc_power_law: 2, O.trunc_madau_dickinson: 3}[type(ra)]
  p.z_grid_res, p.mass_grid_res = int(co.z_grid_res), int(ma.grid_res)
  p.scale_free = int(bool(pop.scale_free))
  p.has_catalog = int(isinstance(pop.gal_cat, O.pixelated_catalog))
  p.z_max = float(co.z_max)
  cos = [co.H0, co.Om0, co.Ok0, co.Or0, co.w0, co.wa, getattr(co, 'Xi0', 1.), getattr(co, 'n', 0.)]
  if isinstance(ma, O.tpl):
    mas = [ma.m_low, ma.m_high, ma.alpha, ma.beta]
  elif isinstance(ma, O.bpl):
    mas = [ma.m_low, ma.m_high, ma.alpha_1, ma.alpha_2, ma.beta, ma.delta_m, ma.break_fraction]
  else:
    mas = [ma.m_low, ma.m_high, ma.lambda_peak, ma.alpha, ma.beta, ma.delta_m, ma.mu_g, ma.sigma_g]
  ras = [ra.gamma, getattr(ra, 'kappa', 0.), getattr(ra, 'zp', 0.), getattr(ra, 'zmax', 0.)]
  for i in range(8):
    p.cosmo[i] = float(cos[i])
    p.mass[i] = float(mas[i]) if i < len(mas) else 0.
  for i in range(4):
    p.rate[i] = float(ras[i])
  p.R0, p.Tobs = float(pop.R0), float(pop.Tobs)
  zr = pop.gal_cat.completeness.z_range if p.has_catalog else (0.073, 1.3)
  p.compl_z0, p.compl_z1 = float(zr[0]), float(zr[1])
  return p


def tables(pop):
  p = pack_params(pop)
  Tc, Tm = p.z_grid_res, p.mass_grid_res
  zt, It, dLt = np.zeros(Tc), np.zeros(Tc), np.zeros(Tc)
  mg, cdf, sc = np.zeros(Tm), np.zeros(Tm), np.zeros(2)
  if lib().orc_tables(C.byref(p), _dp(zt), _dp(It), _dp(dLt), _dp(mg), _dp(cdf), _dp(sc)):
    raise MemoryError('orc_tables')
  return dict(zt=zt, It=It, dLt=dLt, m_grid=mg, cdf_m2=cdf, norm_p_m1=sc[0], fR=sc[1])


def numlike_marg(like, pop, nthreads=0):
  """L_i of every event for an oracle ``hyperlikelihood`` configured with kind_p_gw3d='marginalized'."""
  assert like.pixelated and like.kind_p_gw3d == 'marginalized'
  th = like.theta_gw_det
  gc = pop.gal_cat
  p = pack_params(pop)
  dL, m1, m2, pr = _f64(th.dL), _f64(th.m1det), _f64(th.m2det), _f64(th.pe_prior)
  E, S = dL.shape
  pe_pix = np.ascontiguousarray(th.pixels_pe_opt_nside, dtype=np.int64)
  pixels = np.ascontiguousarray(th.pixels_opt_nsides, dtype=np.int64)
  P = pixels.shape[1]
  zg, pc, gw = _f64(like.z_grids), _f64(gc.p_cat), _f64(th.gw_loc2d_pdf)
  Z = zg.shape[1]
  bw = like.bw_method
  bw_method, bw_scalar = (0, 0.) if bw in (None, 'scott') else ((1, 0.) if bw == 'silverman' else (2, float(bw)))
  cut = float('nan') if like.cut_grid is None else float(like.cut_grid)
  out = np.zeros(E)
  rc = lib().orc_numlike_marg(C.byref(p), E, S, P, Z, _dp(dL), _dp(m1), _dp(m2), _dp(pr), pe_pix.ctypes.data_as(c_lp),
                              pixels.ctypes.data_as(c_lp), _dp(zg), _dp(pc), _dp(gw), cut, int(bool(like.binning)),
                              int(like.num_bins), float(like.pe_neff), bw_method, bw_scalar, int(nthreads), _dp(out))
  if rc:
    raise MemoryError('orc_numlike_marg')
  return out


def numlike_1d(like, pop, nthreads=0):
  """L_i of every event for an oracle ``hyperlikelihood`` in the 1-D (no catalogue) or 'approximate' mode."""
  assert (not like.pixelated) or like.kind_p_gw3d == 'approximate'
  th = like.theta_gw_det
  p = pack_params(pop)
  dL, m1, m2, pr = _f64(th.dL), _f64(th.m1det), _f64(th.m2det), _f64(th.pe_prior)
  E, S = dL.shape
  zg = _f64(like.z_grids)
  Z = zg.shape[1]
  if like.pixelated:
    pc, gw = _f64(pop.gal_cat.p_cat), _f64(th.gw_loc2d_pdf)
    P = pc.shape[1]
    pcp, gwp = _dp(pc), _dp(gw)
  else:
    P, pcp, gwp = 0, None, None
  bw = like.bw_method
  bw_method, bw_scalar = (0, 0.) if bw in (None, 'scott') else ((1, 0.) if bw == 'silverman' else (2, float(bw)))
  cut = float('nan') if like.cut_grid is None else float(like.cut_grid)
  out = np.zeros(E)
  rc = lib().orc_numlike_1d(C.byref(p), E, S, P, Z, _dp(dL), _dp(m1), _dp(m2), _dp(pr), _dp(zg), pcp, gwp, cut,
                            int(bool(like.binning)), int(like.num_bins), float(like.pe_neff), bw_method, bw_scalar,
                            1 if like.kernel == 'gauss' else 0, int(nthreads), _dp(out))
  if rc:
    raise MemoryError('orc_numlike_1d')
  return out


def numlike_full(like, pop, nthreads=0):
  """L_i of every event for an oracle ``hyperlikelihood`` configured with kind_p_gw3d='full' (likelihood.py:211-260)."""
  assert like.pixelated and like.kind_p_gw3d == 'full'
  th = like.theta_gw_det
  p = pack_params(pop)
  dL, m1, m2, pr = _f64(th.dL), _f64(th.m1det), _f64(th.m2det), _f64(th.pe_prior)
  ra, dec, rap, decp = _f64(th.ra), _f64(th.dec), _f64(th.ra_pix), _f64(th.dec_pix)
  E, S = dL.shape
  zg, pc = _f64(like.z_grids), _f64(pop.gal_cat.p_cat)
  P, Z = pc.shape[1], zg.shape[1]
  npx = np.ascontiguousarray(like.neff_pixels, dtype=np.int32)
  bw = like.bw_method
  bw_method, bw_scalar = (0, 0.) if bw in (None, 'scott') else ((1, 0.) if bw == 'silverman' else (2, float(bw)))
  out = np.zeros(E)
  rc = lib().orc_numlike_full(C.byref(p), E, S, P, Z, _dp(dL), _dp(m1), _dp(m2), _dp(pr), _dp(ra), _dp(dec), _dp(rap), _dp(decp),
                              npx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(zg), _dp(pc), float(like.cut_grid), float(like.pe_neff),
                              bw_method, bw_scalar, int(nthreads), _dp(out))
  if rc:
    raise MemoryError('orc_numlike_full')
  return out


def numlike(like, pop, nthreads=0):
  if like.pixelated and like.kind_p_gw3d == 'marginalized':
    return numlike_marg(like, pop, nthreads)
  if like.pixelated and like.kind_p_gw3d == 'full':
    return numlike_full(like, pop, nthreads)
  return numlike_1d(like, pop, nthreads)


def n_exp(sel, pop, nthreads=0):
  """(N_exp, xi, n_eff) of an oracle ``selection_function``."""
  th = sel.theta_inj_det
  p = pack_params(pop)
  dL, m1, m2, pd = _f64(th.dL), _f64(th.m1det), _f64(th.m2det), _f64(th.p_draw)
  out = np.zeros(3)
  neff = float('nan') if sel.N_eff is None else float(sel.N_eff)
  if lib().orc_nexp(C.byref(p), dL.size, _dp(dL), _dp(m1), _dp(m2), _dp(pd), float(sel.N_inj), neff, int(nthreads), _dp(out)):
    raise MemoryError('orc_nexp')
  return out


def compute_all(like, lam, nthreads=0):
  """(log_like_evs, log_like_num, log N_exp, log_hyper) as hyperlikelihood.compute_all (likelihood.py:326-338)."""
  pop = like.population.update(**lam)
  with np.errstate(all='ignore'):
    ll = O.nan_to_num_neginf(np.log(numlike(like, pop, nthreads)))
    log_num = np.sum(ll)
    Nexp = n_exp(like.selection_function, pop, nthreads)[0]
    if not pop.scale_free:
      log_num = log_num + like.nevents * np.log(pop.R0 * pop.Tobs)
      log_hyper = log_num - Nexp
    else:
      log_hyper = log_num - like.nevents * np.log(Nexp)
    return ll, log_num, np.log(Nexp), log_hyper


def max_threads():
  return int(lib().orc_max_threads())
