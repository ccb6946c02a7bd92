"""Shared builders for the tests: the same synthetic inputs fed to the oracle (oracle/) and to the product
(chimera_amd/, HIP through the C ABI)."""
import numpy as np
from chimera_amd import synth
from oracle import chimera_oracle as O

PE_FIELDS = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix',
             'gw_loc2d_pdf', 'pixels_pe_opt_nside')
INJ_FIELDS = ('m1det', 'm2det', 'dL', 'p_draw')

COSMO_KW = dict(H0=70., Om0=0.25, z_max=5.)


def small_config(E=6, S=256, P=4, Z=64, I=2000, seed=7, ragged=True, pixelated=True):
  name = 'C2' if pixelated else 'C1'
  kw = dict(E=E, I=I, S=S, Z=Z)
  if pixelated:
    kw['P'] = P
  cfg, ev, inj = synth.make_config(name, seed=seed, ragged=ragged and pixelated, **kw)
  return cfg, ev, inj


def _models(mod, cosmo='flrw', mass='plp', rate='madau_dickinson', cosmo_kw=None, mass_kw=None, rate_kw=None):
  ckw = dict(COSMO_KW)
  ckw.update(cosmo_kw or {})
  c = getattr(mod['cosmo'], cosmo)(**ckw)
  m = getattr(mod['mass'], mass)(**(mass_kw or {}))
  r = getattr(mod['rate'], rate)(**(rate_kw or {}))
  return c, m, r


def build_oracle(ev, inj, pixelated=True, kind='marginalized', models=None, like_kw=None, pop_kw=None, N_eff=5.):
  mod = dict(cosmo=O, mass=O, rate=O)
  c, m, r = _models(mod, **(models or {}))
  fields = {k: ev[k] for k in PE_FIELDS if k in ev and (pixelated or k in ('m1det', 'm2det', 'dL', 'pe_prior'))}
  th = O.theta_pe_det(**fields)
  gc = O.pixelated_catalog(O.dVdz_completeness(), ev['p_cat'], ev['z_grids'], ev['neff_pixels']) if pixelated else None
  pop = O.population(c, m, r, gal_cat=gc, **(pop_kw or {}))
  sel = O.selection_function(O.theta_inj_det(**{k: inj[k] for k in INJ_FIELDS}), inj['N_inj'], N_eff=N_eff)
  like = O.hyperlikelihood(th, ev['z_grids'], pop, sel, kind_p_gw3d=kind if pixelated else None, **(like_kw or {}))
  return like, pop, sel


def build_product(ev, inj, pixelated=True, kind='marginalized', models=None, like_kw=None, pop_kw=None, N_eff=5.,
                  comm=None):
  import chimera_amd as CH
  from chimera_amd.catalog import dVdz_completeness, pixelated_catalog
  mod = dict(cosmo=CH.cosmo, mass=CH.mass, rate=CH.rate)
  c, m, r = _models(mod, **(models or {}))
  fields = {k: ev[k] for k in PE_FIELDS if k in ev and (pixelated or k in ('m1det', 'm2det', 'dL', 'pe_prior'))}
  th = CH.data.theta_pe_det(**fields)
  gc = None
  if pixelated:
    gc = pixelated_catalog(dVdz_completeness(), p_cat=ev['p_cat'], z_grids=ev['z_grids'], neff_pixels=ev['neff_pixels'])
  pop = CH.population(c, m, r, gal_cat=gc, **(pop_kw or {}))
  sel = CH.selection_function(CH.data.theta_inj_det(**{k: inj[k] for k in INJ_FIELDS}), inj['N_inj'], N_eff=N_eff, comm=comm)
  like = CH.hyperlikelihood(th, ev['z_grids'], pop, sel, kind_p_gw3d=kind if pixelated else None, comm=comm,
                            **(like_kw or {}))
  return like, pop, sel


def neginf_class(x):
  """Values <= -1e300 are '-infinity-class' (-inf and -1.797e308 both mean L_i == 0 or NaN)."""
  return np.asarray(x) <= -1e300


def assert_loglike_close(got, ref, rtol, atol=0.):
  got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
  a, b = neginf_class(got), neginf_class(ref)
  assert np.array_equal(a, b), f"-inf-class mismatch: got {got}, ref {ref}"
  np.testing.assert_allclose(got[~a], ref[~b], rtol=rtol, atol=atol)


# ----------------------------------------------------------------------------------------------------------
# golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py)
# ----------------------------------------------------------------------------------------------------------
import os

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
  with np.load(os.path.join(GOLDEN_DIR, name + '.npz')) as d:
    ev = {k[3:]: d[k] for k in d.files if k.startswith('ev_')}
    inj = {k[4:]: d[k] for k in d.files if k.startswith('inj_')}
    inj['N_inj'] = float(d['N_inj'])
    exp = {k: d[k] for k in ('log_like_evs', 'log_num', 'log_Nexp', 'log_hyper', 'p_gw')}
  return ev, inj, exp


# ----------------------------------------------------------------------------------------------------------
# host mirror of the device-side combination (k_combine), used by the CPU sharding tests
# ----------------------------------------------------------------------------------------------------------
def shard_partials_oracle(like_o, lam, e0, e1, i0, i1):
  """[sum_i nan_to_num(log L_i), nansum dN, sum dN^2] of events [e0,e1) and injections [i0,i1) (oracle)."""
  pop = like_o.population.update(**lam)
  with np.errstate(all='ignore'):
    ll = O.nan_to_num_neginf(np.log(like_o.compute_numlike_evs(pop)))[e0:e1]
    dN = like_o.selection_function.dN(pop)[i0:i1]
    return np.array([np.sum(ll), np.nansum(dN), np.sum(dN**2)])


def combine_partials(partials, E_total, pop, N_inj, N_eff):
  """selection_function.py:38-47 + likelihood.py:298-300,313-316 on the all-reduced partial sums."""
  log_num, s1, s2 = partials
  with np.errstate(all='ignore'):
    xi = s1 / N_inj
    Nexp = pop.Tobs * xi
    if N_eff is not None:
      var = s2 / N_inj**2 - xi**2 / N_inj
      if xi**2 / var < N_eff:
        Nexp = 0.
    if not pop.scale_free:
      log_num = log_num + E_total * np.log(pop.R0 * pop.Tobs)
      return log_num - Nexp
    return log_num - E_total * np.log(Nexp)
