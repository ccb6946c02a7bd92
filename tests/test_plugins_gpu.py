"""Plug-in population models (chimera_amd/population/plugins.py, chm_eval_tabulated): user-written mass / rate / completeness
models evaluated on the host, everything else in the kernels.  GPU tests against the NumPy oracle."""
import numpy as np
import pytest

from oracle import chimera_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _cfg():
  return H.small_config(E=6, S=400, P=4, Z=64, I=3000, seed=19, ragged=True)


def _base_struct():
  from chimera_amd.population._base import base_struct
  return base_struct


def test_plugin_mass_model_reproduces_the_builtin_one():
  """A user-written copy of power-law + peak (its pdf evaluated on the host by the oracle's formula) must give what the
  built-in device model gives, through hyperlikelihood, batch, compute_all and selection_function.N_exp."""
  import chimera_amd as CH
  base_struct = _base_struct()

  class my_plp(base_struct):
    name = 'my_plp'
    default = dict(O.plp.default)

    def p_m1m2(self, m1, m2):
      return O.p_m1m2(O.plp(**self.as_dict), m1, m2)

  cfg, ev, inj = _cfg()
  like_b, pop_b, sel_b = H.build_product(ev, inj)                                   # built-in plp
  like_o, _, _ = H.build_oracle(ev, inj)
  like_u, pop_u, sel_u = H.build_product(ev, inj)
  pop_u = CH.population(pop_b.cosmo, my_plp(), pop_b.rate, gal_cat=pop_b.gal_cat)
  sel_u = CH.selection_function(sel_b.theta_inj_det, N_inj=inj['N_inj'], N_eff=5.)
  like_u = CH.hyperlikelihood(like_b.theta_gw_det, ev['z_grids'], pop_u, sel_u, kind_p_gw3d='marginalized')
  assert like_u._plugins == (True, False, False, False)
  lams = [dict(H0=66., alpha=3.1), dict(H0=72., lambda_peak=0.08, mu_g=31.), dict(H0=80., beta=0.4, m_low=4.2)]
  ref = like_b.batch(lams)
  got = like_u.batch(lams)
  np.testing.assert_allclose(got, ref, rtol=0, atol=1e-8)
  for lam in lams[:2]:
    ro, ru = like_o.compute_all(**lam), like_u.compute_all(**lam)
    H.assert_loglike_close(ru[0], ro[0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ru[2], ro[2], rtol=1e-10)
  np.testing.assert_allclose(sel_u.N_exp(pop_u.update(H0=70.)), sel_b.N_exp(pop_b.update(H0=70.)), rtol=1e-12)
  np.testing.assert_allclose(like_u(H0=np.array([66., 72.])), like_b(H0=np.array([66., 72.])), rtol=0, atol=1e-8)


def test_plugin_rate_and_completeness_against_the_oracle(monkeypatch):
  """A rate law and a completeness model the library does not know, written once and used on both sides."""
  import chimera_amd as CH
  from chimera_amd.catalog import pixelated_catalog
  base_struct = _base_struct()

  def rate_formula(gamma, zc, z):
    return (1. + z)**gamma * np.exp(-z / zc)

  class damped_rate(base_struct):                  # product side
    name = 'damped_rate'
    default = dict(gamma=2.4, zc=1.7)

    def merger_rate(self, z):
      return rate_formula(self.gamma, self.zc, z)

  class damped_rate_o(O._Params):                  # oracle side
    name = 'damped_rate'
    default = dict(gamma=2.4, zc=1.7)

  orig_rate = O.merger_rate
  monkeypatch.setattr(O, 'merger_rate', lambda r, z: rate_formula(r.gamma, r.zc, np.asarray(z, dtype=np.float64))
                      if isinstance(r, damped_rate_o) else orig_rate(r, z))

  class soft_completeness(object):                 # duck-typed completeness: same class on both sides, cosmology functions injected
    def __init__(self, mod, z_half=0.9, width=0.15):
      self.mod, self.z_half, self.width = mod, z_half, width
      self.z_range = np.array([0., z_half])

    def P_compl(self, zgrids):
      return 1. / (1. + np.exp((np.asarray(zgrids) - self.z_half) / self.width))

    def fR(self, cosmo):
      return float(self.mod.Vc_at_z(cosmo, np.array([self.z_half]))[0]) * 0.8

    def p_bkg(self, cosmo, z, distances=None):
      if hasattr(z, 'original_distances'):         # theta_src of the injections (pop_wrapper.py:106)
        return self.mod.dVcdz_at_z(cosmo, np.asarray(z.z), np.asarray(z.original_distances)) * (1. + 0.1 * np.asarray(z.z))
      zz = np.asarray(z, dtype=np.float64)
      if distances is not None:
        return self.mod.dVcdz_at_z(cosmo, zz, distances) * (1. + 0.1 * zz)
      return self.mod.dVcdz_at_z(cosmo, zz) * (1. + 0.1 * zz)

  cfg, ev, inj = _cfg()
  # oracle
  co, mo = O.flrw(**H.COSMO_KW), O.plp()
  gc_o = O.pixelated_catalog(soft_completeness(O), ev['p_cat'], ev['z_grids'], ev['neff_pixels'])
  th_o = O.theta_pe_det(**{k: ev[k] for k in H.PE_FIELDS if k in ev})
  pop_o = O.population(co, mo, damped_rate_o(), gal_cat=gc_o)
  sel_o = O.selection_function(O.theta_inj_det(**{k: inj[k] for k in H.INJ_FIELDS}), inj['N_inj'], N_eff=5.)
  like_o = O.hyperlikelihood(th_o, ev['z_grids'], pop_o, sel_o, kind_p_gw3d='marginalized')
  # product
  cp, mp = CH.cosmo.flrw(**H.COSMO_KW), CH.mass.plp()
  gc_p = pixelated_catalog(soft_completeness(CH.cosmo), p_cat=ev['p_cat'], z_grids=ev['z_grids'], neff_pixels=ev['neff_pixels'])
  th_p = CH.data.theta_pe_det(**{k: ev[k] for k in H.PE_FIELDS if k in ev})
  pop_p = CH.population(cp, mp, damped_rate(), gal_cat=gc_p)
  sel_p = CH.selection_function(CH.data.theta_inj_det(**{k: inj[k] for k in H.INJ_FIELDS}), N_inj=inj['N_inj'], N_eff=5.)
  like_p = CH.hyperlikelihood(th_p, ev['z_grids'], pop_p, sel_p, kind_p_gw3d='marginalized')
  assert like_p._plugins == (False, True, True, False)
  for lam in (dict(H0=68.), dict(H0=75., gamma=1.9, zc=2.5, alpha=3.0)):
    with np.errstate(all='ignore'):
      ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
    H.assert_loglike_close(rp[0], ro[0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rp[2], ro[2], rtol=1e-10)
    np.testing.assert_allclose(rp[3], ro[3], rtol=0, atol=1e-7 * np.sqrt(cfg['E']))
  # the 1-D and approximate modes take the same tables
  like_pa = CH.hyperlikelihood(th_p, ev['z_grids'], pop_p, sel_p, kind_p_gw3d='approximate')
  like_oa = O.hyperlikelihood(th_o, ev['z_grids'], pop_o, sel_o, kind_p_gw3d='approximate')
  with np.errstate(all='ignore'):
    ro, rp = like_oa.compute_all(H0=71.), like_pa.compute_all(H0=71.)
  H.assert_loglike_close(rp[0], ro[0], rtol=1e-9, atol=1e-9)


def test_plugin_cosmology_reproduces_the_builtin_one():
  """A user-written cosmology (a flat LCDM copy whose distance functions run on the host: the oracle's restatement of cosmo.py:122-264)
  enters through the tabulated route -- the (dL, z) table of z_from_dGW, the Jacobian on the event grids and per injection, p_bkg / fR
  through the completeness model -- and must give what the built-in flrw gives: hyperlikelihood, batch, compute_all, N_exp, in the
  marginalized and the 1-D (no catalogue) configurations."""
  import chimera_amd as CH
  from chimera_amd.population.cosmo import plugin_cosmology

  class my_flat_lcdm(plugin_cosmology):
    name = 'my_flat_lcdm'
    default = {'H0': 70., 'Om0': 0.25, 'z_max': 5., 'z_grid_res': 1500}

    def _o(self):
      return O.flrw(H0=self.H0, Om0=self.Om0, z_max=self.z_max, z_grid_res=self.z_grid_res)

    def dL_at_z(self, z):
      return O.dL_at_z(self._o(), z)

    def ddLdz_at_z(self, z, distances=None):
      return O.ddLdz_at_z(self._o(), z, distances)

    def dVcdz_at_z(self, z, distances=None):
      return O.dVcdz_at_z(self._o(), z, distances)

    def Vc_at_z(self, z, distances=None):
      return O.Vc_at_z(self._o(), z, distances)

  cfg, ev, inj = _cfg()
  for pixelated in (True, False):
    if pixelated:
      evx = ev
    else:
      cfgx, evx, injx = H.small_config(E=5, S=300, Z=80, I=3000, seed=12, pixelated=False)
    injx = inj if pixelated else injx
    like_b, pop_b, sel_b = H.build_product(evx, injx, pixelated=pixelated)            # built-in flrw (H0 70, Om0 0.25, z_max 5)
    like_o, _, _ = H.build_oracle(evx, injx, pixelated=pixelated)
    pop_u = CH.population(my_flat_lcdm(), pop_b.mass, pop_b.rate, gal_cat=pop_b.gal_cat)
    sel_u = CH.selection_function(sel_b.theta_inj_det, N_inj=injx['N_inj'], N_eff=5.)
    like_u = CH.hyperlikelihood(like_b.theta_gw_det, evx['z_grids'], pop_u, sel_u, kind_p_gw3d='marginalized' if pixelated else None)
    assert like_u._plugins == (False, False, True, True)
    lams = [dict(H0=66.), dict(H0=74., Om0=0.31, alpha=3.0), dict(H0=81., gamma=2.2)]
    np.testing.assert_allclose(like_u.batch(lams), like_b.batch(lams), rtol=0, atol=1e-8)
    for lam in lams[:2]:
      ro, ru = like_o.compute_all(**lam), like_u.compute_all(**lam)
      H.assert_loglike_close(ru[0], ro[0], rtol=1e-9, atol=1e-9)
      np.testing.assert_allclose(ru[2], ro[2], rtol=1e-10)
      np.testing.assert_allclose(ru[3], ro[3], rtol=0, atol=1e-7 * np.sqrt(len(ro[0])))
    np.testing.assert_allclose(sel_u.N_exp(pop_u.update(H0=70.)), sel_b.N_exp(pop_b.update(H0=70.)), rtol=1e-11)
    # the free functions dispatch to the user's methods (the plum overloads of the reference)
    z = np.array([0.01, 0.3, 1.7])
    np.testing.assert_allclose(CH.cosmo.dL_at_z(pop_u.cosmo, z), CH.cosmo.dL_at_z(pop_b.cosmo, z), rtol=1e-12)
    np.testing.assert_allclose(CH.cosmo.z_from_dGW(pop_u.cosmo, injx['dL'][:50]), CH.cosmo.z_from_dGW(pop_b.cosmo, injx['dL'][:50]), rtol=1e-12)
