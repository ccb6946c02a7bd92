"""Plug-in population models (the reference's open-class extension points, SURVEY 8(b) "plugin fallback").

In CHIMERA a user adds a mass, rate or completeness model by writing a new parameter struct and new ``plum`` overloads of
``p_m1m2`` (mass.py:334-345), ``merger_rate`` (rate.py:96-122) or ``p_bkg`` / ``fR`` (completeness.py:43-67).  Python functions
cannot run inside a kernel, so here such a model is an object WITHOUT the ``_pack`` method of the built-in ones that provides
the function itself:

* mass model           ``model.p_m1m2(m1src, m2src) -> array``      (normalised joint pdf, as mass.py:334-341 returns)
* rate model           ``model.merger_rate(z) -> array``
* completeness model   ``compl.P_compl(zgrids)``, ``compl.fR(cosmo)``, ``compl.p_bkg(cosmo, z_or_theta_src)``; anything that is not
                       ``dVdz_completeness`` (attribute ``builtin = True``) counts as a plug-in
* cosmology            a subclass of ``cosmo.plugin_cosmology`` with ``dL_at_z``, ``ddLdz_at_z``, ``dVcdz_at_z``, ``Vc_at_z`` (Gpc): the host
                       tabulates dL on ``z_grid_interp`` (the table of ``z_from_dGW``), the Jacobian on the event grids and per
                       injection, and p_bkg / fR through the completeness model (cosmo.py:122-264)

plus ``keys`` / ``as_dict`` / ``update(**lambdas)`` like every model (subclass ``base_struct``).  The host evaluates these on
the source-frame quantities of every draw -- ``z = z_from_dGW(cosmo, dL)`` comes from the device, so it is the kernel's own
``z`` -- and hands the values to ``chm_eval_tabulated`` (include/chimera_hip.h: ``chm_tab``); everything else (KDE, integrand,
reductions, the other model pieces) stays in the HIP kernels.  Slow path: the tables travel host -> device on every call.
"""
import numpy as np
from .. import _lib
from ..data import theta_src


def is_plugin_model(model):
  return model is not None and not hasattr(model, '_pack')


def is_plugin_completeness(gal_cat):
  compl = getattr(gal_cat, 'completeness', None)
  if compl is not None:
    return not getattr(compl, 'builtin', False)
  return bool(getattr(gal_cat, 'p_bkg_is_plugin', False))          # empty_catalog(p_bkg=callable)


def population_plugins(pop):
  """(mass, rate, completeness, cosmology) flags of a population.  A plug-in cosmology makes the background term a host quantity too
  (p_bkg = dVc/dz of THAT cosmology), whatever the completeness model."""
  cosmo = is_plugin_model(pop.cosmo)
  return is_plugin_model(pop.mass), is_plugin_model(pop.rate), is_plugin_completeness(pop.gal_cat) or cosmo, cosmo


def _bkg(gal_cat, cosmo, z):
  compl = getattr(gal_cat, 'completeness', None)
  f = compl.p_bkg if compl is not None else gal_cat.p_bkg
  return np.ascontiguousarray(f(cosmo, z), dtype=np.float64)


def build_tab(pops, plugins, ev=None, inj=None):
  """chm_tab for a list of population draws.

  ev  = dict(dL, m1det, m2det (E_loc,S), z_grids (E_loc,Z)) of this shard, or None (no likelihood in the call)
  inj = dict(dL, m1det, m2det (I_loc,))                      of this shard, or None (no selection function in the call)
  Returns (chm_tab, keepalive): the arrays must outlive the call.
  """
  from .cosmo import z_from_dGW, dL_at_z, ddLdz_at_z
  want_m, want_r, want_b = plugins[:3]
  want_c = len(plugins) > 3 and plugins[3]
  nb = len(pops)
  tab, keep = _lib.chm_tab(), []

  def put(field, arrs):
    a = np.ascontiguousarray(np.stack(arrs), dtype=np.float64)
    keep.append(a)
    setattr(tab, field, _lib.dptr(a))

  if ev is not None:
    if want_m:
      out = []
      for p in pops:
        z = z_from_dGW(p.cosmo, ev['dL'])
        out.append(np.asarray(p.mass.p_m1m2(ev['m1det'] / (1. + z), ev['m2det'] / (1. + z)), dtype=np.float64))
      put('pm_samples', out)
    if want_r:
      put('rate_grid', [np.broadcast_to(np.asarray(p.rate.merger_rate(ev['z_grids']), dtype=np.float64), ev['z_grids'].shape) for p in pops])
    if want_b:
      put('bkg_grid', [np.broadcast_to(_bkg(p.gal_cat, p.cosmo, ev['z_grids']), ev['z_grids'].shape) for p in pops])
    if want_c:                                      # likelihood.py:272: ddLdz_at_z(cosmo, z_grids) (1 + z_grids)^2
      put('jac_grid', [np.asarray(ddLdz_at_z(p.cosmo, ev['z_grids']), dtype=np.float64) * (1. + ev['z_grids'])**2 for p in pops])
  if want_c:                                        # the table of z_from_dGW (cosmo.py:260-264)
    put('z_table', [np.asarray(p.cosmo.z_grid_interp, dtype=np.float64) for p in pops])
    put('dL_table', [dL_at_z(p.cosmo, np.asarray(p.cosmo.z_grid_interp, dtype=np.float64)) for p in pops])
  if inj is not None and (want_m or want_r or want_b):
    pm, rt, bk, jc = [], [], [], []
    for p in pops:
      z = z_from_dGW(p.cosmo, inj['dL'])
      if want_m:
        pm.append(np.asarray(p.mass.p_m1m2(inj['m1det'] / (1. + z), inj['m2det'] / (1. + z)), dtype=np.float64))
      if want_r:
        rt.append(np.asarray(p.rate.merger_rate(z), dtype=np.float64))
      if want_b:                                   # pop_wrapper.py:106: p_bkg(cosmo, theta_src) with the original distances
        src = theta_src(m1src=inj['m1det'] / (1. + z), m2src=inj['m2det'] / (1. + z), z=z, original_distances=inj['dL'])
        bk.append(_bkg(p.gal_cat, p.cosmo, src))
        if want_c:                                 # pop_wrapper.py:109: |ddLdz_at_z(cosmo, theta_src)| (1 + z)^2
          jc.append(np.abs(np.asarray(ddLdz_at_z(p.cosmo, src), dtype=np.float64)) * (1. + z)**2)
    if want_m:
      put('pm_inj', pm)
    if want_r:
      put('rate_inj', rt)
    if want_b:
      put('bkg_inj', bk)
    if want_c:
      put('jac_inj', jc)
  if want_b:
    compl = getattr(pops[0].gal_cat, 'completeness', None)
    if compl is not None:                          # (a plug-in cosmology reaches fR through Vc_at_z's dispatch)
      fr = np.ascontiguousarray([float(np.asarray(p.gal_cat.completeness.fR(p.cosmo))) for p in pops], dtype=np.float64)
      keep.append(fr)
      tab.fR = _lib.dptr(fr)
  return tab, keep
