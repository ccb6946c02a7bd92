"""Selection function (reference: CHIMERA/selection_function.py:10-53).

``N_exp(pop)`` = Tobs * nansum(dN/dtheta / p_draw) / N_inj with the N_eff guard, evaluated by the HIP kernels
``k_selection`` + ``k_reduce`` + ``k_combine`` (chimera_amd/csrc/chm_kernels.h) over the detected injections resident
in HBM.  With ``comm=`` the injections are sharded across ranks (CHIMERA/parallel.py:68-73).
"""
import ctypes as C
import numpy as np
from . import _lib
from .data import theta_inj_det
from .parallel import chunk_bounds


class selection_function(object):
  def __init__(self, theta_inj_det, N_inj, N_eff=5., comm=None, device=None):
    self.theta_inj_det = theta_inj_det
    self.N_inj = N_inj
    self.N_eff = N_eff
    self.comm = comm
    self.device = (comm.device if comm is not None else _lib.default_device()) if device is None else device
    self._h = None
    self._options = {}

  # -- device handle -----------------------------------------------------------------------------------
  def _handle(self):
    if self._h is not None:
      return self._h
    th = self.theta_inj_det
    arrs = [_lib.as_f64(getattr(th, k)).ravel() for k in ('dL', 'm1det', 'm2det', 'p_draw')]
    n = arrs[0].size
    if any(a.size != n for a in arrs):
      raise ValueError("theta_inj_det: dL, m1det, m2det, p_draw must have the same length")
    d = _lib.chm_sel_desc()
    d.I = n
    if self.comm is not None and self.comm.nranks > 1:
      d.inj_begin, d.inj_end = chunk_bounds(n, self.comm.nranks, self.comm.rank)
    else:
      d.inj_begin, d.inj_end = 0, n
    d.dL, d.m1det, d.m2det, d.p_draw = (_lib.dptr(a) for a in arrs)
    d.N_inj = float(self.N_inj)
    d.N_eff = float('nan') if self.N_eff is None else float(self.N_eff)
    d.device = self.device
    h = C.c_void_p()
    _lib.check(_lib.lib().chm_sel_create(C.byref(d), C.byref(h)))
    self._h = h
    for name, value in self._options.items():
      _lib.check(_lib.lib().chm_sel_set_option(h, _lib.OPTION[name], int(value)))
    return h

  def set_option(self, name, value):
    """Evaluation option of the device handle (``CHM_OPT_*``, see ``hyperlikelihood.set_option``).  Returns ``self``."""
    if name not in _lib.OPTION:
      raise ValueError(f"selection_function.set_option: unknown option {name!r}")
    if self._h is not None:
      _lib.check(_lib.lib().chm_sel_set_option(self._h, _lib.OPTION[name], int(value)))
    self._options = dict(self._options, **{name: int(value)})
    return self

  def close(self):
    if self._h is not None:
      _lib.lib().chm_sel_destroy(self._h)
      self._h = None

  def lane(self, comm=None):
    """A second evaluation lane on the injections already resident in HBM (``chm_sel_clone``); see ``hyperlikelihood.lane``."""
    import copy
    if self.comm is not None and comm is None:
      raise ValueError("selection_function.lane: a lane of a sharded selection function needs a communicator of its own (comm=)")
    h = self._handle()
    new = copy.copy(self)
    new.comm = comm
    new._options = dict(self._options)
    new._h = C.c_void_p()
    _lib.check(_lib.lib().chm_sel_clone(h, C.byref(new._h)))
    return new

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  def _inj_shard(self):
    """This rank's injections (dL, m1det, m2det) -- what a plug-in model is evaluated on (population/plugins.py)."""
    th = self.theta_inj_det
    n = np.size(th.dL)
    i0, i1 = chunk_bounds(n, self.comm.nranks, self.comm.rank) if (self.comm is not None and self.comm.nranks > 1) else (0, n)
    return {k: _lib.as_f64(getattr(th, k)).ravel()[i0:i1] for k in ('dL', 'm1det', 'm2det')}

  # -- reference surface -------------------------------------------------------------------------------
  def N_exp(self, pop_lambdas):
    """selection_function.py:34-48."""
    p = pop_lambdas.to_params()
    from .population.plugins import population_plugins, build_tab
    plugins = population_plugins(pop_lambdas)
    tab, tab_keep = build_tab([pop_lambdas], plugins, inj=self._inj_shard()) if any(plugins) else (None, None)
    nexp = np.empty(1)
    out = _lib.chm_out()
    out.N_exp = _lib.dptr(nexp)
    comm_h = getattr(self.comm, 'handle', None) if self.comm is not None else None
    host_reduce = self.comm is not None and comm_h is None and self.comm.nranks > 1 and hasattr(self.comm, 'allreduce_sum')
    part = np.empty(3)
    if host_reduce:
      out.partials = _lib.dptr(part)
    if tab is not None:
      _lib.check(_lib.lib().chm_eval_tabulated(None, self._handle(), comm_h, C.byref(p), 1, 0, C.byref(tab), C.byref(out)))
    else:
      _lib.check(_lib.lib().chm_eval(None, self._handle(), comm_h, C.byref(p), 1, 0, C.byref(out)))
    if host_reduce:                                           # HostComm: reduce the two selection sums on the host
      from .parallel import combine_partials
      return combine_partials(self.comm.allreduce_sum(part), 0, self.N_inj, self.N_eff, bool(p.scale_free), p.R0, p.Tobs,
                              has_like=False, has_sel=True)[2]
    return nexp[0]

  def __call__(self, pop_lambdas):
    """selection_function.py:50-53."""
    return self.N_exp(pop_lambdas)
