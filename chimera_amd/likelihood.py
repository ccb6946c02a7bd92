"""Hyper-likelihood (reference: CHIMERA/likelihood.py:14-338) on MI355X.

Same constructor, attributes and methods as the reference class.  The events, their z-grids and the catalogue term
are copied to HBM once (``chm_like_create``); every call evaluates the whole path -- tables, det->src conversion and
population weights, histogram + KDE, interpolation, integrand, trapezoid, pixel and event sums, selection function --
in HIP kernels behind ``chm_eval`` (include/chimera_hip.h).  There is no CPU path.
"""
import ctypes as C
from numbers import Number
import numpy as np
from . import _lib
from .utils.config import logger
from .parallel import chunk_bounds

_BW = {None: 0, 'scott': 0, 'silverman': 1}


def _pix_of_sample(pe_pix, pixels):
  """Position of each sample's HEALPix index in its event's pixel list (-1: in none of the event's pixels);
  the device-side form of ``pe_pix == pixels[i]`` (likelihood.py:179)."""
  pe_pix, pixels = np.asarray(pe_pix), np.asarray(pixels)
  E, P = pixels.shape
  out = np.full(pe_pix.shape, -1, dtype=np.int32)
  for e in range(E):
    valid = pixels[e] != -100
    ids = pixels[e][valid]
    pos = np.flatnonzero(valid)
    order = np.argsort(ids, kind='stable')
    sid = ids[order]
    if sid.size == 0:
      continue
    k = np.clip(np.searchsorted(sid, pe_pix[e]), 0, sid.size - 1)
    hit = sid[k] == pe_pix[e]
    out[e, hit] = pos[order][k[hit]]
  return out


def _vector_length(hyper_lambdas):
  """None for scalar hyper-parameters, else the common length of the array-valued ones."""
  n = None
  for k, v in hyper_lambdas.items():
    if np.ndim(v) > 0:
      m = np.asarray(v).size
      if n is not None and m != n:
        raise ValueError(f"hyper-parameter arrays must have the same length (got {n} and {m} for '{k}')")
      n = m
  return n


class hyperlikelihood(object):
  def __init__(self, theta_gw_det, z_grids, population, selection_function=None, kind_p_gw3d=None, kernel='epan',
               bw_method=None, cut_grid=2.0, binning=True, num_bins=200, pe_neff=2.0, comm=None, device=None, scheme='data'):
    self.theta_gw_det = theta_gw_det
    self.population = population
    self.z_grids = np.ascontiguousarray(z_grids, dtype=np.float64)
    self.selection_function = selection_function
    self.kind_p_gw3d = kind_p_gw3d
    self.kernel = kernel
    self.bw_method = bw_method
    self.cut_grid = cut_grid
    self.binning = binning
    self.num_bins = num_bins
    self.pe_neff = pe_neff
    self.comm = comm
    self.device = (comm.device if comm is not None else _lib.default_device()) if device is None else device
    # Parallel scheme of a likelihood with a communicator (the reference's MPIHyperLike, CHIMERA/parallel.py:13-160):
    #   'data'    events and injections are sharded, every rank evaluates every draw on its shard, one all-reduce of 3 doubles per draw
    #             (parallel.py:94-99, 366-376) -- for large catalogues;
    #   'params'  every rank holds ALL events and injections and evaluates its own chunk of the DRAWS of a batch; the values are gathered
    #             (parallel.py:258-278: zeros(nparams), own slice set, allreduce SUM) -- for catalogues too small to shard (a 12-event shard
    #             is pure launch latency) under a vectorised sampler.
    #   'both'    the ranks form `ngroups` groups (comm = parallel.split(world, ngroups)): inside a group events and injections are sharded as in
    #             'data', the groups take consecutive slices of the DRAWS of a batch (parallel.py:132-224, 306-341, 380-406) -- many ranks, a
    #             catalogue that shards only so far.
    if scheme not in ('data', 'params', 'both'):
      raise ValueError("hyperlikelihood: scheme must be 'data', 'params' or 'both'")
    if scheme == 'both' and not (comm is not None and hasattr(comm, 'world') and hasattr(comm, 'ngroups')):
      raise ValueError("hyperlikelihood: scheme='both' needs the group communicator of chimera_amd.parallel.split(world, ngroups)")
    self.scheme = scheme
    if scheme == 'params' and selection_function is not None and getattr(selection_function, 'comm', None) is not None \
       and selection_function.comm.nranks > 1:
      raise ValueError("hyperlikelihood: scheme='params' replicates the data -- build the selection function without a communicator")

    self.pixelated = True if self.theta_gw_det.pixels_opt_nsides is not None else False     # likelihood.py:79
    self.nevents = len(self.theta_gw_det.dL)
    self.z_int_res = self.z_grids.shape[1]

    if not (bw_method is None or bw_method in ('scott', 'silverman')
            or (isinstance(bw_method, Number) and not isinstance(bw_method, bool))):
      raise ValueError("bw_method should be 'scott', 'silverman', or a scalar")               # math.py:75
    if kernel not in ('epan', 'gauss'):
      raise ValueError("kernel must be 'epan' or 'gauss'")

    if self.pixelated:
      assert self.kind_p_gw3d in ['approximate', 'marginalized', 'full'], \
        "`kind_p_gw3d` must be one of `approximate`, `marginalized`, or `full`"              # likelihood.py:85
      self.max_npixels = self.population.gal_cat.max_npixels
      self.neff_pixels = self.population.gal_cat.neff_pixels
      self.p_gw3d = {'approximate': self.p_gw3dapprox, 'marginalized': self.p_gw3dmarg,
                     'full': self.p_gw3dfull}[self.kind_p_gw3d]
      if self.kind_p_gw3d == 'full':
        logger.info("`king_p_gw3d` has been set to 'full'. Only available kernel is `gaussian`. "
                    "The `binning` option is not available.")
      self._mode = self.kind_p_gw3d
    else:
      self._mode = '1d'
    if comm is not None and comm.nranks > 1 and scheme in ('data', 'both'):
      self._e0, self._e1 = chunk_bounds(self.nevents, comm.nranks, comm.rank)
    else:
      self._e0, self._e1 = 0, self.nevents
    self._handles = {}
    self._options = {}
    from .population.plugins import population_plugins
    self._plugins = population_plugins(self.population)        # (mass, rate, completeness) evaluated on the host?
    if any(self._plugins):
      self.max_draws_per_call = 4                              # the caller-evaluated tables are (draws, events, samples) arrays
    elif self._mode == 'full':
      # the 3-D mode keeps seven (draws, events, samples) arrays of doubles on the device (z, w and the whitened coordinates of k_full_prep):
      # batches are cut so that they stay below ~48 GB
      per_draw = 7 * 8 * max(1, self._e1 - self._e0) * max(1, int(np.shape(self.theta_gw_det.dL)[-1]))
      self.max_draws_per_call = int(max(1, min(256, 48e9 // per_draw)))
    logger.info(f'Created hyperlikelihood model. Using {self.nevents} GW events.')

  # -- device handles ----------------------------------------------------------------------------------
  def _handle(self, mode=None):
    mode = self._mode if mode is None else mode
    if mode in self._handles:
      return self._handles[mode][0]
    th = self.theta_gw_det
    E, S = np.shape(th.dL)
    keep = []

    def f64(a, shape, name):
      a = _lib.as_f64(a)
      if a.shape != shape:
        raise ValueError(f"hyperlikelihood: `{name}` has shape {a.shape}, expected {shape}")
      keep.append(a)
      return _lib.dptr(a)

    d = _lib.chm_like_desc()
    d.E, d.S, d.Z = E, S, self.z_int_res
    d.ev_begin, d.ev_end = self._e0, self._e1
    d.dL, d.m1det, d.m2det = f64(th.dL, (E, S), 'dL'), f64(th.m1det, (E, S), 'm1det'), f64(th.m2det, (E, S), 'm2det')
    d.pe_prior = f64(th.pe_prior, (E, S), 'pe_prior')
    d.z_grids = f64(self.z_grids, (E, self.z_int_res), 'z_grids')
    if mode != '1d':
      gc = self.population.gal_cat
      P = int(self.max_npixels)
      d.P = P
      d.p_cat = f64(gc.p_cat, (E, P, self.z_int_res), 'gal_cat.p_cat')
      d.P_compl = f64(np.asarray(gc.P_compl).reshape(E, self.z_int_res), (E, self.z_int_res), 'gal_cat.P_compl')
      d.gw_loc2d_pdf = f64(th.gw_loc2d_pdf, (E, P), 'gw_loc2d_pdf')
      neff = np.ascontiguousarray(self.neff_pixels, dtype=np.int32)
      if neff.shape != (E,):
        raise ValueError("hyperlikelihood: `neff_pixels` must have shape (Nevents,)")
      keep.append(neff)
      d.neff_pixels = _lib.iptr(neff)
      if mode == 'marginalized':
        pos = np.ascontiguousarray(_pix_of_sample(th.pixels_pe_opt_nside, th.pixels_opt_nsides))
        keep.append(pos)
        d.pix_of_sample = _lib.iptr(pos)
      if mode == 'full':
        d.ra, d.dec = f64(th.ra, (E, S), 'ra'), f64(th.dec, (E, S), 'dec')
        d.ra_pix, d.dec_pix = f64(th.ra_pix, (E, P), 'ra_pix'), f64(th.dec_pix, (E, P), 'dec_pix')
    else:
      d.P = 0
    d.mode = _lib.MODE[mode]
    d.kernel = _lib.KERNEL[self.kernel]
    if isinstance(self.bw_method, Number):
      d.bw_method, d.bw_scalar = 2, float(self.bw_method)
    else:
      d.bw_method, d.bw_scalar = _BW[self.bw_method], 0.
    d.binning = int(bool(self.binning))
    d.num_bins = int(self.num_bins)
    d.cut_grid = float('nan') if self.cut_grid is None else float(self.cut_grid)
    d.pe_neff = float(self.pe_neff)
    d.device = self.device
    h = C.c_void_p()
    _lib.check(_lib.lib().chm_like_create(C.byref(d), C.byref(h)))
    self._handles[mode] = (h, None)
    for name, value in self._options.items():
      _lib.check(_lib.lib().chm_like_set_option(h, _lib.OPTION[name], int(value)))
    return h

  def set_option(self, name, value):
    """Evaluation option of this object's device handles (include/chimera_hip.h, ``CHM_OPT_*``; names: ``_lib.OPTION``), e.g.
    ``set_option('groups', 1)``, ``set_option('timing', 2)`` (``'fused'`` > 0 only with a ``-DCHM_WITH_FUSED`` variant build).  Also handed to the selection function, whose handle takes part in the
    same calls.  ``diag_*`` options need a library built with ``-DCHM_DIAG`` (``ValueError`` otherwise).  Returns ``self``."""
    if name not in _lib.OPTION:
      raise ValueError(f"hyperlikelihood.set_option: unknown option {name!r} (known: {sorted(_lib.OPTION)})")
    for h, _ in self._handles.values():
      _lib.check(_lib.lib().chm_like_set_option(h, _lib.OPTION[name], int(value)))
    self._options = dict(self._options, **{name: int(value)})
    if self.selection_function is not None and hasattr(self.selection_function, 'set_option'):
      self.selection_function.set_option(name, value)
    return self

  def close(self):
    for h, _ in self._handles.values():
      _lib.lib().chm_like_destroy(h)
    self._handles = {}
    self._scalar_state = None

  def lane(self, comm=None):
    """A second evaluation lane on the data this object already holds in HBM (``chm_like_clone``; the selection function gets a
    lane of its own): the same methods, results identical to the last bit, its own streams, tables and workspaces.  One host
    thread per lane -- ``chm_eval`` runs outside the GIL -- keeps two evaluations in flight on one GPU: the per-call fixed costs
    (tables, launch path, reduction tail) of one hide under the kernels of the other, which is what decides the scaling of small
    per-GPU shards (bench.py ``--inflight 2``).  A lane of a sharded likelihood needs its own communicator (``comm=``):
    collectives of one communicator must be issued in the same order by every rank, two host threads do not guarantee that."""
    import copy
    if self.comm is not None and getattr(self.comm, 'nranks', 1) >= 1 and comm is None:
      raise ValueError("hyperlikelihood.lane: a lane of a likelihood with a communicator needs a communicator of its own (comm=)")
    if self.scheme == 'both' and not (hasattr(comm, 'world') and hasattr(comm, 'ngroups')):
      raise ValueError("hyperlikelihood.lane: a lane under scheme='both' needs a group communicator of its own (chimera_amd.parallel.split)")
    self._handle()
    new = copy.copy(self)
    new._handles = {}
    new._scalar_state = None
    new.__dict__.pop('_out_blocks', None)                      # (a lane is driven by its own host thread: its own output rows)
    new._options = dict(self._options)
    new.comm = comm
    for mode, (h, _) in self._handles.items():
      h2 = C.c_void_p()
      _lib.check(_lib.lib().chm_like_clone(h, C.byref(h2)))
      new._handles[mode] = (h2, None)
    if self.selection_function is not None:
      new.selection_function = self.selection_function.lane(comm=comm)
    return new

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  # -- one evaluation through the C ABI ----------------------------------------------------------------
  def _eval(self, pops, want=(), mode=None, with_sel=True, collective=True):
    """Evaluate a list of population draws.  ``want`` subset of {'log_like_evs','numlike_evs','p_gw','partials'}.
    ``collective=False``: this rank's shard only -- no communicator is passed, so the call runs no all-reduce and may be made
    by one rank alone (the inspection calls p_gw*, compute_numlike_evs return per-event arrays of the rank's own events)."""
    nb = len(pops)
    tab, tab_keep = None, None
    if isinstance(pops, C.Array):
      if any(self._plugins):
        raise RuntimeError("plug-in population models need population objects, not a packed chm_params array")
      params = pops
    else:
      params = (_lib.chm_params * nb)(*[p.to_params() for p in pops])
      if any(self._plugins):                                   # plug-in models: evaluate them on the host for this shard
        from .population.plugins import build_tab
        th, e0, e1 = self.theta_gw_det, self._e0, self._e1
        evd = dict(dL=_lib.as_f64(th.dL)[e0:e1], m1det=_lib.as_f64(th.m1det)[e0:e1], m2det=_lib.as_f64(th.m2det)[e0:e1],
                   z_grids=self.z_grids[e0:e1])
        injd = self.selection_function._inj_shard() if (with_sel and self.selection_function is not None) else None
        tab, tab_keep = build_tab(pops, self._plugins, ev=evd, inj=injd)
    h = self._handle(mode)
    El = self._e1 - self._e0
    sel = self.selection_function._handle() if (with_sel and self.selection_function is not None) else None
    comm = self.comm if (collective and self.scheme in ('data', 'both')) else None      # 'params': replicas, nothing to reduce inside a call
    comm_h = getattr(comm, 'handle', None) if comm is not None else None                 # RCCL all-reduce inside chm_eval
    host_reduce = (comm is not None and comm_h is None and comm.nranks > 1 and hasattr(comm, 'allreduce_sum'))
    if not want and tab is None and not host_reduce:
      # the plain call (batch, the samplers): its three output rows and the chm_out that points at them are kept per draw count (building them
      # is ~10 us of ctypes casts per call); the caller gets copies
      cache = self.__dict__.setdefault('_out_blocks', {})
      blk = cache.get(nb)
      if blk is None:
        buf = np.empty((3, nb))
        o_ = _lib.chm_out()
        o_.log_hyper, o_.log_num, o_.N_exp = (_lib.dptr(buf[i]) for i in range(3))
        blk = cache[nb] = (buf, o_)
      buf, out = blk
      res = None
    else:
      res = {'log_hyper': np.empty(nb), 'log_num': np.empty(nb), 'N_exp': np.empty(nb)}
      out = _lib.chm_out()
      out.log_hyper, out.log_num, out.N_exp = (_lib.dptr(res[k]) for k in ('log_hyper', 'log_num', 'N_exp'))
    if 'log_like_evs' in want:
      res['log_like_evs'] = np.empty((nb, El)); out.log_like_evs = _lib.dptr(res['log_like_evs'])
    if 'numlike_evs' in want:
      res['numlike_evs'] = np.empty((nb, El)); out.numlike_evs = _lib.dptr(res['numlike_evs'])
    if 'p_gw' in want:
      m = self._mode if mode is None else mode
      shape = (nb, El, self.z_int_res) if m == '1d' else (nb, El, int(self.max_npixels), self.z_int_res)
      res['p_gw'] = np.empty(shape); out.p_gw = _lib.dptr(res['p_gw'])
    if 'partials' in want:
      res['partials'] = np.empty((nb, 3)); out.partials = _lib.dptr(res['partials'])
    if host_reduce and 'partials' not in res:               # HostComm: the partial sums are reduced and combined on the host
      res['partials'] = np.empty((nb, 3)); out.partials = _lib.dptr(res['partials'])
    if tab is not None:
      _lib.check(_lib.lib().chm_eval_tabulated(h, sel, comm_h, params, nb, self.nevents, C.byref(tab), C.byref(out)))
    else:
      _lib.check(_lib.lib().chm_eval(h, sel, comm_h, params, nb, self.nevents, C.byref(out)))
    if res is None:
      return {'log_hyper': buf[0].copy(), 'log_num': buf[1].copy(), 'N_exp': buf[2].copy()}
    if host_reduce:
      from .parallel import combine_partials
      tot = self.comm.allreduce_sum(res['partials']).reshape(nb, 3)
      sf = self.selection_function if with_sel else None
      for b in range(nb):
        p = params[b]
        res['log_hyper'][b], res['log_num'][b], res['N_exp'][b] = combine_partials(
          tot[b], self.nevents, sf.N_inj if sf is not None else 1., sf.N_eff if sf is not None else None,
          bool(p.scale_free), p.R0, p.Tobs, has_like=True, has_sel=sf is not None)
    return res

  def last_timing(self):
    """HIP-event timings [ms] of the last evaluation: total, tables, samples, kde+integrand, selection, reduce."""
    ms = np.zeros(8)
    sel = self.selection_function._handle() if self.selection_function is not None else None
    _lib.check(_lib.lib().chm_last_timing(self._handle(), sel, _lib.dptr(ms)))
    return ms

  def full_general_pixels(self, nb=1):
    """Diagnostics of kind_p_gw3d='full': (draw, event, pixel) triples of the last evaluation of nb draws that the sample-stationary KDE
    kernel left to the general one (include/chimera_hip.h: chm_like_full_general_pixels)."""
    import ctypes
    n = ctypes.c_int64(0)
    _lib.check(_lib.lib().chm_like_full_general_pixels(self._handle(), int(nb), ctypes.byref(n)))
    return int(n.value)

  # -- reference surface: GW kernels -------------------------------------------------------------------
  def p_gw1d(self, pop_lambdas):
    """likelihood.py:105-144 -> (Nevents, z_int_res)."""
    return self._eval([pop_lambdas], want=('p_gw',), mode='1d', with_sel=False, collective=False)['p_gw'][0]

  def p_gw3dapprox(self, pop_lambdas):
    """likelihood.py:150-154 -> (Nevents, max_npixels, z_int_res)."""
    return self._eval([pop_lambdas], want=('p_gw',), mode='approximate', with_sel=False, collective=False)['p_gw'][0]

  def p_gw3dmarg(self, pop_lambdas):
    """likelihood.py:160-205."""
    return self._eval([pop_lambdas], want=('p_gw',), mode='marginalized', with_sel=False, collective=False)['p_gw'][0]

  def p_gw3dfull(self, pop_lambdas):
    """likelihood.py:211-260."""
    return self._eval([pop_lambdas], want=('p_gw',), mode='full', with_sel=False, collective=False)['p_gw'][0]

  # -- reference surface: numerator --------------------------------------------------------------------
  def compute_numlike_evs(self, pop_lambdas):
    """likelihood.py:266-292 -> (Nevents,) [this rank's events when sharded]."""
    return self._eval([pop_lambdas], want=('numlike_evs',), with_sel=False, collective=False)['numlike_evs'][0]

  def compute_log_likenum(self, pop_lambdas):
    """likelihood.py:294-301.  Sharded: a collective call (every rank makes it; the sum runs over all events)."""
    return self._eval([pop_lambdas], with_sel=False)['log_num'][0]

  # -- reference surface: hyper-likelihood -------------------------------------------------------------
  def compute_log_hyperlike(self, **hyper_lambdas):
    """likelihood.py:307-316.  Array-valued hyper-parameters (the vectorised dict of the reference's sampler glue,
    CHIMERA/utils/emcee_utils.py:54-64) evaluate every draw in one launch sequence and return an array."""
    n = _vector_length(hyper_lambdas)
    if n is None:
      return self._scalar_call(hyper_lambdas)
    return self.batch(hyper_lambdas)

  def _scalar_call(self, lam):
    """One draw, nothing but the value wanted -- the reference's ``like(**lambda)``.  The call is ~0.18 ms of which the device needs ~0.15: the
    host side keeps its parameter block, output block and handles between calls and only patches the hyper-parameters in (what
    ``batch([lam])[0]`` does through freshly made arrays; same ``chm_eval``, same value)."""
    st = getattr(self, '_scalar_state', None)
    sf = self.selection_function
    if st is None or st[0] is not self._handles.get(self._mode, (None,))[0] or st[4] is not (sf._h if sf is not None else None):
      comm = self.comm if self.scheme == 'data' else None
      comm_h = getattr(comm, 'handle', None) if comm is not None else None
      plain = not any(self._plugins) and not (comm is not None and comm_h is None and comm.nranks > 1) \
        and not (self.scheme == 'params' and self.comm is not None and self.comm.nranks > 1) and self.scheme != 'both'
      if not plain:
        return self.batch([lam])[0]
      h = self._handle()
      self._params_array([{}])                                 # (forms the slot table and the base parameter block)
      arr = (_lib.chm_params * 1)()
      res = np.empty(3)
      out = _lib.chm_out()
      out.log_hyper, out.log_num, out.N_exp = (C.cast(C.c_void_p(res.ctypes.data + 8 * i), _lib.c_dp) for i in range(3))
      sel = self.selection_function._handle() if self.selection_function is not None else None
      st = self._scalar_state = (h, arr, res, out, sel, comm_h, C.byref(arr), C.byref(self._base_params), C.sizeof(_lib.chm_params), C.byref(out),
                                 _lib.lib().chm_eval)
    h, arr, res, out, sel, comm_h, parr, pbase, sz, pout, chm_eval = st
    C.memmove(parr, pbase, sz)
    p = arr[0]
    slots = self._slots
    for k, v in lam.items():
      for field, idx, is_int in slots.get(k, ()):
        if idx is None:
          setattr(p, field, int(v) if is_int else float(v))
        else:
          getattr(p, field)[idx] = float(v)
    rc = chm_eval(h, sel, comm_h, arr, 1, self.nevents, pout)
    if rc:
      _lib.check(rc)
    return res[0]

  def __call__(self, **hyper_lambdas):
    """likelihood.py:318-320."""
    return self.compute_log_hyperlike(**hyper_lambdas)

  def compute_all(self, **hyper_lambdas):
    """likelihood.py:326-338 -> (log_like_evs, log_like_num, log N_exp, log_hyper)."""
    pop_lambdas = self.population.update(**hyper_lambdas)
    r = self._eval([pop_lambdas], want=('log_like_evs',))
    with np.errstate(all='ignore'):
      return r['log_like_evs'][0], r['log_num'][0], np.log(r['N_exp'][0]), r['log_hyper'][0]

  # -- batched draws (the reference's 'params' scheme, CHIMERA/parallel.py:258-278) ----------------------
  def _params_array(self, list_of_hyper_lambdas):
    """One chm_params per draw = population.update(**lambda).to_params(), formed by patching a copy of the base struct
    (same values: unknown keys are ignored, every model picks the keys it owns -- pop_wrapper.py:56-64).  A dict of arrays (the vectorised
    dict of the reference's sampler glue, emcee_utils.py:54-64) is packed with one store per hyper-parameter, no per-draw dict."""
    if getattr(self, '_slots', None) is None:
      from .population._base import param_slots
      pop = self.population
      self._slots = param_slots(pop.cosmo, pop.mass, pop.rate)
      self._base_params = pop.to_params()
      self._params_dtype = np.dtype(_lib.chm_params)
      self._base_view = np.frombuffer(self._base_params, dtype=self._params_dtype).copy()
    if isinstance(list_of_hyper_lambdas, dict):
      nb = _vector_length(list_of_hyper_lambdas)
      arr, view = self._params_block(nb)
      for k, v in list_of_hyper_lambdas.items():
        for field, idx, is_int in self._slots.get(k, ()):
          if idx is None:
            view[field] = v
          else:
            view[field][:, idx] = v
      return arr
    nb = len(list_of_hyper_lambdas)
    arr = (_lib.chm_params * nb)()
    if nb < 8:                                                 # few draws: patch ctypes structs directly
      sz = C.sizeof(_lib.chm_params)
      for b, lam in enumerate(list_of_hyper_lambdas):
        C.memmove(C.byref(arr, b * sz), C.byref(self._base_params), sz)
        p = arr[b]
        for k, v in lam.items():
          for field, idx, is_int in self._slots.get(k, ()):
            if idx is None:
              setattr(p, field, int(v) if is_int else float(v))
            else:
              getattr(p, field)[idx] = float(v)
      return arr
    arr, view = self._params_block(nb)                         # every draw starts from the base population
    keys = list_of_hyper_lambdas[0].keys()
    if all(lam.keys() == keys for lam in list_of_hyper_lambdas):
      for k in keys:                                           # one vectorised store per hyper-parameter name
        slots = self._slots.get(k, ())
        if slots:
          vals = [lam[k] for lam in list_of_hyper_lambdas]
          for field, idx, is_int in slots:
            if idx is None:
              view[field] = vals
            else:
              view[field][:, idx] = vals
    else:
      for b, lam in enumerate(list_of_hyper_lambdas):
        for k, v in lam.items():
          for field, idx, is_int in self._slots.get(k, ()):
            if idx is None:
              view[field][b] = v
            else:
              view[field][b, idx] = v
    return arr

  def _params_block(self, nb):
    """nb copies of the base chm_params as a ctypes array + its structured NumPy view (one memmove from a cached image: a broadcast store
    through the structured dtype costs 10 us for 128 draws)."""
    img = self._params_images.get(nb) if hasattr(self, '_params_images') else None
    if img is None:
      if not hasattr(self, '_params_images'):
        self._params_images = {}
      img = self._params_images[nb] = np.tile(self._base_view, nb).tobytes()
    arr = (_lib.chm_params * nb)()
    C.memmove(arr, img, len(img))
    return arr, np.frombuffer(arr, dtype=self._params_dtype)

  #: draws evaluated per launch sequence; longer lists are processed in slices (the per-draw workspaces -- source-frame
  #: z and weights of every sample, per-z factors -- take ~16 B x samples per draw: 65 MB per draw at 1000 events x 4096)
  max_draws_per_call = 256

  def batch(self, list_of_hyper_lambdas):
    """log-hyperlikelihood of several draws in one launch sequence: array of len(list).  Also takes the vectorised form, a dict of
    hyper-parameter arrays of one length (scalars broadcast) -- what ``compute_log_hyperlike(**arrays)`` hands over."""
    if isinstance(list_of_hyper_lambdas, dict):
      n = _vector_length(list_of_hyper_lambdas)
      if n is None:
        raise ValueError("hyperlikelihood.batch: a dict of hyper-parameters needs at least one array-valued entry")
      plain = not any(self._plugins) and self.scheme == 'data' and n <= max(1, int(self.max_draws_per_call))
      if plain:
        return self._eval(self._params_array(list_of_hyper_lambdas))['log_hyper']
      list_of_hyper_lambdas = [{k: (np.asarray(v).reshape(-1)[i] if np.ndim(v) > 0 else v) for k, v in list_of_hyper_lambdas.items()} for i in range(n)]
    lams = list(list_of_hyper_lambdas)
    if self.scheme == 'params' and self.comm is not None and self.comm.nranks > 1:
      return self._batch_over_params(lams)
    if self.scheme == 'both':
      return self._batch_over_both(lams)
    return self._batch_local(lams)

  def _batch_local(self, lams):
    m = max(1, int(self.max_draws_per_call))
    pack = (lambda ls: [self.population.update(**l) for l in ls]) if any(self._plugins) else self._params_array
    if len(lams) <= m:
      return self._eval(pack(lams))['log_hyper']
    return np.concatenate([self._eval(pack(lams[i:i + m]))['log_hyper'] for i in range(0, len(lams), m)])

  def _batch_over_both(self, lams):
    """The reference's 'both' scheme (CHIMERA/parallel.py:380-406): group g of the world evaluates its slice of the draws on its sharded copy of
    the data (the 'data' scheme inside the group: a collective of the group's ranks); the world then assembles all n values -- the first rank
    of every group puts its group's values into a vector of zeros, the vectors are summed over the world (the reference sums the shards'
    partial log-likelihoods in the same all-reduce; here a group's values are already complete: k_combine ran behind the group's all-reduce)."""
    from .parallel import draws_of_group
    n, grp, world = len(lams), self.comm, self.comm.world
    i0, i1 = draws_of_group(n, grp.ngroups, grp.group_id)
    out = np.zeros(n)
    if i0 < i1:
      vals = self._batch_local(lams[i0:i1])
      if grp.rank == 0:
        out[i0:i1] = vals
    return np.asarray(world.allreduce_sum(out)).reshape(n)

  def _batch_over_params(self, lams):
    """The reference's 'params' scheme (CHIMERA/parallel.py:258-278): rank r evaluates the draws [r c, min((r + 1) c, n)), c = ceil(n / R), on
    its full copy of the data; every rank then holds all n values (own slice in a vector of zeros, summed over the ranks)."""
    n, R, r = len(lams), self.comm.nranks, self.comm.rank
    per = (n + R - 1) // R
    i0, i1 = min(r * per, n), min((r + 1) * per, n)
    out = np.zeros(n)
    if i0 < i1:
      out[i0:i1] = self._batch_local(lams[i0:i1])
    return np.asarray(self.comm.allreduce_sum(out)).reshape(n)
