"""ctypes binding of libchimera_hip.so (include/chimera_hip.h).

The library is the only compute backend of this package: there is no CPU fallback.  If it has not been built
(``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C chimera_amd/csrc``) every entry point raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHIMERA_LIB: another build of the same library (A/B of kernel variants, scripts/abl.sh); never a fallback
LIB_PATH = os.environ.get('CHIMERA_LIB') or os.path.join(_HERE, 'lib', 'libchimera_hip.so')

CHM_OK, CHM_E_ARG, CHM_E_HIP, CHM_E_NOMEM, CHM_E_RCCL = 0, -1, -2, -3, -4
MODE = {'1d': 0, 'approximate': 1, 'marginalized': 2, 'full': 3}
KERNEL = {'epan': 0, 'gauss': 1}
NCOSMO, NMASS, NRATE = 8, 8, 4

# function ids of chm_model_eval
(F_E, F_INT_INVE, F_DCR, F_DCT, F_DL, F_DDLDZ, F_DVCDZ, F_VC, F_XI, F_Z_FROM_DGW, F_RATE, F_PM1M2, F_PRIMARY,
 F_SECONDARY, F_SMOOTHING, F_PM1M2_FUSED, F_TPL_CDF, F_GAUSSIAN, F_TRUNC_GAUSSIAN) = range(19)

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)


class chm_params(C.Structure):
  _fields_ = [('cosmo_model', C.c_int32), ('mass_model', C.c_int32), ('rate_model', C.c_int32),
              ('z_grid_res', C.c_int32), ('mass_grid_res', C.c_int32), ('scale_free', C.c_int32),
              ('has_catalog', C.c_int32), ('_pad', C.c_int32),
              ('z_max', C.c_double), ('cosmo', C.c_double * NCOSMO), ('mass', C.c_double * NMASS),
              ('rate', C.c_double * NRATE), ('R0', C.c_double), ('Tobs', C.c_double),
              ('compl_z0', C.c_double), ('compl_z1', C.c_double)]


class chm_like_desc(C.Structure):
  _fields_ = [('E', C.c_int32), ('S', C.c_int32), ('Z', C.c_int32), ('P', C.c_int32),
              ('ev_begin', C.c_int32), ('ev_end', C.c_int32),
              ('dL', c_dp), ('m1det', c_dp), ('m2det', c_dp), ('pe_prior', c_dp), ('ra', c_dp), ('dec', c_dp),
              ('pix_of_sample', c_ip), ('z_grids', c_dp), ('p_cat', c_dp), ('P_compl', c_dp),
              ('gw_loc2d_pdf', c_dp), ('ra_pix', c_dp), ('dec_pix', c_dp), ('neff_pixels', c_ip),
              ('mode', C.c_int32), ('kernel', C.c_int32), ('bw_method', C.c_int32), ('binning', C.c_int32),
              ('num_bins', C.c_int32), ('device', C.c_int32),
              ('bw_scalar', C.c_double), ('cut_grid', C.c_double), ('pe_neff', C.c_double)]


class chm_sel_desc(C.Structure):
  _fields_ = [('I', C.c_int64), ('inj_begin', C.c_int64), ('inj_end', C.c_int64),
              ('dL', c_dp), ('m1det', c_dp), ('m2det', c_dp), ('p_draw', c_dp),
              ('N_inj', C.c_double), ('N_eff', C.c_double), ('device', C.c_int32), ('_pad', C.c_int32)]


class chm_pcat_desc(C.Structure):
  _fields_ = [('E', C.c_int32), ('P', C.c_int32), ('Z', C.c_int32), ('device', C.c_int32),
              ('z_grids', c_dp), ('offsets', C.POINTER(C.c_int64)), ('gal_z', c_dp), ('gal_sig', c_dp), ('gal_w', c_dp),
              ('weight_grid', c_dp)]


class chm_out(C.Structure):
  _fields_ = [('log_hyper', c_dp), ('log_num', c_dp), ('N_exp', c_dp), ('log_like_evs', c_dp),
              ('numlike_evs', c_dp), ('p_gw', c_dp), ('partials', c_dp)]


class chm_tab(C.Structure):
  """Caller-evaluated plug-in models of one call (include/chimera_hip.h: struct chm_tab)."""
  _fields_ = [('pm_samples', c_dp), ('pm_inj', c_dp), ('rate_grid', c_dp), ('rate_inj', c_dp), ('bkg_grid', c_dp),
              ('bkg_inj', c_dp), ('fR', c_dp), ('z_table', c_dp), ('dL_table', c_dp), ('jac_grid', c_dp), ('jac_inj', c_dp)]


SYMBOLS = ['chm_version', 'chm_device_count', 'chm_last_error', 'chm_like_create', 'chm_like_destroy',
           'chm_sel_create', 'chm_sel_destroy', 'chm_like_clone', 'chm_sel_clone', 'chm_eval', 'chm_eval_tabulated', 'chm_model_eval', 'chm_model_tables',
           'chm_comm_unique_id', 'chm_comm_init_rank', 'chm_comm_destroy', 'chm_comm_allreduce_sum', 'chm_comm_nranks',
           'chm_device_synchronize', 'chm_device_pci_bus_id',
           'chm_last_timing', 'chm_like_full_general_pixels', 'chm_pcat_compute', 'chm_kde2d_pixels',
           'chm_kde1d', 'chm_binning1d', 'chm_gkde_nd', 'chm_gkde_nd_log', 'chm_trapz', 'chm_cumtrapz',
           'chm_like_set_option', 'chm_sel_set_option', 'chm_diag_build', 'chm_has_fused', 'chm_comm_set_ticket', 'chm_comm_ticket_reset', 'chm_comm_ticket_skip',
           'chm_comm_ticket_timeout', 'chm_comm_ticket_wait', 'chm_comm_ticket_done']

# options of a handle (include/chimera_hip.h: CHM_OPT_*); ids >= 100 need a library built with -DCHM_DIAG
OPTION = {'serial': 1, 'groups': 2, 'fused': 3, 'timing': 4, 'graph_max_nb': 5, 'spin_wait': 6,
          'diag_full_chain': 100, 'diag_no_dense_node': 101, 'diag_marg_generic': 102, 'diag_samples_generic': 103,
          'diag_selection_generic': 104, 'diag_no_grid_prep': 105, 'diag_zf_full': 106, 'diag_kde_ipw': 107, 'diag_samp_cpb': 108,
          'diag_self_blocks': 109, 'diag_few_nb': 110, 'diag_no_zero_copy': 111, 'diag_no_zf_sel': 112, 'diag_host_prof': 113, 'diag_fused_nw': 114, 'diag_poison': 115}

_lib = None


def lib():
  """Load libchimera_hip.so once; raise (never fall back) if it is missing."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise RuntimeError(f"chimera_amd: HIP library not built ({LIB_PATH} missing). Build it with "
                       "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback.")
  L = C.CDLL(LIB_PATH)
  vp = C.c_void_p
  L.chm_version.restype = C.c_char_p
  L.chm_last_error.restype = C.c_char_p
  L.chm_device_count.restype = C.c_int
  L.chm_like_create.argtypes = [C.POINTER(chm_like_desc), C.POINTER(vp)]
  L.chm_like_destroy.argtypes = [vp]
  L.chm_sel_create.argtypes = [C.POINTER(chm_sel_desc), C.POINTER(vp)]
  L.chm_like_clone.argtypes = [vp, C.POINTER(vp)]
  L.chm_sel_clone.argtypes = [vp, C.POINTER(vp)]
  L.chm_sel_destroy.argtypes = [vp]
  L.chm_eval.argtypes = [vp, vp, vp, C.POINTER(chm_params), C.c_int32, C.c_int64, C.POINTER(chm_out)]
  L.chm_eval_tabulated.argtypes = [vp, vp, vp, C.POINTER(chm_params), C.c_int32, C.c_int64, C.POINTER(chm_tab), C.POINTER(chm_out)]
  L.chm_model_eval.argtypes = [C.POINTER(chm_params), C.c_int32, c_dp, c_dp, C.c_int64, c_dp, C.c_int32]
  L.chm_model_tables.argtypes = [C.POINTER(chm_params), c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int32]
  L.chm_comm_unique_id.argtypes = [C.c_char_p]
  L.chm_comm_init_rank.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
  L.chm_comm_destroy.argtypes = [vp]
  L.chm_comm_allreduce_sum.argtypes = [vp, c_dp, C.c_int32]
  L.chm_comm_nranks.argtypes = [vp]
  L.chm_device_synchronize.argtypes = [C.c_int32]
  L.chm_last_timing.argtypes = [vp, vp, c_dp]
  L.chm_like_full_general_pixels.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64)]
  L.chm_pcat_compute.argtypes = [C.POINTER(chm_params), C.POINTER(chm_pcat_desc), c_dp]
  L.chm_kde2d_pixels.argtypes = [C.c_int32, C.c_int32, C.c_int32, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp, C.c_int32]
  i32, i64, f64 = C.c_int32, C.c_int64, C.c_double
  L.chm_kde1d.argtypes = [c_dp, c_dp, i64, c_dp, i64, i32, i32, f64, c_dp, i32]
  L.chm_binning1d.argtypes = [c_dp, c_dp, i64, i32, c_dp, c_dp, i32]
  L.chm_gkde_nd.argtypes = [c_dp, c_dp, i32, i64, c_dp, i64, i32, f64, c_dp, i32]
  L.chm_gkde_nd_log.argtypes = [c_dp, c_dp, i32, i64, c_dp, i64, i32, f64, c_dp, i32]
  L.chm_trapz.argtypes = [c_dp, c_dp, i64, i32, i32, c_dp, i32]
  L.chm_cumtrapz.argtypes = [c_dp, c_dp, i32, c_dp, i32]
  L.chm_like_set_option.argtypes = [vp, i32, i64]
  L.chm_sel_set_option.argtypes = [vp, i32, i64]
  L.chm_diag_build.argtypes = []
  L.chm_has_fused.argtypes = []
  L.chm_comm_set_ticket.argtypes = [vp, i64]
  L.chm_comm_ticket_reset.argtypes = [i64]
  L.chm_comm_ticket_skip.argtypes = [i64]
  try:
    L.chm_comm_ticket_timeout.argtypes = [i64]
    L.chm_comm_ticket_wait.argtypes = [i64]
    L.chm_comm_ticket_done.argtypes = [i64]
  except AttributeError:
    if 'CHIMERA_LIB' not in os.environ:        # (an older variant build selected for a same-box A/B may lack the round-6 entries; the release library may not)
      raise
  L.chm_device_pci_bus_id.argtypes = [i32, C.c_char_p, i32]
  for name in SYMBOLS:
    if name not in ('chm_version', 'chm_last_error') and (hasattr(L, name) or 'CHIMERA_LIB' not in os.environ):
      getattr(L, name).restype = C.c_int
  _lib = L
  return L


def check(rc):
  """Map a CHM_E_* return code to the reference-side exception type (ValueError for arguments, RuntimeError else)."""
  if rc == CHM_OK:
    return
  msg = lib().chm_last_error().decode()
  if rc == CHM_E_ARG:
    raise ValueError(msg)
  if rc == CHM_E_NOMEM:
    raise MemoryError(msg)
  raise RuntimeError(msg)


def default_device():
  for k in ('CHIMERA_DEVICE', 'LOCAL_RANK'):
    if k in os.environ:
      return int(os.environ[k])
  return 0


def as_f64(a):
  return np.ascontiguousarray(a, dtype=np.float64)


def dptr(a):
  return None if a is None else a.ctypes.data_as(c_dp)


def iptr(a):
  return None if a is None else a.ctypes.data_as(c_ip)
