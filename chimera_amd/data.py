"""Data containers of the hyper-likelihood path (reference: CHIMERA/data.py:15-59).

Plain attribute containers with the reference's field names and ``.update(**kw)`` (which returns a new object, as
``eqx.tree_at`` does in data.py:16-25).  Arrays are kept as NumPy arrays on the host; the HIP handle copies them to
HBM once (see likelihood.py).  File loaders / HEALPix pixelisation (data.py:70-484) are outside the hot path.
"""
import numpy as np


class theta_generic(object):
  _fields = ()

  def __init__(self, **kwargs):
    unknown = set(kwargs) - set(self._fields)
    if unknown:
      raise TypeError(f"{self.__class__.__name__}: unexpected field(s) {sorted(unknown)}")
    for f in self._fields:
      v = kwargs.get(f, None)
      if v is not None and not isinstance(v, dict):
        v = np.asarray(v)
      setattr(self, f, v)
    self.__post_init__()

  def __post_init__(self):
    pass

  def update(self, **kwargs):
    d = {f: getattr(self, f) for f in self._fields}
    for k in kwargs:
      if k not in self._fields:
        raise AttributeError(f"{self.__class__.__name__} has no field '{k}'")
    d.update(kwargs)
    return self.__class__(**d)

  def __repr__(self):
    parts = []
    for f in self._fields:
      v = getattr(self, f)
      if v is not None:
        parts.append(f"{f}={getattr(v, 'shape', v)}")
    return f"{self.__class__.__name__}({', '.join(parts)})"


class theta_pe_det(theta_generic):
  """data.py:27-47."""
  _fields = ('m1det', 'm2det', 'dL', 'phi', 'theta', 'ra', 'dec', 'pe_prior', 'pixels_pe_all_nsides', 'opt_nsides',
             'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside')

  def __post_init__(self):
    if self.pe_prior is None and self.dL is not None:
      self.pe_prior = np.ones_like(self.dL, dtype=np.float64)


class theta_inj_det(theta_generic):
  """data.py:49-53."""
  _fields = ('m1det', 'm2det', 'dL', 'p_draw')


class theta_src(theta_generic):
  """data.py:55-59."""
  _fields = ('m1src', 'm2src', 'z', 'original_distances')


theta_pe_datasets = ['m1det', 'm2det', 'dL', 'pe_prior']
theta_pe_pixelated_datasets = ['m1det', 'm2det', 'dL', 'pe_prior', 'ra', 'dec', 'theta', 'phi', 'opt_nsides',
                               'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside']


def save_npz(fname, obj):
  """Store a theta_* container as ``.npz`` (stand-in for the reference's HDF5 layout, data.py:61-64, io.py:7-66)."""
  np.savez(fname, **{f: getattr(obj, f) for f in obj._fields
                     if getattr(obj, f) is not None and not isinstance(getattr(obj, f), dict)})


def load_npz(fname, cls=theta_pe_det):
  with np.load(fname) as d:
    return cls(**{k: d[k] for k in d.files if k in cls._fields})


################
# DATA LOADING #   (reference: CHIMERA/data.py:70-236)
################
import ctypes as _C
from . import _lib
from .utils import angles
from .utils.io import save_set, load_set, load_data_h5
from .utils.config import logger


def load_galaxy_catalog(file_path, parameters=['ra_gal', 'dec_gal', 'z_cgal'], units='rad', backend='numpy'):
  """data.py:70-105."""
  if units not in ['rad', 'deg']:
    raise ValueError("units must be either 'rad' or 'deg'")
  data = load_data_h5(file_path, require_keys=parameters)
  result = {'ra': data['ra_gal'], 'dec': data['dec_gal'], 'z': data['z_cgal']}
  if units == 'rad':
    result['ra'] = np.deg2rad(result['ra'])
    result['dec'] = np.deg2rad(result['dec'])
  return result


def _process_selection(n, max_n, name):
  """What a loader's ``nevents`` / ``nsamples`` / ``ninj`` argument selects (the rules of data.py:218-233): everything for ``None``, the listed
  indices for an index array, and for a count that many entries drawn without replacement (``np.random.choice``, so that a seeded
  session picks what the reference picks) in ascending order -- or everything, with a warning, when the count exceeds what is there."""
  if n is None:
    return slice(None)
  if np.ndim(n) > 0:
    return np.asarray(n)
  if isinstance(n, (bool, np.bool_)) or not isinstance(n, (int, np.integer)):
    raise ValueError(f"Invalid selection for {name}: must be None, list or int")
  if n <= max_n:
    return np.sort(np.random.choice(max_n, n, replace=False))
  logger.warning(f"Requested more {name} than available. Using all {max_n}.")
  return slice(None)


def load_gw_pe_samples(file_ev_pe, parameters=['dL', 'm1det', 'm2det', 'phi', 'theta'], group='posteriors', nevents=None,
                       nsamples=None, return_struct=True):
  """data.py:107-148."""
  data = load_data_h5(file_ev_pe, group_h5=group, require_keys=parameters)
  event_idx = _process_selection(nevents, data['dL'].shape[0], 'events')
  sample_idx = _process_selection(nsamples, data['dL'].shape[1], 'samples')
  result = {k: np.asarray(data[k][event_idx][:, sample_idx]) for k in parameters}
  if {'theta', 'phi'}.issubset(parameters):
    ra, dec = angles.ra_dec_from_th_phi(result['theta'], result['phi'])
    result.update(ra=ra, dec=dec)
  return theta_pe_det(**result) if return_struct else result


def load_injection_data(file_inj, snr_cut=None, ninj=None, group=None, key_mapping=None, return_struct=True):
  """data.py:150-216."""
  defaults = {'m1s': 'm1src', 'm2s': 'm2src', 'm1d': 'm1det', 'm2d': 'm2det', 'dL': 'dL', 'z': 'z', 'snr': 'SNR_net',
              'log_pdraw': 'log_p_draw_nospin'}
  keys = {**defaults, **(key_mapping or {})}
  data = load_data_h5(file_inj, group_h5=group, require_keys=[keys[k] for k in ['dL', 'snr', 'log_pdraw']])
  keep = data[keys['snr']] > snr_cut if snr_cut else slice(None)
  m1d = data[keys['m1d']] if keys['m1d'] in data else data[keys['m1s']] * (1 + data[keys['z']])
  m2d = data[keys['m2d']] if keys['m2d'] in data else data[keys['m2s']] * (1 + data[keys['z']])
  assert (m1d[keep] > 0).all() and (m2d[keep] > 0).all(), "Masses must be positive"
  assert (data[keys['dL']][keep] > 0).all(), "Distances must be positive"
  assert (m2d[keep] <= m1d[keep]).all(), "Primary mass must be >= secondary mass"
  inj_data = {'m1det': m1d[keep], 'm2det': m2d[keep], 'dL': data[keys['dL']][keep]}
  inj_idx = _process_selection(ninj, len(inj_data['m1det']), 'injections')
  result = {k: np.asarray(v[inj_idx]) for k, v in inj_data.items()}
  prior = np.exp(data[keys['log_pdraw']][keep][inj_idx])
  return theta_inj_det(**result, p_draw=prior) if return_struct else (result, prior)


################
# PIXELIZATION #   (reference: CHIMERA/data.py:239-404)
################
theta_pe_pixelated_groups = ['pixels_pe_all_nsides']


def sky_conf_pixels(healpix_pe, sky_conf, nside):
  """The pixels inside the ``sky_conf`` credible area of EVERY event at once (the per-event rule of data.py:239-260): with p the fraction of
  an event's samples per pixel, sorted downwards and summed up, the threshold is the fraction at which the running sum first reaches
  ``sky_conf``; the event keeps every pixel with p >= threshold, in ascending pixel order.

  One ``np.unique`` over the keys ``event * npix + pixel`` of all (event, sample) pairs gives every event's occupied pixels and their counts
  (the map of 12 nside^2 zeros per event the reference fills is never formed); each event's threshold then comes from its own short
  stretch of that array.  Returns a list of E index arrays."""
  pix = np.atleast_2d(np.asarray(healpix_pe, dtype=np.int64))
  E, S = pix.shape
  npix = int(angles.nside2npix(nside))
  keys, counts = np.unique(np.arange(E, dtype=np.int64)[:, None] * npix + pix, return_counts=True)
  bounds = np.searchsorted(keys, np.arange(E + 1, dtype=np.int64) * npix)
  out = []
  for e in range(E):
    a, b = bounds[e], bounds[e + 1]
    occupied, frac = keys[a:b] - e * npix, counts[a:b] / S
    downwards = np.sort(frac)[::-1]
    first = int(np.searchsorted(np.cumsum(downwards), sky_conf))
    if first < downwards.size:
      out.append(occupied[frac >= downwards[first]])
    else:                                                   # the occupied pixels never reach the level: the threshold falls on an empty pixel (p = 0)
      out.append(np.arange(npix, dtype=np.int64))
  return out


def compute_sky_conf_event(healpix_pe, sky_conf, nside):
  """data.py:246-260 for one event: pixels whose sample fraction reaches the sky-confidence threshold."""
  return sky_conf_pixels(np.asarray(healpix_pe)[None, :], sky_conf, nside)[0]


def _pad_arr_list(array_list, pad_value):
  """Ragged rows -> one (rows, longest) array, short rows filled with ``pad_value`` (what data.py:406-420 returns for 1-D rows): a boolean
  mask of the filled cells takes all rows in one assignment."""
  rows = [np.asarray(a) for a in array_list]
  lengths = np.array([r.shape[0] for r in rows])
  padded = np.full((len(rows), int(lengths.max())), pad_value, dtype=rows[0].dtype)
  padded[np.arange(padded.shape[1])[None, :] < lengths[:, None]] = np.concatenate(rows)
  return padded


def gw_loc2d_pdf_at_pixels(ra, dec, ra_pix, dec_pix, npix, device=None):
  """2-D Gaussian KDE of each event's (ra, dec) samples at its pixel centres (data.py:343-345) on the GPU."""
  ra, dec = _lib.as_f64(ra), _lib.as_f64(dec)
  rp, dp = _lib.as_f64(ra_pix), _lib.as_f64(dec_pix)
  E, S = ra.shape
  P = rp.shape[1]
  n = np.ascontiguousarray(npix, dtype=np.int32)
  out = np.full((E, P), -100.)
  dev = _lib.default_device() if device is None else device
  _lib.check(_lib.lib().chm_kde2d_pixels(E, S, P, _lib.dptr(ra), _lib.dptr(dec), _lib.dptr(rp), _lib.dptr(dp), _lib.iptr(n),
                                         _lib.dptr(out), dev))
  return out


def pixelize_gw_catalog(theta_gw, nside_list, mean_npixels_event, sky_conf, nest=False, prefix=None, ret_datastruct=True):
  """data.py:262-392: HEALPix indices of the samples for every nside, per-event optimal nside (number of pixels inside the
  sky-confidence area closest to ``mean_npixels_event``), the event pixels, their centres, the 2-D localisation density at
  the centres (GPU) and the pixel of each sample (samples outside the area go to the nearest event pixel)."""
  num_events = theta_gw.dL.shape[0]
  ra, dec = np.asarray(theta_gw.ra, dtype=np.float64), np.asarray(theta_gw.dec, dtype=np.float64)
  pixels_pe_all_nsides = {}
  for nside in nside_list:
    logger.info(f"Precomputing Healpix pixels (NSIDE={nside}, NEST={nest})")
    pixels_pe_all_nsides[f"nside_{nside}"] = angles.find_pix_RAdec(ra, dec, nside, nest)
  conf_pixels = {nside: sky_conf_pixels(pixels_pe_all_nsides[f"nside_{nside}"], sky_conf, nside) for nside in nside_list}     # every event at once
  pixel_count_matrix = np.array([[len(conf_pixels[nside][e]) for nside in nside_list] for e in range(num_events)])
  best = np.argmin(np.abs(pixel_count_matrix - mean_npixels_event), axis=1)
  opt_nsides = np.array(nside_list)[best]
  event_pixels = [conf_pixels[int(opt_nsides[e])][e] for e in range(num_events)]
  # (the reference calls find_ra_dec without `nest`, data.py:311 -- RING centres for NESTED indices; here the ordering is passed on)
  pixel_ra, pixel_dec = zip(*[angles.find_ra_dec(event_pixels[e], nside=opt_nsides[e], nest=nest) for e in range(num_events)])
  pe_samples_pixels = np.zeros(ra.shape, dtype=np.int64)
  for e in range(num_events):
    sample_pix = pixels_pe_all_nsides[f"nside_{opt_nsides[e]}"][e]
    valid = np.isin(sample_pix, event_pixels[e])
    sep = angles.angular_separation_from_LOS(ra[e][:, None], dec[e][:, None], pixel_ra[e][None, :], pixel_dec[e][None, :])
    pe_samples_pixels[e] = np.where(valid, sample_pix, event_pixels[e][np.argmin(sep, axis=1)])
  padded_event_pixels = _pad_arr_list(event_pixels, pad_value=-100)
  padded_pixel_ra = _pad_arr_list(list(pixel_ra), pad_value=-100.)
  padded_pixel_dec = _pad_arr_list(list(pixel_dec), pad_value=-100.)
  npix = np.array([len(p) for p in event_pixels], dtype=np.int32)
  padded_pixel_probs = gw_loc2d_pdf_at_pixels(ra, dec, padded_pixel_ra, padded_pixel_dec, npix)
  theta_gw_pixelated = theta_gw.update(pixels_pe_all_nsides=pixels_pe_all_nsides, opt_nsides=opt_nsides,
                                       pixels_opt_nsides=padded_event_pixels, ra_pix=padded_pixel_ra, dec_pix=padded_pixel_dec,
                                       gw_loc2d_pdf=padded_pixel_probs, pixels_pe_opt_nside=pe_samples_pixels)
  if prefix is not None:
    print_list = "-".join(map(str, nside_list))
    # the reference writes HDF5 (data.py:368); without h5py the same sets go to .npz
    from .utils import io as _io
    ext = ".h5" if _io.h5py is not None else ".npz"
    fname = prefix + f"_pixelated_nsidelist{print_list}_meanpixels{mean_npixels_event}_skyconf{sky_conf}_nest{nest}{ext}"
    save_set(theta_gw_pixelated, fname, datasets=[d for d in theta_pe_pixelated_datasets if getattr(theta_gw_pixelated, d) is not None],
             groups=theta_pe_pixelated_groups)
  return theta_gw_pixelated


def load_pixelated_gw_catalog(fname):
  """data.py:395-404."""
  if str(fname).endswith(('.h5', '.hdf5')):
    from .utils import io as _io
    _io._need_h5py(fname)
    with _io.h5py.File(fname, 'r') as f:
      avail = set(f.keys())
  else:
    with np.load(fname) as f:
      avail = set(f.files)
  datasets = [d for d in theta_pe_pixelated_datasets if d in avail]
  return load_set(theta_pe_det(), fname, attrs=[], datasets=datasets, groups=theta_pe_pixelated_groups)


def compute_localization_areas(theta, phi, percentile=0.9, unit='deg2'):
  """data.py:426-451."""
  thetas, phis = np.atleast_2d(theta), np.atleast_2d(phi)
  area = np.zeros(thetas.shape[0])
  for e in range(thetas.shape[0]):
    th, ph = thetas[e], phis[e]
    s2t, s2p = np.cov(th, th)[0, 0], np.cov(ph, ph)[0, 0]
    cov2 = np.cov(th, ph)[0, 1]**2
    one_sigma = 2 * np.pi * np.abs(np.sin(np.mean(th))) * np.sqrt(s2t * s2p - cov2)
    area[e] = -np.log(1 - percentile / 100) * one_sigma * (180 / np.pi)**2
  return area


def compute_localization_volumes(theta, phi, dL, cosmo_min, cosmo_max, percentile=90):
  """data.py:452-484 -> localisation volume of every event [Gpc^3]: the event's sky area (``compute_localization_areas``, in steradians) times
  the comoving shell between the redshifts of the lower / upper ``(100 - percentile) / 2`` distance percentiles, the nearer edge under
  ``cosmo_min`` and the farther one under ``cosmo_max``, per unit solid angle:  A (Vc(z_max; cosmo_max) - Vc(z_min; cosmo_min)) / 4 pi.
  (The reference body names undefined objects -- ``flrw.z_from_dGW``, ``flrw.V_at_z``, ``cosmo_param_min`` -- and cannot run; this is what it
  spells out, on ``cosmo.z_from_dGW`` / ``cosmo.Vc_at_z``, which evaluate on the GPU through ``chm_model_eval``.)"""
  from .population import cosmo as _cosmo
  dL = np.atleast_2d(np.asarray(dL, dtype=np.float64))
  steradians = compute_localization_areas(theta, phi, percentile) / (180. / np.pi)**2
  tail = (100. - percentile) / 2.
  d_near, d_far = np.percentile(dL, tail, axis=1), np.percentile(dL, 100. - tail, axis=1)
  v_near = _cosmo.Vc_at_z(cosmo_min, _cosmo.z_from_dGW(cosmo_min, d_near))
  v_far = _cosmo.Vc_at_z(cosmo_max, _cosmo.z_from_dGW(cosmo_max, d_far))
  return steradians * (np.asarray(v_far) - np.asarray(v_near)) / (4. * np.pi)
