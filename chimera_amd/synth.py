"""
Seeded synthetic inputs for the hyper-likelihood path: mock dark-siren events with
posterior samples, a pixelated sky patch per event, a per-pixel galaxy-catalogue redshift prior ``p_cat``
and a detected-injection set.  Pure NumPy, self-contained: it is a *data* generator (the reference's real
inputs are Zenodo HDF5 products, examples/test1dgalaxies.ipynb cell 1, unavailable here) and is not part
of the evaluated path.  Its private flat-LCDM helper exists only to place mock sources at plausible distances.

Field names and layouts follow the reference containers (CHIMERA/data.py:27-59):
padded per-pixel arrays carry the sentinel -100.
"""
import numpy as np

SEED = 20250926
C_KM = 299792.458

CONFIGS = {
  # name: (E, P, Z, S, I, cosmology)
  'C1': dict(E=10, P=1, Z=500, S=4096, I=100_000, pixelated=False),
  'C2': dict(E=100, P=16, Z=500, S=4096, I=100_000, pixelated=True),
  'C3': dict(E=1000, P=32, Z=1000, S=4096, I=100_000, pixelated=True),
  'C4': dict(E=69, P=16, Z=500, S=4096, I=1_000_000, pixelated=True),
  'C5': dict(E=10000, P=32, Z=1000, S=4096, I=100_000, pixelated=True),
}


class _Fid(object):
  """Private fiducial flat-LCDM (H0=70, Om0=0.25) on a fine grid; distances in Gpc."""

  def __init__(self, H0=70., Om0=0.25, zmax=12., n=60_000):
    self.H0, self.Om0 = H0, Om0
    self.dH = C_KM * 1e-3 / H0
    self.z = np.concatenate([[0.], np.logspace(-6, np.log10(zmax), n - 1)])
    E = self.E(self.z)
    self.dC = self.dH * np.concatenate([[0.], np.cumsum(0.5 * (1 / E[1:] + 1 / E[:-1]) * np.diff(self.z))])
    self.dLt = self.dC * (1 + self.z)

  def E(self, z):
    return np.sqrt(self.Om0 * (1 + z)**3 + 1 - self.Om0)

  def dL(self, z):
    return np.interp(z, self.z, self.dLt)

  def z_of_dL(self, dL, H0=None):
    if H0 is None:
      return np.interp(dL, self.dLt, self.z)
    return np.interp(dL * (H0 / self.H0), self.dLt, self.z)

  def dVdz(self, z):
    dC = np.interp(z, self.z, self.dC)
    return 4 * np.pi * self.dH * dC**2 / self.E(z)

  def ddLdz(self, z):
    dC = np.interp(z, self.z, self.dC)
    return dC + self.dH * (1 + z) / self.E(z)


def _md(z, gamma=2.7, kappa=3., zp=2.):
  return (1 + z)**gamma / (1 + ((1 + z) / (1 + zp))**(gamma + kappa))


def _sample_pdf(rng, x, pdf, n):
  cdf = np.concatenate([[0.], np.cumsum(0.5 * (pdf[1:] + pdf[:-1]) * np.diff(x))])
  cdf /= cdf[-1]
  return np.interp(rng.random(n), cdf, x)


def _plp_pdf(m, alpha=3.4, mlow=5.1, mhigh=87., lam=0.039, mu=34., sig=3.6, dm=4.8):
  pl = np.where((m >= mlow) & (m <= mhigh), m**(-alpha), 0.)
  pl /= (mhigh**(1 - alpha) - mlow**(1 - alpha)) / (1 - alpha)
  g = np.exp(-0.5 * ((m - mu) / sig)**2) / (sig * np.sqrt(2 * np.pi))
  x = np.clip(m - mlow, 1e-12, None)
  with np.errstate(all='ignore'):
    sm = np.where(m <= mlow, 0., np.where(m >= mlow + dm, 1., 1. / (1. + np.exp(dm / x + dm / (x - dm - 1e-12)))))
  return ((1 - lam) * pl + lam * g) * sm


def _detected(rng, m1, m2, z, dL, snr_thr=8.):
  """Toy detection: SNR proxy = 36 (Mc_det/25)^(5/6) / dL[Gpc] * Theta, Theta ~ Beta(2,4); kept if above snr_thr."""
  mc = (m1 * m2)**0.6 / (m1 + m2)**0.2 * (1 + z)
  snr = 36. * (mc / 25.)**(5. / 6.) / dL * rng.beta(2., 4., len(z))
  return snr > snr_thr


def _draw_population(rng, fid, n, z_range):
  zz = np.linspace(z_range[0], z_range[1], 4000)
  z = _sample_pdf(rng, zz, _md(zz) / (1 + zz) * fid.dVdz(zz), n)
  mm = np.linspace(5.1, 87., 8000)
  m1 = _sample_pdf(rng, mm, _plp_pdf(mm), n)
  beta, mlow = 1.1, 5.1     # m2 | m1 ~ m2^beta on [mlow, m1] (smoothing ignored for the draw)
  m2 = (mlow**(beta + 1) + rng.random(n) * (m1**(beta + 1) - mlow**(beta + 1)))**(1 / (beta + 1))
  return z, m1, m2


def make_events(E, S, P=None, seed=SEED, z_range=(0.001, 1.25), ragged=False, frac_outside=0.02,
                sigma_dL=0.15, sigma_m=0.05):
  """Posterior samples for E mock events (+ pixelisation when P is not None).

  Returns a dict: m1det, m2det, dL, pe_prior, ra, dec (E,S) f64; and, if pixelated, pixels_pe_opt_nside (E,S) i64,
  pixels_opt_nsides (E,P) i64 [-100 pad], ra_pix, dec_pix, gw_loc2d_pdf (E,P) f64 [-100 pad], neff_pixels (E,) i32;
  plus z_true, m1_true, m2_true for sanity checks.
  """
  rng = np.random.default_rng(seed)
  fid = _Fid()
  z_true, m1, m2 = np.empty(0), np.empty(0), np.empty(0)
  while len(z_true) < E:                      # population draws that pass the same toy detection as the injections
    z_, m1_, m2_ = _draw_population(rng, fid, 40 * E + 1000, z_range)
    det = _detected(rng, m1_, m2_, z_, fid.dL(z_))
    z_true, m1, m2 = (np.concatenate([a, b[det]]) for a, b in ((z_true, z_), (m1, m1_), (m2, m2_)))
  z_true, m1, m2 = z_true[:E], m1[:E], m2[:E]
  dL_true = fid.dL(z_true)

  dL = dL_true[:, None] * (1 + sigma_dL * rng.standard_normal((E, S)))
  dL = np.where(dL < 0.05 * dL_true[:, None], 0.05 * dL_true[:, None], dL)
  m1d = (m1 * (1 + z_true))[:, None] * (1 + sigma_m * rng.standard_normal((E, S)))
  m2d = (m2 * (1 + z_true))[:, None] * (1 + sigma_m * rng.standard_normal((E, S)))
  m1d, m2d = np.maximum(m1d, m2d), np.minimum(m1d, m2d)
  out = dict(m1det=m1d, m2det=m2d, dL=dL, pe_prior=dL**2, z_true=z_true, m1_true=m1, m2_true=m2)

  ra0 = rng.uniform(0.5, 5.5, E)
  dec0 = rng.uniform(-0.8, 0.8, E)
  sig_sky = rng.uniform(0.01, 0.04, E)
  ra = ra0[:, None] + sig_sky[:, None] * rng.standard_normal((E, S)) / np.cos(dec0)[:, None]
  dec = dec0[:, None] + sig_sky[:, None] * rng.standard_normal((E, S))
  ra_true = ra0 + 0.5 * sig_sky * rng.standard_normal(E) / np.cos(dec0)
  dec_true = dec0 + 0.5 * sig_sky * rng.standard_normal(E)
  out.update(ra=ra, dec=dec)
  if P is None:
    return out

  ncol = int(np.ceil(np.sqrt(P)))
  neff = rng.integers(max(1, P // 2), P + 1, E) if ragged else np.full(E, P)
  pixels = np.full((E, P), -100, dtype=np.int64)
  ra_pix = np.full((E, P), -100.)
  dec_pix = np.full((E, P), -100.)
  pdf2d = np.full((E, P), -100.)
  pe_pix = np.empty((E, S), dtype=np.int64)
  host_pix = np.zeros(E, dtype=np.int64)
  for e in range(E):
    n = int(neff[e])
    k = np.arange(n)
    nrow = int(np.ceil(n / ncol))
    step = 3.2 * sig_sky[e] / max(ncol - 1, 1)
    gx = (k % ncol - (ncol - 1) / 2) * step
    gy = (k // ncol - (nrow - 1) / 2) * step
    ra_pix[e, :n] = ra0[e] + gx / np.cos(dec0[e])
    dec_pix[e, :n] = dec0[e] + gy
    pixels[e, :n] = 100_000 * (e + 1) + rng.permutation(5 * P)[:n]          # arbitrary distinct HEALPix-like ids
    dx = (ra[e][:, None] - ra_pix[e, :n][None, :]) * np.cos(dec0[e])
    dy = dec[e][:, None] - dec_pix[e, :n][None, :]
    near = np.argmin(dx**2 + dy**2, axis=1)
    pe_pix[e] = pixels[e, near]
    host_pix[e] = np.argmin(((ra_true[e] - ra_pix[e, :n]) * np.cos(dec0[e]))**2 + (dec_true[e] - dec_pix[e, :n])**2)
    nout = int(frac_outside * S)
    if nout:
      pe_pix[e, rng.choice(S, nout, replace=False)] = 7                      # samples outside the sky-confidence area
    # make sure every pixel keeps at least two samples (an empty pixel is NaN in the reference: likelihood.py:180-192)
    for j in range(n):
      if np.sum(pe_pix[e] == pixels[e, j]) < 2:
        pe_pix[e, rng.choice(S, 2, replace=False)] = pixels[e, j]
    r2 = (gx**2 + gy**2) / sig_sky[e]**2
    pdf2d[e, :n] = np.exp(-0.5 * r2) / (2 * np.pi * sig_sky[e]**2 / np.cos(dec0[e]))
  out.update(pixels_pe_opt_nside=pe_pix, pixels_opt_nsides=pixels, ra_pix=ra_pix, dec_pix=dec_pix,
             gw_loc2d_pdf=pdf2d, neff_pixels=neff.astype(np.int32), host_pix=host_pix,
             opt_nsides=np.where(np.arange(E) % 2 == 0, 64, 128))
  return out


def make_z_grids(dL, Z, H0_prior=(20., 200.)):
  """Per-event uniform z grids bracketing the samples for every H0 in the prior
  (the same construction as CHIMERA/population/pop_wrapper.py:159-207, on the private fiducial cosmology)."""
  fid = _Fid()
  dL_max = dL.max(axis=1) * 2
  dL_min = np.maximum(dL.min(axis=1) * 0.5, 1e-8)
  z_min = fid.z_of_dL(dL_min, H0_prior[0])
  z_max = fid.z_of_dL(dL_max, H0_prior[1])
  t = np.arange(Z) / (Z - 1)
  return z_min[:, None] * (1 - t) + z_max[:, None] * t


def make_p_cat(z_grids, neff_pixels, P, seed=SEED + 1, ngal_mean=30, z_err=0.001, z_lim=(0.073, 1.3),
               z_host=None, host_pix=None):
  """Per-pixel catalogue term: sum of galaxy Gaussians x dV/dz, each normalised on the event grid
  (what CHIMERA/catalog/catalog.py:212-221 produces), -100 padded to P pixels.  Returns (p_cat (E,P,Z), N_gal (E,))."""
  rng = np.random.default_rng(seed)
  fid = _Fid()
  E, Z = z_grids.shape
  p_cat = np.full((E, P, Z), -100.)
  N_gal = np.zeros(E, dtype=np.int64)
  zz = np.linspace(z_lim[0], z_lim[1], 4000)
  pdfz = fid.dVdz(zz)
  for e in range(E):
    zg = z_grids[e]
    dv = fid.dVdz(zg)
    n = int(neff_pixels[e])
    ng = rng.poisson(ngal_mean, n)
    zgal = _sample_pdf(rng, zz, pdfz, int(ng.sum()))
    owner = np.repeat(np.arange(n), ng)
    if z_host is not None:                      # the true host is a catalogue member
      zgal = np.concatenate([zgal, [z_host[e]]])
      owner = np.concatenate([owner, [host_pix[e]]])
    sel = (zgal > zg[0]) & (zgal < zg[-1])
    owner = owner[sel]
    zgal = zgal[sel]
    sig = z_err * (1 + zgal)
    g = np.exp(-0.5 * ((zg[:, None] - zgal[None, :]) / sig[None, :])**2) / (sig[None, :] * np.sqrt(2 * np.pi))
    g *= dv[:, None]
    norm = 0.5 * ((g[1:] + g[:-1]) * np.diff(zg)[:, None]).sum(0)
    with np.errstate(all='ignore'):
      g = g / norm[None, :]
    g[~np.isfinite(g)] = 0.
    for j in range(n):
      m = owner == j
      p_cat[e, j] = g[:, m].sum(1) / max(m.sum(), 1) if m.any() else 0.
    N_gal[e] = len(zgal)
  return p_cat, N_gal


def make_galaxy_sample(ev, z_grids, seed=SEED + 3, ngal_mean=30, z_lim=(0.073, 1.3), frac_foreign=0.2):
  """A galaxy sample in the layout `pixelated_catalog` consumes (CHIMERA/catalog/catalog.py:105-136): redshifts `z` and,
  for every nside in ``ev['opt_nsides']``, the HEALPix-like index of each galaxy (`pix<nside>`; -1 where the galaxy lies
  in no event pixel at that nside).  Galaxies follow dV/dz; a fraction lies outside every event pixel."""
  rng = np.random.default_rng(seed)
  fid = _Fid()
  E, P = ev['pixels_opt_nsides'].shape
  zz = np.linspace(z_lim[0], z_lim[1], 4000)
  pdfz = fid.dVdz(zz)
  nsides = np.unique(ev['opt_nsides'])
  zs, pix = [], {int(ns): [] for ns in nsides}
  for e in range(E):
    n = int(ev['neff_pixels'][e])
    ng = rng.poisson(ngal_mean, n)
    z = _sample_pdf(rng, zz, pdfz, int(ng.sum()))
    ids = np.repeat(ev['pixels_opt_nsides'][e, :n], ng)
    zs.append(z)
    for ns in nsides:
      pix[int(ns)].append(ids if ev['opt_nsides'][e] == ns else np.full(ids.shape, -1, dtype=np.int64))
  nf = int(frac_foreign * sum(len(z) for z in zs))
  zs.append(_sample_pdf(rng, zz, pdfz, nf))
  for ns in nsides:
    pix[int(ns)].append(np.full(nf, -1, dtype=np.int64))
  gal = {'z': np.concatenate(zs)}
  perm = rng.permutation(len(gal['z']))
  gal['z'] = gal['z'][perm]
  for ns in nsides:
    gal[f'pix{int(ns)}'] = np.concatenate(pix[int(ns)])[perm]
  return gal


def make_injections(I, seed=SEED + 2, oversample=40):
  """Detected injections with an analytic draw density.

  Draw: z ~ p(z) prop. to (1+z)^-3 dV/dz on [1e-3, 1.4]; m1 ~ m^-2.35 on [4, 110]; m2|m1 ~ uniform[4, m1];
  detection by the same toy SNR proxy as the events.  p_draw is the density in (m1det, m2det, dL[Gpc]).
  Returns dict(m1det, m2det, dL, p_draw (I,), N_inj = number of draws consumed).
  """
  rng = np.random.default_rng(seed)
  fid = _Fid()
  zz = np.linspace(1e-3, 1.4, 6000)
  pz = (1 + zz)**-3. * fid.dVdz(zz)
  pz /= (0.5 * (pz[1:] + pz[:-1]) * np.diff(zz)).sum()
  a, lo, hi = 2.35, 4., 110.
  keep = dict(m1det=[], m2det=[], dL=[], p_draw=[])
  n_drawn, n_kept = 0, 0
  while n_kept < I:
    n = min(int(oversample * (I - n_kept)) + 1000, 4_000_000)
    z = _sample_pdf(rng, zz, pz, n)
    m1 = (lo**(1 - a) + rng.random(n) * (hi**(1 - a) - lo**(1 - a)))**(1 / (1 - a))
    m2 = lo + rng.random(n) * (m1 - lo)
    dL = fid.dL(z)
    idx = np.flatnonzero(_detected(rng, m1, m2, z, dL))
    if n_kept + len(idx) >= I:                  # stop exactly at I detections; count only the draws consumed
      idx = idx[:I - n_kept]
      n_drawn += idx[-1] + 1
    else:
      n_drawn += n
    z, m1, m2, dL = z[idx], m1[idx], m2[idx], dL[idx]
    p_m1 = m1**(-a) * (1 - a) / (hi**(1 - a) - lo**(1 - a))
    p = p_m1 / (m1 - lo) * np.interp(z, zz, pz) / (fid.ddLdz(z) * (1 + z)**2)
    keep['m1det'].append(m1 * (1 + z))
    keep['m2det'].append(m2 * (1 + z))
    keep['dL'].append(dL)
    keep['p_draw'].append(p)
    n_kept += len(idx)
  out = {k: np.concatenate(v) for k, v in keep.items()}
  out['N_inj'] = float(n_drawn)
  return out


def make_config(name='C3', seed=SEED, ragged=False, E=None, I=None, **override):
  """All arrays for one BASELINE.json configuration (optionally shrunk via E=/I=/S=/P=/Z=)."""
  cfg = dict(CONFIGS[name])
  cfg.update({k: v for k, v in dict(E=E, I=I, **override).items() if v is not None})
  ev = make_events(cfg['E'], cfg['S'], cfg['P'] if cfg['pixelated'] else None, seed=seed, ragged=ragged)
  ev['z_grids'] = make_z_grids(ev['dL'], cfg['Z'])
  if cfg['pixelated']:
    ev['p_cat'], ev['N_gal'] = make_p_cat(ev['z_grids'], ev['neff_pixels'], cfg['P'], seed=seed + 1,
                                          z_host=ev['z_true'], host_pix=ev['host_pix'])
  inj = make_injections(cfg['I'], seed=seed + 2)
  return cfg, ev, inj
