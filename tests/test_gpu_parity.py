"""GPU parity tests: the HIP path (through the C ABI / ctypes) against the NumPy oracle on the same seeded inputs.

Stated fp64 tolerance: tables rtol 1e-13; per-event L_i rtol 1e-9; p_gw rtol 1e-9 (+ tiny atol at
the edge of the Epanechnikov support); log-hyperlikelihood atol 1e-7*sqrt(E).
"""
import os
import numpy as np
import pytest

from tests import helpers as H
from oracle import chimera_oracle as O
import chimera_amd as CH

pytestmark = pytest.mark.gpu

RTOL_L = 1e-9


@pytest.fixture(scope='module')
def cfg_pix():
  return H.small_config(E=6, S=256, P=4, Z=64, I=3000, seed=11, ragged=True)


@pytest.fixture(scope='module')
def cfg_1d():
  return H.small_config(E=5, S=300, Z=80, I=3000, seed=12, pixelated=False)


def test_tables_and_model_functions():
  import chimera_amd as CH
  for cname, kw in [('flrw', dict(H0=67., Om0=0.31, z_max=5.)), ('flrw', dict(H0=80., Om0=0.3, Ok0=0.05, w0=-0.9, wa=0.2)),
                    ('flrw', dict(H0=60., Om0=0.3, Ok0=-0.04, Or0=1e-4)), ('mg_flrw', dict(H0=70., Xi0=1.8, n=1.9, z_max=5.))]:
    co, cp = getattr(O, cname)(**kw), getattr(CH.cosmo, cname)(**kw)
    np.testing.assert_allclose(cp.z_grid_interp, co.z_grid_interp, rtol=1e-13, atol=0)
    np.testing.assert_allclose(cp.integral_invE_interp, co.integral_invE_interp, rtol=1e-13, atol=1e-300)
    z = np.concatenate([[0., 1e-12, 1e-5], np.linspace(0.001, co.z_max * 1.05, 301)])
    dL = O.dL_at_z(co, z)
    for fn in ('E_at_z', 'int_invE_at_z', 'dCr_at_z', 'dCt_at_z', 'dL_at_z', 'ddLdz_at_z', 'dVcdz_at_z'):
      np.testing.assert_allclose(getattr(CH.cosmo, fn)(cp, z), getattr(O, fn)(co, z), rtol=2e-13, atol=1e-300, err_msg=fn)
    for fn in ('ddLdz_at_z', 'dVcdz_at_z'):
      np.testing.assert_allclose(getattr(CH.cosmo, fn)(cp, z[3:], dL[3:]), getattr(O, fn)(co, z[3:], dL[3:]), rtol=2e-13, err_msg=fn)
    # Vc: the curved-space expression (cosmo.py:178-184) subtracts two nearly equal terms, so one ulp in asinh/asin is
    # amplified by ~(dH/(sqrt|Ok0| dCt))^2: compare it in absolute terms there, to rtol in the flat case.
    vtol = dict(rtol=2e-13, atol=1e-300) if co.Ok0 == 0 else dict(rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(CH.cosmo.Vc_at_z(cp, z), O.Vc_at_z(co, z), err_msg='Vc_at_z', **vtol)
    np.testing.assert_allclose(CH.cosmo.Vc_at_z(cp, z[3:], dL[3:]), O.Vc_at_z(co, z[3:], dL[3:]), err_msg='Vc_at_z(d)', **vtol)
    d = np.concatenate([[0., 1e-9], np.linspace(0.01, 1.2 * dL.max(), 400)])
    np.testing.assert_allclose(CH.cosmo.z_from_dGW(cp, d), O.z_from_dGW(co, d), rtol=2e-13, atol=1e-300)
  m1 = np.linspace(3., 100., 389)
  m2 = m1 * np.linspace(0.05, 1.1, 389)
  for mname, kw in [('plp', {}), ('plp', dict(alpha=2.1, mu_g=30., delta_m=3.)), ('tpl', {}), ('bpl', {}), ('tpl', dict(alpha=1.0))]:
    mo, mp = getattr(O, mname)(**kw), getattr(CH.mass, mname)(**kw)
    np.testing.assert_allclose(mp.m_grid, mo.m_grid, rtol=1e-13)
    # the first entries sit in the smoothing window, where exp(-delta_m/(m - m_low)...) amplifies one ulp of m_grid
    np.testing.assert_allclose(mp.cdf_m2_conditioned, mo.cdf_m2_conditioned, rtol=1e-13, atol=1e-16 * mo.cdf_m2_conditioned[-1])
    np.testing.assert_allclose(mp.norm_p_m1, mo.norm_p_m1, rtol=1e-13)
    np.testing.assert_allclose(CH.mass.p_m1m2(mp, m1, m2), O.p_m1m2(mo, m1, m2), rtol=1e-12, atol=1e-300, err_msg=mname)
    np.testing.assert_allclose(CH.mass.primary_mass_pdf_notnorm(mp, m1), O.primary_mass_pdf_notnorm(mo, m1), rtol=1e-12, atol=1e-300)
    # the reduced-operation form used inside the per-sample kernels (p_m1m2_fused), incl. the edges of every window
    from chimera_amd.population._base import make_params, model_eval
    from chimera_amd import _lib
    edge = np.array([mo.m_low, mo.m_low + getattr(mo, 'delta_m', 0.), mo.m_high, np.nextafter(mo.m_low, 0.), np.nextafter(mo.m_high, 1e3),
                     getattr(mo, 'mu_g', 30.) + 5. * getattr(mo, 'sigma_g', 1.)])
    M1 = np.concatenate([m1, np.repeat(edge, edge.size), edge, edge])
    M2 = np.concatenate([m2, np.tile(edge, edge.size), edge * 0.999999, edge * 0.5])
    pf = model_eval(make_params(mass=mp), _lib.F_PM1M2_FUSED, M1, M2)
    np.testing.assert_allclose(pf, O.p_m1m2(mo, M1, M2), rtol=1e-12, atol=1e-300, err_msg=mname + ' (fused)')
  z = np.linspace(0., 3., 200)
  for rname in ('power_law', 'madau_dickinson', 'trunc_power_law', 'trunc_madau_dickinson'):
    np.testing.assert_allclose(CH.rate.merger_rate(getattr(CH.rate, rname)(), z), O.merger_rate(getattr(O, rname)(), z), rtol=1e-13)
  comp_o, comp_p = O.dVdz_completeness(), CH.completeness.dVdz_completeness()
  np.testing.assert_allclose(comp_p.fR(CH.cosmo.flrw(H0=67.)), comp_o.fR(O.flrw(H0=67.)), rtol=1e-12)


def _compare(like_p, like_o, lam, E, check_pgw=True):
  ro = like_o.compute_all(**lam)
  rp = like_p.compute_all(**lam)
  H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
  if np.isfinite(ro[3]):
    np.testing.assert_allclose(rp[1], ro[1], rtol=0, atol=1e-7 * np.sqrt(E))
    np.testing.assert_allclose(rp[2], ro[2], rtol=1e-10)
    np.testing.assert_allclose(rp[3], ro[3], rtol=0, atol=1e-7 * np.sqrt(E))
    np.testing.assert_allclose(like_p(**lam), ro[3], rtol=0, atol=1e-7 * np.sqrt(E))
  if check_pgw:
    pop_o, pop_p = like_o.population.update(**lam), like_p.population.update(**lam)
    go = like_o.p_gw3d(pop_o) if like_o.pixelated else like_o.p_gw1d(pop_o)
    gp = like_p.p_gw3d(pop_p) if like_p.pixelated else like_p.p_gw1d(pop_p)
    if like_o.pixelated:        # padded pixels (p >= neff_pixels) are masked out of the integrand (likelihood.py:274-277):
      valid = np.arange(go.shape[1])[None, :] < np.asarray(like_o.neff_pixels)[:, None]       # compare the real ones
      go, gp = go[valid], gp[valid]
    fin = np.isfinite(go)
    assert np.array_equal(fin, np.isfinite(gp))
    np.testing.assert_allclose(gp[fin], go[fin], rtol=1e-9, atol=1e-9 * np.max(np.abs(go[fin])))
    no, npp = like_o.compute_numlike_evs(pop_o), like_p.compute_numlike_evs(pop_p)
    f2 = np.isfinite(no)
    np.testing.assert_allclose(npp[f2], no[f2], rtol=RTOL_L, atol=1e-300)


@pytest.mark.parametrize('kind', ['marginalized', 'approximate', 'full'])
def test_pixelated_modes(cfg_pix, kind):
  cfg, ev, inj = cfg_pix
  like_o, _, _ = H.build_oracle(ev, inj, kind=kind)
  like_p, _, _ = H.build_product(ev, inj, kind=kind)
  for lam in (dict(H0=70.), dict(H0=55., alpha=3.0), dict(H0=95., gamma=2.0, mu_g=32.)):
    _compare(like_p, like_o, lam, cfg['E'])


@pytest.mark.parametrize('kernel', ['epan', 'gauss'])
def test_1d_mode(cfg_1d, kernel):
  cfg, ev, inj = cfg_1d
  kw = dict(kernel=kernel)
  like_o, _, _ = H.build_oracle(ev, inj, pixelated=False, like_kw=kw)
  like_p, _, _ = H.build_product(ev, inj, pixelated=False, like_kw=kw)
  for lam in (dict(H0=70.), dict(H0=110., lambda_peak=0.1)):
    _compare(like_p, like_o, lam, cfg['E'])


@pytest.mark.parametrize('like_kw', [dict(binning=False), dict(bw_method='silverman'), dict(bw_method=0.3),
                                     dict(cut_grid=None), dict(num_bins=37, cut_grid=1.0), dict(kernel='gauss')])
def test_kde_options(cfg_pix, like_kw):
  cfg, ev, inj = cfg_pix
  for kind in ('marginalized', 'approximate'):
    like_o, _, _ = H.build_oracle(ev, inj, kind=kind, like_kw=like_kw)
    like_p, _, _ = H.build_product(ev, inj, kind=kind, like_kw=like_kw)
    _compare(like_p, like_o, dict(H0=72.), cfg['E'])


@pytest.mark.parametrize('models', [dict(cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.8, n=1.9)), dict(mass='tpl'), dict(mass='bpl'),
                                    dict(rate='power_law'), dict(rate='trunc_madau_dickinson', rate_kw=dict(zmax=2.5)),
                                    dict(cosmo_kw=dict(Ok0=0.03, w0=-0.95, wa=0.1))])
def test_model_families(cfg_pix, models):
  cfg, ev, inj = cfg_pix
  like_o, _, _ = H.build_oracle(ev, inj, models=models)
  like_p, _, _ = H.build_product(ev, inj, models=models)
  _compare(like_p, like_o, dict(H0=68.), cfg['E'], check_pgw=False)


def test_not_scale_free_and_neff_guards(cfg_pix):
  cfg, ev, inj = cfg_pix
  pk = dict(scale_free=False, R0=20., Tobs=2.)
  like_o, _, _ = H.build_oracle(ev, inj, pop_kw=pk)
  like_p, _, _ = H.build_product(ev, inj, pop_kw=pk)
  _compare(like_p, like_o, dict(H0=70., R0=25.), cfg['E'], check_pgw=False)
  # pe_neff guard: a huge threshold zeroes every event -> L_i = 0 -> -1.797e308 each -> sum = -inf  
  like_o, _, _ = H.build_oracle(ev, inj, like_kw=dict(pe_neff=1e9))
  like_p, _, _ = H.build_product(ev, inj, like_kw=dict(pe_neff=1e9))
  ro, rp = like_o.compute_all(H0=70.), like_p.compute_all(H0=70.)
  assert np.all(rp[0] == -np.finfo(np.float64).max) and np.all(ro[0] == rp[0])
  assert rp[1] == -np.inf and ro[1] == -np.inf
  # N_eff guard: N_exp = 0 when the injections are too few effective samples   (selection_function.py:43-47)
  like_o, pop_o, sel_o = H.build_oracle(ev, inj, N_eff=1e12)
  like_p, pop_p, sel_p = H.build_product(ev, inj, N_eff=1e12)
  assert sel_o.N_exp(pop_o) == 0. and sel_p.N_exp(pop_p) == 0.
  like_o, pop_o, sel_o = H.build_oracle(ev, inj, N_eff=None)
  like_p, pop_p, sel_p = H.build_product(ev, inj, N_eff=None)
  np.testing.assert_allclose(sel_p.N_exp(pop_p), sel_o.N_exp(pop_o), rtol=1e-11)


def test_batch_matches_single(cfg_pix):
  cfg, ev, inj = cfg_pix
  like_p, _, _ = H.build_product(ev, inj)
  lams = [dict(H0=h) for h in (50., 60., 70., 80., 90.)]
  single = np.array([like_p(**l) for l in lams])
  batched = like_p.batch(lams)
  np.testing.assert_array_equal(batched, single)


def test_p_gw1d_of_a_pixelated_likelihood(cfg_pix):
  """hyperlikelihood.p_gw1d on an object built with a galaxy catalogue (likelihood.py:105-144 needs the samples only): the 1-D
  handle carries no completeness array -- this used to dereference a null pointer on the device."""
  cfg, ev, inj = cfg_pix
  for kind in ('approximate', 'marginalized'):
    like_o, pop_o, _ = H.build_oracle(ev, inj, kind=kind)
    like_p, pop_p, _ = H.build_product(ev, inj, kind=kind)
    a, b = like_o.p_gw1d(pop_o.update(H0=71.)), like_p.p_gw1d(pop_p.update(H0=71.))
    np.testing.assert_allclose(b, a, rtol=1e-9, atol=1e-9 * np.max(a))


def test_long_batches_are_sliced(cfg_pix):
  cfg, ev, inj = cfg_pix
  like, _, _ = H.build_product(ev, inj)
  lams = [dict(H0=float(h)) for h in np.linspace(60., 80., 23)]
  whole = like.batch(lams)
  like.max_draws_per_call = 5                     # 23 draws -> slices of 5, 5, 5, 5, 3
  np.testing.assert_array_equal(like.batch(lams), whole)
  np.testing.assert_array_equal(like(H0=np.linspace(60., 80., 23)), whole)


def test_compute_z_grids(cfg_pix):
  import chimera_amd as CH
  cfg, ev, inj = cfg_pix
  th_o, th_p = O.theta_pe_det(dL=ev['dL']), CH.data.theta_pe_det(dL=ev['dL'])
  for cname, kw, prior in [('flrw', dict(H0=70., Om0=0.25, z_max=5.), {'H0': [20, 200]}),
                           ('mg_flrw', dict(H0=70., z_max=5.), {'H0': [40, 120], 'Xi0': [0.5, 3.], 'n': [0.5, 3.]})]:
    zo = O.compute_z_grids(getattr(O, cname)(**kw), th_o, cosmo_prior=prior, z_int_res=50)
    zp = CH.compute_z_grids(getattr(CH.cosmo, cname)(**kw), th_p, cosmo_prior=prior, z_int_res=50)
    np.testing.assert_allclose(zp, zo, rtol=1e-12)


# ----------------------------------------------------------------------------------------------------------
# sharding (device partials) and RCCL plumbing on one GPU
# ----------------------------------------------------------------------------------------------------------
class _FakeRank(object):
  """Shard selector without a communicator: chm_eval then returns this shard's partial sums un-reduced."""
  handle = None

  def __init__(self, nranks, rank):
    self.nranks, self.rank, self.device = nranks, rank, 0


@pytest.mark.parametrize('kind', ['marginalized', 'full'])
def test_sharded_partials_add_up(cfg_pix, kind):
  cfg, ev, inj = cfg_pix
  like, pop, sel = H.build_product(ev, inj, kind=kind)
  lam = dict(H0=66.)
  whole = like._eval([like.population.update(**lam)], want=('partials', 'log_like_evs'))
  parts, evs = [], []
  for r in range(3):
    lk, _, _ = H.build_product(ev, inj, kind=kind, comm=_FakeRank(3, r))
    res = lk._eval([lk.population.update(**lam)], want=('partials', 'log_like_evs'))
    parts.append(res['partials'][0]); evs.append(res['log_like_evs'][0])
  np.testing.assert_array_equal(np.concatenate(evs), whole['log_like_evs'][0])     # same events, same device arithmetic
  np.testing.assert_allclose(np.sum(parts, axis=0), whole['partials'][0], rtol=1e-13)
  like_o, pop_o, sel_o = H.build_oracle(ev, inj, kind=kind)
  got = H.combine_partials(np.sum(parts, axis=0), cfg['E'], pop_o.update(**lam), inj['N_inj'], 5.)
  np.testing.assert_allclose(got, like_o(**lam), rtol=0, atol=1e-7 * np.sqrt(cfg['E']))


@pytest.mark.parametrize('kind', ['marginalized', 'approximate', 'full', None])
def test_event_groups_on_two_streams_give_the_one_lane_values(kind):
  """[r3] Shards of >= 500 events are evaluated in event groups that alternate between two streams when a call carries more than 8 draws
  (the sample stage of one group beside the GW kernel of the previous one).  The values must be those of the same call with one group
  (set_option('groups', 1)), bit for bit -- every event's log-likelihood and the hyper-likelihood -- and agree with the C restatement."""
  from oracle import oracle_c as OC
  pix = kind is not None
  cfg, ev, inj = H.small_config(E=640, S=96, P=3, Z=40 if kind != 'full' else 400, I=4000, seed=31, ragged=True, pixelated=pix)
  like, _, _ = H.build_product(ev, inj, pixelated=pix, kind=kind)
  lams = [dict(H0=float(h)) for h in np.linspace(62., 78., 11)]
  pops = [like.population.update(**l) for l in lams]
  grouped = like._eval(pops, want=('log_like_evs',))
  like.set_option('groups', 1)
  one = like._eval(pops, want=('log_like_evs',))
  like.set_option('groups', 0)
  np.testing.assert_array_equal(grouped['log_like_evs'], one['log_like_evs'])
  np.testing.assert_array_equal(grouped['log_hyper'], one['log_hyper'])
  np.testing.assert_array_equal(like.batch(lams), one['log_hyper'])
  like_o, _, _ = H.build_oracle(ev, inj, pixelated=pix, kind=kind)
  ref = OC.compute_all(like_o, lams[4], nthreads=8)
  H.assert_loglike_close(grouped['log_like_evs'][4], ref[0], rtol=RTOL_L, atol=1e-9)
  np.testing.assert_allclose(grouped['log_hyper'][4], ref[3], rtol=0, atol=1e-7 * np.sqrt(cfg['E']))


@pytest.mark.parametrize('case', ['wide_kernels', 'narrow_kernels', 'jittered_grid', 'many_samples', 'coarse_grid', 'long_stretch'])
def test_full_mode_sample_stationary_and_general_kernels(case):
  """[r3] kind_p_gw3d='full' runs the sample-stationary KDE kernel (k_full_kde_chain: the power sums of a sample are carried from one chunk of
  the grid to the next, no exp per chunk) and leaves to the general kernel the pixels it cannot do.  Cases: Scott bandwidth (every sample
  starts at the first chunk); a bandwidth of 0.12 on a 900-point grid (the stretch inside the mask spans ~100 kernel widths: most samples
  start at a later chunk -- the 'waiting' path -- and leave the 37-width window again); a grid jittered by 1e-7 of its step (not uniform:
  general kernel); 4500 samples per event (more than a thread block keeps in registers: walked in two sets); a 60-point grid with narrow kernels
  (a chunk spans > 15 widths: general kernel, one exp per pair); a 5000-point grid (> 1024 points inside the mask: general kernel).  Every case against the NumPy oracle (all pairs, one exp each) to the stated
  1e-9, with the kernel that ran checked through chm_like_full_general_pixels (the two kernels against each other: tests/test_zz_variant_builds.py,
  which needs the diagnostic build of the library)."""
  kw = dict(E=3, S=700, P=3, Z=900, I=1500, seed=41)
  like_kw = {}
  if case == 'narrow_kernels':
    like_kw = dict(bw_method=0.12)
  if case == 'many_samples':
    kw.update(S=4500, E=2, P=2, Z=600)
  if case == 'coarse_grid':
    kw.update(Z=60); like_kw = dict(bw_method=0.05)
  if case == 'long_stretch':                         # > 1024 grid points inside the mask (the stretch does not fit the kernel's LDS rows): general kernel
    kw.update(Z=5000, E=2, P=2, S=300)
  cfg, ev, inj = H.small_config(ragged=True, **kw)
  if case == 'jittered_grid':
    zg = np.array(ev['z_grids'])
    dz = np.diff(zg, axis=1).mean(axis=1, keepdims=True)
    zg[:, 1:-1] += 1e-7 * dz * np.random.default_rng(5).uniform(-1., 1., size=zg[:, 1:-1].shape)
    ev = dict(ev, z_grids=zg)
  like_o, _, _ = H.build_oracle(ev, inj, kind='full', like_kw=like_kw)
  like_p, _, _ = H.build_product(ev, inj, kind='full', like_kw=like_kw)
  lam = dict(H0=69.)
  _compare(like_p, like_o, lam, cfg['E'])
  like_p._eval([like_p.population.update(**lam)], want=('log_like_evs',))
  general = like_p.full_general_pixels(1)
  npix = int(np.sum(ev['neff_pixels']))
  if case in ('wide_kernels', 'narrow_kernels', 'many_samples'):
    assert general == 0, (case, general, npix)
  else:
    assert general == npix, (case, general, npix)


def test_rccl_single_rank_communicator(cfg_pix):
  from chimera_amd.parallel import Comm
  cfg, ev, inj = cfg_pix
  comm = Comm(1, 0, device=0)
  from chimera_amd import _lib
  assert _lib.lib().chm_comm_nranks(comm.handle) == 1
  x = np.array([1.5, -2.25, 1e300])
  np.testing.assert_array_equal(comm.allreduce_sum(x), x)
  like, _, _ = H.build_product(ev, inj, comm=comm)
  ref, _, _ = H.build_product(ev, inj)
  assert like(H0=70.) == ref(H0=70.)
  # [r3] the scalar call keeps its HIP-graph replay under a communicator: the first call of a configuration runs eagerly, the second is
  # captured (the graph ends at the rank's partial sums, ncclAllReduce + k_combine follow on the stream), later ones are replayed --
  # every one of them must give the single-process value to the last bit, and the batched call the same numbers
  hs = [66., 71.5, 68.25, 74., 69.125, 72.75]
  got = [like(H0=h) for h in hs]
  want = [ref(H0=h) for h in hs]
  assert got == want, (got, want)
  np.testing.assert_array_equal(like.batch([dict(H0=h) for h in hs]), np.asarray(want))
  comm.close()


def test_two_evaluations_in_flight_on_lanes_of_one_handle(cfg_pix):
  """[r3] hyperlikelihood.lane(): a second evaluation lane on the same resident data (chm_like_clone / chm_sel_clone).  Every lane
  gives the parent's numbers to the last bit -- scalar calls (graph replay), batches, per-event outputs -- also when two host threads
  keep one evaluation per lane in flight at the same time, and the shared arrays outlive whichever handle is destroyed first."""
  from concurrent.futures import ThreadPoolExecutor
  cfg, ev, inj = cfg_pix
  like, _, _ = H.build_product(ev, inj)
  hs = np.linspace(62., 78., 12)
  want = np.array([like(H0=float(h)) for h in hs])
  lane = like.lane()
  assert lane._handles and lane.selection_function is not like.selection_function
  got = np.array([lane(H0=float(h)) for h in hs])
  np.testing.assert_array_equal(got, want)
  np.testing.assert_array_equal(lane.batch([dict(H0=float(h)) for h in hs]), want)
  np.testing.assert_array_equal(lane.compute_all(H0=70.)[0], like.compute_all(H0=70.)[0])
  # two threads, one lane each, many overlapping calls: scalar calls on one, batches on the other
  with ThreadPoolExecutor(max_workers=2) as ex:
    for rep in range(6):
      fa = ex.submit(lambda: [like(H0=float(h)) for h in hs])
      fb = ex.submit(lambda: lane.batch([dict(H0=float(h)) for h in hs[::-1]]))
      np.testing.assert_array_equal(np.array(fa.result()), want)
      np.testing.assert_array_equal(fb.result(), want[::-1])
  # the parent goes first: the lane keeps the uploaded arrays alive
  like.selection_function.close(); like.close()
  np.testing.assert_array_equal(np.array([lane(H0=float(h)) for h in hs]), want)
  lane.selection_function.close(); lane.close()


def test_options_select_streams_and_launch_paths_not_results(cfg_pix):
  """[r4] chm_like_set_option: one stream / one event group / no graph replay / no timing events / no spinning wait are choices of scheduling --
  every combination returns the default's values bit for bit (scalar calls, batches beyond the few-draw limit, per-event outputs); unknown
  options and out-of-range values raise ValueError; a clone starts with its source's options."""
  cfg, ev, inj = cfg_pix
  like, _, _ = H.build_product(ev, inj)
  hs = np.linspace(61., 79., 11)
  lams = [dict(H0=float(h)) for h in hs]
  want_b = like.batch(lams)
  want_s = np.array([like(H0=float(h)) for h in hs[:4]])
  want_e = like.compute_all(H0=70.)[0]
  for name, value, back in (('serial', 1, 0), ('groups', 1, 0), ('groups', 3, 0), ('graph_max_nb', 0, 8), ('timing', 0, 1), ('timing', 2, 1), ('spin_wait', 0, 1)):
    like.set_option(name, value)
    np.testing.assert_array_equal(like.batch(lams), want_b, err_msg=name)
    np.testing.assert_array_equal(np.array([like(H0=float(h)) for h in hs[:4]]), want_s, err_msg=name)
    np.testing.assert_array_equal(like.compute_all(H0=70.)[0], want_e, err_msg=name)
    like.set_option(name, back)
  assert like.last_timing()[0] > 0.
  like.set_option('timing', 0); like.batch(lams)
  assert np.all(like.last_timing()[:7] == 0.)
  like.set_option('timing', 1)
  with pytest.raises(ValueError):
    like.set_option('no_such_option', 1)
  with pytest.raises(ValueError):
    like.set_option('groups', 129)
  with pytest.raises(ValueError):
    like.set_option('fused', 7)
  like.set_option('serial', 1)
  lane = like.lane()
  assert lane._options == like._options and lane._options['serial'] == 1
  np.testing.assert_array_equal(lane.batch(lams), want_b)
  lane.selection_function.close(); lane.close()
  like.close()


def test_ticketed_collectives_of_two_lanes_keep_their_order(cfg_pix):
  """[r4] Two evaluation lanes, each with a (one-rank) RCCL communicator of its own and a host thread of its own: chm_comm_set_ticket makes the lanes
  hand their all-reduces to the device in ticket order.  The thread holding the HIGHER ticket is started first and must wait for the lower one;
  values equal the communicator-free ones bit for bit; a call that fails before its collective passes the turn on instead of blocking the next."""
  import threading, time
  from chimera_amd.parallel import Comm
  cfg, ev, inj = cfg_pix
  ref, _, _ = H.build_product(ev, inj)
  c0, c1 = Comm(1, 0, device=0), Comm(1, 0, device=0)
  like, _, _ = H.build_product(ev, inj, comm=c0)
  lane = like.lane(comm=c1)
  hs = np.linspace(62., 78., 8)
  lams = [dict(H0=float(h)) for h in hs]
  want = ref.batch(lams)
  Comm.reset_tickets(0)
  order, res = [], {}

  def run(ln, comm, ticket, delay):
    time.sleep(delay)
    comm.set_ticket(ticket)
    res[ticket] = ln.batch(lams)
    order.append(ticket)
  for rep in range(3):
    ta = threading.Thread(target=run, args=(lane, c1, 2 * rep + 1, 0.))          # the later ticket starts first ...
    tb = threading.Thread(target=run, args=(like, c0, 2 * rep, 0.15))            # ... and waits for the earlier one
    ta.start(); tb.start(); ta.join(60); tb.join(60)
    assert not ta.is_alive() and not tb.is_alive()
    assert order[-2:] == [2 * rep, 2 * rep + 1], order
    np.testing.assert_array_equal(res[2 * rep], want); np.testing.assert_array_equal(res[2 * rep + 1], want)
  # a ticketed call that fails in its argument checks (nb = 0 is refused) still passes the turn on
  from chimera_amd import _lib
  import ctypes as C
  c0.set_ticket(6)
  out = _lib.chm_out()
  assert _lib.lib().chm_eval(like._handle(), None, c0.handle, like._params_array(lams), 0, cfg['E'], C.byref(out)) == _lib.CHM_E_ARG
  c1.set_ticket(7)
  np.testing.assert_array_equal(lane.batch(lams), want)                          # (would hang if ticket 6 had been lost)
  lane.selection_function.close(); lane.close(); like.selection_function.close(); like.close(); ref.close()
  c0.close(); c1.close()


def _hostcomm_worker(rank, world, addr, outdir):
  import os, sys
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  from chimera_amd.parallel import HostComm, Rendezvous
  from tests import helpers as HH
  rd = Rendezvous(world, rank, address=addr, timeout=120.)
  cfg, ev, inj = HH.small_config(E=9, S=300, P=4, Z=60, I=3001, seed=41, ragged=True)
  comm = HostComm(world, rank, device=0, rendezvous=rd)
  out = []
  for pop_kw in ({}, dict(scale_free=False, R0=12., Tobs=1.5)):
    like, pop, sel = HH.build_product(ev, inj, comm=comm, pop_kw=pop_kw)
    lams = [dict(H0=64.), dict(H0=70., alpha=3.0), dict(H0=81., gamma=2.0)]
    out.append(np.concatenate([like.batch(lams), [like(**lams[1]), sel.N_exp(pop.update(**lams[2]))]]))
    like.close(); sel.close()
  np.save(os.path.join(outdir, f'rank{rank}.npy'), np.array(out))
  rd.barrier()
  rd.close()


def _spawn(target, world, args):
  import multiprocessing as mp
  ctx = mp.get_context('spawn')                              # fresh interpreters: the parent has initialised the GPU
  procs = [ctx.Process(target=target, args=(r, world) + tuple(args)) for r in range(world)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(280)
    assert p.exitcode == 0, f"rank process exited with {p.exitcode}"


@pytest.mark.timeout(300)
def test_two_ranks_share_one_gpu_through_the_host_communicator(tmp_path):
  """The N > 1 path end to end on the device: two processes shard events and injections (both on GPU 0), reduce the three
  partial sums per draw through the host communicator (Rendezvous sockets) and must each obtain the single-process result."""
  _spawn(_hostcomm_worker, 2, (str(tmp_path / 'rdzv.sock'), str(tmp_path)))
  r0, r1 = np.load(tmp_path / 'rank0.npy'), np.load(tmp_path / 'rank1.npy')
  np.testing.assert_array_equal(r0, r1)
  cfg, ev, inj = H.small_config(E=9, S=300, P=4, Z=60, I=3001, seed=41, ragged=True)
  for k, pop_kw in enumerate(({}, dict(scale_free=False, R0=12., Tobs=1.5))):
    like, pop, sel = H.build_product(ev, inj, pop_kw=pop_kw)
    lams = [dict(H0=64.), dict(H0=70., alpha=3.0), dict(H0=81., gamma=2.0)]
    ref = np.concatenate([like.batch(lams), [like(**lams[1]), sel.N_exp(pop.update(**lams[2]))]])
    np.testing.assert_allclose(r0[k], ref, rtol=1e-12)


def _params_worker(rank, world, addr, outdir):
  import os, sys
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import chimera_amd as CH
  from chimera_amd.parallel import HostComm, Rendezvous
  from tests import helpers as HH
  rd = Rendezvous(world, rank, address=addr, timeout=120.)
  cfg, ev, inj = HH.small_config(E=9, S=300, P=4, Z=60, I=3001, seed=41, ragged=True)
  comm = HostComm(world, rank, device=0, rendezvous=rd)
  like0, pop, sel = HH.build_product(ev, inj)                # replicas: the selection function carries no communicator
  like = CH.hyperlikelihood(like0.theta_gw_det, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', comm=comm, scheme='params')
  lams = [dict(H0=float(h)) for h in np.linspace(62., 79., 5)]
  out = np.concatenate([like.batch(lams), [like(H0=70.5)], like(H0=np.array([66., 67., 68.]))])
  np.save(os.path.join(outdir, f'rank{rank}.npy'), out)
  like.close(); sel.close()
  rd.barrier()
  rd.close()


@pytest.mark.timeout(300)
def test_params_scheme_two_replicas_share_one_gpu(tmp_path):
  """[r4] scheme='params' (CHIMERA/parallel.py:258-278) end to end on the device: two processes, each with ALL events and injections on GPU 0,
  evaluate their own chunk of the draws of a batch and gather the values over the host communicator -- every rank the single-process values,
  bit for bit (each value is computed by exactly one rank)."""
  _spawn(_params_worker, 2, (str(tmp_path / 'rdzv.sock'), str(tmp_path)))
  r0, r1 = np.load(tmp_path / 'rank0.npy'), np.load(tmp_path / 'rank1.npy')
  np.testing.assert_array_equal(r0, r1)
  cfg, ev, inj = H.small_config(E=9, S=300, P=4, Z=60, I=3001, seed=41, ragged=True)
  like, _, _ = H.build_product(ev, inj)
  lams = [dict(H0=float(h)) for h in np.linspace(62., 79., 5)]
  ref = np.concatenate([like.batch(lams), [like(H0=70.5)], like(H0=np.array([66., 67., 68.]))])
  np.testing.assert_array_equal(r0, ref)


def _both_worker(rank, world, addr, outdir):
  import os, sys
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import chimera_amd as CH
  from chimera_amd.parallel import HostComm, Rendezvous, split
  from tests import helpers as HH
  rd = Rendezvous(world, rank, address=addr, timeout=120.)
  cfg, ev, inj = HH.small_config(E=9, S=300, P=4, Z=60, I=3001, seed=41, ragged=True)
  wcomm = HostComm(world, rank, device=0, rendezvous=rd)
  grp = split(wcomm, 2)                                       # two parameter batches of world / 2 ranks
  like0, pop, sel = HH.build_product(ev, inj, comm=grp)       # events and injections sharded over the GROUP
  like = CH.hyperlikelihood(like0.theta_gw_det, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', comm=grp, scheme='both')
  lams = [dict(H0=float(h)) for h in np.linspace(62., 79., 5)]
  out = np.concatenate([like.batch(lams), [like(H0=70.5)], like(H0=np.array([66., 67., 68.]))])
  np.save(os.path.join(outdir, f'rank{rank}.npy'), out)
  like.close(); like0.close(); sel.close()
  rd.barrier()
  grp.close()
  rd.close()


@pytest.mark.timeout(300)
def test_both_scheme_two_groups_of_two_ranks_share_one_gpu(tmp_path):
  """[r4] scheme='both' (CHIMERA/parallel.py:132-224, 380-406) end to end on the device: four processes on GPU 0 form two groups of two; a group
  shards events and injections over its ranks (partial sums reduced over the group's own socket star) and evaluates its slice of the draws;
  the world assembles the values.  Every rank holds the single-process values (to rounding: the shards' sums are added in another order)."""
  _spawn(_both_worker, 4, (str(tmp_path / 'rdzv.sock'), str(tmp_path)))
  rs = [np.load(tmp_path / f'rank{r}.npy') for r in range(4)]
  for r in rs[1:]:
    np.testing.assert_array_equal(r, rs[0])
  cfg, ev, inj = H.small_config(E=9, S=300, P=4, Z=60, I=3001, seed=41, ragged=True)
  like, _, _ = H.build_product(ev, inj)
  lams = [dict(H0=float(h)) for h in np.linspace(62., 79., 5)]
  ref = np.concatenate([like.batch(lams), [like(H0=70.5)], like(H0=np.array([66., 67., 68.]))])
  np.testing.assert_allclose(rs[0], ref, rtol=1e-12)


def test_split_of_a_one_rank_rccl_world_gives_a_working_group_communicator():
  """[r4] parallel.split on an RCCL world: the group's unique id is created by rank 0 and the group communicator carries the 'both' scheme
  (one rank = one group here: the collective inside chm_eval runs through RCCL, the assembly over the world too)."""
  import chimera_amd as CH
  from chimera_amd.parallel import Comm, split
  world = Comm(1, 0, device=0)
  grp = split(world, 1)
  assert (grp.group_id, grp.ngroups, grp.nranks, grp.rank) == (0, 1, 1, 0) and grp.handle and grp.handle.value != world.handle.value
  cfg, ev, inj = H.small_config(E=5, S=128, P=3, Z=40, I=1501, seed=9, ragged=True)
  like0, pop, sel = H.build_product(ev, inj, comm=grp)
  like = CH.hyperlikelihood(like0.theta_gw_det, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', comm=grp, scheme='both')
  ref, _, _ = H.build_product(ev, inj)
  lams = [dict(H0=float(h)) for h in (64., 70., 77.)]
  np.testing.assert_array_equal(like.batch(lams), ref.batch(lams))
  assert like(H0=70.) == ref(H0=70.)
  like.close(); like0.close(); sel.close(); grp.close(); world.close()


def _rccl_worker(rank, world, addr, outdir):
  import os, sys
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  from chimera_amd import _lib
  from chimera_amd.parallel import Comm, Rendezvous
  from tests import helpers as HH
  rd = Rendezvous(world, rank, address=addr, timeout=120.)
  comm = Comm(world, rank, device=rank, rendezvous=rd)        # one GPU per rank; the unique id travels over the sockets
  assert _lib.lib().chm_comm_nranks(comm.handle) == world
  np.testing.assert_array_equal(comm.allreduce_sum(np.array([rank + 1., 2.])), [world * (world + 1) / 2, 2. * world])
  cfg, ev, inj = HH.small_config(E=9, S=300, P=4, Z=60, I=3001, seed=41, ragged=True)
  out = []
  for pop_kw in ({}, dict(scale_free=False, R0=12., Tobs=1.5)):
    like, pop, sel = HH.build_product(ev, inj, comm=comm, pop_kw=pop_kw)
    lams = [dict(H0=64.), dict(H0=70., alpha=3.0), dict(H0=81., gamma=2.0)]
    out.append(np.concatenate([like.batch(lams), [like(**lams[1]), sel.N_exp(pop.update(**lams[2]))]]))
    like.close(); sel.close()
  np.save(os.path.join(outdir, f'rank{rank}.npy'), np.array(out))
  rd.barrier()
  comm.close()
  rd.close()


@pytest.mark.timeout(300)
def test_real_multi_rank_rccl_communicator_when_several_gpus_are_visible(tmp_path):
  """ncclAllReduce inside chm_eval with nranks > 1: one process per GPU, events and injections sharded, every rank must obtain the
  single-process value.  Needs >= 2 visible GPUs (gpurun boxes expose one: skipped there; the driver's 8-GPU node runs it)."""
  from chimera_amd import _lib
  ndev = _lib.lib().chm_device_count()
  if ndev < 2:
    pytest.skip(f"{ndev} GPU visible: a multi-rank RCCL communicator needs one device per rank")
  world = min(ndev, 4)
  _spawn(_rccl_worker, world, (str(tmp_path / 'rdzv.sock'), str(tmp_path)))
  rs = [np.load(tmp_path / f'rank{r}.npy') for r in range(world)]
  for r in rs[1:]:
    np.testing.assert_array_equal(r, rs[0])
  cfg, ev, inj = H.small_config(E=9, S=300, P=4, Z=60, I=3001, seed=41, ragged=True)
  for k, pop_kw in enumerate(({}, dict(scale_free=False, R0=12., Tobs=1.5))):
    like, pop, sel = H.build_product(ev, inj, pop_kw=pop_kw)
    lams = [dict(H0=64.), dict(H0=70., alpha=3.0), dict(H0=81., gamma=2.0)]
    ref = np.concatenate([like.batch(lams), [like(**lams[1]), sel.N_exp(pop.update(**lams[2]))]])
    np.testing.assert_allclose(rs[0][k], ref, rtol=1e-12)


# ----------------------------------------------------------------------------------------------------------
# size-independent properties at a BASELINE-sized shape (300 events x 32 pixels x 1000 z-bins x 4096 samples)
# ----------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def cfg_big():
  from chimera_amd import synth
  return synth.make_config('C3', E=300, I=100_000)


def test_full_size_properties(cfg_big):
  cfg, ev, inj = cfg_big
  E = cfg['E']
  like, pop, sel = H.build_product(ev, inj)
  lam = dict(H0=70.)
  base = like.compute_all(**lam)
  assert np.all(np.isfinite(base[0])) and np.isfinite(base[3])
  # (1) event permutation invariance: the per-event values move with the events, the sum is unchanged
  perm = np.random.default_rng(0).permutation(E)
  evp = {k: (v[perm] if hasattr(v, 'shape') and v.shape[:1] == (E,) else v) for k, v in ev.items()}
  likep, _, _ = H.build_product(evp, inj)
  rp = likep.compute_all(**lam)
  np.testing.assert_array_equal(rp[0], base[0][perm])
  np.testing.assert_allclose(rp[3], base[3], rtol=0, atol=1e-9 * E)
  # (2) batch of draws == one draw at a time; repeated evaluation is reproducible bit for bit
  lams = [dict(H0=h) for h in (62., 70., 78.)]
  np.testing.assert_array_equal(like.batch(lams), np.array([like(**l) for l in lams]))
  assert like(**lam) == base[3]
  # (3) linearity in the sky-localisation density: doubling gw_loc2d_pdf doubles every L_i   (likelihood.py:194)
  ev2 = dict(ev); ev2['gw_loc2d_pdf'] = np.where(ev['gw_loc2d_pdf'] != -100., 2. * ev['gw_loc2d_pdf'], -100.)
  like2, _, _ = H.build_product(ev2, inj)
  np.testing.assert_allclose(like2.compute_all(**lam)[0], base[0] + np.log(2.), rtol=0, atol=1e-12)
  # (4) the selection term: N_exp scales linearly with 1/N_inj and with R0
  n1 = sel.N_exp(pop)
  sel2 = type(sel)(sel.theta_inj_det, N_inj=2 * inj['N_inj'], N_eff=None)
  np.testing.assert_allclose(sel2.N_exp(pop), n1 / 2, rtol=1e-14)
  np.testing.assert_allclose(sel.N_exp(pop.update(R0=3.)), 3 * n1, rtol=1e-14)
  # (5) shards add up
  parts = []
  for r in range(4):
    lk, _, _ = H.build_product(ev, inj, comm=_FakeRank(4, r))
    parts.append(lk._eval([lk.population.update(**lam)], want=('partials',))['partials'][0])
  whole = like._eval([like.population.update(**lam)], want=('partials',))['partials'][0]
  np.testing.assert_allclose(np.sum(parts, axis=0), whole, rtol=1e-12)
  # (6) approximate vs marginalized agree to a few per cent in log-likelihood per event on well-sampled events
  likea, _, _ = H.build_product(ev, inj, kind='approximate')
  ra = likea.compute_all(**lam)
  assert np.median(np.abs(ra[0] - base[0])) < 0.2


def test_full_size_against_oracle_sample(cfg_big):
  """The first 6 events of the big configuration, checked one by one against the oracle."""
  cfg, ev, inj = cfg_big
  n = 6
  sub = {k: (v[:n] if hasattr(v, 'shape') and v.shape[:1] == (cfg['E'],) else v) for k, v in ev.items()}
  like_p, _, _ = H.build_product(ev, inj)
  like_o, _, _ = H.build_oracle(sub, inj)
  for lam in (dict(H0=70.), dict(H0=84., alpha=3.0)):
    H.assert_loglike_close(like_p.compute_all(**lam)[0][:n], like_o.compute_all(**lam)[0], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize('kind', ['marginalized', 'approximate'])
def test_full_size_every_event_against_the_c_oracle(cfg_big, kind):
  """All 300 events x 32 pixels x 1000 z-bins x 4096 samples + 1e5 injections against the plain-C restatement
  (oracle/chimera_oracle_c.c, OpenMP; itself checked against the NumPy oracle in tests/test_oracle_c.py)."""
  import os
  from oracle import oracle_c as OC
  cfg, ev, inj = cfg_big
  like_p, _, _ = H.build_product(ev, inj, kind=kind)
  like_o, _, _ = H.build_oracle(ev, inj, kind=kind)
  nthr = min(16, os.cpu_count() or 1)
  for lam in (dict(H0=67.), dict(H0=88., lambda_peak=0.08, gamma=2.0)):
    rp = like_p.compute_all(**lam)
    rc = OC.compute_all(like_o, lam, nthreads=nthr)
    H.assert_loglike_close(rp[0], rc[0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rp[2], rc[2], rtol=1e-10)
    np.testing.assert_allclose(rp[3], rc[3], rtol=0, atol=1e-7 * np.sqrt(cfg['E']))


def test_full_size_modified_propagation_against_the_c_oracle(cfg_big):
  """BASELINE.json configs[4] (C5) physics at full per-event size: modified GW propagation (Xi0, n) varied per draw, 300 events x 32
  pixels x 1000 z-bins x 4096 samples, every event and the selection term against the plain-C restatement."""
  import os
  from oracle import oracle_c as OC
  cfg, ev, inj = cfg_big
  models = dict(cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.8, n=1.9))
  like_p, _, _ = H.build_product(ev, inj, models=models)
  like_o, _, _ = H.build_oracle(ev, inj, models=models)
  nthr = min(16, os.cpu_count() or 1)
  lams = [dict(H0=72., Xi0=2.4, n=1.5), dict(H0=61., Xi0=0.7, n=2.6)]
  batch = like_p.batch(lams)
  for i, lam in enumerate(lams):
    rp = like_p.compute_all(**lam)
    rc = OC.compute_all(like_o, lam, nthreads=nthr)
    H.assert_loglike_close(rp[0], rc[0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rp[2], rc[2], rtol=1e-10)
    np.testing.assert_allclose(rp[3], rc[3], rtol=0, atol=1e-7 * np.sqrt(cfg['E']))
    assert batch[i] == rp[3]
  like_p.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('name', ['C1', 'C2', 'C4'])
def test_baseline_configurations_at_full_size_against_the_c_oracle(name):
  """BASELINE.json configs[0], [1] and [3] at their full sizes (C1: 10 events, 1-D, 500 z-bins; C2: 100 events x 16 pixels x 500
  z-bins; C4: 69 events x 16 pixels x 500 z-bins with 1e6 detected injections), every event and the selection term against the
  plain-C restatement -- C3 is covered above and by every default bench.py run (`parity_full_size`), C5 by
  profiles/r01/bench_C5.json's run of the same kernels on 10 000 events."""
  import os
  from chimera_amd import synth
  from oracle import oracle_c as OC
  cfg, ev, inj = synth.make_config(name)
  pix = cfg['pixelated']
  like_p, _, _ = H.build_product(ev, inj, pixelated=pix)
  like_o, _, _ = H.build_oracle(ev, inj, pixelated=pix)
  nthr = min(16, os.cpu_count() or 1)
  for lam in (dict(H0=67.), dict(H0=91., alpha=2.9, gamma=3.1)):
    rp = like_p.compute_all(**lam)
    rc = OC.compute_all(like_o, lam, nthreads=nthr)
    assert np.all(np.isfinite(rc[0]))
    H.assert_loglike_close(rp[0], rc[0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rp[2], rc[2], rtol=1e-10)
    np.testing.assert_allclose(rp[3], rc[3], rtol=0, atol=1e-7 * np.sqrt(cfg['E']))
  like_p.close()


# ----------------------------------------------------------------------------------------------------------
# catalogue term computed on the GPU (pixelated_catalog.precompute_p_cat, catalog.py:152-231)
# ----------------------------------------------------------------------------------------------------------
def test_precompute_p_cat_matches_oracle():
  import chimera_amd as CH
  from chimera_amd import synth
  from chimera_amd.catalog import pixelated_catalog, dVdz_completeness
  cfg, ev, inj = synth.make_config('C2', E=6, S=64, P=4, Z=96, I=100, ragged=True)
  gal = synth.make_galaxy_sample(ev, ev['z_grids'], ngal_mean=12)
  w = np.random.default_rng(5).uniform(0.5, 2., gal['z'].size)
  th = CH.data.theta_pe_det(dL=ev['dL'], pixels_opt_nsides=ev['pixels_opt_nsides'], ra_pix=ev['ra_pix'], opt_nsides=ev['opt_nsides'])
  for z_err in (0.01, 0.001):            # the second is under-resolved by the grid: non-finite rows -> 0 (catalog.py:172)
    gc = pixelated_catalog(dVdz_completeness(), cosmo=CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.), z_grids=ev['z_grids'],
                           data_gal=gal, data_gw_pixelated=th, z_err=z_err, weights=w)
    co = O.flrw(H0=70., Om0=0.25, z_max=5.)
    for e in range(cfg['E']):
      ns = ev['opt_nsides'][e]
      good = ev['pixels_opt_nsides'][e][ev['pixels_opt_nsides'][e] != -100]
      isin = np.isin(gal[f'pix{ns}'], good)
      pc, ngal = O.compute_p_cat_event(ev['z_grids'][e], gal['z'][isin], z_err * (1 + gal['z'][isin]), w[isin], gal[f'pix{ns}'][isin],
                                       good, cfg['P'], co)
      np.testing.assert_allclose(gc.p_cat[e], pc, rtol=1e-10, atol=1e-300, err_msg=f'event {e} z_err {z_err}')
      assert gc.N_gal[e] == ngal
    assert gc.P_compl.shape == (cfg['E'], 1, 96) and gc.max_npixels == cfg['P']
    np.testing.assert_array_equal(gc.neff_pixels, ev['neff_pixels'])


def test_precompute_p_cat_with_the_background_of_a_plugin_completeness():
  """[r3] pixelated_catalog(sumgauss='pbkg') with a user-written completeness model (CHIMERA/catalog/catalog.py:164-171, 223-231): the
  galaxy Gaussians are weighted by the model's own p_bkg(cosmo, z), evaluated on the host on the event grids and handed to k_pcat
  (chm_pcat_desc.weight_grid).  Against the oracle's _sum_gaussians_pbkg; with p_bkg = dVc/dz it must reproduce the 'dVdz' catalogue."""
  import chimera_amd as CH
  from chimera_amd import synth
  from chimera_amd.catalog import pixelated_catalog, dVdz_completeness

  class tilted_completeness(object):                     # a plug-in: the reference's interface P_compl / fR / p_bkg (completeness.py:22-67)
    def __init__(self, tilt): self.tilt = tilt
    def P_compl(self, zgrids): return np.where(np.asarray(zgrids) < 0.9, 1., 0.)
    def fR(self, cosmo, normalized=False): return 1.
    def p_bkg(self, cosmo, z):
      z = np.asarray(z, dtype=np.float64)
      return CH.cosmo.dVcdz_at_z(cosmo, z) * (1. + z) ** self.tilt

  cfg, ev, inj = synth.make_config('C2', E=5, S=64, P=4, Z=80, I=100, ragged=True)
  gal = synth.make_galaxy_sample(ev, ev['z_grids'], ngal_mean=10)
  w = np.random.default_rng(8).uniform(0.5, 2., gal['z'].size)
  th = CH.data.theta_pe_det(dL=ev['dL'], pixels_opt_nsides=ev['pixels_opt_nsides'], ra_pix=ev['ra_pix'], opt_nsides=ev['opt_nsides'])
  cp, co = CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.), O.flrw(H0=70., Om0=0.25, z_max=5.)
  kw = dict(cosmo=cp, z_grids=ev['z_grids'], data_gal=gal, data_gw_pixelated=th, z_err=0.01, weights=w)
  for tilt in (-1.7, 0.):
    gc = pixelated_catalog(tilted_completeness(tilt), sumgauss='pbkg', **kw)
    for e in range(cfg['E']):
      ns = ev['opt_nsides'][e]
      good = ev['pixels_opt_nsides'][e][ev['pixels_opt_nsides'][e] != -100]
      isin = np.isin(gal[f'pix{ns}'], good)
      pc, ngal = O.compute_p_cat_event(ev['z_grids'][e], gal['z'][isin], 0.01 * (1 + gal['z'][isin]), w[isin], gal[f'pix{ns}'][isin], good, cfg['P'],
                                       co, p_bkg=lambda c, z: O.dVcdz_at_z(c, z) * (1. + z) ** tilt)
      np.testing.assert_allclose(gc.p_cat[e], pc, rtol=1e-10, atol=1e-300, err_msg=f'event {e} tilt {tilt}')
      assert gc.N_gal[e] == ngal
  ref = pixelated_catalog(dVdz_completeness(), **kw)       # tilt 0: p_bkg is dVc/dz, the default weight of the kernel
  np.testing.assert_allclose(gc.p_cat, ref.p_cat, rtol=1e-12, atol=1e-300)
  other = pixelated_catalog(tilted_completeness(-1.7), sumgauss='pbkg', **kw)
  live = ref.p_cat > 0
  assert np.max(np.abs(other.p_cat[live] / ref.p_cat[live] - 1.)) > 1e-3          # the tilt does change the catalogue term


# ----------------------------------------------------------------------------------------------------------
# ragged / odd shapes: odd S and Z (scalar load paths), a single pixel, an odd pixel count (half-empty wave), many bins per
# lane, very few bins, a single event, more than one sample chunk per event
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('shape', [dict(E=3, S=101, P=5, Z=51), dict(E=1, S=64, P=1, Z=33), dict(E=4, S=2100, P=3, Z=40),
                                   dict(E=2, S=300, P=7, Z=64)])
@pytest.mark.parametrize('like_kw', [dict(), dict(num_bins=1100), dict(num_bins=3)])
def test_odd_shapes(shape, like_kw):
  cfg, ev, inj = H.small_config(I=999, seed=17, ragged=True, **shape)
  for kind in ('marginalized', 'approximate', 'full'):
    if kind == 'full' and like_kw:
      continue
    like_o, _, _ = H.build_oracle(ev, inj, kind=kind, like_kw=like_kw)
    like_p, _, _ = H.build_product(ev, inj, kind=kind, like_kw=like_kw)
    ro, rp = like_o.compute_all(H0=71.), like_p.compute_all(H0=71.)
    H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
    if np.isfinite(ro[3]):
      np.testing.assert_allclose(rp[3], ro[3], rtol=0, atol=1e-7 * np.sqrt(cfg['E']))


def test_degenerate_pixels_give_minus_infinity_class(cfg_pix):
  """A pixel whose samples all have zero weight (or that holds no sample) makes the reference's KDE 0/0 = NaN, hence the
  event's likelihood NaN -> log-likelihood -inf (likelihood.py:180-192, 296-297).  Same on the GPU."""
  cfg, ev, inj = cfg_pix
  ev2 = dict(ev)
  pe = ev['pixels_pe_opt_nside'].copy()
  e = 1
  victim = ev['pixels_opt_nsides'][e, 0]
  pe[e][pe[e] == victim] = 7                     # nobody lives in pixel 0 of event 1 any more
  ev2['pixels_pe_opt_nside'] = pe
  like_o, _, _ = H.build_oracle(ev2, inj)
  like_p, _, _ = H.build_product(ev2, inj)
  ro, rp = like_o.compute_all(H0=70.), like_p.compute_all(H0=70.)
  assert ro[0][e] == -np.inf and rp[0][e] == -np.inf
  H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
  assert rp[3] == -np.inf and ro[3] == -np.inf


@pytest.mark.parametrize('mass,cosmo', [('plp', 'flrw'), ('tpl', 'flrw'), ('bpl', 'mg_flrw'), ('plp', 'mg_flrw')])
def test_random_hyperparameter_draws_against_the_c_oracle(mass, cosmo):
  """Twelve random draws over every free hyper-parameter of the models (cosmology, mass, rate), evaluated in one batch on the
  GPU and one by one by the C oracle: per-event log-likelihoods and the hyper-likelihood must agree for each draw."""
  import os
  from oracle import oracle_c as OC
  cfg, ev, inj = H.small_config(E=24, S=1024, P=6, Z=200, I=20000, seed=23, ragged=True)
  models = dict(mass=mass, cosmo=cosmo)
  like_p, _, _ = H.build_product(ev, inj, models=models)
  like_o, _, _ = H.build_oracle(ev, inj, models=models)
  rng = np.random.default_rng(99)
  lams = []
  for _ in range(12):
    lam = dict(H0=rng.uniform(55., 95.), Om0=rng.uniform(0.15, 0.45), gamma=rng.uniform(1., 4.), kappa=rng.uniform(2., 5.),
               zp=rng.uniform(1., 3.), m_low=rng.uniform(3.5, 6.), m_high=rng.uniform(70., 110.), beta=rng.uniform(0., 2.5))
    if cosmo == 'mg_flrw':
      lam.update(Xi0=rng.uniform(0.5, 3.), n=rng.uniform(0.5, 3.))
    if mass == 'plp':
      lam.update(alpha=rng.uniform(2., 4.5), lambda_peak=rng.uniform(0.01, 0.2), mu_g=rng.uniform(28., 40.), sigma_g=rng.uniform(2., 6.),
                 delta_m=rng.uniform(2., 7.))
    elif mass == 'bpl':
      lam.update(alpha_1=rng.uniform(1., 2.5), alpha_2=rng.uniform(3., 7.), break_fraction=rng.uniform(0.2, 0.7), delta_m=rng.uniform(2., 7.))
    else:
      lam.update(alpha=rng.uniform(1.5, 4.))
    lams.append(lam)
  got = like_p.batch(lams)
  nthr = min(16, os.cpu_count() or 1)
  for i, lam in enumerate(lams):
    rc = OC.compute_all(like_o, lam, nthreads=nthr)
    if np.isfinite(rc[3]):
      np.testing.assert_allclose(got[i], rc[3], rtol=0, atol=1e-7 * np.sqrt(cfg['E']), err_msg=str(lam))
    else:
      assert not np.isfinite(got[i]) or got[i] < -1e300
    if i % 4 == 0:
      H.assert_loglike_close(like_p.compute_all(**lam)[0], rc[0], rtol=RTOL_L, atol=1e-9)


@pytest.mark.parametrize('seed', range(6))
def test_random_configurations_against_the_numpy_oracle(seed):
  """Random shapes, modes, KDE options, models and hyper-parameters (five configurations per seed) against the NumPy oracle."""
  rng = np.random.default_rng(1000 + seed)
  for _ in range(5):
    pixelated = rng.random() < 0.8
    kind = rng.choice(['marginalized', 'marginalized', 'approximate', 'full']) if pixelated else None
    E, S = int(rng.integers(1, 6)), int(rng.integers(40, 700))
    P, Z = int(rng.integers(1, 7)), int(rng.integers(12, 90))
    cfg, ev, inj = H.small_config(E=E, S=S, P=P, Z=Z, I=int(rng.integers(300, 3000)), seed=int(rng.integers(1, 10**6)),
                                  ragged=bool(rng.random() < 0.5), pixelated=pixelated)
    like_kw = dict(num_bins=int(rng.choice([3, 17, 64, 200, 333])), pe_neff=float(rng.choice([2., 5., 50.])))
    if kind != 'full':
      like_kw['cut_grid'] = [None, 1.0, 2.0, 3.5][int(rng.integers(0, 4))]
      like_kw['bw_method'] = [None, 'scott', 'silverman', 0.25][int(rng.integers(0, 4))]
      like_kw['binning'] = bool(rng.random() < 0.75)
    if kind in (None, 'approximate'):
      like_kw['kernel'] = str(rng.choice(['epan', 'gauss']))
    models = dict(mass=str(rng.choice(['plp', 'tpl', 'bpl'])), cosmo=str(rng.choice(['flrw', 'mg_flrw'])),
                  rate=str(rng.choice(['power_law', 'madau_dickinson', 'trunc_power_law', 'trunc_madau_dickinson'])))
    if models['rate'].startswith('trunc'):
      models['rate_kw'] = dict(zmax=float(rng.uniform(1.5, 4.)))
    pop_kw = dict(scale_free=bool(rng.random() < 0.7), R0=float(rng.uniform(5., 40.)), Tobs=float(rng.uniform(0.5, 3.)))
    N_eff = [None, 5.][int(rng.integers(0, 2))]
    lam = dict(H0=float(rng.uniform(55., 95.)), Om0=float(rng.uniform(0.2, 0.4)), gamma=float(rng.uniform(1., 3.5)),
               m_low=float(rng.uniform(3.5, 6.)), m_high=float(rng.uniform(75., 110.)), beta=float(rng.uniform(0., 2.)))
    if models['cosmo'] == 'mg_flrw':
      lam.update(Xi0=float(rng.uniform(0.6, 2.5)), n=float(rng.uniform(0.5, 2.5)))
    desc = f"kind={kind} shape=({E},{S},{P},{Z}) like_kw={like_kw} models={models} pop_kw={pop_kw} N_eff={N_eff} lam={lam}"
    like_o, _, _ = H.build_oracle(ev, inj, pixelated=pixelated, kind=kind, models=models, like_kw=like_kw, pop_kw=pop_kw, N_eff=N_eff)
    like_p, _, _ = H.build_product(ev, inj, pixelated=pixelated, kind=kind, models=models, like_kw=like_kw, pop_kw=pop_kw, N_eff=N_eff)
    with np.errstate(all='ignore'):
      ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
    try:
      H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
      if np.isfinite(ro[3]):
        np.testing.assert_allclose(rp[2], ro[2], rtol=1e-10)
        np.testing.assert_allclose(rp[3], ro[3], rtol=1e-12, atol=1e-7 * np.sqrt(E))
    except AssertionError as err:
      raise AssertionError(desc + '\n' + str(err))
    like_p.close()


@pytest.mark.parametrize('kind', ['marginalized', 'approximate'])
def test_nan_and_out_of_range_samples_behave_like_the_reference(cfg_pix, kind):
  """A NaN distance makes the event's z statistics NaN (jnp.min / max / std propagate it), hence L_i = NaN -> -inf; samples
  with masses outside the population's range simply carry zero weight.  The v_max/v_min + NaN-vote forms must agree."""
  cfg, ev, inj = cfg_pix
  ev2 = dict(ev)
  dL = ev['dL'].copy(); m1 = ev['m1det'].copy()
  dL[0, 5] = np.nan                              # event 0: one NaN distance
  m1[2, :40] = 1e4                               # event 2: forty samples far above m_high -> weight 0
  m1[3, 7] = np.nan                              # event 3: one NaN mass -> NaN weight -> sums NaN
  ev2['dL'], ev2['m1det'] = dL, m1
  like_o, _, _ = H.build_oracle(ev2, inj, kind=kind)
  like_p, _, _ = H.build_product(ev2, inj, kind=kind)
  with np.errstate(all='ignore'):
    ro, rp = like_o.compute_all(H0=70.), like_p.compute_all(H0=70.)
  assert H.neginf_class(ro[0][0]) and H.neginf_class(ro[0][3]) and np.isfinite(ro[0][2])
  H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)


@pytest.mark.parametrize('kind', ['marginalized', 'full'])
def test_nan_hyperparameters_behave_like_the_reference(cfg_pix, kind):
  """[r4] A NaN hyper-parameter (and an infinite cosmological or mass one) must come out as it does from the reference's arithmetic: NaN
  weights -> NaN sums -> L_i = NaN -> log L_i in the -inf class, log of the selection bias -inf, total NaN -- or, for a parameter the
  configured models never read, the unchanged finite answer.  The fast kernels drop NaNs in places (v_min / v_max clamps, saturating
  conversions) and rely on the NaN having been caught elsewhere: this is the check that it is.  (Infinite rate parameters: the next test.)"""
  cfg, ev, inj = cfg_pix
  like_o, _, _ = H.build_oracle(ev, inj, kind=kind)
  like_p, _, _ = H.build_product(ev, inj, kind=kind)
  cases = [(n, np.nan) for n in ('H0', 'Om0', 'alpha', 'beta', 'ml', 'mh', 'mu_g', 'sigma_g', 'lambda_peak', 'delta_m', 'gamma', 'kappa', 'zp')]
  cases += [(n, np.inf) for n in ('H0', 'Om0', 'alpha', 'beta', 'mu_g', 'sigma_g', 'lambda_peak', 'delta_m')]
  seen_nan = seen_finite = 0
  for name, bad in cases:
    with np.errstate(all='ignore'):
      ro, rp = like_o.compute_all(**{name: bad}), like_p.compute_all(**{name: bad})
    try:
      H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
      np.testing.assert_allclose(rp[1], ro[1], rtol=1e-10, atol=0, equal_nan=True)
      np.testing.assert_allclose(rp[3], ro[3], rtol=1e-10, atol=1e-8, equal_nan=True)
    except AssertionError as err:
      raise AssertionError(f'{name} = {bad}\n{err}')
    seen_nan += bool(np.isnan(ro[3]).all()); seen_finite += bool(np.isfinite(ro[3]).all())
  assert seen_nan >= 10 and seen_finite >= 2
  like_p.close()


@pytest.mark.parametrize('kind', [None, 'marginalized'])
def test_a_sample_within_an_ulp_of_a_bin_edge_lands_in_one_of_the_two_neighbouring_bins(kind):
  """[r5] The documented exception to the 1e-9 on L_i, pinned: binning1d takes idx = floor((z - lo)/(hi - lo) * B) (CHIMERA/utils/math.py:41), device
  and oracle compute z to a few ulp of each other, so a sample whose quotient sits within an ulp of an integer may land in either of two neighbouring
  bins -- and with few samples per bin that moves L_i by far more than 1e-9.  The distance of one sample is bisected (with the oracle's own arithmetic)
  down to the two ADJACENT doubles d_a < d_b between which the oracle's bin index flips; the oracle's L_i at d_a and at d_b are the two admissible
  answers (everything else moves by 1e-16).  The device must give one of them at either distance -- either neighbouring bin, nothing else."""
  from oracle import chimera_oracle as O
  pix = kind is not None
  cfg, ev, inj = H.small_config(E=3, S=96, P=2, Z=64, I=1500, seed=31, ragged=False, pixelated=pix)
  B, e, j = 17, 1, 40
  like_kw = dict(num_bins=B, pe_neff=1.)
  ev = dict(ev); ev['dL'] = np.array(ev['dL'], dtype=np.float64, copy=True)
  if pix:                                                        # a sample in the middle of a well-filled pixel (not the pixel's largest z: that one defines the range)
    ids = ev['pixels_pe_opt_nside'][e]
    mine = np.flatnonzero(ids == ids[j])
    assert len(mine) > 8
    j = int(mine[np.argsort(ev['dL'][e, mine])[len(mine) // 2]])

  def quotient(d):
    """t = (z_j - lo)/(hi - lo) * B of sample j of event e at distance d: the oracle's operations (math.py:36-41 on the pixel's masked samples)"""
    ev['dL'][e, j] = d
    lo_, pop, _ = H.build_oracle(ev, inj, pixelated=pix, kind=kind, like_kw=like_kw)
    th, w = O.get_theta_src_and_weights(pop.update(H0=70.) if hasattr(pop, 'update') else pop, lo_.theta_gw_det)
    z = np.asarray(th.z[e])
    if pix:
      m = ev['pixels_pe_opt_nside'][e] == ev['pixels_pe_opt_nside'][e, j]
      zm = np.where(m, z, z.min())                               # likelihood.py:180
    else:
      zm = z
    lo, hi = zm.min(), zm.max()
    return (z[j] - lo) / (hi - lo) * B

  d0 = float(ev['dL'][e, j])
  for f in (1.004, 1.008, 1.015, 1.03):                              # the narrowest bracket with a bin edge inside
    a, b = d0 / f, d0 * f
    ta, tb = quotient(a), quotient(b)
    if np.floor(ta) < np.floor(tb):
      break
  assert np.floor(ta) < np.floor(tb) < B - 1 and ta > 1.             # an edge in between, the sample is neither the range's lower nor its upper end
  edge = np.floor(ta) + 1.
  while np.nextafter(a, np.inf) < b:                                 # bisection on the doubles: quotient(a) < edge <= quotient(b)
    mid = 0.5 * (a + b)
    if mid <= a or mid >= b:
      break
    if quotient(mid) < edge:
      a = mid
    else:
      b = mid
  assert np.nextafter(a, np.inf) == b and np.floor(quotient(a)) == edge - 1. and np.floor(quotient(b)) == edge
  vals_o, vals_p = [], []
  for d in (a, b):
    ev['dL'][e, j] = d
    lo_, _, _ = H.build_oracle(ev, inj, pixelated=pix, kind=kind, like_kw=like_kw)
    lp_, _, sp_ = H.build_product(ev, inj, pixelated=pix, kind=kind, like_kw=like_kw)
    vals_o.append(lo_.compute_all(H0=70.)[0]); vals_p.append(lp_.compute_all(H0=70.)[0])
    lp_.close(); sp_.close()
  oa, ob = vals_o[0][e], vals_o[1][e]
  assert abs(oa - ob) > 1e-7 * abs(oa)                               # the flip is visible: 100x the stated tolerance
  for vp in vals_p:
    assert min(abs(vp[e] - oa), abs(vp[e] - ob)) <= 1e-9 * abs(oa), (vp[e], oa, ob)
    for other in (0, 2):                                             # the other events do not notice
      np.testing.assert_allclose(vp[other], vals_o[0][other], rtol=1e-9)


@pytest.mark.parametrize('kind', ['marginalized', 'approximate', 'full', None])
@pytest.mark.parametrize('rate', ['madau_dickinson', 'power_law', 'trunc_power_law', 'trunc_madau_dickinson'])
def test_infinite_rate_parameters_behave_like_the_reference(kind, rate):
  """[r5] gamma, kappa, z_p = +-inf (CHIMERA/population/rate.py:96-122 under NumPy / XLA: C99 pow, one quotient, one product): kappa = +inf and
  z_p = +-inf leave a finite rate ((1+z)^gamma below z_p, 0 above; the plain power law), gamma = +inf an infinite one -- 0 * inf = NaN wherever
  p_gw vanishes on an event grid, so every L_i is NaN (log L_i in the -inf class) and N_exp = inf --, gamma = -inf and kappa = -inf NaN throughout.
  exp(y log x) on the device gives NaN for all of them; such a draw takes merger_rate_special and the kernels that report the poisoned grids.
  Scalar and batched calls (a batch mixes ordinary and infinite draws: the whole call takes the general kernels) must agree."""
  pixelated = kind is not None
  cfg, ev, inj = H.small_config(E=6, S=256, P=4, Z=64, I=3000, seed=11, ragged=True, pixelated=pixelated)
  models = dict(rate=rate)
  like_o, _, _ = H.build_oracle(ev, inj, pixelated=pixelated, kind=kind, models=models)
  like_p, _, _ = H.build_product(ev, inj, pixelated=pixelated, kind=kind, models=models)
  names = ['gamma'] + (['kappa', 'zp'] if 'madau' in rate else [])
  cases = [{n: v} for n in names for v in (np.inf, -np.inf)] + [{'H0': 68.}]
  seen = set()
  res_o = []
  for lam in cases:
    with np.errstate(all='ignore'):
      ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
    res_o.append(ro[3])
    try:
      H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
      np.testing.assert_allclose(rp[1], ro[1], rtol=1e-10, atol=0, equal_nan=True)
      np.testing.assert_allclose(rp[2], ro[2], rtol=1e-10, atol=0, equal_nan=True)
      np.testing.assert_allclose(rp[3], ro[3], rtol=1e-10, atol=1e-8, equal_nan=True)
    except AssertionError as err:
      raise AssertionError(f'{rate} {kind} {lam}\n{err}')
    seen.add('nan' if np.isnan(ro[3]) else ('finite' if np.isfinite(ro[3]) else 'inf'))
  assert 'finite' in seen and ('nan' in seen or 'inf' in seen)
  # one batch with ordinary and infinite draws side by side == the scalar calls
  with np.errstate(all='ignore'):
    got = like_p.batch(cases)
  np.testing.assert_allclose(got, np.array(res_o), rtol=1e-10, atol=1e-8, equal_nan=True)
  # the free function through chm_model_eval
  import chimera_amd as CH
  from oracle import chimera_oracle as O
  z = np.array([0., 0.3, 1., 2., 4.63, 7.])
  for lam in cases[:-1]:
    rp_, ro_ = getattr(CH.rate, rate)(**lam), getattr(O, rate)(**lam)
    with np.errstate(all='ignore'):
      np.testing.assert_allclose(CH.rate.merger_rate(rp_, z), O.merger_rate(ro_, z), rtol=1e-13, equal_nan=True, err_msg=str(lam))
  like_p.close()


@pytest.mark.parametrize('kind', ['approximate', None])
def test_infinite_rate_meets_a_kde_that_vanishes_nowhere(kind):
  """[r5] gamma = +inf makes the rate factor +inf at every grid point with z > 0.  Where p_gw vanishes somewhere on the event grid the reference's
  trapezoid holds a 0 * inf = NaN and L_i is NaN -- but a GAUSSIAN kernel without cut_grid vanishes nowhere, every product is +inf, L_i = +inf and
  log L_i comes out as +1.797e308 (nan_to_num).  The 1-D integrand kernel forms the reference's products over the whole grid for such a draw
  (scripts/fuzz_parity.py seed 6001464: the blanket NaN of event_poisoned() gave -inf there)."""
  pixelated = kind is not None
  cfg, ev, inj = H.small_config(E=5, S=512, P=3, Z=46, I=2000, seed=19, ragged=True, pixelated=pixelated)
  models = dict(rate='power_law', mass='tpl')
  for like_kw in (dict(kernel='gauss', cut_grid=None, binning=True, num_bins=17, bw_method='scott'), dict(kernel='gauss', cut_grid=None, binning=False),
                  dict(kernel='epan', cut_grid=None, binning=True, num_bins=17), dict(kernel='gauss', cut_grid=2.0)):
    like_o, _, _ = H.build_oracle(ev, inj, pixelated=pixelated, kind=kind, models=models, like_kw=like_kw, pop_kw=dict(scale_free=False))
    like_p, _, _ = H.build_product(ev, inj, pixelated=pixelated, kind=kind, models=models, like_kw=like_kw, pop_kw=dict(scale_free=False))
    for lam in (dict(gamma=np.inf), dict(gamma=np.inf, H0=55.), dict(gamma=-np.inf), dict(gamma=2.)):
      with np.errstate(all='ignore'):
        ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
        a, bt = like_p(**lam), like_p.batch([lam, dict(H0=71.)])[0]
      np.testing.assert_array_equal(np.asarray(rp[0]) >= 1e300, np.asarray(ro[0]) >= 1e300, err_msg=f'{like_kw} {lam}: +inf class {rp[0]} against {ro[0]}')
      H.assert_loglike_close(np.where(np.asarray(rp[0]) >= 1e300, 0., rp[0]), np.where(np.asarray(ro[0]) >= 1e300, 0., ro[0]), rtol=RTOL_L, atol=1e-9)
      np.testing.assert_allclose(rp[3], ro[3], rtol=1e-10, atol=1e-8, equal_nan=True, err_msg=f'{like_kw} {lam}')
      assert (a == bt) or (np.isnan(a) and np.isnan(bt))
    if like_kw.get('kernel') == 'gauss' and like_kw.get('cut_grid') is None and like_kw.get('binning') and kind is None:
      with np.errstate(all='ignore'):
        assert np.any(np.asarray(like_o.compute_all(gamma=np.inf)[0]) >= 1e300), like_o.compute_all(gamma=np.inf)[0]     # the case the test is about does occur
    like_p.close()


@pytest.mark.parametrize('kind', ['marginalized', 'approximate', 'full', None])
def test_unphysical_cosmology_behaves_like_the_reference(kind):
  """A strongly closed universe with E(z)^2 < 0 beyond z ~ 2: the distance tables carry NaNs and dL(z) is not monotonic.  The
  reference's integrand is 0 * NaN = NaN wherever the event grid reaches the NaN region -- also outside the KDE's support,
  which the kernels skip -- so those events are -inf; the table search must follow the reference's bisection on the
  unsorted table.  (Found by tests/tools/fuzz_extreme.py.)"""
  lam = {'H0': 20.475122477377834, 'Om0': 0.04178692963304655, 'Ok0': -0.27540350807880004, 'Xi0': 0.28514090098152856,
         'n': 2.349145907232978, 'gamma': 1.54, 'kappa': 4.18, 'zp': 4.63, 'm_low': 6.16, 'm_high': 163.4, 'beta': -2.9,
         'alpha': 2.86, 'lambda_peak': 0.91, 'mu_g': 42.2, 'sigma_g': 6.9, 'delta_m': 6.7}
  pixelated = kind is not None
  cfg, ev, inj = H.small_config(E=16, S=512, P=5, Z=120, I=8000, seed=77, ragged=True, pixelated=pixelated)
  models = dict(mass='plp', cosmo='mg_flrw')
  like_p, _, _ = H.build_product(ev, inj, pixelated=pixelated, kind=kind, models=models)
  like_o, _, _ = H.build_oracle(ev, inj, pixelated=pixelated, kind=kind, models=models)
  with np.errstate(all='ignore'):
    ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
  assert np.any(np.isneginf(ro[0])) and np.any(np.isfinite(ro[0]))           # some grids reach the NaN region, some do not
  H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
  np.testing.assert_array_equal(np.isneginf(rp[0]), np.isneginf(ro[0]))
  np.testing.assert_allclose(rp[2], ro[2], rtol=1e-10)


@pytest.mark.parametrize('case', [
  ('approximate', {}, 909250, 'tpl', 'flrw',
   {'H0': 26.518558671537647, 'Om0': 0.871725516050871, 'gamma': 6.4229324865850135, 'kappa': 3.427411651741881, 'zp': 4.856931881940628,
    'm_low': 8.351102188184093, 'm_high': 51.295931649573774, 'beta': 0.6654668225932907, 'Ok0': -0.03427301592113163, 'alpha': 8.560190290864174}),
  ('marginalized', {'cut_grid': None, 'num_bins': 31}, 605202, 'plp', 'mg_flrw',
   {'H0': 173.0737421847404, 'Om0': 0.27016236329203797, 'gamma': 1.9623967506108637, 'kappa': 6.005486513875636, 'zp': 1.1259641290540021,
    'm_low': 3.1865187151476166, 'm_high': 69.80316153457956, 'beta': -0.7449746903909409, 'w0': -1.904242589617507, 'wa': 0.19610915312668364,
    'Xi0': 5.642797377880129, 'n': 3.192431016954092, 'alpha': 8.38535074257262, 'lambda_peak': 0.8860688900006598, 'mu_g': 13.037789838841249,
    'sigma_g': 11.113989312380001, 'delta_m': 2.1056043915480016})])
def test_kde_is_an_exact_zero_above_the_last_weighted_bin(case):
  """A handful of samples carry all the weight, the catalogue term is sizeable only far above them: the dense kernel sum is an exact
  zero there, and a prefix-sum difference that is off by one ulp of the total (1e-16 of the peak) would decide log L_i (it came
  out at -48 / -53 against the reference's -112 / -722).  The bin ranges are clipped to the last lane chunk that holds weight.
  (Found by tests/tools/fuzz_extreme_modes.py.)"""
  kind, like_kw, seed, mass, cosmo, lam = case
  cfg, ev, inj = H.small_config(E=6, S=300, P=3, Z=50, I=3000, seed=seed, ragged=True)
  models = dict(mass=mass, cosmo=cosmo)
  like_p, _, _ = H.build_product(ev, inj, kind=kind, like_kw=like_kw, models=models)
  like_o, _, _ = H.build_oracle(ev, inj, kind=kind, like_kw=like_kw, models=models)
  with np.errstate(all='ignore'):
    ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
  assert np.nanmin(ro[0][ro[0] > -1e300]) < -100.                             # the event in question is there
  H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)


def test_weights_spanning_many_decades_fall_back_to_the_dense_kernel_sum():
  """Weights from 1e-271 to 1e-7 with n_eff = 4: the light bins above the bulk are below the rounding of the prefix sums (1e-16 of the
  weight summed so far), yet they alone meet the catalogue term, so they decide log L_i (-44.128; the prefix form gave -44.134).
  The general KDE evaluation switches to the reference's dense sum where the bins in reach hold < 1e-4 of the weight below them.
  (Found by tests/tools/fuzz_extreme_modes.py.)"""
  lam = {'H0': 109.89167374980342, 'Om0': 0.8346657563998994, 'gamma': 7.807241135391621, 'kappa': 3.30528277152718, 'zp': 4.6230440999507145,
         'm_low': 8.989582612260971, 'm_high': 186.04784716991537, 'beta': 4.559581991531361, 'w0': -1.9249785711872833, 'wa': -0.9677983626855067,
         'Xi0': 4.436192160268282, 'n': 3.364281867399951, 'alpha': 2.696354492046777, 'lambda_peak': 0.7837617293451652, 'mu_g': 57.98524711980541,
         'sigma_g': 13.949408055627696, 'delta_m': 11.754930403127213}
  cfg, ev, inj = H.small_config(E=6, S=300, P=3, Z=50, I=3000, seed=909250, ragged=True)
  models = dict(mass='plp', cosmo='mg_flrw')
  like_p, _, _ = H.build_product(ev, inj, kind='approximate', models=models)
  like_o, _, _ = H.build_oracle(ev, inj, kind='approximate', models=models)
  with np.errstate(all='ignore'):
    ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
  assert -45. < ro[0][1] < -44.
  H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)


def _decades_case():
  """Mock events whose sample weights fall by 60 decades from the nearest to the farthest posterior sample (through pe_prior, which is
  data) while the catalogue term lives only in the upper 40 % of each event's redshift range: the light bins far above the bulk --
  1e-20 ... 1e-50 of the weight summed below them -- alone decide L_i."""
  cfg, ev, inj = H.small_config(E=4, S=2000, P=3, Z=80, I=2000, seed=77, ragged=False)
  ev = dict(ev)
  dL = ev['dL']
  rank = np.argsort(np.argsort(dL, axis=1), axis=1) / (dL.shape[1] - 1.)
  ev['pe_prior'] = dL**2 * 10.**(60. * rank)
  like_o, pop_o, _ = H.build_oracle(ev, inj)
  z = O.get_theta_src_and_weights(pop_o.update(H0=70.), like_o.theta_gw_det)[0].z
  pc = np.array(ev['p_cat'])
  for e in range(cfg['E']):
    m = ev['z_grids'][e] < np.quantile(z[e], 0.6)
    pc[e][:, m] = np.where(pc[e][:, m] != -100., 0., -100.)
  ev['p_cat'] = pc
  return cfg, ev, inj


def test_standard_marginalized_kernel_takes_the_dense_sum_where_the_prefix_differences_are_rounding():
  """The production GW kernel (binning, cut_grid set, marginalized: kde_sub_item) on weights spanning 60 decades: nodes whose bins in
  reach hold < 1e-4 of the weight below them take the reference's dense kernel sum (math.py:77-81) over the pixel's samples, so
  log L_i (-41 ... -49 here) agrees with the oracle to the stated tolerance.  (That the same events come out wrong with the fallback
  switched off -- i.e. that the case does exercise the limit of the prefix-sum form -- is checked on the diagnostic build of the library,
  tests/test_zz_variant_builds.py: the release library has no such switch.)"""
  cfg, ev, inj = _decades_case()
  like_o, _, _ = H.build_oracle(ev, inj)
  with np.errstate(all='ignore'):
    ro = like_o.compute_all(H0=70.)
  assert np.sum(ro[0] < -35.) >= 2 and np.all(np.isfinite(ro[0]))
  like_p, _, _ = H.build_product(ev, inj)
  rp = like_p.compute_all(H0=70.)
  H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
  from chimera_amd import _lib
  if _lib.lib().chm_has_fused():                             # (-DCHM_WITH_FUSED variant: the fused event kernel applies the same bound and redo, chm_fused.h)
    like_p.set_option('fused', 2)
    rf = like_p.compute_all(H0=70.)
    H.assert_loglike_close(rf[0], ro[0], rtol=RTOL_L, atol=1e-9)
  else:
    with pytest.raises(ValueError):                           # [r5] the release library carries no fused kernel
      like_p.set_option('fused', 2)
  with pytest.raises(ValueError):                             # the release library refuses the diagnostic switch
    like_p.set_option('diag_no_dense_node', 1)
  like_p.close()


def test_nan_tail_of_the_distance_table_and_the_scan_search():
  """An unphysical closed universe whose 1/E turns NaN inside the table: jnp.cumsum carries the NaN only from the first NaN term
  on, jnp.interp at its own nodes turns dL NaN one node earlier (0/dx * NaN), and searchsorted (method 'scan') on the NaN-tailed
  table follows its own probe sequence -- N_exp and every event must still agree.  (Found by tests/tools/fuzz_extreme_modes.py.)"""
  lam = {'H0': 172.90880151995614, 'Om0': 0.03371921860963653, 'gamma': -1.06608079317212, 'kappa': 3.136588508730526, 'zp': 2.6460173899198813,
         'm_low': 5.780300710566562, 'm_high': 114.15291593846199, 'beta': 5.123153973860013, 'Ok0': -0.2685268289581615, 'alpha': 1.6484774521175334}
  like_kw = dict(cut_grid=None, num_bins=31)
  cfg, ev, inj = H.small_config(E=6, S=300, P=3, Z=50, I=3000, seed=231555, ragged=True)
  models = dict(mass='tpl', cosmo='flrw')
  like_p, pop_p, sel_p = H.build_product(ev, inj, like_kw=like_kw, models=models)
  like_o, pop_o, sel_o = H.build_oracle(ev, inj, like_kw=like_kw, models=models)
  with np.errstate(all='ignore'):
    ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
    co = pop_o.update(**lam).cosmo
    assert np.isnan(co.integral_invE_interp).any()
    zo = O.z_from_dGW(co, inj['dL'])
    zp = CH.cosmo.z_from_dGW(pop_p.update(**lam).cosmo, inj['dL'])
  np.testing.assert_allclose(zp, zo, rtol=1e-12, atol=0, equal_nan=True)
  np.testing.assert_allclose(rp[2], ro[2], rtol=1e-10)
  H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
  np.testing.assert_array_equal(np.isneginf(rp[0]), np.isneginf(ro[0]))


def test_events_dropping_out_one_by_one_give_the_reference_value_classes():
  """H0 scanned beyond the range the z grids were built for: events lose their likelihood one by one, and log_num / log_hyper go
  ordinary -> exactly -1.79769313e+308 (one such event) -> -inf (two or more), the three classes of the reference notebook's
  `res_H0` (tests/golden/ref_notebook_res_H0.json, tests/test_oracle_pins.py)."""
  cfg, ev, inj = H.small_config(E=5, S=200, P=3, Z=60, I=3000, seed=21, ragged=True)
  like_o, _, _ = H.build_oracle(ev, inj)
  like_p, _, _ = H.build_product(ev, inj)
  big = -np.finfo(np.float64).max
  seen = set()
  for h in np.concatenate([np.linspace(2., 20., 19), np.linspace(200., 900., 15), [70.]]):
    with np.errstate(all='ignore'):
      ro, rp = like_o.compute_all(H0=float(h)), like_p.compute_all(H0=float(h))
    H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
    np.testing.assert_array_equal(rp[0] == big, ro[0] == big)
    n_dead = int(np.sum(ro[0] == big))
    if n_dead == 0:
      np.testing.assert_allclose(rp[3], ro[3], rtol=0, atol=1e-7 * np.sqrt(cfg['E'])); seen.add('ordinary')
    elif n_dead == 1:
      assert rp[1] == big and rp[3] == big and ro[3] == big; seen.add('sentinel')
    else:
      assert np.isneginf(rp[1]) and np.isneginf(rp[3]) and np.isneginf(ro[3]); seen.add('-inf')
  assert seen == {'ordinary', 'sentinel', '-inf'}


def test_vectorised_call_and_sampler_glue(cfg_pix):
  from chimera_amd.utils.emcee_utils import generate_dict, make_log_prob
  cfg, ev, inj = cfg_pix
  like, _, _ = H.build_product(ev, inj)
  H0 = np.array([55., 65., 75.]); al = np.array([3.0, 3.4, 3.8])
  vec = like(H0=H0, alpha=al, gamma=2.7)
  one = np.array([like(H0=h, alpha=a, gamma=2.7) for h, a in zip(H0, al)])
  np.testing.assert_array_equal(vec, one)
  theta = np.stack([H0, al], axis=1)
  assert generate_dict(theta, ['H0', 'alpha'])['alpha'].tolist() == al.tolist()
  lp = make_log_prob(like, ['H0', 'alpha'], priors=[[20., 70.], [1., 5.]])(theta)
  assert lp[2] == -np.inf and np.all(lp[:2] == np.array([like(H0=h, alpha=a) for h, a in zip(H0[:2], al[:2])]))
  with pytest.raises(ValueError):
    like(H0=np.array([60., 70.]), alpha=np.array([3., 3.1, 3.2]))


def test_alternating_z_max_in_scalar_calls_never_replays_another_z_max(cfg_pix):
  """[r5] k_znodes takes z_max by value and k_tables reads the cached nodes: a HIP graph captured for one z_max must not be replayed for
  another.  Scalar calls A, A, A (eager, capture, replay), then B, A, B, A, B, B, B, A: every value must be the one a FRESH handle gives for that z_max
  (a z_max convergence scan with the reference-shaped call; before, the third 'A' after a 'B' replayed B's nodes under A's parameters).  The same with
  an option changed after the capture (the option epoch is part of the key)."""
  cfg, ev, inj = cfg_pix
  seq = [5., 5., 5., 7.5, 5., 7.5, 5., 7.5, 7.5, 7.5, 5.]
  like, _, _ = H.build_product(ev, inj)
  got = [like(H0=69., z_max=zm) for zm in seq]
  want = {}
  for zm in (5., 7.5):
    fresh, _, sf = H.build_product(ev, inj)
    want[zm] = fresh(H0=69., z_max=zm)
    fresh.close(); sf.close()
  assert want[5.] != want[7.5]                                 # the table's nodes move with z_max: the two values differ in the last digits at least
  np.testing.assert_array_equal(np.array(got), np.array([want[zm] for zm in seq]))
  like_o, _, _ = H.build_oracle(ev, inj)
  for zm in (5., 7.5):
    np.testing.assert_allclose(want[zm], like_o(H0=69., z_max=zm), rtol=0, atol=1e-7 * np.sqrt(cfg['E']))
  # a replayed graph must not outlive the options it was captured under
  for _ in range(3):
    a = like(H0=71.)
  like.set_option('serial', 1)
  assert like(H0=71.) == a
  like.set_option('serial', 0)
  assert like(H0=71.) == a
  like.close()


@pytest.mark.parametrize('mname', ['plp', 'bpl', 'tpl'])
def test_pdf_joint_and_marg_matches_the_oracle(mname):
  """[r5] CHIMERA/population/mass.py:351-362, the reference's plotting helper: joint pdf on a mesh of [m_low, m_high]^2 (chm_model_eval) and its two
  trapezoid-normalised marginals (chm_trapz), against the oracle's restatement; each marginal integrates to 1."""
  mp, mo = getattr(CH.mass, mname)(), getattr(O, mname)()
  dp, do = CH.mass.pdf_joint_and_marg(mp, res=(600, 250)), O.pdf_joint_and_marg(mo, res=(600, 250))
  assert dp['p_joint'].shape == (250, 600) and dp['m1mesh'].shape == (250, 600)
  for k in ('m1', 'm2', 'm1mesh', 'm2mesh'):
    np.testing.assert_allclose(dp[k], do[k], rtol=1e-15)
  np.testing.assert_allclose(dp['p_joint'], do['p_joint'], rtol=1e-11, atol=1e-300)
  for k, x in (('p_m1_marg', 'm1'), ('p_m2_marg', 'm2')):
    np.testing.assert_allclose(dp[k], do[k], rtol=1e-11, atol=1e-300)
    if mname != 'tpl':                                         # (tpl has no smoothing: p_m1m2(m_low, m_low) = m^beta / cdf(m_low) = x / 0 = inf, the reference's own
      assert abs(np.trapezoid(dp[k], dp[x]) - 1.) < 1e-12      #  marginals are inf / inf = NaN at the first node and 0 elsewhere -- reproduced, compared above)
    else:
      assert np.isnan(do[k][0]) and np.isnan(dp[k][0])


def test_bench_starts_its_ranks_itself_without_a_launcher():
  """[r5] `bench.py --gpus 2 --host-comm` with no WORLD_SIZE in the environment: the process becomes the launcher, two ranks come up as children
  (here on ONE GPU, through the host sockets: a rehearsal, and the line says so), rank 0's JSON line is the process's stdout and carries n_gpus = 2,
  the world size RCCL / the sockets saw and the PCI bus id of every rank's device."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
  p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--host-comm', '--events', '24', '--inj', '3000', '--nbatch', '12',
                      '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-single-call'], cwd=root, env=env,
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
  assert p.returncode == 0, p.stderr[-1500:]
  line = json.loads(p.stdout.strip().split('\n')[-1])
  assert line['n_gpus'] == 2 and line['multi_gpu']['world'] == 2 and len(line['multi_gpu']['pci_bus_ids']) == 2
  assert 'rehearsal' in line['multi_gpu']['collective'] and np.isfinite(line['value'])
  # [r6] the guarded second leg with two evaluations in flight per rank (two lanes, each with a socket star of its own under --host-comm) ran and agrees
  leg = line['multi_gpu']['inflight2']
  assert 'error' not in leg and leg['lanes'] == 2 and leg['ms_per_step'] > 0 and leg['last_log_hyper'] == line['last_log_hyper'], leg


def test_last_timing_reports_the_stages_of_an_eager_call(cfg_pix):
  """bench.py's roofline block divides by these HIP-event durations: a batched (eager) call must report positive times for the whole
  evaluation, the sample stage, the GW kernel and the selection function; a graph-replayed scalar call carries no events (zeros)."""
  cfg, ev, inj = cfg_pix
  like, _, _ = H.build_product(ev, inj)
  like.batch([dict(H0=60. + i) for i in range(12)])
  ms = like.last_timing()
  assert ms[0] > 0 and ms[2] == 0 and ms[3] == 0, ms         # [r6] the default call carries the events of the whole evaluation only
  like.set_option('timing', 2)                                # per-kernel events on request
  like.batch([dict(H0=60. + i) for i in range(12)])
  ms = like.last_timing()
  assert ms[0] > 0 and ms[2] > 0 and ms[3] > 0 and ms[4] > 0 and ms[0] >= ms[3], ms
  for _ in range(4):
    like(H0=70.)
  assert not np.any(like.last_timing()[:7] > 0)


@pytest.mark.parametrize('models', [dict(), dict(mass='bpl'), dict(mass='tpl', rate='power_law'), dict(cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.6, n=2.1))],
                         ids=['flrw-plp (k_selection_fast)', 'flrw-bpl', 'flrw-tpl-powerlaw', 'mg_flrw (k_selection_fast<., MG>)'])
def test_selection_function_with_hostile_injections(cfg_pix, models):
  """Injections the direct-index front end cannot key (zero, negative, NaN, infinite distances), distances below and beyond the table,
  masses outside the population, an odd count (the last pair is half empty), duplicates: N_exp and the N_eff guard follow the
  reference's nansum / sum (selection_function.py:38-47) in the fast and in the general kernel, batched and one draw at a time."""
  cfg, ev, inj = cfg_pix
  inj = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in inj.items()}
  n = len(inj['dL']) - (1 - len(inj['dL']) % 2)              # odd number of injections
  for k in ('dL', 'm1det', 'm2det', 'p_draw'):
    inj[k] = inj[k][:n]
  inj['dL'][:12] = [0., -3., 1e-12, 1e-6, 2e6, 7e4, 1e-300, 5e-324, 1e300, inj['dL'][20], inj['dL'][20], 3e-3]
  inj['m1det'][30:36] *= 60.
  inj['m2det'][36:42] *= 1e-3
  for bad in (None, np.nan, np.inf):
    if bad is not None:
      inj['dL'][13] = bad
    for N_eff in (None, 5.):
      _, pop_o, sel_o = H.build_oracle(ev, inj, models=models, N_eff=N_eff)
      like_p, pop_p, sel_p = H.build_product(ev, inj, models=models, N_eff=N_eff)
      lams = [dict(H0=58.), dict(H0=70.), dict(H0=93., Om0=0.4)] + [dict(H0=60. + j) for j in range(9)] + [dict(H0=66., Xi0=0.7, n=0.6), dict(Xi0=3.1, n=2.9)]      # (Xi0, n: read by mg_flrw only)
      with np.errstate(all='ignore'):
        ref = np.array([sel_o.N_exp(pop_o.update(**l)) for l in lams])
      one = np.array([sel_p.N_exp(pop_p.update(**l)) for l in lams])
      np.testing.assert_allclose(one, ref, rtol=1e-10, equal_nan=True)
      allp = like_p.compute_all(**lams[1])
      with np.errstate(all='ignore'):
        assert np.isclose(np.exp(allp[2]), ref[1], rtol=1e-10, equal_nan=True) or (ref[1] == 0. and allp[2] == -np.inf)
      like_p.close(); sel_p.close()


@pytest.mark.parametrize('kind', ['marginalized', 'approximate'])
def test_an_event_grid_that_does_not_ascend_is_evaluated_point_by_point_like_the_reference(cfg_pix, kind):
  """[r6] The standard GW kernel ends its grid loop at the first pass beyond the KDE's support and every kernel takes the k-range of an
  event from the two ends of [lb, ub] -- both assume the reference's ascending linspace grids (pop_wrapper.py:207).  A caller-supplied grid with a
  descending step inside the support (two neighbouring points swapped; a whole stretch reversed) goes to the kernels that evaluate every grid
  point, as the reference's arithmetic (pointwise interpolation, jnp.trapezoid with negative steps) does."""
  cfg, ev, inj = cfg_pix
  ev = dict(ev)
  zg = np.array(ev['z_grids'], copy=True)
  Z = zg.shape[1]
  # swap two neighbours in the middle of event 0's grid, reverse a stretch of event 1's
  zg[0, Z // 2], zg[0, Z // 2 + 1] = zg[0, Z // 2 + 1], zg[0, Z // 2]
  zg[1, Z // 3: Z // 3 + 9] = zg[1, Z // 3: Z // 3 + 9][::-1].copy()
  ev['z_grids'] = zg
  like_o, _, _ = H.build_oracle(ev, inj, kind=kind)
  like_p, _, _ = H.build_product(ev, inj, kind=kind)
  for lam in (dict(H0=70.), dict(H0=58., alpha=3.0)):
    _compare(like_p, like_o, lam, cfg['E'], check_pgw=True)
  np.testing.assert_allclose(like_p.batch([dict(H0=70.), dict(H0=58., alpha=3.0)])[0], like_p(H0=70.), rtol=0, atol=0)


def test_results_behind_completion_flags_have_arrived_when_the_call_returns():
  """[r6] Calls of every size complete through flags the last kernel stores in pinned memory behind the results.  A fuzz campaign (~20 000 flagged calls)
  caught two calls returning the content of a freshly allocated result block: the flag had overtaken the results on their way to host memory.  The host
  now marks the block with a NaN payload no result carries and waits until the results have replaced it.  Stress: 20 000 calls of 1-5 draws on one pair
  of handles, and 200 fresh pairs (first-sight calls right after the allocation of the block) -- every value must be the bits of the first evaluation."""
  cfg, ev, inj = H.small_config(E=2, S=128, P=2, Z=32, I=300, seed=11)
  like, _, sel = H.build_product(ev, inj)
  lams = [dict(H0=60. + 3. * i) for i in range(5)]
  want = np.array([like(**l) for l in lams])
  assert np.all(np.isfinite(want))
  bad = 0
  for it in range(5000):
    for n in (1, 2, 3, 5):
      got = like.batch(lams[:n]) if n > 1 else np.array([like(**lams[0])])
      bad += int(not np.array_equal(got, want[:n]))
  assert bad == 0, f"{bad} of 20000 calls returned something else than the first evaluation's bits"
  # calls of many draws (128: the bench's shape; the draws read from pinned memory by k_tables, 384 results + 128 flags written back)
  big = dict(H0=np.linspace(58., 82., 128))
  want_big = like.batch(big)
  assert np.array_equal(want_big[::32], np.array([like(H0=float(h)) for h in big['H0'][::32]]))
  for it in range(3000):
    got = like.batch(big)
    if not np.array_equal(got, want_big):
      bad += 1
  assert bad == 0, f"{bad} of 3000 128-draw calls differ from the first"
  # two lanes on the same resident data, each driven by its own host thread (its own result block and flags)
  import threading
  lane2 = like.lane()
  errs = []

  def drive(ln, n):
    for it in range(3000):
      got = ln.batch(lams[:n])
      if not np.array_equal(got, want[:n]):
        errs.append((n, it, got))
  ts = [threading.Thread(target=drive, args=(like, 2)), threading.Thread(target=drive, args=(lane2, 5))]
  [t.start() for t in ts]; [t.join() for t in ts]
  assert not errs, errs[:3]
  lane2.selection_function.close(); lane2.close()
  like.close(); sel.close()
  for it in range(200):
    lk, _, sl = H.build_product(ev, inj)
    a = lk(**lams[0])
    b = lk.batch(lams[:2])
    c = lk.batch(lams[:5] + lams[:5])
    assert a == want[0] and np.array_equal(b, want[:2]) and np.array_equal(c, np.concatenate([want, want])), (it, a, b, c)
    lk.close(); sl.close()
