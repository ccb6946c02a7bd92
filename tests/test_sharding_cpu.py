"""N > 1 path on CPU: two gloo ranks shard events and injections with the product's own partition rule
(chimera_amd.parallel.chunk_bounds), each rank forms the three partial sums the HIP path forms per shard
([sum_i log L_i, nansum dN, sum dN^2], here with the oracle as the per-shard evaluator), one SUM all-reduce, and the
shared combination -- the result must equal the single-process evaluation.  (On the GPU the same three doubles are
all-reduced by RCCL inside chm_eval; tests/test_gpu_parity.py::test_sharded_partials checks the device partials.)"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
  return p


def _worker(rank, world, port, outdir):
  sys.path.insert(0, ROOT)
  import torch
  import torch.distributed as dist
  from chimera_amd.parallel import chunk_bounds
  from tests import helpers as H
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  cfg, ev, inj = H.small_config(E=7, S=96, P=3, Z=32, I=501, seed=31, ragged=True)
  results = []
  for kind, pop_kw in (('marginalized', {}), ('approximate', dict(scale_free=False, R0=15., Tobs=2.))):
    like, pop0, sel = H.build_oracle(ev, inj, kind=kind, pop_kw=pop_kw)
    for lam in (dict(H0=70.), dict(H0=61., alpha=3.0)):
      e0, e1 = chunk_bounds(cfg['E'], world, rank)
      i0, i1 = chunk_bounds(cfg['I'], world, rank)
      part = torch.from_numpy(H.shard_partials_oracle(like, lam, e0, e1, i0, i1))
      dist.all_reduce(part, op=dist.ReduceOp.SUM)
      pop = like.population.update(**lam)
      sharded = H.combine_partials(part.numpy(), cfg['E'], pop, inj['N_inj'], sel.N_eff)
      results.append((sharded, like(**lam)))
  np.save(os.path.join(outdir, f'rank{rank}.npy'), np.array(results))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharding_matches_single_process(tmp_path):
  import torch.multiprocessing as mp
  world = 2
  mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  r0, r1 = np.load(tmp_path / 'rank0.npy'), np.load(tmp_path / 'rank1.npy')
  np.testing.assert_array_equal(r0, r1)                                   # every rank forms the same answer
  np.testing.assert_allclose(r0[:, 0], r0[:, 1], rtol=1e-13)              # sharded == single process
  assert np.all(np.isfinite(r0))


def test_partials_with_zero_likelihood_events_overflow_like_the_reference():
  """Two events with L_i = 0 contribute -1.797e308 each; their sum is -inf on one rank or across ranks."""
  big = -np.finfo(np.float64).max
  assert big + big == -np.inf and (big + 1.0) + (big - 2.0) == -np.inf


def test_host_combination_of_all_reduced_partials():
  """parallel.combine_partials (the host form of the device's combination, used by HostComm) against the reference formulas
  (selection_function.py:38-47, likelihood.py:298-300, 313-316), including the N_eff guard and the non-scale-free branch."""
  sys.path.insert(0, ROOT)
  from chimera_amd.parallel import combine_partials
  from tests import helpers as H
  from oracle import chimera_oracle as O
  rng = np.random.default_rng(3)
  for _ in range(50):
    E, N_inj = int(rng.integers(1, 2000)), float(rng.integers(10**4, 10**7))
    s1 = float(rng.uniform(1., 1e4)); s2 = float(s1**2 / rng.uniform(2., 2000.))
    part = np.array([float(rng.uniform(-5e3, 0.)), s1, s2])
    scale_free, N_eff = bool(rng.random() < 0.5), [None, 5., 1e9][int(rng.integers(0, 3))]
    pop = O.population(O.flrw(), O.plp(), O.madau_dickinson(), R0=float(rng.uniform(1., 50.)), Tobs=float(rng.uniform(0.5, 3.)),
                       scale_free=scale_free)
    with np.errstate(all='ignore'):
      want = H.combine_partials(part, E, pop, N_inj, N_eff)
      got = combine_partials(part, E, N_inj, N_eff, scale_free, pop.R0, pop.Tobs)
    if np.isfinite(want):
      np.testing.assert_allclose(got[0], want, rtol=1e-14)
    else:
      assert got[0] == want or (np.isnan(got[0]) and np.isnan(want))


# ----------------------------------------------------------------------------------------------------------
# the product's own control plane (chimera_amd.parallel.Rendezvous / HostComm: sockets, no torch)
# ----------------------------------------------------------------------------------------------------------
def _rdzv_worker(rank, world, addr, outdir):
  sys.path.insert(0, ROOT)
  from chimera_amd.parallel import Rendezvous, HostComm, chunk_bounds, combine_partials
  from tests import helpers as H
  rd = Rendezvous(world, rank, address=addr, timeout=60.)
  rec = {}
  rec['bcast'] = rd.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
  x = np.array([rank + 1.0, -0.5 * rank, 1e300 if rank == 1 else 1.0])
  rec['sum'] = rd.allreduce_sum(x)
  rec['max'] = rd.allreduce_max(x)
  rd.barrier()
  # the sharded evaluation as bench.py / hyperlikelihood run it with a HostComm: partition, three partial sums per shard (the oracle is
  # the per-shard evaluator here, the HIP library on the GPU), one SUM all-reduce over the sockets, the shared combination
  comm = HostComm(world, rank, device=0, rendezvous=rd)
  cfg, ev, inj = H.small_config(E=7, S=96, P=3, Z=32, I=501, seed=31, ragged=True)
  like, pop0, sel = H.build_oracle(ev, inj, kind='marginalized')
  res = []
  for lam in (dict(H0=70.), dict(H0=61., alpha=3.0)):
    e0, e1 = chunk_bounds(cfg['E'], world, rank)
    i0, i1 = chunk_bounds(cfg['I'], world, rank)
    tot = comm.allreduce_sum(H.shard_partials_oracle(like, lam, e0, e1, i0, i1))
    pop = like.population.update(**lam)
    got = combine_partials(tot, cfg['E'], inj['N_inj'], sel.N_eff, pop.scale_free, pop.R0, pop.Tobs)[0]
    res.append((got, like(**lam)))
  rec['res'] = np.array(res)
  np.savez(os.path.join(outdir, f'rank{rank}.npz'), bcast=np.frombuffer(rec['bcast'], dtype=np.uint8), s=rec['sum'], m=rec['max'], res=rec['res'])
  rd.barrier()
  comm.close()
  rd.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world,tcp', [(2, False), (3, False), (2, True)])
def test_rendezvous_star_and_host_comm_sharding(tmp_path, world, tcp):
  """chimera_amd.parallel.Rendezvous over a Unix-domain socket (one node) or TCP: broadcast of the 128-byte id, SUM / MAX
  all-reduce (same bits on every rank), barrier; and the HostComm-sharded evaluation equals the single-process one."""
  import multiprocessing as mp
  ctx = mp.get_context('spawn')
  addr = ('127.0.0.1', _free_port()) if tcp else str(tmp_path / 'rdzv.sock')
  procs = [ctx.Process(target=_rdzv_worker, args=(r, world, addr, str(tmp_path))) for r in range(world)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(120)
    assert p.exitcode == 0
  recs = [np.load(tmp_path / f'rank{r}.npz') for r in range(world)]
  want_sum = np.sum([[r + 1.0, -0.5 * r, 1e300 if r == 1 else 1.0] for r in range(world)], axis=0)
  want_max = np.max([[r + 1.0, -0.5 * r, 1e300 if r == 1 else 1.0] for r in range(world)], axis=0)
  for rec in recs:
    assert bytes(rec['bcast']) == bytes(range(128))
    np.testing.assert_array_equal(rec['s'], recs[0]['s'])
    np.testing.assert_allclose(rec['s'], want_sum, rtol=1e-15)
    np.testing.assert_array_equal(rec['m'], want_max)
    np.testing.assert_array_equal(rec['res'], recs[0]['res'])
    np.testing.assert_allclose(rec['res'][:, 0], rec['res'][:, 1], rtol=1e-13)
  if not tcp:
    assert not os.path.exists(addr)                          # the hub removes its socket path


def test_default_rendezvous_address_is_per_launcher():
  sys.path.insert(0, ROOT)
  from chimera_amd import parallel
  old = dict(os.environ)
  try:
    os.environ.pop('CHIMERA_COMM_ADDR', None)
    os.environ['MASTER_PORT'] = '29999'
    a = parallel.default_address()
    assert isinstance(a, str) and '29999' in a and str(os.getppid()) in a
    os.environ['CHIMERA_COMM_ADDR'] = 'node7:4242'
    assert parallel.default_address() == ('node7', 4242)
  finally:
    os.environ.clear(); os.environ.update(old)

# ----------------------------------------------------------------------------------------------------------
# the 'params' scheme (CHIMERA/parallel.py:258-278): replicas, the draws of a batch split over the ranks, values gathered
# ----------------------------------------------------------------------------------------------------------
def _params_worker(rank, world, addr, outdir):
  sys.path.insert(0, ROOT)
  import chimera_amd as CH
  from chimera_amd.parallel import Rendezvous, HostComm
  from tests import helpers as H
  rd = Rendezvous(world, rank, address=addr, timeout=60.)
  comm = HostComm(world, rank, device=0, rendezvous=rd)
  cfg, ev, inj = H.small_config(E=5, S=64, P=3, Z=24, I=301, seed=19, ragged=True)
  like_o, _, _ = H.build_oracle(ev, inj)
  like, pop, sel = H.build_product(ev, inj)                   # (constructing the product object needs no GPU; its evaluator is replaced below)
  like = CH.hyperlikelihood(like.theta_gw_det, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', comm=comm, scheme='params')
  assert (like._e0, like._e1) == (0, cfg['E'])               # replicas: every rank holds every event
  calls = []

  def local(lams):                                            # the per-rank evaluator: the oracle here, the HIP library on the GPU
    calls.append(len(lams))
    return np.array([like_o(**l) for l in lams])
  like._batch_local = local
  lams = [dict(H0=float(h)) for h in np.linspace(60., 80., 7)]
  got = like.batch(lams)
  one = like(H0=71.)                                          # a scalar call: rank 0 evaluates, every rank receives
  np.savez(os.path.join(outdir, f'rank{rank}.npz'), got=got, one=one, calls=np.array(calls))
  rd.barrier()
  comm.close()
  rd.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world', [2, 3])
def test_params_scheme_splits_the_draws_and_gathers_the_values(tmp_path, world):
  """scheme='params': rank r evaluates the draws [r c, min((r + 1) c, n)), c = ceil(n / R) (parallel.py:262-264), and every rank ends up with
  all n values -- equal on every rank to the last bit and equal to the single-process values."""
  import multiprocessing as mp
  ctx = mp.get_context('spawn')
  addr = str(tmp_path / 'rdzv.sock')
  procs = [ctx.Process(target=_params_worker, args=(r, world, addr, str(tmp_path))) for r in range(world)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(200)
    assert p.exitcode == 0
  sys.path.insert(0, ROOT)
  from tests import helpers as H
  cfg, ev, inj = H.small_config(E=5, S=64, P=3, Z=24, I=301, seed=19, ragged=True)
  like_o, _, _ = H.build_oracle(ev, inj)
  want = np.array([like_o(H0=float(h)) for h in np.linspace(60., 80., 7)])
  recs = [np.load(tmp_path / f'rank{r}.npz') for r in range(world)]
  per = -(-7 // world)
  for r, rec in enumerate(recs):
    np.testing.assert_array_equal(rec['got'], recs[0]['got'])
    np.testing.assert_array_equal(rec['got'], want)
    assert float(rec['one']) == like_o(H0=71.)
    n_own = max(0, min((r + 1) * per, 7) - min(r * per, 7))
    assert list(rec['calls']) == ([n_own] if n_own else []) + ([1] if r == 0 else [])


def test_params_scheme_argument_checks():
  sys.path.insert(0, ROOT)
  import chimera_amd as CH
  from tests import helpers as H
  cfg, ev, inj = H.small_config(E=3, S=32, P=2, Z=16, I=101, seed=2)
  like, pop, sel = H.build_product(ev, inj)
  with pytest.raises(ValueError, match="scheme"):
    CH.hyperlikelihood(like.theta_gw_det, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', scheme='both')

  class FakeComm:
    nranks, rank, device, handle = 2, 0, 0, None
  sel.comm = FakeComm()
  with pytest.raises(ValueError, match="replicates the data"):
    CH.hyperlikelihood(like.theta_gw_det, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', comm=FakeComm(), scheme='params')


# ----------------------------------------------------------------------------------------------------------
# the 'both' scheme (CHIMERA/parallel.py:132-224, 306-341, 380-406): groups of ranks; data sharded inside a group, draws split over the groups
# ----------------------------------------------------------------------------------------------------------
def test_group_layout_and_draw_slices_follow_the_reference_rule():
  sys.path.insert(0, ROOT)
  from chimera_amd.parallel import group_layout, draws_of_group
  # 8 ranks in 3 parameter batches: sizes 3, 3, 2 (parallel.py:138-141), consecutive ranks (parallel.py:142-147)
  lay = [group_layout(8, r, 3) for r in range(8)]
  assert lay == [(0, 0, 3), (0, 1, 3), (0, 2, 3), (1, 0, 3), (1, 1, 3), (1, 2, 3), (2, 0, 2), (2, 1, 2)]
  assert [group_layout(4, r, 4) for r in range(4)] == [(r, 0, 1) for r in range(4)]
  assert [group_layout(4, r, 1) for r in range(4)] == [(0, r, 4) for r in range(4)]
  for bad in (0, 9):
    with pytest.raises(ValueError):
      group_layout(8, 0, bad)
  # 7 draws over 3 batches: 3, 2, 2 (parallel.py:311-321); fewer draws than batches: the last batches get none
  assert [draws_of_group(7, 3, g) for g in range(3)] == [(0, 3), (3, 5), (5, 7)]
  assert [draws_of_group(1, 3, g) for g in range(3)] == [(0, 1), (1, 1), (1, 1)]
  assert [draws_of_group(0, 2, g) for g in range(2)] == [(0, 0), (0, 0)]


def _both_worker(rank, world, ngroups, addr, outdir):
  sys.path.insert(0, ROOT)
  import chimera_amd as CH
  from chimera_amd.parallel import Rendezvous, HostComm, split, chunk_bounds, group_layout
  from tests import helpers as H
  rd = Rendezvous(world, rank, address=addr, timeout=60.)
  wcomm = HostComm(world, rank, device=0, rendezvous=rd)
  grp = split(wcomm, ngroups)
  g, r, n = group_layout(world, rank, ngroups)
  assert (grp.group_id, grp.rank, grp.nranks, grp.ngroups) == (g, r, n, ngroups) and grp.world is wcomm
  cfg, ev, inj = H.small_config(E=7, S=64, P=3, Z=24, I=301, seed=23, ragged=True)
  like_o, _, sel_o = H.build_oracle(ev, inj)
  like, pop, sel = H.build_product(ev, inj, comm=grp)         # (no GPU is touched: the per-rank evaluator is replaced below)
  like = CH.hyperlikelihood(like.theta_gw_det, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', comm=grp, scheme='both')
  assert (like._e0, like._e1) == chunk_bounds(cfg['E'], n, r)   # events sharded INSIDE the group
  i0, i1 = chunk_bounds(cfg['I'], n, r)
  calls = []

  def local(lams):                                            # the 'data' scheme inside the group, the oracle as the shard evaluator
    calls.append(len(lams))
    out = []
    for lam in lams:
      part = grp.allreduce_sum(H.shard_partials_oracle(like_o, lam, like._e0, like._e1, i0, i1)) if n > 1 else \
        H.shard_partials_oracle(like_o, lam, like._e0, like._e1, i0, i1)
      out.append(H.combine_partials(np.asarray(part), cfg['E'], like_o.population.update(**lam), inj['N_inj'], sel_o.N_eff))
    return np.array(out)
  like._batch_local = local
  lams = [dict(H0=float(h)) for h in np.linspace(60., 80., 7)]
  got = like.batch(lams)
  one = like(H0=71.)                                          # a scalar call: group 0 evaluates, every rank receives
  np.savez(os.path.join(outdir, f'rank{rank}.npz'), got=got, one=one, calls=np.array(calls), group=np.array([g, r, n]))
  rd.barrier()
  grp.close()
  wcomm.close()
  rd.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world,ngroups', [(4, 2), (5, 2), (3, 3)])
def test_both_scheme_shards_inside_groups_and_splits_the_draws_over_them(tmp_path, world, ngroups):
  """scheme='both': the world splits into `ngroups` groups of consecutive ranks (own socket star each), a group evaluates its slice of the draws
  with events and injections sharded over its ranks, the world assembles the values: every rank holds all n values, equal to the last bit on
  every rank and equal to the single-process values to rounding (the shards' sums are added in another order)."""
  import multiprocessing as mp
  ctx = mp.get_context('spawn')
  addr = str(tmp_path / 'rdzv.sock')
  procs = [ctx.Process(target=_both_worker, args=(r, world, ngroups, addr, str(tmp_path))) for r in range(world)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(200)
    assert p.exitcode == 0
  sys.path.insert(0, ROOT)
  from chimera_amd.parallel import draws_of_group
  from tests import helpers as H
  cfg, ev, inj = H.small_config(E=7, S=64, P=3, Z=24, I=301, seed=23, ragged=True)
  like_o, _, _ = H.build_oracle(ev, inj)
  want = np.array([like_o(H0=float(h)) for h in np.linspace(60., 80., 7)])
  recs = [np.load(tmp_path / f'rank{r}.npz') for r in range(world)]
  for rec in recs:
    np.testing.assert_array_equal(rec['got'], recs[0]['got'])
    np.testing.assert_allclose(rec['got'], want, rtol=1e-12)
    np.testing.assert_allclose(float(rec['one']), like_o(H0=71.), rtol=1e-12)
    g = int(rec['group'][0])
    a, b = draws_of_group(7, ngroups, g)
    assert list(rec['calls']) == ([b - a] if b > a else []) + ([1] if g == 0 else [])


# ---- [r6] the collective sequencer of a rank with several evaluation lanes (chm_comm_set_ticket & co.; host code: no GPU needed) ----
def _ticket_lib():
  from chimera_amd import _lib
  L = _lib.lib()
  L.chm_comm_ticket_reset(0)
  L.chm_comm_ticket_timeout(120000)
  return L


def test_a_skipped_ticket_never_lets_a_higher_one_overtake_a_lower_one():
  """chm_comm_ticket_skip(k) used to serve ticket k + 1 at once, in front of a ticket k - 1 still on its way to its collective on this
  rank only -- the cross-rank order the tickets exist for.  Tickets 0, 1, 2: 1 is forfeited first, 2 must still wait for 0."""
  import threading
  import time
  L = _ticket_lib()
  order = []

  def lane(ticket, delay):
    time.sleep(delay)
    assert L.chm_comm_ticket_wait(ticket) == 0
    order.append(ticket)
    assert L.chm_comm_ticket_done(ticket) == 0

  t2 = threading.Thread(target=lane, args=(2, 0.))
  t2.start()
  assert L.chm_comm_ticket_skip(1) == 0                     # the step with ticket 1 failed before its call
  time.sleep(0.3)
  assert order == [], "ticket 2 was served while ticket 0 had not enqueued its collective"
  t0 = threading.Thread(target=lane, args=(0, 0.))
  t0.start()
  t0.join(10); t2.join(10)
  assert order == [0, 2]
  # forfeiting tickets out of order, then the one in front: everything behind it is through at once
  L.chm_comm_ticket_reset(10)
  for k in (13, 12, 11):
    L.chm_comm_ticket_skip(k)
  t0 = time.time()
  assert L.chm_comm_ticket_wait(10) == 0 and L.chm_comm_ticket_done(10) == 0
  assert L.chm_comm_ticket_wait(14) == 0 and L.chm_comm_ticket_done(14) == 0
  assert time.time() - t0 < 1.0
  L.chm_comm_ticket_reset(0)


def test_a_ticket_whose_turn_never_comes_fails_once_and_forfeits():
  """The timeout is a run-time setting (chm_comm_ticket_timeout); a ticket that times out fails ONCE with CHM_E_RCCL (round 5: the destructor of the
  call waited a second time) and is forfeited, so that the tickets behind it are served as soon as the ones in front of it are."""
  import time
  from chimera_amd import _lib
  L = _ticket_lib()
  assert L.chm_comm_ticket_timeout(0) != 0                   # refused
  assert L.chm_comm_ticket_timeout(200) == 0
  t0 = time.time()
  rc = L.chm_comm_ticket_wait(1)                             # ticket 0 never comes
  dt = time.time() - t0
  assert rc == _lib.CHM_E_RCCL and 0.15 < dt < 2.0, (rc, dt)
  assert b'timeout' in L.chm_last_error()
  # ticket 1 is forfeited: once ticket 0 is through, ticket 2 does not wait for it
  assert L.chm_comm_ticket_wait(0) == 0 and L.chm_comm_ticket_done(0) == 0
  t0 = time.time()
  assert L.chm_comm_ticket_wait(2) == 0 and L.chm_comm_ticket_done(2) == 0
  assert time.time() - t0 < 0.15
  L.chm_comm_ticket_timeout(120000)
  L.chm_comm_ticket_reset(0)
