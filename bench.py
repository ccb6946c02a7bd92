#!/usr/bin/env python3
"""bench.py -- hyper-likelihood throughput on MI355X (BASELINE.json metric).

A step = one call of ``hyperlikelihood.batch`` with ``--nbatch`` (default 128) different hyper-parameter draws, i.e. nbatch
full hyperposterior evaluations (each: tables + det->src + weights + histogram/KDE + integrand + trapz + selection
function + reduce) over the C3 workload: 1000 events x 32 pixels x 1000 z-bins x 4096 samples/event, 1e5 detected
injections, PowerLaw+Peak + Madau-Dickinson + flat-LCDM, kind_p_gw3d='marginalized', binning(200), cut_grid=2 --
with the inputs already resident in HBM.  Every draw uses a different H0 (tables rebuilt for every draw, as in the
reference's H0 scans, examples/test1dgalaxies.ipynb cell 11).  value = evaluations per second = steps * nbatch / time;
the latency of a single-draw call (nbatch = 1, the reference's scalar call) is reported as ``single_call_ms``.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

N > 1: events and injections are sharded across ranks (strong scaling: the total workload is fixed), one RCCL
all-reduce of 3 * nbatch doubles per step inside chm_eval.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

KERNEL_NAMES = {'marginalized': 'k_kde_marg_sub2<32, 4>', 'full': 'k_full_kde'}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s measured copy)


def algorithmic_bytes(E, S, P, Z, I, B, pixelated=True, full=False):
  """every input read once, nothing materialised (fp64 values, int32 pixel index)."""
  b = E * S * (4 * 8 + (4 if pixelated else 0)) + (E * S * 16 if full else 0)
  b += (E * P * Z * 8 if pixelated else 0) + E * Z * 8 * 2 + E * P * 8 * 3 + I * 32 + E * 8
  return b


def kde_kernel_bytes(E, S, P, Z):
  """Algorithmic bytes of ONE launch of the dominant kernel (k_kde_integrate): z, w (fp64) and the pixel index
  (int32) of every sample once, p_cat once, the event grid and the three per-z factors once, per-pixel scalars."""
  return E * S * (8 + 8 + 4) + E * P * Z * 8 + E * Z * 8 * 4 + E * P * 8 * 3


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--config', default='C3')
  ap.add_argument('--mode', default='marginalized')
  ap.add_argument('--nbatch', type=int, default=128, help='hyper-parameter draws per chm_eval call (hyperlikelihood.batch)')
  ap.add_argument('--events', type=int, default=None, help='shrink the number of events (debug)')
  ap.add_argument('--inj', type=int, default=None, help='shrink the number of injections (debug)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--single-call', action='store_true', help='also time the scalar one-draw call (extra launches after the timed region)')
  ap.add_argument('--host-comm', action='store_true', help='reduce the partial sums through the host (gloo) instead of RCCL')
  ap.add_argument('--force-comm', action='store_true', help='build the gloo group and the RCCL communicator even for one rank (rehearses the N > 1 path)')
  ap.add_argument('--cpu-events', type=int, default=1000, help='events of the workload the CPU baseline evaluates (1000 = all of C3, ~15 s)')
  args = ap.parse_args()

  # stdout must carry ONE JSON line: gloo ("[Gloo] Rank 0 is connected ...") and RCCL (its version banner) print to fd 1 when
  # they come up, so fd 1 points at stderr for the whole run and the JSON line goes to the saved descriptor
  sys.stdout.flush()
  real_stdout = os.dup(1)
  os.dup2(2, 1)
  rank = int(os.environ.get('RANK', 0))
  world = int(os.environ.get('WORLD_SIZE', 1))
  local_rank = int(os.environ.get('LOCAL_RANK', 0))
  if world != args.gpus and world > 1:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
  # one rank per GPU; on a box with fewer GPUs than ranks (a rehearsal with --host-comm) the ranks wrap around the devices
  try:
    import torch
    ndev = torch.cuda.device_count()              # counting devices does not initialise the GPU
  except ImportError:
    ndev = 0
  device = local_rank % ndev if ndev > 0 else local_rank
  os.environ.setdefault('CHIMERA_DEVICE', str(device))

  dist = None
  if world > 1 or args.force_comm:
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('gloo', rank=rank, world_size=world)      # control plane only (barrier, max-of-times, id exchange)

  import chimera_amd as CH
  from chimera_amd import synth, _lib
  from chimera_amd.catalog import dVdz_completeness, pixelated_catalog
  from chimera_amd.parallel import Comm

  t0 = time.time()
  cfg, ev, inj = synth.make_config(args.config, E=args.events, I=args.inj)
  E, S, P, Z, I = cfg['E'], cfg['S'], cfg['P'], cfg['Z'], cfg['I']
  pixelated = cfg['pixelated']
  t_gen = time.time() - t0

  comm, comm_kind = None, None
  if world > 1 or args.force_comm:
    from chimera_amd.parallel import HostComm
    err = None
    try:
      comm = None if args.host_comm else Comm(world, rank, device)          # one RCCL rank per GPU
    except Exception as e:                                                        # noqa: BLE001 -- any failure -> host fallback
      err = e
    import torch
    flag = torch.tensor([0 if comm is not None else 1], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.SUM)                                   # every rank must take the same branch
    if int(flag[0]) > 0:
      if comm is not None:
        comm.close()
      if err is not None:
        print(f"[bench] rank {rank}: RCCL communicator failed ({err}); partial sums go through the host (gloo)", file=sys.stderr)
      comm, comm_kind = HostComm(world, rank, device), "host (gloo) all-reduce of 3*nbatch doubles"
    else:
      comm_kind = "RCCL all-reduce of 3*nbatch doubles inside chm_eval"
  mg = args.config == 'C5'                      # BASELINE.json configs[4]: modified GW propagation (Xi0, n)
  cosmo = CH.cosmo.mg_flrw(H0=70., Om0=0.25, z_max=5., Xi0=1.8, n=1.9) if mg else CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.)
  mass = CH.mass.plp()
  rate = CH.rate.madau_dickinson(gamma=2.7, kappa=3., zp=2.)
  pe_fields = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix',
               'gw_loc2d_pdf', 'pixels_pe_opt_nside')
  if pixelated:
    th = CH.data.theta_pe_det(**{k: ev[k] for k in pe_fields})
    gal_cat = pixelated_catalog(dVdz_completeness(z_range=[0.073, 1.3]), p_cat=ev['p_cat'], z_grids=ev['z_grids'],
                                neff_pixels=ev['neff_pixels'])
    kind = args.mode
  else:
    th = CH.data.theta_pe_det(**{k: ev[k] for k in ('m1det', 'm2det', 'dL', 'pe_prior')})
    gal_cat, kind = None, None
  pop = CH.population(cosmo, mass, rate, gal_cat=gal_cat, scale_free=True)
  sel = CH.selection_function(CH.data.theta_inj_det(**{k: inj[k] for k in ('m1det', 'm2det', 'dL', 'p_draw')}),
                              N_inj=inj['N_inj'], N_eff=5., comm=comm)
  like = CH.hyperlikelihood(th, ev['z_grids'], pop, sel, kind_p_gw3d=kind, kernel='epan', bw_method=None, cut_grid=2,
                            binning=True, num_bins=200, comm=comm)

  # one-time hand-over of the host buffers (pixel sort + H2D copies in chm_like_create / chm_sel_create); NOT part of `value`
  t0 = time.time()
  like._handle(); sel._handle()
  t_upload = time.time() - t0
  nb = args.nbatch
  H0s = np.linspace(55., 95., 4099)          # a different H0 for every draw of every step

  Xi0s = np.linspace(0.6, 3.0, 4099)

  def lambdas(step):
    if mg:                                        # a different (Xi0, H0) for every draw
      return [dict(Xi0=float(Xi0s[(step * nb + j) % len(Xi0s)]), H0=float(H0s[(7 * (step * nb + j)) % len(H0s)])) for j in range(nb)]
    return [dict(H0=float(H0s[(step * nb + j) % len(H0s)])) for j in range(nb)]

  def sync():
    try:
      import torch
      if torch.cuda.is_available():
        torch.cuda.synchronize(device)
    except ImportError:
      pass
    if dist is not None:
      dist.barrier()

  # the hyper-parameter draws of every step are the sampler's output, prepared before the timed region; packing them into
  # chm_params (hyperlikelihood._params_array) is part of the call and stays inside it
  draws = [lambdas(k) for k in range(args.warmup + args.steps)]
  vals = []
  for w in range(args.warmup):
    vals.append(like.batch(draws[w]))
  # Python's cyclic collector off the clock: with torch imported (~170 000 tracked objects) its generational passes cost 0.4 ms
  # per step on average (measured, scripts/host_overhead.py) -- nothing in a step creates reference cycles
  import gc
  gc.collect()
  gc.freeze()
  sync()
  kt = np.zeros(8)
  t1 = time.perf_counter()
  for k in range(args.steps):
    vals.append(like.batch(draws[args.warmup + k]))          # synchronous: returns after the HIP stream has drained
    kt += like.last_timing()
  sync()
  dt = time.perf_counter() - t1
  if dist is not None:
    import torch
    tt = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt[0])
  kt /= max(args.steps, 1)
  single_ms = None
  if args.single_call:      # latency of the reference-style scalar call (one draw per call), outside the timed region
    for j in range(3):
      like(H0=float(H0s[-1 - j]))
    sync()
    t2 = time.perf_counter()
    nsingle = 10
    for j in range(nsingle):
      like(H0=float(H0s[-10 - j]))
    sync()
    single_ms = 1e3 * (time.perf_counter() - t2) / nsingle
  traffic = None
  tf = os.path.join(ROOT, 'profiles', 'r01', 'pmc_traffic.json')
  if os.path.exists(tf):
    try:
      with open(tf) as f:
        tj = json.load(f)
      if (tj.get('E') == (like._e1 - like._e0) and tj.get('kernel') == KERNEL_NAMES.get(kind) and cfg['P'] == tj.get('P')
          and cfg['Z'] == tj.get('Z') and tj.get('nbatch') == nb):      # measured on this workload, this many draws per call
        traffic = tj['bytes_per_draw'] * nb
    except Exception:
      traffic = None

  if rank == 0:
    evals = args.steps * nb
    value = evals / dt
    El = like._e1 - like._e0
    kb = kde_kernel_bytes(El, S, P if pixelated else 1, Z) * nb
    kde_ms = kt[3]
    ach = kb / (kde_ms * 1e-3) / 1e9 if kde_ms > 0 else 0.
    path_bytes = algorithmic_bytes(E, S, P, Z, I, 200, pixelated, kind == 'full')
    out = {
      "metric": "log-likelihood evals/sec (full hyperposterior call), N_ev x N_pix x N_z",
      "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
      "dtype": "f64", "data": "synthetic (seed 20250926; chimera_amd/synth.py)",
      "config": {"workload": f"{args.config}: {E} events x {P} pixels x {Z} z-bins, {S} samples/event, {I} detected injections, "
                             f"PLP + Madau-Dickinson + {'modified-GW-propagation (Xi0, n) flat-LCDM' if mg else 'flat-LCDM'}, {kind or '1d'}, binning 200, cut_grid 2",
                 "E": E, "P": P, "Z": Z, "S": S, "I": I, "kind_p_gw3d": kind, "nbatch": nb,
                 "parallelism": f"events+injections sharded over {world} GPU(s)" + (f"; {comm_kind}" if comm_kind else ""),
                 "cells_per_s": value * E * max(P, 1) * Z},
      "single_call_ms": single_ms,
      "roofline": {"bound": "hbm", "kernel": KERNEL_NAMES.get(kind, "k_kde1d+k_integrate_1d"),
                   "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                   "traffic": traffic, "traffic_source": "profiles/r01/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" if traffic else None,
                   "bytes_per_launch": kb, "kernel_ms": kde_ms,
                   "note": "achieved = ALGORITHMIC bytes of the launch (every input of the kernel once: 370.7 MB per draw at C3) / its "
                           "HIP-event duration; the kernel moves far fewer bytes than that (see traffic): it reads only the part of "
                           "each p_cat row inside the KDE's support, and the draws of a call share p_cat and samples through L2 / "
                           "Infinity Cache -- so frac can approach or exceed 1 without the HBM being saturated; the kernel is "
                           "fp64-VALU / latency bound (DESIGN.md section 4)",
                   "path_bytes_per_eval": path_bytes,
                   "path_frac": path_bytes * nb / (kt[0] * 1e-3) / 1e9 / HBM_PEAK_GBS if kt[0] > 0 else None,
                   "stage_ms": {"eval": kt[0], "tables": kt[1], "samples": kt[2], "kde_integrate": kt[3],
                                "selection": kt[4], "reduce": kt[5], "events_wall": kt[6], "event_groups": kt[7]}},
      "setup_s": {"synthetic": t_gen, "upload_once": t_upload,
                  "note": "upload_once = chm_like_create + chm_sel_create (host pixel sort, log(m_det), H2D of the shard); every "
                          "evaluation afterwards moves ~350 B of parameters per draw host->device and 24 B back"},
      "last_log_hyper": float(np.asarray(vals[-1]).ravel()[-1]),
    }
    if world == 1 and not args.no_cpu_baseline:
      out["cpu_baseline"] = cb = cpu_baseline(cfg, ev, inj, kind, args.cpu_events)
      if cb.get("log_hyper_H0_67") is not None:      # full-size parity check of the timed path against the CPU port
        g = float(like(H0=67.))
        out["parity_full_size"] = {"H0": 67., "log_hyper_hip": g, "log_hyper_cpu_port": cb["log_hyper_H0_67"],
                                   "abs_diff": abs(g - cb["log_hyper_H0_67"]), "tolerance": 1e-7 * float(np.sqrt(E))}
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(out) + '\n').encode())
  like.close()
  sel.close()
  if comm is not None:
    comm.close()
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


def cpu_baseline(cfg, ev, inj, kind, n_ev, threads=None, numpy_events=48):
  """CPU baselines on this host, same workload, same algorithm as the reference (dense G x B kernel sums):
  * value: the plain-C / OpenMP restatement (oracle/chimera_oracle_c.c) on `threads` cores (the GPU box's CPU share is 16),
    n_ev events (default: the whole workload) + all injections, 2 evaluations with different H0 (tables rebuilt each time);
  * numpy_1core: the NumPy restatement (oracle/chimera_oracle.py) on one core, on the first `numpy_events` events, scaled."""
  for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(k, '1')
  from oracle import chimera_oracle as O
  E = cfg['E']
  fields = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf',
            'pixels_pe_opt_nside')

  def build(n):
    n = min(n, E)
    sub = {k: (v[:n] if hasattr(v, 'shape') and v.shape[:1] == (E,) else v) for k, v in ev.items()}
    th = O.theta_pe_det(**{k: sub[k] for k in fields if k in sub})
    gc = O.pixelated_catalog(O.dVdz_completeness(), sub['p_cat'], sub['z_grids'], sub['neff_pixels']) if cfg['pixelated'] else None
    pop = O.population(O.flrw(H0=70., Om0=0.25, z_max=5.), O.plp(), O.madau_dickinson(), gal_cat=gc)
    sel = O.selection_function(O.theta_inj_det(**{k: inj[k] for k in ('m1det', 'm2det', 'dL', 'p_draw')}), inj['N_inj'])
    return n, O.hyperlikelihood(th, sub['z_grids'], pop, sel, kind_p_gw3d=kind), pop, sel

  out = {}
  # NumPy, one core, bounded sample
  n_np, like, pop, sel = build(numpy_events)
  popu = pop.update(H0=67.)
  t0 = time.perf_counter(); like.compute_log_likenum(popu); t_ev = time.perf_counter() - t0
  t0 = time.perf_counter(); sel.N_exp(popu); t_sel = time.perf_counter() - t0
  np1 = {"value": 1.0 / (t_ev * E / n_np + t_sel), "cores": 1,
         "sample": f"oracle/chimera_oracle.py, first {n_np} of {E} events ({t_ev:.2f} s) + all {cfg['I']} injections ({t_sel:.3f} s), "
                   f"1 evaluation, event time scaled x{E / n_np:.1f}"}
  if kind != 'marginalized':
    out = dict(np1, unit="evals/s", kind="port", host_cpus=os.cpu_count())
    return out
  # C / OpenMP, `threads` cores, the whole workload
  from oracle import oracle_c as OC
  threads = threads or min(16, os.cpu_count() or 1)
  n_c, like, pop, sel = build(n_ev)
  OC.numlike_marg(like, pop.update(H0=70.), nthreads=threads) if n_c <= 64 else None      # warm the library on small runs only
  H0s = (67., 61., 73., 79.)                       # different H0 per evaluation: the tables are rebuilt each time
  t0 = time.perf_counter()
  vals = [OC.compute_all(like, dict(H0=h), nthreads=threads)[3] for h in H0s]
  t_c = (time.perf_counter() - t0) / len(H0s)
  val = vals[0]
  t_full = t_c * (E / n_c) if n_c < E else t_c
  out = {"value": 1.0 / t_full, "unit": "evals/s", "cores": threads, "kind": "port",
         "sample": f"oracle/chimera_oracle_c.c (C + OpenMP, {threads} threads), {n_c} of {E} events + all {cfg['I']} injections, "
                   f"{len(H0s)} evaluations at different H0 ({t_c:.2f} s each; {len(H0s) * t_c * threads:.0f} core-seconds in all)",
         "host_cpus": os.cpu_count(), "numpy_1core": np1, "log_hyper_H0_67": float(val) if n_c == E else None}
  return out


if __name__ == '__main__':
  main()
