#!/usr/bin/env python3
"""bench.py -- hyper-likelihood throughput on MI355X (BASELINE.json metric).

A step = one call of ``hyperlikelihood.batch`` with ``--nbatch`` (default 128) different hyper-parameter draws, i.e. nbatch
full hyperposterior evaluations (each: tables + det->src + weights + histogram/KDE + integrand + trapz + selection
function + reduce) over the C3 workload: 1000 events x 32 pixels x 1000 z-bins x 4096 samples/event, 1e5 detected
injections, PowerLaw+Peak + Madau-Dickinson + flat-LCDM, kind_p_gw3d='marginalized', binning(200), cut_grid=2 --
with the inputs already resident in HBM.  Every draw uses a different H0 (tables rebuilt for every draw, as in the
reference's H0 scans, examples/test1dgalaxies.ipynb cell 11).  value = evaluations per second = steps * nbatch / time;
the latency of the reference-shaped scalar call (one draw per call) is reported as ``single_call_ms``.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

N > 1: events and injections are sharded across ranks (strong scaling: the total workload is fixed), one RCCL
all-reduce of 3 * nbatch doubles per step inside chm_eval.  Rank 0 prints ONE JSON line.  [r6] After the timed region an N > 1 run adds
a guarded leg with two evaluations in flight per rank (multi_gpu.inflight2: a hang costs the leg, not the line) and its line carries the
roofline block of rank 0's shard (PMC counters of the whole workload scaled by the shard's share, the probe ceilings); an N = 1 run adds
short legs of the other BASELINE configurations and call modes (extra.configs: C1, C2, C4, approximate, full, one-draw kernel times).  No PyTorch anywhere: the
launcher only provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT; the ranks meet over chimera_amd.parallel.Rendezvous
(a Unix-domain socket) for the RCCL unique id, the barriers and the max-over-ranks of the wall time.

Roofline bookkeeping: the two kernels that make up ~90 % of a step -- the sample stage ``k_samples`` and
the marginalized GW kernel ``k_kde_marg_sub2`` -- are bound by VALU ISSUE.  [r4] The headline ``roofline.frac`` is the USEFUL fraction: the
issue cycles of the launch's fp64 add / mul / fma instructions (PMC class counters x 4 cycles) over the cycles 1024 SIMDs offer at 2.4 GHz
during the launch -- moves, selects, lane reads, integer and address arithmetic do not count as achieved work; ``issue_busy_frac`` (every
VALU instruction at its issue cost) says how full the issue ports were, ``min_inst`` what the kernel would need at the least per unit of
work and how far above that it is.  Each kernel is priced with the issue costs measured on the card
(scripts/issue_cost.hip -> profiles/r03/issue_cost.txt: 2 cycles per wave64 instruction for a few simple 32-bit opcodes, 16 for fp64
reciprocal / square root, 8 for fp32 transcendentals, 4 for EVERYTHING else -- fp64 arithmetic, v_mov_b64, v_cndmask, DPP moves,
v_readlane, 64-bit integer operations alike): busy cycles = PMC instruction counts of one launch (committed passes of this very command,
profiles/rNN/pmc_per_launch*.json) x those costs, with the share of 2-cycle opcodes taken from the static mix of the kernel's hot loop
(scripts/isa_mix.py, stored in the same file).  frac = busy cycles / (1024 SIMDs x 2.4 GHz x the launch's LIVE HIP-event duration); the
same at the clock the chip held under the profile, and the real fp64 flops (FMA = 2, add / mul = 1: SQ_INSTS_VALU_FLOPS_FP64) against
78.6 TFLOP/s are printed beside it.  The PMC file carries the sha256 of the gfx950 code object it was collected from: when the loaded
library's differs, no fraction is printed.  The HBM view (unique bytes of the launch and PMC fabric traffic against 8 TB/s) is kept.
[r6] roofline.frac_of_sustained = the launch's work per second over what the SAME kernel body sustains on a cache-resident workload
(profiles/rNN/probe_ceilings.json, scripts/run_probes.py: the production bodies replayed for > 1 s at the kernels' own occupancy) -- a measured
ceiling in the place of round 5's paper count; roofline.hbm_call_frac = the algorithmic bytes of one evaluation over the scalar call's wall time.
"""
import argparse
import ctypes as C_
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# ranks of one node share device memory handles through dmabuf only on this image's driver (RCCL's intra-node transports): the launcher's environment
# normally carries this; a rank started without it would fail in ncclCommInitRank with hipIpcGetMemHandle: invalid argument.  Read at HSA start-up,
# i.e. at the first HIP call of the process -- nothing has loaded the library yet.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s measured copy)
N_SIMD, CLK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMDs; max shader clock (MI355X_MICROARCH.md, chip-level parameters)
ISSUE_PEAK_TCYC = N_SIMD * CLK_HZ / 1e12             # VALU issue cycles per second of the whole chip at the maximum clock: 2.4576e12
FP64_PEAK_TFLOPS = N_SIMD * CLK_HZ / 4 * 64 * 2 / 1e12    # one fp64 FMA per lane per 4 cycles: 78.6 TFLOP/s (the public fp64 vector figure)
# issue cycles per wave64 instruction on one SIMD (profiles/r03/issue_cost.txt, measured by scripts/issue_cost.hip)
CYC_FAST, CYC_VALU, CYC_TRANS32, CYC_TRANS64 = 2, 4, 8, 16
NPART, SAMPLE_WPB, SAMPLE_CHUNK, NEVSTAT = 16, 8, 4096, 12    # workspace record sizes of chm_kernels.h


def algorithmic_bytes(E, S, P, Z, I, B, pixelated=True, full=False):
  """every input of ONE evaluation read once, nothing materialised (fp64 values, int32 pixel index)."""
  b = E * S * (4 * 8 + (4 if pixelated else 0)) + (E * S * 16 if full else 0)
  b += (E * P * Z * 8 if pixelated else 0) + E * Z * 8 * 2 + E * P * 8 * 3 + I * 32 + E * 8
  return b


def gw_kernel_unique_bytes(E, S, P, Z, nb):
  """Unique bytes of ONE launch of the marginalized GW kernel over nb draws: arrays shared by the draws (p_cat, the event grids,
  segment offsets, sky densities) once; per-draw arrays (z and w of every sample, the two per-z factors on the grid, event
  statistics, per-pixel results) nb times."""
  shared = E * P * Z * 8 + E * Z * 8 + E * (P + 1) * 4 + E * P * 8 + E * 4
  per_draw = E * S * 16 + E * Z * 16 + E * NEVSTAT * 8 + E * P * 8
  return shared + nb * per_draw


def sample_kernel_unique_bytes(E, S, nb, Tc=1500, Tm=1000):
  """Unique bytes of ONE launch of the sample stage: the six per-sample inputs once (dL, m1det, m2det, 1/pe_prior, log m1det,
  log m2det -- shared by the draws); per draw the tables (zt, dLt, m_grid, cdf_m2), the (z, w) output and the partial records."""
  nc = -(-S // SAMPLE_CHUNK) * SAMPLE_WPB
  return E * S * 48 + E * 16 + nb * ((2 * Tc + 2 * Tm) * 8 + E * S * 16 + E * nc * NPART * 8)


def code_object_sha256(lib_path):
  """sha256 of the gfx950 code object inside a HIP shared library (the clang offload bundle's gfx950 entry) -- the key that ties a
  committed PMC file to the binary it was collected from (scripts/isa_mix.py computes the same)."""
  import hashlib
  import struct
  try:
    d = open(lib_path, 'rb').read()
    i = d.find(b'__CLANG_OFFLOAD_BUNDLE__')
    n = struct.unpack_from('<Q', d, i + 24)[0]
    o = i + 32
    for _ in range(n):
      off, size, tl = struct.unpack_from('<QQQ', d, o)
      o += 24
      triple = d[o:o + tl]
      o += tl
      if b'gfx950' in triple:
        return hashlib.sha256(d[i + off:i + off + size]).hexdigest()
  except Exception:                                 # noqa: BLE001
    pass
  return None


def load_pmc(keys, sha=None):
  """Newest committed PMC summary (profiles/rNN/pmc_per_launch*.json, written by scripts/collect_profiles.py from separate
  rocprofv3 --pmc passes of this very command) whose workload keys match; None otherwise.  Returns (path, json, fresh): fresh is
  False when the file names another code object than the loaded library's (then no fraction is derived from it)."""
  best = None
  for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'pmc_per_launch*.json'))):
    try:
      with open(f) as fh:
        j = json.load(fh)
    except Exception:
      continue
    w = j.get('workload')
    if isinstance(w, dict) and all(w.get(k, 0 if k == 'fused' else None) == v for k, v in keys.items()):
      fresh = bool(sha) and j.get('code_object_sha256') == sha
      if best is None or fresh or not best[2]:               # a file collected from the loaded binary wins over a newer one that was not
        best = (os.path.relpath(f, ROOT), j, fresh)
  return best


def load_probe_ceilings(sha):
  """[r6] Measured ceilings of the two hot kernels (scripts/run_probes.py with the -DCHM_PROBE build: the production BODIES of k_kde_marg_sub2 and
  k_samples_fast replayed for > 1 s on a cache-resident workload -- what their instruction streams sustain on the card, at the clock the board holds,
  with HBM out of the picture).  Newest profiles/rNN/probe_ceilings.json; (path, json, fresh): fresh = collected beside the loaded release binary."""
  best = None
  for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'probe_ceilings.json'))):
    try:
      with open(f) as fh:
        j = json.load(fh)
    except Exception:                                 # noqa: BLE001
      continue
    fresh = bool(sha) and j.get('release_code_object_sha256') == sha
    if best is None or fresh or not best[2]:
      best = (os.path.relpath(f, ROOT), j, fresh)
  return best


def pmc_kernel(pmc, prefix):
  if pmc is None:
    return None
  for name, c in pmc[1].get('kernels', {}).items():
    if name.startswith(prefix):
      st = (pmc[1].get('static_mix') or {}).get(name)
      return dict(c, name=name, static=st)
  return None


def quartiles(x):
  x = np.sort(np.asarray(x, dtype=float))
  return float(np.median(x)), float(np.percentile(x, 25)), float(np.percentile(x, 75))


# What a kernel needs AT THE LEAST per unit of work, in wave64 VALU instructions per 64 units (one per lane) -- the floor `achieved / minimal`
# is quoted against ("Minimal instruction counts", derives each figure):
#   sample stage, per sample: z(dL) from the direct-index table 14, 1/(1+z) and the two source masses 9, log(1+z) 10, four exps of the mass
#     model at 13 + their arguments 12, the cdf interpolant 10, smoothing windows 14, one quotient 8, prior + four statistics 9, stores 2 = 140
#   GW kernel, per PAIR of pixels (one wave): 5.1 rounds of 32 samples x 13 (bin index + atomic) = 66, prefix sums of 200 bins x 3 moments
#     (7 bins per lane x 5 + four 5-level scans x 3) = 95, bandwidth and node constants 50, 4.3 passes x (2 points x 2 nodes x 12 + interpolation
#     and integrand 2 x 11 + loop 6) = 327, final scans and stores 32 = 570
#   selection kernel, per injection: the sample stage's 140 minus stores and statistics (11) + E(z), the rate powers and the Jacobian 95 = 224
MIN_INST = {'k_samples': (140., 'sample'), 'k_kde_marg_sub2': (570., 'pair of pixels'), 'k_selection': (224., 'injection')}


def kernel_roofline(label, prefix, ms, unique_bytes, pmc, units=None, probe=None, work=None, scale=None):
  """One kernel against its ceilings.  ms: live HIP-event duration of one launch; units: units of work per launch in waves (MIN_INST);
  probe / work: the measured ceilings (load_probe_ceilings) and the launch's work in the probe's unit (pairs of pixels, samples)."""
  k = pmc_kernel(pmc, prefix)
  if k and scale:                                     # counters of a launch over the whole workload -> this shard's launch (cycle counts and times of the profiled launch stay: the clock)
    k = {kk: (v * scale if (isinstance(v, (int, float)) and (kk.startswith(('SQ_INSTS', 'SQ_WAVES', 'FETCH_SIZE', 'WRITE_SIZE', 'TCC_')))) else v) for kk, v in k.items()}
  fresh = bool(pmc and pmc[2])
  sec = ms * 1e-3
  out = {"kernel": k['name'] if k else prefix, "stage": label, "kernel_ms": ms, "bound": "valu-issue",
         "unique_bytes_per_launch": unique_bytes,
         "hbm_unique_GBs": unique_bytes / sec / 1e9 if sec > 0 else None,
         "hbm_unique_frac": unique_bytes / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else None,
         "pmc_matches_loaded_code_object": fresh}
  if scale:
    out["pmc_counters_scaled_by"] = scale
  if k and sec > 0 and fresh:
    insts = k.get('SQ_INSTS_VALU')
    if insts:
      st = (k.get('static') or {}).get('hot_loop') or {}
      nv = st.get('valu_total') or 0
      fast_share = (st.get('fast', 0) / nv) if nv else 0.
      n_t64 = k.get('SQ_INSTS_VALU_TRANS_F64')
      if n_t64 is None:
        n_t64 = insts * (st.get('f64_trans', 0) / nv if nv else 0.)
      n_t32 = insts * (st.get('trans32', 0) / nv if nv else 0.)
      n_fast = insts * fast_share
      cycles = CYC_FAST * n_fast + CYC_TRANS64 * n_t64 + CYC_TRANS32 * n_t32 + CYC_VALU * (insts - n_fast - n_t64 - n_t32)
      n_amf = sum(k.get(c, 0.) for c in ('SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_FMA_F64'))
      flops = k.get('SQ_INSTS_VALU_FLOPS_FP64')
      out.update({"valu_inst_per_launch": insts, "valu_issue_cycles_per_launch": cycles, "cycles_per_valu_inst": cycles / insts,
                  "static_hot_loop": {kk: st.get(kk) for kk in ('valu_total', 'f64', 'f64_amf', 'f64_trans', 'fast', 'mov', 'cndmask', 'lane', 'valu', 'salu', 'lds', 'vmem')} if st else None,
                  "fp64_add_mul_fma_share_pmc": n_amf / insts if n_amf else None,
                  "valu_busy_Tcycle_s": cycles / sec / 1e12, "valu_busy_frac": cycles / sec / 1e12 / ISSUE_PEAK_TCYC,
                  # [r4] what the kernel ACHIEVES: issue cycles of its fp64 add / mul / fma instructions (4 each) over the cycles on offer
                  "useful_Tcycle_s": CYC_VALU * n_amf / sec / 1e12 if n_amf else None,
                  "useful_frac": CYC_VALU * n_amf / sec / 1e12 / ISSUE_PEAK_TCYC if n_amf else None,
                  "fp64_TFLOPs_real": flops * 64 / sec / 1e12 if flops else None,
                  "fp64_frac_of_78.6_TFLOPs": flops * 64 / sec / 1e12 / FP64_PEAK_TFLOPS if flops else None})
      mi = next((v for kk, v in MIN_INST.items() if prefix.startswith(kk)), None)
      if mi and units:
        # [r5] labelled for what it is: a paper count of the author's, not a measurement -- a diagnostic beside `frac`, not evidence
        out["min_inst"] = {"per_unit_minimal_paper_estimate": mi[0], "unit": mi[1], "per_unit_achieved": insts / units,
                           "achieved_over_paper_estimate": insts / units / mi[0],
                           "note": "the minimum is an ESTIMATE made on paper (DESIGN section 4), not a measured microkernel"}
      if k.get('GRBM_GUI_ACTIVE') and k.get('profiled_ms'):
        clk = k['GRBM_GUI_ACTIVE'] / 8 / (k['profiled_ms'] * 1e-3)         # 8 XCDs count the launch's cycles
        out["clock_GHz_under_profile"] = clk / 1e9
        out["valu_busy_frac_at_held_clock"] = (cycles / scale if scale else cycles) / (N_SIMD * clk * k['profiled_ms'] * 1e-3)      # (of the profiled launch)
    if k.get('FETCH_SIZE') is not None and k.get('WRITE_SIZE') is not None:
      # gfx950: FETCH_SIZE tallies 64 B per 128-B request of wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM)
      traffic = (2 * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024
      out.update({"traffic_bytes_per_launch": traffic, "hbm_traffic_GBs": traffic / sec / 1e9,
                  "hbm_traffic_frac": traffic / sec / 1e9 / HBM_PEAK_GBS})
  insts_ = k.get('SQ_INSTS_VALU') if (k and fresh) else None
  pk = next((v for kk, v in ((probe[1].get('kernels') or {}).items() if probe else ()) if prefix.startswith(kk) or kk.startswith(prefix)), None)
  if pk and work and probe[2] and sec > 0:
    # [r6] the launch against what the SAME body sustains on a cache-resident workload (scripts/run_probes.py): work per second, and -- where the PMC
    # pass of this command counted the launch's VALU instructions -- wave-instructions per second against the probe's
    out["sustained"] = {"unit": pk.get('unit'), "work_per_launch": work, "achieved_per_s": work / sec, "probe_per_s": pk.get('units_per_s'),
                        "frac_of_sustained": work / sec / pk['units_per_s'] if pk.get('units_per_s') else None,
                        "valu_winst_per_s": insts_ / sec if insts_ else None, "probe_valu_winst_per_s": pk.get('valu_winst_per_s'),
                        "valu_rate_over_probe": insts_ / sec / pk['valu_winst_per_s'] if (insts_ and pk.get('valu_winst_per_s')) else None,
                        "probe_clock_GHz": pk.get('clock_GHz'), "source": probe[0]}
  return out


def visible_gpus_sysfs():
  """GPUs of this node as the kernel driver lists them (/sys/class/kfd/kfd/topology/nodes/*/properties: a node with simd_count > 0 is a GPU),
  cut down to the devices ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES leave -- read from sysfs, so that the LAUNCHER never opens the HIP runtime
  (a process that holds a device context beside rank 0 for the whole run).  0 when there is no kfd topology (no AMD GPU driver: no GPU)."""
  base = '/sys/class/kfd/kfd/topology/nodes'
  try:
    n = 0
    for d in os.listdir(base):
      try:
        with open(os.path.join(base, d, 'properties')) as f:
          props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
      except OSError:
        continue
      if int(props.get('simd_count', 0)) > 0:
        n += 1
  except OSError:
    return 0
  for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
    v = os.environ.get(var)
    if v is not None:
      n = min(n, len([x for x in v.split(',') if x.strip() != '']))
  return n


def spawn_ranks(n):
  """Launcher of last resort: n child processes `python3 bench.py <same arguments>` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
  set (one rank per GPU, the ranks meet over chimera_amd.parallel.Rendezvous as under torch.distributed.run).  Rank 0's stdout -- the JSON line --
  is this process's.  The launcher makes NO HIP call: the GPUs are counted from sysfs.  Returns the exit status: non-zero when any rank failed
  (the first rank to fail ends the others: a rank that died leaves its peers in the rendezvous or in a collective), or when the node shows fewer
  GPUs than ranks and the call does not ask for the host-socket rehearsal."""
  import subprocess
  ndev = visible_gpus_sysfs()
  if ndev < n and '--host-comm' not in sys.argv:
    print(f"bench.py: --gpus {n} but {ndev} GPU(s) visible and no launcher environment (WORLD_SIZE): refusing to run fewer ranks under an "
          f"{n}-GPU label (use --host-comm for a rehearsal on fewer GPUs)", file=sys.stderr)
    return 2
  # MASTER_PORT is only a TAG here: the ranks meet on a Unix-domain socket whose name carries it and this launcher's pid (parallel.default_address);
  # nothing binds the number, so there is no port to lose between a probe and its use
  port = 20000 + os.getpid() % 40000
  procs = []
  for r in range(n):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                  stdout=None if r == 0 else subprocess.DEVNULL))
  rc = 0
  try:
    live = list(procs)
    while live and rc == 0:
      time.sleep(0.05)
      for p_ in list(live):
        r_ = p_.poll()
        if r_ is not None:
          live.remove(p_)
          rc = rc or r_
    if rc:                                              # a rank failed: the others would wait for it in the rendezvous (300 s) or in RCCL
      for p_ in live:
        p_.terminate()
      t_end = time.time() + 10.
      for p_ in live:
        try:
          p_.wait(max(0.1, t_end - time.time()))
        except subprocess.TimeoutExpired:
          pass
  finally:
    for p_ in procs:
      if p_.poll() is None:
        p_.kill()
  return rc


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
  ap.add_argument('--no-single-call', action='store_true', help='skip the scalar one-draw call timing (runs after the timed region)')
  ap.add_argument('--single-calls', type=int, default=40, help='scalar calls timed for single_call_ms (median + IQR)')
  ap.add_argument('--no-graph', action='store_true', help='no HIP-graph replay of few-draw calls (keeps the per-kernel HIP-event timings for --nbatch <= 8)')
  ap.add_argument('--serial', action='store_true', help='every kernel of a call on one stream (CHM_OPT_SERIAL; per-kernel durations under a profiler)')
  ap.add_argument('--groups', type=int, default=0, help='event groups alternating between two streams (CHM_OPT_GROUPS; 0 = automatic, 1 = one group)')
  ap.add_argument('--fused', type=int, default=0, help='fused event kernel (CHM_OPT_FUSED; a -DCHM_WITH_FUSED variant build only -- the release library refuses values > 0): 0 never, 1 calls of <= 8 draws, 2 every call')
  ap.add_argument('--host-comm', action='store_true', help='reduce the partial sums through the host sockets instead of RCCL (a rehearsal: the line then says so; never a fallback)')
  ap.add_argument('--force-comm', action='store_true', help='build the rendezvous and the RCCL communicator even for one rank (rehearses the N > 1 path)')
  ap.add_argument('--inflight', type=int, default=1, help='evaluations in flight per rank: 2 = two lanes (hyperlikelihood.lane) driven by two host threads, the steps alternate between them')
  ap.add_argument('--no-inflight2', action='store_true', help='N > 1: skip the second leg with two evaluations in flight per rank (multi_gpu.inflight2)')
  ap.add_argument('--no-extra', action='store_true', help='N = 1: skip the short legs of the other BASELINE configurations / modes (extra.configs)')
  ap.add_argument('--cpu-events', type=int, default=1000, help='events of the workload the CPU baseline evaluates (1000 = all of C3)')
  ap.add_argument('--cpu-evals', type=int, default=20, help='timed CPU evaluations (median + IQR)')
  args = ap.parse_args()

  # [r5] `python3 bench.py --gpus N` WITHOUT a torchrun-style launcher (no WORLD_SIZE in the environment): this process becomes the launcher --
  # it starts the N ranks as child processes BEFORE anything touches a GPU and exits with their status; it never runs one rank under an N-GPU
  # label (such a call used to print a line with n_gpus = 1 -- a driver launching it plainly would have recorded a flat curve)
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    raise SystemExit(spawn_ranks(args.gpus))

  # stdout must carry ONE JSON line: RCCL prints its version banner to fd 1 when it comes up, so fd 1 points at stderr for the
  # whole run and the JSON line goes to the saved descriptor
  sys.stdout.flush()
  real_stdout = os.dup(1)
  os.dup2(2, 1)
  rank = int(os.environ.get('RANK', 0))
  world = int(os.environ.get('WORLD_SIZE', 1))
  local_rank = int(os.environ.get('LOCAL_RANK', 0))
  if world != args.gpus and world > 1:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

  # few-draw calls are replayed from a HIP graph, which carries no timing events: the timed loop keeps its per-kernel timings (its calls
  # stay eager when nbatch <= 8), the scalar-call latency is measured on the graph path unless --no-graph
  graph_max_nb = 0 if args.no_graph else (1 if 1 < args.nbatch <= 8 else None)
  import chimera_amd as CH
  from chimera_amd import synth, _lib
  from chimera_amd.catalog import dVdz_completeness, pixelated_catalog
  from chimera_amd.parallel import Comm, HostComm, Rendezvous

  L = _lib.lib()
  ndev = L.chm_device_count()
  if ndev < 1:
    raise SystemExit("bench.py: no HIP device visible (the library has no CPU path)")
  # one rank per GPU; on a box with fewer GPUs than ranks (a rehearsal with --host-comm) the ranks wrap around the devices
  device = local_rank % ndev
  os.environ['CHIMERA_DEVICE'] = str(device)

  rdzv = Rendezvous(world, rank) if (world > 1 or args.force_comm) else None

  t0 = time.time()
  cfg, ev, inj = synth.make_config(args.config, E=args.events, I=args.inj)
  E, S, P, Z, I = cfg['E'], cfg['S'], cfg['P'], cfg['Z'], cfg['I']
  pixelated = cfg['pixelated']
  t_gen = time.time() - t0

  comm, comm_kind = None, None
  if rdzv is not None:
    err = None
    try:
      comm = None if args.host_comm else Comm(world, rank, device, rendezvous=rdzv)    # one RCCL rank per GPU
    except Exception as e:                                                        # noqa: BLE001 -- any failure -> host fallback
      err = e
    nfail = int(rdzv.allreduce_sum(np.array([0. if comm is not None else 1.]))[0])    # every rank must take the same branch
    nccl_count = None
    if nfail > 0 and not args.host_comm:
      # [r4] no silent fallback: a multi-GPU `value` printed by this script means RCCL carried the all-reduce.  A rank whose communicator
      # does not come up ends the job with a non-zero exit code on every rank (the host-socket path is a rehearsal one asks for: --host-comm)
      if err is not None:
        print(f"[bench] rank {rank}: RCCL communicator failed: {err}", file=sys.stderr)
      if comm is not None:
        comm.close()
      rdzv.barrier()
      rdzv.close()
      raise SystemExit(f"bench.py: RCCL communicator unavailable on {nfail} of {world} rank(s); no line is printed (use --host-comm to rehearse over the host sockets)")
    if args.host_comm:
      comm = HostComm(world, rank, device, rendezvous=rdzv)
      comm_kind = "host-socket all-reduce (--host-comm: a rehearsal, not a scaling measurement)"
    else:
      nccl_count = nr = L.chm_comm_nranks(comm.handle)
      assert nr == world, (nr, world)
      comm_kind = f"RCCL all-reduce of 3*nbatch doubles inside chm_eval, ncclCommCount={nr}"
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
  # evaluation options of the handles (include/chimera_hip.h: CHM_OPT_*; the library reads no environment variable)
  if args.serial:
    like.set_option('serial', 1)
  if args.groups:
    like.set_option('groups', args.groups)
  if args.fused:
    like.set_option('fused', args.fused)
  if graph_max_nb is not None:
    like.set_option('graph_max_nb', graph_max_nb)
  nb = args.nbatch
  if nb <= 8 and graph_max_nb is not None:
    like.set_option('timing', 2)                      # few draws per call with the graph replay off (--no-graph, or 1 < nbatch <= 8: the timed calls stay eager): per-kernel HIP-event times of every step (the one-lane pass below is for calls of many draws)
  H0s = np.linspace(55., 95., 4099)          # a different H0 for every draw of every step
  Xi0s = np.linspace(0.6, 3.0, 4099)

  def lambdas(step, n=None):
    """The draws of one step as the sampler hands them over: the vectorised dict of hyper-parameter arrays (emcee_utils.py:54-64)."""
    n = nb if n is None else n
    j = step * n + np.arange(n)
    if mg:                                        # a different (Xi0, H0) for every draw
      return dict(Xi0=Xi0s[j % len(Xi0s)].copy(), H0=H0s[(7 * j) % len(H0s)].copy())
    return dict(H0=H0s[j % len(H0s)].copy())

  def sync():
    _lib.check(L.chm_device_synchronize(device))
    if rdzv is not None:
      rdzv.barrier()

  # the hyper-parameter draws of every step are the sampler's output, prepared before the timed region; packing them into
  # chm_params (hyperlikelihood._params_array) is part of the call and stays inside it
  draws = [lambdas(k) for k in range(args.warmup + args.steps)]
  vals = []
  # --inflight 2: a second lane on the same resident data (chm_like_clone / chm_sel_clone: own streams, tables, workspaces; with a
  # communicator its own RCCL communicator), one host thread per lane, the steps alternate between the lanes
  lanes, lane_comms, pool = [like], [], None

  def lane_comm(i):
    """The communicator of lane i >= 1.  RCCL: a communicator of its own (the job's rendezvous carries its id once).  Host sockets: a socket star of
    its OWN as well -- the lanes' host threads reduce concurrently, and two threads on one socket interleave their messages."""
    if not isinstance(comm, HostComm):
      return Comm(world, rank, device, rendezvous=rdzv)
    a = rdzv.address
    r2 = Rendezvous(world, rank, address=(a + f'.lane{i}') if isinstance(a, str) else (a[0], a[1] + 10 + i))
    hc = HostComm(world, rank, device, rendezvous=r2)
    hc._own = True                                          # (closed with the communicator)
    return hc

  if args.inflight > 1:
    from concurrent.futures import ThreadPoolExecutor
    for i in range(1, args.inflight):
      lc = None
      if comm is not None:
        lc = lane_comm(i)
        lane_comms.append(lc)
      lanes.append(like.lane(comm=lc))
    pool = ThreadPoolExecutor(max_workers=args.inflight)
  for w in range(args.warmup):
    for ln in (lanes if w < 2 else lanes[:1]):             # every lane allocates its workspaces outside the timed region
      vals.append(ln.batch(draws[w]))
  import gc
  gc.collect()
  gc.freeze()                                   # nothing in a step creates reference cycles: keep the cyclic collector off the clock
  sync()
  kt = np.zeros(8)
  step_s = []
  t1 = time.perf_counter()
  if pool is None:
    for k in range(args.steps):
      ta = time.perf_counter()
      vals.append(like.batch(draws[args.warmup + k]))          # synchronous: returns when the last kernel has stored the results (completion flags)
      step_s.append(time.perf_counter() - ta)
      if nb <= 8 and graph_max_nb is not None:
        kt += like.last_timing()                               # (the --no-graph run of few draws: per-kernel times of every step)
    # [r6] HIP-event times of the LAST step only: reading them waits for the call's final event (an interrupt-driven wait of tens of microseconds
    # that round 5 paid after every step of the timed region)
    if not (nb <= 8 and graph_max_nb is not None):
      kt += like.last_timing() * max(args.steps, 1)
  else:
    from collections import deque
    # [r4] step k carries ticket k on every rank: the lanes' all-reduces (one RCCL communicator per lane, one host thread per lane) are handed
    # to the device in step order everywhere (chm_comm_set_ticket; the order two host threads reach ncclAllReduce in is otherwise free)
    ticketed = comm is not None and hasattr(comm, 'set_ticket')
    if ticketed:
      Comm.reset_tickets(0)

    def run_step(ln, k):
      if ticketed:
        ln.comm.set_ticket(k)
      return ln.batch(draws[args.warmup + k])
    pending = deque()
    for k in range(args.steps):
      if len(pending) == args.inflight:
        vals.append(pending.popleft().result())
      pending.append(pool.submit(run_step, lanes[k % args.inflight], k))
    while pending:
      vals.append(pending.popleft().result())
    for ln in lanes:
      _lib.check(L.chm_device_synchronize(device))
  sync()
  dt = time.perf_counter() - t1
  # per-kernel durations for the roofline block: a few more steps with every event kernel on ONE lane (CHM_GROUPS=1: the library's default
  # for large shards overlaps the sample stage of one event group with the GW kernel of the previous one on two streams, where a kernel's
  # HIP-event span is not its duration) -- outside the timed region, same draws
  kt_timed_eval = kt[0] / max(args.steps, 1)
  # [r6] also for N > 1 (every rank runs the same collective calls on its shard): the N > 1 line then carries the roofline block of rank 0's shard
  if pool is None and nb > 8:                            # (calls of few draws are a single chain anyway)
    like.set_option('groups', 1)
    like.set_option('timing', 2)                         # per-kernel events (the default call carries the whole evaluation's two only)
    kt = np.zeros(8)
    ntot = max(4, min(args.steps, 24))               # as sustained as the timed region (the chip clocks higher in short bursts): the last half counts
    nser = 0
    for k in range(ntot):
      like.batch(draws[args.warmup + (k % max(args.steps, 1))])
      if k >= ntot // 2:
        kt += like.last_timing(); nser += 1
    like.set_option('groups', args.groups)
    like.set_option('timing', 1)
    kt *= max(args.steps, 1) / nser
    sync()
  dt_rank = dt
  multi_info = None
  if rdzv is not None:
    dt = float(rdzv.allreduce_max(np.array([dt]))[0])        # max over ranks
    dt_min = -float(rdzv.allreduce_max(np.array([-dt_rank]))[0])
    # the collective of one step on its own: 3 * nbatch doubles through the communicator the evaluations used (after the timed region)
    ar = []
    if comm is not None and hasattr(comm, 'allreduce_sum'):
      x = np.zeros(3 * nb)
      for j in range(25):
        rdzv.barrier()
        ta = time.perf_counter()
        comm.allreduce_sum(x)
        ar.append(1e6 * (time.perf_counter() - ta))
    # [r5] the PCI bus id of every rank's device: N distinct ids = N ranks on N GPUs (each rank fills its row of a (world, 32) table of character codes)
    buf = C_.create_string_buffer(64)
    _lib.check(L.chm_device_pci_bus_id(device, buf, 64))
    tab_ids = np.zeros((world, 32))
    code = np.frombuffer(buf.value[:32], dtype=np.uint8)
    tab_ids[rank, :len(code)] = code
    tab_ids = np.asarray(rdzv.allreduce_sum(tab_ids.ravel())).reshape(world, 32)
    bus_ids = [bytes(int(v) for v in row if v > 0).decode() for row in tab_ids]
    if not args.host_comm and len(set(bus_ids)) != world:
      raise SystemExit(f"bench.py: {world} ranks on {len(set(bus_ids))} distinct GPUs ({bus_ids}): no line is printed")
    multi_info = {"collective": "host sockets (--host-comm rehearsal)" if args.host_comm else "RCCL", "ncclCommCount": nccl_count, "world": world,
                  "pci_bus_ids": bus_ids, "distinct_gpus": len(set(bus_ids)),
                  "rank_ms_per_step": {"max": 1e3 * dt / max(args.steps, 1), "min": 1e3 * dt_min / max(args.steps, 1)},
                  "allreduce_us": {"median": float(np.median(ar[5:])), "calls": len(ar) - 5, "doubles": 3 * nb,
                                   "note": "host call to return, incl. H2D / D2H of the buffer: an upper bound on what the in-stream collective adds"} if len(ar) > 5 else None,
                  "inflight": args.inflight}
  kt /= max(args.steps, 1)

  single = None
  if not args.no_single_call:      # latency of the reference-shaped scalar call (one draw per call), outside the timed region
    for j in range(5):
      like(H0=float(H0s[-1 - j]))
    sync()
    ts = []
    for j in range(args.single_calls):
      ta = time.perf_counter()
      like(H0=float(H0s[-10 - j]))
      ts.append(1e3 * (time.perf_counter() - ta))
    sync()
    med, q1, q3 = quartiles(ts)
    b_alg = algorithmic_bytes(E, S, P, Z, I, 200, pixelated, kind == 'full')
    single = {"median_ms": med, "q25_ms": q1, "q75_ms": q3, "calls": len(ts), "evals_per_s": 1e3 / med,
              # [r4] the whole call against the HBM roofline: SURVEY 8(d)'s algorithmic bytes of one evaluation over the call's wall time
              "algorithmic_bytes": b_alg, "hbm_GBs": b_alg / (med * 1e-3) / 1e9, "hbm_frac": b_alg / (med * 1e-3) / 1e9 / HBM_PEAK_GBS}

  # [r6] N > 1 (or --force-comm): a second leg with TWO evaluations in flight per rank, after the timed region -- a second lane on the same resident
  # shard with a communicator of its own, one host thread per lane, the steps alternate between the lanes and carry tickets (chm_comm_set_ticket).
  # The tables, launch path and reduction tail of one call then run under the kernels of the other: what is left of the non-scaling part of a
  # 125-event shard's step.  `value` stays the inflight-1 figure; the leg is reported as multi_gpu.inflight2.  It runs in a thread of its own under a
  # deadline: a lane that hangs (a collective that never completes) costs the leg, not the line -- the process then prints and leaves without the
  # collectives of an orderly shutdown.
  hard_exit = False
  if rdzv is not None and comm is not None and args.inflight == 1 and not args.no_inflight2 and multi_info is not None:
    import threading
    leg = {}

    def inflight2_leg():
      from collections import deque
      from concurrent.futures import ThreadPoolExecutor
      lc, err = None, None
      try:
        lc = lane_comm(args.inflight)
      except Exception as e:                                  # noqa: BLE001
        err = e
      if int(rdzv.allreduce_sum(np.array([0. if lc is not None else 1.]))[0]) > 0:        # every rank takes the same branch
        leg['error'] = f"second communicator unavailable ({err})"
        return
      lane2 = like.lane(comm=lc)
      lanes2 = [like, lane2]
      ticketed = hasattr(comm, 'set_ticket')
      if ticketed:
        Comm.ticket_timeout(20.)
        Comm.reset_tickets(0)
      tk = [0]

      def run_step(ln, k, ticket):
        if ticketed:
          ln.comm.set_ticket(ticket)
        return ln.batch(draws[args.warmup + (k % max(args.steps, 1))])
      try:
        for w in range(2):                                    # both lanes allocate their workspaces before the clock starts
          for ln in lanes2:
            run_step(ln, w, tk[0]); tk[0] += 1
        sync()
        with ThreadPoolExecutor(max_workers=2) as pool2:
          pending = deque()
          ta = time.perf_counter()
          for k in range(args.steps):
            if len(pending) == 2:
              pending.popleft().result()
            pending.append(pool2.submit(run_step, lanes2[k % 2], k, tk[0])); tk[0] += 1
          last = None
          while pending:
            last = pending.popleft().result()
          sync()
          dt2 = time.perf_counter() - ta
        dt2 = float(rdzv.allreduce_max(np.array([dt2]))[0])
        leg.update({"ms_per_step": 1e3 * dt2 / max(args.steps, 1), "value": args.steps * nb / dt2, "steps": args.steps, "lanes": 2,
                    "last_log_hyper": float(np.asarray(last).ravel()[-1]) if last is not None else None,
                    "note": "two evaluations in flight per rank (two lanes, two communicators, ticketed collectives), measured after the timed region; "
                            "`value` of the line is the inflight-1 figure"})
      finally:
        leg['_cleanup'] = (lane2, lc)

    th = threading.Thread(target=inflight2_leg, daemon=True)
    th.start()
    th.join(90.)
    if th.is_alive():
      leg = {"error": "the leg did not finish within 90 s (a lane or a collective hung); the line is printed without it"}
      hard_exit = True
    elif 'ms_per_step' not in leg:
      leg.setdefault('error', 'the leg raised: see stderr')
      hard_exit = 'second communicator unavailable' not in leg['error']
    cl = leg.pop('_cleanup', None)
    if cl is not None and not hard_exit:
      ln2, lc2 = cl
      if ln2.selection_function is not None:
        ln2.selection_function.close()
      ln2.close()
      lane_comms.append(lc2)
    multi_info["inflight2"] = leg

  if rank == 0:
    evals = args.steps * nb
    value = evals / dt
    El = like._e1 - like._e0
    lib_sha = code_object_sha256(_lib.LIB_PATH)
    pmc = load_pmc(dict(config=args.config, E=El, P=P, Z=Z, S=S, nbatch=nb, mode=kind or '1d', n_gpus=1, fused=args.fused), lib_sha)
    pmc_scale = None
    if pmc is None and world > 1:
      # [r6] a shard of an N-GPU run: the PMC passes of the WHOLE workload on one GPU, per-launch counters scaled by the shard's share of the events (the
      # counts per pair of pixels / per sample do not depend on how many events a launch holds); labelled in the line
      pmc = load_pmc(dict(config=args.config, E=E, P=P, Z=Z, S=S, nbatch=nb, mode=kind or '1d', n_gpus=1, fused=args.fused), lib_sha)
      pmc_scale = El / float(E) if pmc is not None else None
    probe = load_probe_ceilings(lib_sha)
    kernels = []
    if kind == 'marginalized' and args.fused >= 2:
      # the fused event kernel does the sample stage, the per-z factors and the GW kernel of every (event, draw) in one launch: its span is kt[3]
      kernels.append(kernel_roofline("fused event kernel (samples + statistics + histograms + per-z factors + KDE + integrand)", "k_marg_fused", kt[3],
                                     El * S * 49 + El * P * Z * 8 + El * Z * 8 + nb * ((2 * 1500 + 2 * 1000) * 8 + El * Z * 16), pmc))
    elif kind == 'marginalized':
      kernels.append(kernel_roofline("marginalized GW kernel (histogram + KDE + interp + integrand + trapz)", "k_kde_marg_sub2", kt[3],
                                     gw_kernel_unique_bytes(El, S, P, Z, nb), pmc, units=El * P / 2. * nb, probe=probe, work=El * ((P + 1) // 2) * nb, scale=pmc_scale))
    full_pairs = None
    if kind == 'full':
      kernels.append(kernel_roofline("3-D Gaussian KDE + integrand (sample-stationary kernel; the general kernel's share of the stage is its empty blocks)", "k_full_kde_chain", kt[3], El * S * 32 * nb + El * P * Z * 8, pmc))
      # sample x query pairs of one step (E npix Z_eff S): the masked stretch of every event grid, [min z - c std, max z + c std]
      # (likelihood.py:222-225), from the source-frame z of the last step's draws (z_from_dGW on the device, outside the timed region)
      full_pairs = 0
      for jd in range(nb):
        lam = {k_: float(v_[jd]) for k_, v_ in draws[-1].items()}
        zz = np.asarray(CH.cosmo.z_from_dGW(cosmo.update(**lam), ev['dL'][like._e0:like._e1]))
        zlo = zz.min(axis=1) - 2. * zz.std(axis=1); zhi = zz.max(axis=1) + 2. * zz.std(axis=1)
        zg = ev['z_grids'][like._e0:like._e1]
        nmask = np.sum((zg <= zhi[:, None]) & (zg >= zlo[:, None]), axis=1)
        full_pairs += int(np.sum(nmask * np.asarray(ev['neff_pixels'][like._e0:like._e1]))) * S
      kf = kernels[-1]
      sec = kt[3] * 1e-3
      # the power-sum march costs 5 fp64 instructions (4 cycles each) per 4 pairs and lane (4 fma + 1 multiply): peak = 614.4 G wave-inst/s x 64 lanes / 1.25
      pk = N_SIMD * CLK_HZ / CYC_VALU / 1e9 * 64 / 1.25
      # [r5] what the card sustains on the march ALONE (scripts/march_probe.hip, profiles/r05/march_probe.txt: 20.1-20.4 Tpair/s with one, two or three
      # waves per SIMD, the clock down to 1.9 GHz under the fp64 FMAs) -- the reachable fraction of `peak_Gpairs_s` is 0.65
      march_alone = 20420.
      kf.update({"pairs_per_launch": full_pairs, "Gpairs_s": full_pairs / sec / 1e9 if sec > 0 else None,
                 "peak_Gpairs_s": pk, "pair_frac": full_pairs / sec / 1e9 / pk if sec > 0 else None,
                 "march_alone_Gpairs_s_measured_r05": march_alone, "frac_of_march_alone": full_pairs / sec / 1e9 / march_alone if sec > 0 else None})
    if not (kind == 'marginalized' and args.fused >= 2):        # (the fused event kernel has no sample stage of its own)
      kernels.append(kernel_roofline("sample stage (z(dL), source-frame masses, population weights, event statistics)", "k_samples", kt[2],
                                     sample_kernel_unique_bytes(El, S, nb), pmc, units=El * S * nb / 64., probe=probe, work=float(El) * S * nb, scale=pmc_scale))
    # the selection kernel runs on its own stream beside the event kernels (its span there is not a kernel duration): timed standalone
    # here, after the timed region, as the selection-only call chm_eval(NULL, sel, ...) of the same draws
    sel_ms = None
    if world == 1:
      import ctypes as C
      pa = like._params_array(draws[-1])
      nexp = np.empty(nb)
      o = _lib.chm_out(); o.N_exp = _lib.dptr(nexp)
      ms = np.zeros(8)
      acc = []
      sel.set_option('timing', 2)
      for j in range(6):
        _lib.check(L.chm_eval(None, sel._handle(), None, pa, nb, 0, C.byref(o)))
        _lib.check(L.chm_last_timing(None, sel._handle(), _lib.dptr(ms)))
        if j >= 2 and ms[4] > 0:
          acc.append(ms[4])
      sel.set_option('timing', 1)
      if acc:
        sel_ms = float(np.median(acc))
        kernels.append(kernel_roofline("selection function (dN/dtheta per injection, two sums); standalone selection-only call", "k_selection", sel_ms,
                                       I * 48 + nb * (2 * 1500 + 2 * 1000) * 8, pmc, units=I * nb / 64.))
    dom = max(kernels, key=lambda k_: k_["kernel_ms"] or 0.) if kind != 'full' else kernels[0]
    path_bytes = algorithmic_bytes(E, S, P, Z, I, 200, pixelated, kind == 'full')
    med, q1, q3 = quartiles(step_s) if step_s else (None, None, None)
    roof = {"bound": "valu-issue", "kernel": dom["kernel"],
            "achieved": dom.get("useful_Tcycle_s") if kind != 'full' else dom.get("Gpairs_s"),
            "peak": ISSUE_PEAK_TCYC if kind != 'full' else dom.get("peak_Gpairs_s"),
            "unit": "Tcycle/s (issue cycles of fp64 add / mul / fma instructions; peak: the VALU issue cycles of 1024 SIMDs at 2.4 GHz)" if kind != 'full' else "Gpair/s",
            "frac": dom.get("useful_frac") if kind != 'full' else dom.get("pair_frac"),
            # [r6] the dominant kernel against the MEASURED ceiling of its own body (cache-resident probe): replaces round 5's paper count as the yardstick
            "frac_of_sustained": (dom.get("sustained") or {}).get("frac_of_sustained"),
            "useful_frac": dom.get("useful_frac"), "issue_busy_frac": dom.get("valu_busy_frac"),
            "issue_busy_frac_at_held_clock": dom.get("valu_busy_frac_at_held_clock"), "min_inst": dom.get("min_inst"),
            "fp64_TFLOPs_real": dom.get("fp64_TFLOPs_real"), "fp64_peak_TFLOPs": FP64_PEAK_TFLOPS,
            "traffic": dom.get("traffic_bytes_per_launch"),
            "traffic_source": (pmc[0] + (" (separate rocprofv3 --pmc passes of this command)" if not pmc_scale else
                                         f" (PMC passes of the WHOLE workload on one GPU; per-launch counters scaled by this rank's share of the events, {pmc_scale:.4f})")) if pmc else None,
            "code_object_sha256": lib_sha, "pmc_matches_loaded_code_object": bool(pmc and pmc[2]),
            "kernel_ms": dom["kernel_ms"],
            "hbm": {"unique_bytes_per_launch": dom["unique_bytes_per_launch"], "achieved": dom["hbm_unique_GBs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": dom["hbm_unique_frac"], "traffic_frac": dom.get("hbm_traffic_frac")},
            "kernels": kernels,
            # SURVEY 8(d)'s call-level figure: the algorithmic bytes of ONE evaluation over the scalar call's wall time against 8 TB/s
            "hbm_call_frac": single["hbm_frac"] if single else None,
            "note": "both dominant kernels are bound by VALU issue.  frac = useful_frac = issue cycles of the launch's fp64 add / mul / fma instructions "
                    "(PMC class counters of the committed passes of this command x 4 cycles, profiles/r03/issue_cost.txt) / (1024 SIMDs x 2.4 GHz x the "
                    "launch's LIVE HIP-event duration): what the kernel achieves.  issue_busy_frac prices EVERY VALU instruction at its measured issue "
                    "cost (4 cycles; a few simple 32-bit opcodes 2, fp64 rcp/sqrt 16) -- how full the issue ports are, moves and selects included; "
                    "..._at_held_clock uses the clock under the profile.  min_inst: instructions per unit of work against a PAPER ESTIMATE of the minimum (a diagnostic, not a measurement).  "
                    "fp64_TFLOPs_real counts FMA = 2, add / mul = 1 (PMC).  No fraction is printed when the PMC file was collected from another code "
                    "object than the one loaded.  hbm.* is the same launch against 8 TB/s: unique bytes (shared inputs once, per-draw arrays x nbatch) "
                    "and PMC fabric traffic; the call-level HBM fraction of the scalar call is single_call.hbm_frac",
            "path_bytes_per_eval": path_bytes,
            "stage_ms": {"note": "per-kernel times from steps with the event kernels on one lane (CHM_GROUPS=1) after the timed region; "
                                 "eval_timed = HIP-event time of a step inside the timed region (event groups on two lanes)",
                         "eval_timed": kt_timed_eval, "eval": kt[0], "tables": kt[1], "samples": kt[2], "kde_integrate": kt[3],
                         "selection": kt[4], "selection_standalone": sel_ms, "reduce": kt[5], "events_wall": kt[6], "event_groups": kt[7]}}
    out = {
      "metric": "log-likelihood evals/sec (full hyperposterior call), N_ev x N_pix x N_z",
      "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
      "dtype": "f64", "data": "synthetic (seed 20250926; chimera_amd/synth.py)",
      "config": {"workload": f"{args.config}: {E} events x {P} pixels x {Z} z-bins, {S} samples/event, {I} detected injections, "
                             f"PLP + Madau-Dickinson + {'modified-GW-propagation (Xi0, n) flat-LCDM' if mg else 'flat-LCDM'}, {kind or '1d'}, binning 200, cut_grid 2",
                 "E": E, "P": P, "Z": Z, "S": S, "I": I, "kind_p_gw3d": kind, "nbatch": nb, "inflight": args.inflight, "fused": args.fused, "groups": args.groups, "serial": bool(args.serial),
                 "parallelism": f"events+injections sharded over {world} GPU(s)" + (f"; {comm_kind}" if comm_kind else ""),
                 "cells_per_s": value * E * max(P, 1) * Z},
      "step_ms": {"median": 1e3 * med, "q25": 1e3 * q1, "q75": 1e3 * q3, "n": len(step_s)} if step_s else None,
      "multi_gpu": multi_info,
      "single_call_ms": single["median_ms"] if single else None,
      "single_call": single,
      "roofline": roof,
      "setup_s": {"synthetic": t_gen, "upload_once": t_upload,
                  "note": "upload_once = chm_like_create + chm_sel_create (host pixel sort, log(m_det), H2D of the shard); every "
                          "evaluation afterwards moves ~350 B of parameters per draw host->device and 24 B back"},
      "last_log_hyper": float(np.asarray(vals[-1]).ravel()[-1]),
    }
    if world == 1 and not args.no_extra and args.config == 'C3' and kind == 'marginalized' and args.events is None and args.inj is None and nb == 128 and args.inflight == 1 \
       and not (args.serial or args.groups or args.fused or args.no_graph):
      # [r6] the other BASELINE configurations and call modes in the SAME driver-run line (everything but C3 / marginalized / 128 draws was
      # builder-run evidence): short legs after the timed region, inputs synthesised at full size (C5's 4.2 GB synthesis is left to the GPU test)
      t_x = time.time()
      out["extra"] = {"configs": extra_legs(CH, synth, _lib, L, device, cfg, ev, inj, like, H0s), "note":
                      "3 warm-up + 10 timed steps each (hyperlikelihood.batch of nbatch draws, different H0 per draw), then 20 scalar calls; "
                      "one_draw_kernels: HIP-event times of the C3 kernels at ONE draw per call (graph replay off) against the HBM roofline"}
      out["extra"]["wall_s"] = time.time() - t_x
    if world == 1 and not args.no_cpu_baseline:
      out["cpu_baseline"] = cb = cpu_baseline(cfg, ev, inj, kind, args.cpu_events, n_evals=args.cpu_evals)
      if cb.get("value"):
        out["vs_cpu_baseline"] = value / cb["value"]
      if cb.get("log_hyper_H0_67") is not None:      # full-size parity check of the timed path against the CPU port
        g = float(like(H0=67.))
        out["parity_full_size"] = {"H0": 67., "log_hyper_hip": g, "log_hyper_cpu_port": cb["log_hyper_H0_67"],
                                   "abs_diff": abs(g - cb["log_hyper_H0_67"]), "tolerance": 1e-7 * float(np.sqrt(E))}
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(out) + '\n').encode())
  if hard_exit:                                     # the inflight-2 leg hung or failed on this rank: no collective of an orderly shutdown can be trusted
    sys.stderr.flush()
    os._exit(0)
  try:
    if pool is not None:
      pool.shutdown()
    for ln in lanes[1:]:
      if ln.selection_function is not None:
        ln.selection_function.close()
      ln.close()
    like.close()
    sel.close()
    if rdzv is not None:
      rdzv.barrier()
    for lc in lane_comms:
      lc.close()
    if comm is not None:
      comm.close()
    if rdzv is not None:
      rdzv.close()
  except Exception as e:                              # noqa: BLE001
    if multi_info is not None and 'inflight2' in multi_info:
      # a peer left without the shutdown collectives (its inflight-2 leg hung): the line is out, nothing is left to measure
      print(f"[bench] rank {rank}: shutdown after the inflight-2 leg: {e}", file=sys.stderr)
      sys.stderr.flush()
      os._exit(0)
    raise


def extra_legs(CH, synth, _lib, L, device, cfg3, ev3, inj3, like3, H0s, steps=10, warmup=3, single_calls=20):
  """Short legs of the other BASELINE configurations / modes on one GPU (after the timed region of the headline).  Every leg: evals/s and ms per step
  of `hyperlikelihood.batch` at its draw count, and the median scalar call."""
  from chimera_amd.catalog import dVdz_completeness, pixelated_catalog
  pe_fields = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside')

  def build(cfg, ev, inj, kind):
    if cfg['pixelated']:
      th = CH.data.theta_pe_det(**{k: ev[k] for k in pe_fields})
      gal_cat = pixelated_catalog(dVdz_completeness(z_range=[0.073, 1.3]), p_cat=ev['p_cat'], z_grids=ev['z_grids'], neff_pixels=ev['neff_pixels'])
    else:
      th = CH.data.theta_pe_det(**{k: ev[k] for k in ('m1det', 'm2det', 'dL', 'pe_prior')})
      gal_cat, kind = None, None
    pop = CH.population(CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.), CH.mass.plp(), CH.rate.madau_dickinson(gamma=2.7, kappa=3., zp=2.), gal_cat=gal_cat, scale_free=True)
    sel = CH.selection_function(CH.data.theta_inj_det(**{k: inj[k] for k in ('m1det', 'm2det', 'dL', 'p_draw')}), N_inj=inj['N_inj'], N_eff=5.)
    like = CH.hyperlikelihood(th, ev['z_grids'], pop, sel, kind_p_gw3d=kind, kernel='epan', bw_method=None, cut_grid=2, binning=True, num_bins=200)
    return like, sel

  def leg(name, cfg, ev, inj, kind, nb_leg):
    like, sel = build(cfg, ev, inj, kind)
    try:
      draws = [dict(H0=H0s[(k * nb_leg + np.arange(nb_leg)) % len(H0s)].copy()) for k in range(warmup + steps)]
      for k in range(warmup):
        like.batch(draws[k])
      _lib.check(L.chm_device_synchronize(device))
      ta = time.perf_counter()
      for k in range(steps):
        v = like.batch(draws[warmup + k])
      _lib.check(L.chm_device_synchronize(device))
      dt = time.perf_counter() - ta
      for j in range(4):
        like(H0=float(H0s[-1 - j]))
      ts = []
      for j in range(single_calls):
        tb = time.perf_counter()
        like(H0=float(H0s[-10 - j]))
        ts.append(1e3 * (time.perf_counter() - tb))
      return {"workload": f"{name}: {cfg['E']} events x {cfg['P']} pixels x {cfg['Z']} z-bins, {cfg['S']} samples/event, {cfg['I']} injections, {kind or '1d'}",
              "nbatch": nb_leg, "steps": steps, "evals_per_s": steps * nb_leg / dt, "ms_per_step": 1e3 * dt / steps,
              "single_call_ms": float(np.median(ts)), "last_log_hyper": float(np.asarray(v).ravel()[-1])}
    finally:
      like.close(); sel.close()

  res = {}
  for name in ('C1', 'C2', 'C4'):
    t0 = time.time()
    cfg, ev, inj = synth.make_config(name)
    t_gen = time.time() - t0
    res[name] = leg(name, cfg, ev, inj, 'marginalized' if cfg['pixelated'] else None, 128)
    res[name]["synthetic_s"] = t_gen
    del ev, inj
  res['C3_approximate'] = leg('C3', cfg3, ev3, inj3, 'approximate', 128)
  res['C3_full'] = leg('C3', cfg3, ev3, inj3, 'full', 4)
  # the C3 kernels at ONE draw per call: HBM is the applicable bound there (graph replay off, so that the call carries its timing events)
  E, S, P, Z, I = cfg3['E'], cfg3['S'], cfg3['P'], cfg3['Z'], cfg3['I']
  like3.set_option('graph_max_nb', 0)
  like3.set_option('timing', 2)
  try:
    kt = np.zeros(8); n = 0
    for k in range(14):
      like3.batch(dict(H0=H0s[k:k + 1].copy()))
      if k >= 4:
        kt += like3.last_timing(); n += 1
    kt /= n
    ub_s, ub_g = sample_kernel_unique_bytes(E, S, 1), gw_kernel_unique_bytes(E, S, P, Z, 1)
    res['one_draw_kernels'] = {"samples_us": 1e3 * kt[2], "gw_kernel_us": 1e3 * kt[3], "tables_us": 1e3 * kt[1], "eval_us": 1e3 * kt[0],
                               "samples_unique_bytes": ub_s, "gw_unique_bytes": ub_g,
                               "samples_hbm_frac_unique": ub_s / (kt[2] * 1e-3) / 1e9 / HBM_PEAK_GBS if kt[2] > 0 else None,
                               "gw_hbm_frac_unique": ub_g / (kt[3] * 1e-3) / 1e9 / HBM_PEAK_GBS if kt[3] > 0 else None,
                               "note": "eager one-draw calls; gw_kernel_us spans the GW kernel + fix-up (HIP events around both)"}
  finally:
    like3.set_option('timing', 1)
    like3.set_option('graph_max_nb', 8)
  return res


def cpu_baseline(cfg, ev, inj, kind, n_ev, threads=None, numpy_events=48, n_evals=20):
  """CPU baselines on this host, same workload, same algorithm as the reference (dense G x B kernel sums):
  * value: the plain-C / OpenMP restatement (oracle/chimera_oracle_c.c) on `threads` cores (the GPU box's CPU share is 16),
    n_ev events (default: the whole workload) + all injections, `n_evals` evaluations with different H0 (tables rebuilt each
    time), median and quartiles of the per-evaluation time;
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
  n_np, like, pop, sel = build(numpy_events if kind != 'full' else 2)
  popu = pop.update(H0=67.)
  t0 = time.perf_counter(); like.compute_log_likenum(popu); t_ev = time.perf_counter() - t0
  t0 = time.perf_counter(); sel.N_exp(popu); t_sel = time.perf_counter() - t0
  np1 = {"value": 1.0 / (t_ev * E / n_np + t_sel), "cores": 1,
         "sample": f"oracle/chimera_oracle.py, first {n_np} of {E} events ({t_ev:.2f} s) + all {cfg['I']} injections ({t_sel:.3f} s), "
                   f"1 evaluation, event time scaled x{E / n_np:.1f}"}
  # C / OpenMP, `threads` cores
  from oracle import oracle_c as OC
  threads = threads or min(16, os.cpu_count() or 1)
  if kind == 'full':                               # ~1e8 exp per event: a bounded sample of the events, scaled
    n_ev, n_evals = min(n_ev, 48), min(n_evals, 5)
  n_c, like, pop, sel = build(n_ev)
  H0s = np.linspace(61., 79., max(n_evals, 1))      # different H0 per evaluation: the tables are rebuilt each time
  H0s[0] = 67.
  OC.compute_all(like, dict(H0=70.), nthreads=threads) if n_c <= 64 else None      # warm the library on small runs only
  ts, vals = [], []
  for h in H0s:
    t0 = time.perf_counter()
    vals.append(OC.compute_all(like, dict(H0=float(h)), nthreads=threads)[3])
    ts.append(time.perf_counter() - t0)
  scale = (E / n_c) if n_c < E else 1.
  med, q1, q3 = quartiles(ts)
  out = {"value": 1.0 / (med * scale), "unit": "evals/s", "cores": threads, "kind": "port",
         "eval_s": {"median": med * scale, "q25": q1 * scale, "q75": q3 * scale, "n": len(ts)},
         "sample": f"oracle/chimera_oracle_c.c (C + OpenMP, {threads} threads), {n_c} of {E} events + all {cfg['I']} injections, "
                   f"{len(ts)} evaluations at different H0 (median {med:.3f} s each{'' if n_c == E else f', scaled x{scale:.1f}'}; "
                   f"{sum(ts) * threads:.0f} core-seconds in all)",
         "host_cpus": os.cpu_count(), "numpy_1core": np1, "log_hyper_H0_67": float(vals[0]) if n_c == E else None}
  return out


if __name__ == '__main__':
  main()
