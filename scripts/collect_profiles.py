#!/usr/bin/env python3
"""Collect a round's profiles on the GPU box (run through gpurun):

  python3 scripts/collect_profiles.py r02 [--tag C3] [--skip-trace] [-- <bench.py arguments>]

Writes under gpurun_out/profiles_<round>/ (copy what is to be judged into profiles/<round>/):
  kernel_stats<tag>.csv          rocprofv3 --kernel-trace --stats summary of `python3 bench.py <args>` (default streams)
  kernel_stats_serial<tag>.csv   the same with `--serial --groups 1` (every kernel on one stream: standalone durations)
  pmc_per_launch<tag>.json       per kernel, per launch: PMC counters from SEPARATE rocprofv3 --pmc passes (counters only, no
                                 tracing flags besides --kernel-trace), the kernel's average duration in those passes, and the
                                 workload keys bench.py matches against
  bench<tag>.json                the bench line of the same command (run last, so that its roofline block reads the fresh PMC file)
This process never touches the GPU; every profiled program is started as `rocprofv3 ... -- python3 bench.py ...`.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
  "FETCH_SIZE GRBM_GUI_ACTIVE",
  "WRITE_SIZE",
  "TCC_HIT_sum TCC_MISS_sum",
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES",
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES",
  # [r3] the VALU stream by class: fp64 add / mul / fma / transcendental, 32- and 64-bit integer, conversions (bench.py prices fp64 at 4
  # and everything else at 2 cycles per wave64 instruction; scripts/issue_cost.hip measures those costs on the card)
  "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT",
  "SQ_INSTS_VALU_FLOPS_FP64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM",
]


def run(cmd, log, env=None, timeout=900):
  with open(log, 'w') as f:
    return subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, env=env, timeout=timeout, cwd=ROOT)


def short(name):
  return name.split('(')[0].replace('void ', '').strip()


def main():
  argv = sys.argv[1:]
  bench_args = []
  if '--' in argv:
    i = argv.index('--')
    argv, bench_args = argv[:i], argv[i + 1:]
  rnd = argv[0] if argv and not argv[0].startswith('-') else 'r02'
  tag = ''
  if '--tag' in argv:
    tag = '_' + argv[argv.index('--tag') + 1]
  skip_trace = '--skip-trace' in argv
  only = [int(x) for x in argv[argv.index('--passes') + 1].split(',')] if '--passes' in argv else None
  out = os.path.join(ROOT, 'gpurun_out', f'profiles_{rnd}')
  os.makedirs(out, exist_ok=True)
  os.environ['TMPDIR'] = '/tmp'
  env = dict(os.environ)
  base = ['python3', 'bench.py'] + bench_args
  quick = base + ['--no-cpu-baseline', '--no-single-call', '--no-extra']

  if not skip_trace:
    for serial in (False, True):
      d = os.path.join(out, 'trace_serial' if serial else 'trace')
      rc = run(['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '--'] + quick + (['--serial', '--groups', '1'] if serial else []),
               os.path.join(out, f"trace{'_serial' if serial else ''}{tag}.log"), env=env)
      print('kernel-trace', 'serial' if serial else 'default', 'rc', rc, flush=True)
      for f in glob.glob(os.path.join(d, '*', '*kernel_stats.csv')):
        shutil.copy(f, os.path.join(out, f"kernel_stats{'_serial' if serial else ''}{tag}.csv"))
      shutil.rmtree(d, ignore_errors=True)

  kernels = collections.defaultdict(dict)
  for pi, counters in enumerate(PASSES):
    if only is not None and pi not in only:
      continue
    d = os.path.join(out, f'pmc{pi}')
    # --groups 1: one launch of every kernel per step covers the whole workload (the default splits large shards into event groups on two
    # streams) -- the launch the roofline block of bench.py times after its timed region
    rc = run(['rocprofv3', '--kernel-trace', '--pmc'] + counters.split() + ['--output-format', 'csv', '-d', d, '--'] + quick +
             ['--steps', '3', '--warmup', '1', '--groups', '1'], os.path.join(out, f'pmc{pi}{tag}.log'), env=env)
    print('pmc pass', pi, counters, 'rc', rc, flush=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
      for r in csv.DictReader(open(f)):
        agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, '*', '*kernel_trace.csv')):
      for r in csv.DictReader(open(f)):
        dur[short(r['Kernel_Name'])].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6)
    for k, v in agg.items():
      kernels[k].update({c: sum(x) / len(x) for c, x in v.items()})
      kernels[k].setdefault('launches_per_pass', len(next(iter(v.values()))))
      if k in dur and 'GRBM_GUI_ACTIVE' in v:
        kernels[k]['profiled_ms'] = sum(dur[k]) / len(dur[k])            # duration in the pass that carried GRBM_GUI_ACTIVE
    shutil.rmtree(d, ignore_errors=True)

  # workload keys from a plain bench line of the same arguments (also the bench line kept with the profiles); the PMC file goes
  # to profiles/<round>/ of THIS copy first so that the line's roofline block is computed from it
  prof_dir = os.path.join(ROOT, 'profiles', rnd)
  os.makedirs(prof_dir, exist_ok=True)
  probe = subprocess.run(base + ['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-single-call', '--no-extra'], cwd=ROOT, capture_output=True, text=True,
                         timeout=900)
  line = [l for l in probe.stdout.strip().split('\n') if l.startswith('{')]
  cfg = json.loads(line[-1])['config'] if line else {}
  wl = {"config": (bench_args[bench_args.index('--config') + 1] if '--config' in bench_args else 'C3'), "E": cfg.get('E'), "P": cfg.get('P'),
        "Z": cfg.get('Z'), "S": cfg.get('S'), "nbatch": cfg.get('nbatch'), "mode": cfg.get('kind_p_gw3d') or '1d', "n_gpus": 1, "fused": cfg.get('fused', 0)}
  # [r3] tie the counters to the binary they were collected from, and keep the static instruction mix of its hot loops beside them
  sys.path.insert(0, os.path.join(ROOT, 'scripts'))
  import isa_mix
  lib = os.environ.get('CHIMERA_LIB') or os.path.join(ROOT, 'chimera_amd', 'lib', 'libchimera_hip.so')
  static, _, _ = isa_mix.analyse(lib, [short(k) for k in kernels], with_loops=True)
  for k, e in static['kernels'].items():                      # the three largest loops are enough to read the hot loops off
    e['loops'] = e.get('loops', [])[:3]
  if only is not None:                                        # a partial collection: merge into what an earlier call of this round wrote for the SAME binary
    try:
      old = json.load(open(os.path.join(prof_dir, f'pmc_per_launch{tag}.json')))
      if old.get('code_object_sha256') == static['code_object_sha256']:
        for k, v in old.get('kernels', {}).items():
          kernels[k] = dict(v, **kernels.get(k, {}))
    except Exception:                                         # noqa: BLE001
      pass
  pmc = {"workload": wl, "command": ' '.join(base), "passes": PASSES, "code_object_sha256": static['code_object_sha256'],
         "issue_cost_model": static['issue_cost_model'], "static_mix": static['kernels'],
         "note": "per kernel, averages per launch over the launches of a pass; FETCH_SIZE / WRITE_SIZE in KiB (rocprofv3); SQ_* in the "
                 "counters' own units (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles); every pass is its own run",
         "kernels": kernels}
  for dst in (os.path.join(out, f'pmc_per_launch{tag}.json'), os.path.join(prof_dir, f'pmc_per_launch{tag}.json')):
    with open(dst, 'w') as f:
      json.dump(pmc, f, indent=1, sort_keys=True)
  with open(os.path.join(out, f'bench{tag}.json'), 'w') as fo, open(os.path.join(out, f'bench{tag}.err'), 'w') as fe:
    rc = subprocess.call(base, stdout=fo, stderr=fe, cwd=ROOT, timeout=1200)
  print('bench rc', rc)
  try:
    b = json.loads(open(os.path.join(out, f'bench{tag}.json')).read().strip().split('\n')[-1])
    print('value', b['value'], 'ms/step', b['ms_per_step'], 'single', b.get('single_call_ms'))
    for k in b['roofline']['kernels']:
      print({x: k.get(x) for x in ('kernel', 'kernel_ms', 'useful_frac', 'valu_busy_frac', 'valu_busy_frac_at_held_clock', 'fp64_TFLOPs_real', 'cycles_per_valu_inst', 'hbm_unique_frac', 'hbm_traffic_frac', 'clock_GHz_under_profile')})
  except Exception as e:                                             # noqa: BLE001
    print('bench line not parsed:', e)
  for k, v in kernels.items():
    if 'SQ_INSTS_VALU' in v:
      print(k[:50], {c: round(v[c], 1) for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVES') if c in v})


if __name__ == '__main__':
  main()
