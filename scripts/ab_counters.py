#!/usr/bin/env python3
"""Same-box A/B of library builds WITH counters (run through gpurun):

  python3 scripts/ab_counters.py OUT.txt base pack exp32 ... [-- <bench.py arguments>]

For every build (base = the release library, NAME = chimera_amd/lib/variants/libchimera_hip_NAME.so, selected with CHIMERA_LIB): two plain bench runs
(ms per step, per-kernel HIP-event times of the one-lane pass), one rocprofv3 --kernel-trace --stats run with every kernel on one stream (average
kernel durations) and two separate --pmc passes (LDS / wait counters; SQ instruction counters) of 3 steps with one event group.  Prints one table row
per (build, hot kernel).  This process never touches the GPU."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = ["SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE",
          "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"]
HOT = ('k_kde_marg_sub2', 'k_samples_fast', 'k_selection_fast', 'k_zfactors', 'k_tables')


def short(name):
  return name.split('(')[0].replace('void ', '').strip()


def main():
  argv = sys.argv[1:]
  bench_args = []
  if '--' in argv:
    i = argv.index('--')
    argv, bench_args = argv[:i], argv[i + 1:]
  out_txt, libs = argv[0], argv[1:]
  os.environ['TMPDIR'] = '/tmp'
  tmp = os.path.join(ROOT, 'gpurun_out', 'abc_tmp')
  rows = []
  quick = ['python3', 'bench.py', '--no-cpu-baseline', '--no-single-call', '--no-extra'] + bench_args
  plain = collections.defaultdict(list)
  for rep in range(2):
    for lib in libs:
      env = dict(os.environ, CHIMERA_NO_REBUILD='1')
      if lib != 'base':
        env['CHIMERA_LIB'] = os.path.join(ROOT, 'chimera_amd', 'lib', 'variants', f'libchimera_hip_{lib}.so')
      p = subprocess.run(quick + ['--steps', '30', '--warmup', '3'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
      try:
        j = json.loads(p.stdout.strip().split('\n')[-1])
        s = j['roofline']['stage_ms']
        plain[lib].append((j['ms_per_step'], s['samples'], s['kde_integrate'], j['last_log_hyper']))
        print(f"{lib:12s} rep{rep + 1} ms_per_step={j['ms_per_step']:.4f} samples={s['samples']:.4f} kde={s['kde_integrate']:.4f} last={j['last_log_hyper']!r}", flush=True)
      except Exception as e:                          # noqa: BLE001
        print(lib, 'plain run failed', e, p.stderr[-800:], flush=True)
  for lib in libs:
    env = dict(os.environ, CHIMERA_NO_REBUILD='1')
    if lib != 'base':
      env['CHIMERA_LIB'] = os.path.join(ROOT, 'chimera_amd', 'lib', 'variants', f'libchimera_hip_{lib}.so')
    kern = collections.defaultdict(dict)
    d = os.path.join(tmp, 'trace')
    shutil.rmtree(d, ignore_errors=True)
    rc = subprocess.call(['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '--'] + quick + ['--steps', '10', '--warmup', '2', '--serial', '--groups', '1'],
                         cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    for f in glob.glob(os.path.join(d, '*', '*kernel_stats.csv')):
      for r in csv.DictReader(open(f)):
        kern[short(r['Name'])]['avg_us'] = float(r['AverageNs']) / 1e3
        kern[short(r['Name'])]['calls'] = int(r['Calls'])
    shutil.rmtree(d, ignore_errors=True)
    for pi, counters in enumerate(PASSES):
      d = os.path.join(tmp, f'pmc{pi}')
      shutil.rmtree(d, ignore_errors=True)
      rc = subprocess.call(['rocprofv3', '--kernel-trace', '--pmc'] + counters.split() + ['--output-format', 'csv', '-d', d, '--'] + quick + ['--steps', '3', '--warmup', '1', '--groups', '1'],
                           cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
      agg = collections.defaultdict(lambda: collections.defaultdict(list))
      for f in glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
          agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
      for k, v in agg.items():
        kern[k].update({c: sum(x) / len(x) for c, x in v.items()})
      shutil.rmtree(d, ignore_errors=True)
      print(lib, 'pmc pass', pi, 'rc', rc, flush=True)
    for k, v in sorted(kern.items()):
      if k.startswith(HOT):
        rows.append((lib, k, v))
  with open(os.path.join(ROOT, out_txt), 'w') as f:
    f.write('# same-box A/B with counters: ' + ' '.join(sys.argv[1:]) + '\n')
    f.write('# plain runs (30 steps): ms per step | samples ms | GW kernel ms (HIP events of the one-lane pass) | log_hyper of the last draw\n')
    for lib in libs:
      for i, r in enumerate(plain[lib]):
        f.write(f"{lib:12s} rep{i + 1} ms_per_step={r[0]:.4f} samples={r[1]:.4f} kde={r[2]:.4f} last={r[3]!r}\n")
    f.write('# per kernel: rocprofv3 average duration (serial, one group) and PMC counters per launch (3-step passes, one group)\n')
    cols = ['avg_us', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_WAVE_CYCLES']
    f.write(f"{'build':12s} {'kernel':42s} " + ' '.join(f'{c:>20s}' for c in cols) + f" {'conflict/idx':>12s} {'lds_busy':>9s}\n")
    for lib, k, v in rows:
      conf = v.get('SQ_LDS_BANK_CONFLICT', 0) / v['SQ_LDS_IDX_ACTIVE'] if v.get('SQ_LDS_IDX_ACTIVE') else float('nan')
      busy = v.get('SQ_LDS_IDX_ACTIVE', 0) / (v['GRBM_GUI_ACTIVE'] / 8 * 256) if v.get('GRBM_GUI_ACTIVE') else float('nan')
      f.write(f"{lib:12s} {k[:42]:42s} " + ' '.join(f"{v.get(c, float('nan')):20.6g}" for c in cols) + f" {conf:12.3f} {busy:9.3f}\n")
  print(open(os.path.join(ROOT, out_txt)).read())


if __name__ == '__main__':
  main()
