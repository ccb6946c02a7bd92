#!/usr/bin/env python3
"""Memory-path counters of the hot kernels for several library builds on one box (run through gpurun):

  python3 scripts/pmc_memory_path.py OUT.txt base plain nost ... [-- <bench.py arguments>]

base = the release library, NAME = chimera_amd/lib/variants/libchimera_hip_NAME.so.  One rocprofv3 --kernel-trace --pmc run of 3 steps (one event
group) per build and counter pass; per kernel and launch: the average of every counter and of the kernel's duration.  This process never touches the GPU."""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (a pass of TA_* counters -- TA_BUSY_avr, TA_ADDR_STALLED_BY_TC_CYCLES_sum, TA_DATA_STALLED_BY_TC_CYCLES_sum -- did not finish within seven minutes on the
#  box of gpurun call r06z and was killed there: the texture-addresser and TCP counters are left out)
PASSES = ["SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE",
          "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum",
          "TCC_EA0_WRREQ_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_BUSY_avr"]
HOT = ('k_kde_marg_sub2', 'k_samples_fast')


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
  tmp = os.path.join(ROOT, 'gpurun_out', 'pmcmp_tmp')
  quick = ['python3', 'bench.py', '--no-cpu-baseline', '--no-single-call', '--no-extra', '--steps', '3', '--warmup', '1', '--groups', '1'] + bench_args
  res = collections.defaultdict(dict)
  for lib in libs:
    env = dict(os.environ, CHIMERA_NO_REBUILD='1')
    if lib != 'base':
      env['CHIMERA_LIB'] = os.path.join(ROOT, 'chimera_amd', 'lib', 'variants', f'libchimera_hip_{lib}.so')
    for pi, counters in enumerate(PASSES):
      d = os.path.join(tmp, f'pmc{pi}')
      shutil.rmtree(d, ignore_errors=True)
      try:
        rc = subprocess.call(['rocprofv3', '--kernel-trace', '--pmc'] + counters.split() + ['--output-format', 'csv', '-d', d, '--'] + quick,
                             cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150)
      except subprocess.TimeoutExpired:
        print(lib, 'pmc pass', pi, 'did not finish in 150 s: stopping here (no further GPU step after a killed one)', flush=True)
        raise SystemExit(3)
      agg = collections.defaultdict(lambda: collections.defaultdict(list))
      for f in glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
          agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
      dur = collections.defaultdict(list)
      for f in glob.glob(os.path.join(d, '*', '*kernel_trace.csv')):
        for r in csv.DictReader(open(f)):
          dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
      for k, v in agg.items():
        if k.startswith(HOT):
          res[(lib, k)].update({c: sum(x) / len(x) for c, x in v.items()})
          res[(lib, k)][f'us_pass{pi}'] = sum(dur[k]) / max(1, len(dur[k]))
      shutil.rmtree(d, ignore_errors=True)
      print(lib, 'pmc pass', pi, 'rc', rc, flush=True)
  with open(os.path.join(ROOT, out_txt), 'w') as f:
    f.write('# memory-path counters per launch: ' + ' '.join(sys.argv[1:]) + '\n')
    for (lib, k), v in sorted(res.items(), key=lambda kv: (kv[0][1], kv[0][0])):
      f.write(f'{lib:8s} {k}\n')
      for c, x in sorted(v.items()):
        f.write(f'    {c:44s} {x:16.6g}\n')
  print(open(os.path.join(ROOT, out_txt)).read())


if __name__ == '__main__':
  main()
