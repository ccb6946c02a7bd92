#!/usr/bin/env python3
"""Timeline of the kernels of one scalar call (one draw per call, HIP-graph replay) from a rocprofv3 kernel trace:

  python3 scripts/timeline_scalar.py [out.txt [bench.py arguments ...]]          (on the GPU box; never touches the GPU itself)
                                     e.g. out.txt --events 125 --inj 12500: the per-rank share of an 8-GPU run of C3

Runs `rocprofv3 --kernel-trace -- python3 bench.py --nbatch 1 ...`, takes the LAST complete call of the trace (k_tables ... k_combine /
k_reduce_final) and prints start / end of every kernel relative to the call's first kernel, plus the idle gaps on the critical path.
"""
import csv, glob, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'timeline_scalar.txt')
d = os.path.join(ROOT, 'gpurun_out', 'tl_trace')
shutil.rmtree(d, ignore_errors=True)
os.environ['TMPDIR'] = '/tmp'
cmd = ['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', d, '--', 'python3', 'bench.py', '--nbatch', '1', '--steps', '200', '--warmup', '20',
       '--no-cpu-baseline', '--no-single-call', '--no-extra'] + sys.argv[2:]
subprocess.call(cmd, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
rows = []
for f in glob.glob(os.path.join(d, '*', '*kernel_trace.csv')):
  for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')))
rows.sort()
# calls start at k_tables; only full evaluations count (bench.py ends with a few selection-only calls: chm_eval(NULL, sel, ...))
st = [i for i, r in enumerate(rows) if r[2].startswith('k_tables')]
calls = [rows[a:b] for a, b in zip(st, st[1:] + [len(rows)]) if any(r[2].startswith('k_kde_marg') for r in rows[a:b])]
calls = calls[-61:-1]                                        # the last complete ones
lines = []
per_call = sorted((c[-1][1] - c[0][0]) * 1e-3 for c in calls)
period = sorted((b[0][0] - a[0][0]) * 1e-3 for a, b in zip(calls[:-1], calls[1:]))
call = calls[-1]
t0 = call[0][0]
lines.append('one scalar call (a late one of the trace)%s; times in us relative to the start of k_tables' % ((' [bench.py ' + ' '.join(sys.argv[2:]) + ']') if sys.argv[2:] else ''))
for s_, e_, n_ in call:
  lines.append('%8.1f %8.1f  %6.1f us  %s' % ((s_ - t0) * 1e-3, (e_ - t0) * 1e-3, (e_ - s_) * 1e-3, n_))
lines.append('first kernel start -> last kernel end: %.1f us; median over %d calls %.1f us' % ((call[-1][1] - t0) * 1e-3, len(per_call), per_call[len(per_call) // 2]))
lines.append('start-to-start of consecutive calls (median): %.1f us' % period[len(period) // 2])
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
shutil.rmtree(d, ignore_errors=True)
