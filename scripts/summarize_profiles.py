#!/usr/bin/env python3
"""Tables of DESIGN.md / BASELINE.md straight from the committed evidence of a round (no GPU needed):

  python3 scripts/summarize_profiles.py r03            print the tables
  python3 scripts/summarize_profiles.py r03 --write    also replace the blocks between <!-- profiles:BEGIN name --> / <!-- profiles:END name -->
                                                       in DESIGN.md and BASELINE.md

Sources: profiles/rNN/bench*.json (the bench lines), kernel_stats_serial*.csv (rocprofv3 --kernel-trace --stats, one lane), pmc_per_launch*.json
(separate --pmc passes + the code object's sha256 + the static hot-loop mix).  Every number in the generated blocks can be re-derived from those files.
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(rnd, tag=''):
  f = os.path.join(ROOT, 'profiles', rnd, f'bench{tag}.json')
  if not os.path.exists(f):
    return None
  lines = [l for l in open(f).read().strip().split('\n') if l.startswith('{')]
  return json.loads(lines[-1]) if lines else None


def kstats(rnd, tag=''):
  f = os.path.join(ROOT, 'profiles', rnd, f'kernel_stats_serial{tag}.csv')
  out = {}
  if os.path.exists(f):
    for r in csv.DictReader(open(f)):
      out[r['Name'].split('(')[0].replace('void ', '').strip()] = (float(r['AverageNs']) * 1e-3, int(r['Calls']))
  return out


def pmc(rnd, tag=''):
  f = os.path.join(ROOT, 'profiles', rnd, f'pmc_per_launch{tag}.json')
  return json.load(open(f)) if os.path.exists(f) else None


def find(d, prefix):
  for k, v in d.items():
    if k.startswith(prefix):
      return k, v
  return None, None


def kernel_table(rnd):
  ks, k1 = kstats(rnd), kstats(rnd, '_nbatch1')
  p, b = pmc(rnd), bench(rnd)
  cfg = b['config']
  E, P, S, nb, I = cfg['E'], cfg['P'], cfg['S'], cfg['nbatch'], cfg['I']
  units = {'k_kde_marg_sub2': ('pair of pixels', E * P / 2 * nb), 'k_samples_fast': ('sample', E * S * nb / 64.), 'k_selection_fast': ('injection', I * nb / 64.),
           'k_zfactors': (None, None), 'k_marg_fixup': (None, None), 'k_event_stats': (None, None), 'k_tables': (None, None), 'k_reduce_final': (None, None)}
  rk = {k['kernel'].split('<')[0]: k for k in b['roofline']['kernels']}
  rows = ['| Kernel | ms per 128 draws (one lane, rocprofv3 average) | µs at 1 draw | VALU wave-instructions per launch (PMC) | per unit | issue cycles / instruction | useful (fp64 add / mul / fma) | VALU busy at 2.4 GHz / at the held clock | real fp64 TFLOP/s |',
          '|---|---|---|---|---|---|---|---|---|']
  for pref in ('k_tables', 'k_samples_fast', 'k_event_stats', 'k_zfactors', 'k_kde_marg_sub2', 'k_marg_fixup', 'k_selection_fast', 'k_reduce_final'):
    n, v = find(ks, pref)
    n1, v1 = find(k1, pref if pref != 'k_zfactors' else 'k_zf')
    if n1 is None and pref == 'k_zfactors':
      n1, v1 = find(k1, 'k_zfactors')
    pk = find(p['kernels'], pref)[1] if p else None
    insts = pk.get('SQ_INSTS_VALU') if pk else None
    unit, cnt = units[pref]
    r = rk.get(pref)
    rows.append('| `%s` | %s | %s | %s | %s | %s | %s | %s | %s |' % (
      n or pref, '%.3f' % (v[0] * 1e-3) if v else '—', '%.1f' % v1[0] if v1 else '—',
      '%.4g' % insts if insts else '—', ('%.0f per %s' % (insts / cnt, unit)) if insts and cnt else '—',
      '%.2f' % r['cycles_per_valu_inst'] if r and r.get('cycles_per_valu_inst') else '—',
      '%.3f' % r['useful_frac'] if r and r.get('useful_frac') else '—',
      ('%.2f / %.2f' % (r['valu_busy_frac'], r['valu_busy_frac_at_held_clock'])) if r and r.get('valu_busy_frac_at_held_clock') else '—',
      '%.1f' % r['fp64_TFLOPs_real'] if r and r.get('fp64_TFLOPs_real') else '—'))
  sha = p.get('code_object_sha256', '?')[:16] if p else '?'
  rows.append('')
  rows.append(f'(code object `{sha}…`; the busy fractions use the live duration of the sustained one-lane pass of `bench.json`: '
              + ', '.join('`%s` %.3f ms' % (k['kernel'].split('<')[0], k['kernel_ms']) for k in b['roofline']['kernels']) + ')')
  return '\n'.join(rows)


def headline(rnd):
  b, b1 = bench(rnd), bench(rnd, '_nbatch1')
  out = []
  r = b['roofline']
  cb = b.get('cpu_baseline') or {}
  out.append('| Quantity | Value | Source |')
  out.append('|---|---|---|')
  out.append('| **Throughput, 128 draws per call** (bench default) | **%.0f evals/s** — %.3f ms per step (median %.3f, quartiles %.3f / %.3f) = %.0f µs per evaluation; %.3g ev·px·z cells/s | `bench.json` |'
             % (b['value'], b['ms_per_step'], b['step_ms']['median'], b['step_ms']['q25'], b['step_ms']['q75'], 1e3 * b['ms_per_step'] / b['config']['nbatch'], b['config']['cells_per_s']))
  s = b['single_call']
  out.append('| Scalar call `like(**λ)` (HIP-graph replay) | **%.4f ms** (quartiles %.4f / %.4f) = %.0f evals/s | `bench.json: single_call` |' % (s['median_ms'], s['q25_ms'], s['q75_ms'], s['evals_per_s']))
  if cb:
    out.append('| CPU baseline: C/OpenMP restatement, %d threads | %.2f evals/s (median %.3f s per evaluation, %d evaluations) | `bench.json: cpu_baseline` |' % (cb['cores'], cb['value'], cb['eval_s']['median'], cb['eval_s']['n']))
    out.append('| CPU baseline: NumPy oracle, 1 core | %.3f evals/s | `cpu_baseline.numpy_1core` |' % cb['numpy_1core']['value'])
    out.append('| GPU / CPU (context, not credit) | %.0f× the %d-thread C port | `vs_cpu_baseline` |' % (b['vs_cpu_baseline'], cb['cores']))
  pf = b.get('parity_full_size')
  if pf:
    out.append('| Full-size parity inside the bench run | `log_hyper(H0=67)`: HIP %.13f, C port %.13f (abs diff %.1e; tolerance %.0e) | `parity_full_size` |' % (pf['log_hyper_hip'], pf['log_hyper_cpu_port'], pf['abs_diff'], pf['tolerance']))
  dk = [k for k in r['kernels'] if k['kernel'] == r['kernel']][0]
  if 'issue_busy_frac' in r:                               # round 4 on: frac = the USEFUL fraction (fp64 add / mul / fma issue cycles), the busy fraction beside it
    mi = r.get('min_inst') or {}
    out.append('| `roofline` of the driver line | bound **%s**, kernel `%s`: **frac %.3f** = issue cycles of its fp64 add / mul / fma instructions, %.3f of %.4f Tcycle/s (useful work); '
               'issue ports busy with ANY VALU instruction %.3f (%.3f at the %.2f GHz held); %.0f VALU instructions per %s against a paper estimate of the minimum of %.0f (x %.2f); real fp64 %.1f of 78.6 TFLOP/s; '
               'HBM: unique bytes %.2f GB per launch = %.2f of 8 TB/s, PMC traffic %.2f GB = %.2f | `bench.json: roofline` |'
               % (r['bound'], r['kernel'], r['frac'] or 0., r['achieved'] or 0., r['peak'], r['issue_busy_frac'] or 0., r.get('issue_busy_frac_at_held_clock') or 0., dk.get('clock_GHz_under_profile') or 0.,
                  mi.get('per_unit_achieved') or 0., mi.get('unit') or 'unit', (mi.get('per_unit_minimal_paper_estimate') or mi.get('per_unit_minimal') or 0.), (mi.get('achieved_over_paper_estimate') or mi.get('achieved_over_minimal') or 0.),
                  r['fp64_TFLOPs_real'] or 0., r['hbm']['unique_bytes_per_launch'] / 1e9, r['hbm']['frac'], (r['traffic'] or 0) / 1e9, r['hbm']['traffic_frac'] or 0.))
    if s.get('hbm_frac') is not None:
      out.append('| Scalar call against the HBM roofline (call level) | %.1f MB of algorithmic bytes in %.4f ms = %.2f TB/s = **%.3f of 8 TB/s** | `single_call.hbm_frac` |'
                 % (s['algorithmic_bytes'] / 1e6, s['median_ms'], s['hbm_GBs'] / 1e3, s['hbm_frac']))
  else:
    out.append('| `roofline` of the driver line | bound **%s**, kernel `%s`: achieved %.3f of %.4f Tcycle/s = **frac %.3f** (%.3f at the %.2f GHz held); real fp64 %.1f of 78.6 TFLOP/s; HBM: unique bytes %.2f GB per launch = %.2f of 8 TB/s, PMC traffic %.2f GB = %.2f | `bench.json: roofline` |'
               % (r['bound'], r['kernel'], r['achieved'], r['peak'], r['frac'], r['frac_at_held_clock'] or 0., dk.get('clock_GHz_under_profile') or 0.,
                  r['fp64_TFLOPs_real'] or 0., r['hbm']['unique_bytes_per_launch'] / 1e9, r['hbm']['frac'], (r['traffic'] or 0) / 1e9, r['hbm']['traffic_frac'] or 0.))
  # [r6] the kernels against the MEASURED ceilings of their own bodies (scripts/run_probes.py -> probe_ceilings.json) and the LDS side of the GW kernel
  for k in r['kernels']:
    su = k.get('sustained')
    if su and su.get('frac_of_sustained'):
      out.append('| `%s` against the measured ceiling of its body (cache-resident probe) | %.4g %s/s of %.4g sustained = **%.3f**; VALU wave-instructions: %s of %.4g /s (probe clock %.2f GHz) | `roofline.kernels[].sustained`, `%s` |'
                 % (k['kernel'], su['achieved_per_s'], su['unit'], su['probe_per_s'], su['frac_of_sustained'],
                    ('%.4g' % su['valu_winst_per_s']) if su.get('valu_winst_per_s') else 'n/a', su.get('probe_valu_winst_per_s') or 0., su.get('probe_clock_GHz') or 0., su.get('source')))
  pj = pmc(rnd)
  if pj:
    n_, gk = find(pj['kernels'], 'k_kde_marg_sub2')
    if gk and gk.get('SQ_LDS_IDX_ACTIVE') and gk.get('GRBM_GUI_ACTIVE'):
      cyc = gk['GRBM_GUI_ACTIVE'] / 8 * 256
      out.append('| LDS pipe of `%s` | `SQ_LDS_IDX_ACTIVE` %.4g of %.4g CU-cycles = **%.2f busy**, of which bank conflicts %.4g = %.0f %%; `SQ_INSTS_LDS` %.4g per launch | `pmc_per_launch.json` |'
                 % (n_, gk['SQ_LDS_IDX_ACTIVE'], cyc, gk['SQ_LDS_IDX_ACTIVE'] / cyc, gk.get('SQ_LDS_BANK_CONFLICT', 0.), 100. * gk.get('SQ_LDS_BANK_CONFLICT', 0.) / gk['SQ_LDS_IDX_ACTIVE'], gk.get('SQ_INSTS_LDS', 0.)))
  if r.get('hbm_call_frac') is not None:
    out.append('| `roofline.hbm_call_frac` (SURVEY 8(d): algorithmic bytes of one evaluation over the scalar call) | **%.3f of 8 TB/s** | `bench.json: roofline` |' % r['hbm_call_frac'])
  for k in r['kernels'][1:]:
    if k.get('valu_busy_frac'):
      mi = k.get('min_inst') or {}
      out.append('| `%s` | %.3f ms per launch; useful (fp64 add / mul / fma) %s; %.4g VALU instructions x %.2f cycles -> busy %.2f at 2.4 GHz, %.2f at the %.2f GHz held%s; real fp64 %.1f TFLOP/s; PMC traffic %.2f GB -> %.2f of 8 TB/s | `roofline.kernels` |'
                 % (k['kernel'], k['kernel_ms'], ('**%.3f**' % k['useful_frac']) if k.get('useful_frac') else '—', k['valu_inst_per_launch'], k['cycles_per_valu_inst'], k['valu_busy_frac'],
                    k.get('valu_busy_frac_at_held_clock') or 0., k.get('clock_GHz_under_profile') or 0.,
                    ('; %.0f instructions per %s, paper estimate of the minimum %.0f (x %.2f)' % (mi['per_unit_achieved'], mi['unit'], mi.get('per_unit_minimal_paper_estimate', mi.get('per_unit_minimal')), mi.get('achieved_over_paper_estimate', mi.get('achieved_over_minimal')))) if mi else '',
                    k.get('fp64_TFLOPs_real') or 0., (k.get('traffic_bytes_per_launch') or 0) / 1e9, k.get('hbm_traffic_frac') or 0.))
  if b1:
    for k in b1['roofline']['kernels']:
      if k.get('traffic_bytes_per_launch'):
        out.append('| `%s` at ONE draw per launch (HBM is the applicable bound) | %.1f µs; PMC traffic %.0f MB -> %.2f TB/s = **%.2f of 8 TB/s** (unique bytes %.0f MB -> %.2f) | `bench_nbatch1.json` |'
                   % (k['kernel'], 1e3 * k['kernel_ms'], k['traffic_bytes_per_launch'] / 1e6, k['hbm_traffic_GBs'] / 1e3, k['hbm_traffic_frac'], k['unique_bytes_per_launch'] / 1e6, k['hbm_unique_frac']))
  for tag, label in (('_approximate', "`kind_p_gw3d='approximate'`, 128 draws per call"), ('_full', "`kind_p_gw3d='full'`, 4 draws per call"), ('_C1', 'C1 (10 events, 1-D)'), ('_C2', 'C2 (100 ev × 16 px × 500 z)'),
                     ('_C4', 'C4 (69 ev × 16 px × 500 z, 1e6 injections)'), ('_C5', 'C5 (10 000 ev × 32 px × 1000 z, mg_flrw) on ONE GPU, 16 draws per call')):
    x = bench(rnd, tag)
    if x:
      extra = ''
      if tag == '_full':
        k = x['roofline']['kernels'][0]
        extra = ('; `%s` %.2f ms per launch = %.2f Tpair/s = %.3f of the power-sum ceiling' % (k['kernel'], k['kernel_ms'], k['Gpairs_s'] / 1e3, k['pair_frac'])) if k.get('Gpairs_s') else ''
      out.append('| %s | %.0f evals/s (%.3f ms per step; scalar call %.3f ms)%s | `bench%s.json` |' % (label, x['value'], x['ms_per_step'], x.get('single_call_ms') or 0., extra, tag))
  ex = (b.get('extra') or {}).get('configs') or {}
  lab = {'C1': 'C1 (10 events, 1-D)', 'C2': 'C2 (100 ev × 16 px × 500 z)', 'C4': 'C4 (69 ev × 16 px × 500 z, 1e6 injections)',
         'C3_approximate': "C3, `kind_p_gw3d='approximate'`", 'C3_full': "C3, `kind_p_gw3d='full'`"}
  for key in ('C1', 'C2', 'C4', 'C3_approximate', 'C3_full'):
    x = ex.get(key)
    if x:
      out.append('| %s -- leg of the SAME driver-run line, %d draws per call | %.0f evals/s (%.3f ms per step; scalar call %.3f ms) | `bench.json: extra.configs.%s` |'
                 % (lab[key], x['nbatch'], x['evals_per_s'], x['ms_per_step'], x['single_call_ms'], key))
  od = ex.get('one_draw_kernels')
  if od:
    out.append('| C3 kernels at ONE draw per call (leg of the driver-run line; HBM is the applicable bound) | sample stage %.1f µs = %.2f of 8 TB/s (unique bytes), GW kernel + fix-up %.1f µs = %.2f; tables %.1f µs; whole eager call %.1f µs | `bench.json: extra.configs.one_draw_kernels` |'
               % (od['samples_us'], od['samples_hbm_frac_unique'] or 0., od['gw_kernel_us'], od['gw_hbm_frac_unique'] or 0., od['tables_us'], od['eval_us']))
  for tag, label in (('_fused', 'FUSED event kernel (`--fused 2`: off by default), 128 draws per call'), ('_fused_nbatch1', 'FUSED event kernel, one draw per call')):
    x = bench(rnd, tag)
    if x:
      k = x['roofline']['kernels'][0]
      out.append('| %s | %.0f evals/s (%.3f ms per step; scalar call %.3f ms); `%s` %.3f ms per launch, PMC traffic %.2f GB per launch | `bench%s.json`, `ab_fused_event_kernel.txt` |'
                 % (label, x['value'], x['ms_per_step'], x.get('single_call_ms') or 0., k['kernel'], k['kernel_ms'], (k.get('traffic_bytes_per_launch') or 0) / 1e9, tag))
  out.append('| One-time hand-over of the host arrays | %.2f s; not part of `value` | `bench.json: setup_s.upload_once` |' % b['setup_s']['upload_once'])
  return '\n'.join(out)


def main():
  rnd = sys.argv[1] if len(sys.argv) > 1 else 'r03'
  blocks = {f'{rnd}-kernels': kernel_table(rnd), f'{rnd}-headline': headline(rnd)}
  for name, txt in blocks.items():
    print(f'--- {name}\n{txt}\n')
  if '--write' in sys.argv:
    for doc in ('DESIGN.md', 'BASELINE.md'):
      p = os.path.join(ROOT, doc)
      s = open(p).read()
      for name, txt in blocks.items():
        pat = re.compile(r'(<!-- profiles:BEGIN %s -->\n)(?:.*?\n)??(<!-- profiles:END %s -->)' % (re.escape(name), re.escape(name)), re.S)
        if pat.search(s):
          s = pat.sub(lambda m: m.group(1) + txt + '\n' + m.group(2), s)
      open(p, 'w').write(s)


if __name__ == '__main__':
  main()
