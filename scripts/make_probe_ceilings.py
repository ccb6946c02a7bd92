#!/usr/bin/env python3
"""profiles/rNN/probe_ceilings.json from one probe session:  python3 scripts/make_probe_ceilings.py gpurun_out/r06d r06

Reads <dir>/probe_E4_nb4.json (scripts/run_probes.py: work per second of the two kernel bodies on the cache-resident workload) and <dir>/probe_pmc.txt (the
same launches under rocprofv3 --pmc: VALU wave-instructions per second, clock, LDS counters) and ties them to the RELEASE library of this tree (sha256 of
its gfx950 code object: bench.py prints roofline.frac_of_sustained only when the loaded library is that binary)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def main():
  d, rnd = sys.argv[1], sys.argv[2]
  pr = json.load(open(os.path.join(ROOT, d, 'probe_E4_nb4.json')))
  pmc = {}
  for line in open(os.path.join(ROOT, d, 'probe_pmc.txt')):
    m = re.match(r'(?:void )?(k_probe_\w+)<', line)
    if not m:
      continue
    vals = dict(re.findall(r'(\w+)=([0-9.e+]+)', line))
    w = re.search(r'VALU winst/s = ([0-9.e+]+)', line)
    c = re.search(r'clock GHz = ([0-9.]+)', line)
    ms = re.search(r'median ms \(second half\) ([0-9.]+)', line)
    pmc[m.group(1)] = dict({k: float(v) for k, v in vals.items()}, valu_winst_per_s=float(w.group(1)) if w else None, clock_GHz=float(c.group(1)) if c else None,
                           launch_ms=float(ms.group(1)) if ms else None)
  kernels = {}
  for name, probe_kernel in (('k_kde_marg_sub2', 'k_probe_gw'), ('k_samples_fast', 'k_probe_samples')):
    full = next(k for k in pr if k.startswith(name))
    e = pr[full]
    c = pmc.get(probe_kernel, {})
    lds_busy = c['SQ_LDS_IDX_ACTIVE'] / (c['GRBM_GUI_ACTIVE'] / 8 * 256) if c.get('GRBM_GUI_ACTIVE') and c.get('SQ_LDS_IDX_ACTIVE') else None
    kernels[name] = {"body_of": full, "unit": e['unit'], "units_per_s": e['units_per_s'], "launches": e['launches'], "launch_ms": e['launch_ms_median_second_half'],
                     "total_s": e['total_s'], "valu_winst_per_s": c.get('valu_winst_per_s'), "clock_GHz": c.get('clock_GHz'),
                     "valu_issue_busy_at_4_cycles": c['valu_winst_per_s'] * 4 / (1024 * c['clock_GHz'] * 1e9) if c.get('valu_winst_per_s') and c.get('clock_GHz') else None,
                     "lds_busy": lds_busy, "lds_conflict_share": c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'] if c.get('SQ_LDS_IDX_ACTIVE') else None}
  out = {"release_code_object_sha256": bench.code_object_sha256(os.path.join(ROOT, 'chimera_amd', 'lib', 'libchimera_hip.so')),
         "workload": pr['workload'], "kernels": kernels,
         "how": "scripts/run_probes.py on the -DCHM_PROBE build of the same sources: scripts/gw_loop_probe.hip / scripts/sample_body_probe.hip replay the production "
                "bodies (kde_marg_sub2_body<32, 4, 200, false>, samples_fast_body<2, false, false>) on a few-MB resident workload in launches of ~0.12 s for > 1 s; "
                "second pass under rocprofv3 --pmc for the instruction and LDS counters"}
  dst = os.path.join(ROOT, 'profiles', rnd, 'probe_ceilings.json')
  with open(dst, 'w') as f:
    json.dump(out, f, indent=1)
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  main()
