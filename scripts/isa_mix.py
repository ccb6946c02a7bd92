#!/usr/bin/env python3
"""Static instruction mix of the gfx950 code object inside libchimera_hip.so (runs on the CPU; no GPU needed).

  python3 scripts/isa_mix.py [--lib PATH] [--kernel SUBSTR ...] [--loops] [--json OUT] [--dump KERNEL_SUBSTR]

* finds the clang offload bundle in the shared library, takes its gfx950 entry, prints the entry's sha256 (the key that ties a PMC
  file to the binary it was collected from: scripts/collect_profiles.py stores it, bench.py compares it with the library it loaded);
* disassembles it with llvm-objdump --mcpu=gfx950 and classifies every instruction of every kernel:
    f64      VALU that computes in fp64 (v_*_f64 arithmetic, compares, conversions to/from f64): half-rate on the SIMD-32 (4 cycles
             per wave64 instruction; transcendentals v_rcp/v_rsq/v_sqrt_f64 listed separately as f64_trans)
    mov      v_mov_b32 / v_mov_b64 / v_accvgpr_* (register moves, DPP moves included)
    cndmask  v_cndmask_b32
    lane     v_readlane / v_writelane / v_readfirstlane (SGPR spill traffic and cross-lane broadcasts)
    valu     every other VALU (integer, 32-bit float, compares of integers, bit operations): 2 cycles per wave64 instruction
    salu, smem, lds, vmem, wait (s_waitcnt / s_nop), branch
* with --loops: every natural loop (a backward branch to a label) of the selected kernels with its own mix, innermost loops first,
  so that the hot loops of k_kde_marg_sub2 / k_samples_fast can be read off without a GPU.
Issue-cost model = what scripts/issue_cost.hip measured on the card (profiles/r03/issue_cost.txt): cycles = 2 n_fast + 16 n_f64_trans +
8 n_trans32 + 4 n_other, where "fast" are the few simple 32-bit opcodes that run at the SIMD-32's full rate (v_mov_b32, v_add/sub_u32,
v_and_b32, v_ashrrev_i32, v_fma/mul_f32 ...); everything else -- fp64 arithmetic as well as v_mov_b64, v_cndmask, v_med3, DPP moves,
v_readlane, compares, conversions, 64-bit integer operations -- occupies the SIMD for 4 cycles per wave64 instruction.
"""
import collections
import hashlib
import json
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
# Issue cycles per wave64 instruction on one SIMD, measured on the card (profiles/r03/issue_cost.txt): only the simplest 32-bit VALU
# operations run at the SIMD-32's full rate; fp64 arithmetic, 64-bit moves and integer operations, v_cndmask, v_med3, shifts by a VGPR
# amount, DPP moves, v_readlane / v_writelane, compares and conversions all take 4 cycles; fp64 reciprocal / square root 16, fp32 exp/log 8.
CYC_FAST, CYC_VALU, CYC_TRANS32, CYC_F64_TRANS = 2, 4, 8, 16
CYC_F64 = CYC_VALU
FAST = ('v_mov_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_ashrrev_i32', 'v_lshrrev_b32',
        'v_fma_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_fmac_f32', 'v_mac_f32')      # measured or same VOP2 family as a measured one
TRANS32 = ('v_exp_f32', 'v_log_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_sin_f32', 'v_cos_f32')

TRANS64 = ('v_rcp_f64', 'v_rsq_f64', 'v_sqrt_f64')
# fp64 flops per lane of one instruction (FMA = 2; add / mul / min / max = 1; the rest are not arithmetic flops)
FLOPS64 = {'v_fma_f64': 2, 'v_fmac_f64': 2, 'v_add_f64': 1, 'v_mul_f64': 1, 'v_max_f64': 1, 'v_min_f64': 1, 'v_div_fmas_f64': 2, 'v_rcp_f64': 1, 'v_rsq_f64': 1,
           'v_sqrt_f64': 1}


def code_object(lib):
  """(bytes of the gfx950 code object, sha256 hex) from the clang offload bundle of a HIP shared library."""
  d = open(lib, 'rb').read()
  i = d.find(b'__CLANG_OFFLOAD_BUNDLE__')
  if i < 0:
    raise SystemExit(f'{lib}: no offload bundle')
  n = struct.unpack_from('<Q', d, i + 24)[0]
  o = i + 32
  for _ in range(n):
    off, size, tl = struct.unpack_from('<QQQ', d, o)
    o += 24
    triple = d[o:o + tl]
    o += tl
    if b'gfx950' in triple:
      co = d[i + off:i + off + size]
      return co, hashlib.sha256(co).hexdigest()
  raise SystemExit(f'{lib}: no gfx950 entry in the bundle')


def classify(mn):
  base = mn
  for suf in ('_e32', '_e64', '_dpp', '_sdwa', '_e64_dpp'):
    if base.endswith(suf):
      base = base[:-len(suf)]
  if base.startswith('v_'):
    if base.startswith(TRANS64):
      return 'f64_trans'
    if 'f64' in base:
      return 'f64'
    if base.startswith(('v_mov_b', 'v_accvgpr')):
      return 'mov'
    if base.startswith('v_cndmask'):
      return 'cndmask'
    if base.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')):
      return 'lane'
    if base.startswith('v_mfma'):
      return 'mfma'
    return 'valu'
  if base.startswith(('s_waitcnt', 's_nop', 's_sleep')):
    return 'wait'
  if base.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc', 's_swappc', 's_barrier')):
    return 'branch'
  if base.startswith(('s_load', 's_buffer_load', 's_store', 's_dcache', 's_memtime', 's_memrealtime')):
    return 'smem'
  if base.startswith('s_'):
    return 'salu'
  if base.startswith('ds_'):
    return 'lds'
  if base.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
    return 'vmem'
  return 'other'


VALU_CLASSES = ('f64', 'f64_trans', 'mov', 'cndmask', 'lane', 'valu', 'mfma')


AMF64 = ('v_fma_f64', 'v_fmac_f64', 'v_add_f64', 'v_mul_f64', 'v_div_fmas_f64')      # what the PMC counters SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 can see


def summarize(insts):
  c = collections.Counter(k for _, _, k, _ in insts)
  c['f64_amf'] = sum(1 for _, mn, k, _ in insts if k == 'f64' and mn.startswith(AMF64))
  nv = sum(c[k] for k in VALU_CLASSES)
  def base(mn):
    return re.sub(r'_(e32|e64|sdwa)$', '', mn)
  c['fast'] = sum(1 for _, mn, k, ops in insts if k in VALU_CLASSES and base(mn) in FAST and not re.search(r',\s*s\d+\s*$', ops))    # (a v_mov_b32 from an SGPR takes 4)
  c['trans32'] = sum(1 for _, mn, k, _ in insts if base(mn) in TRANS32)
  cyc = CYC_FAST * c['fast'] + CYC_F64_TRANS * c['f64_trans'] + CYC_TRANS32 * c['trans32'] + CYC_VALU * (nv - c['fast'] - c['f64_trans'] - c['trans32'])
  fl = 0
  for _, mn, k, _ in insts:
    b = re.sub(r'_(e32|e64|dpp|sdwa)$', '', mn)
    fl += FLOPS64.get(b, 0)
  out = {k: c[k] for k in ('f64', 'f64_amf', 'f64_trans', 'fast', 'trans32', 'mov', 'cndmask', 'lane', 'valu', 'salu', 'smem', 'lds', 'vmem', 'wait', 'branch') if c[k]}
  out.update({'valu_total': nv, 'f64_share': round((c['f64'] + c['f64_trans']) / nv, 4) if nv else None, 'issue_cycles': cyc,
              'cycles_per_valu': round(cyc / nv, 3) if nv else None, 'f64_flop_per_lane': fl})
  return out


def parse(co_path):
  txt = subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', co_path], stdout=subprocess.PIPE, text=True, check=True).stdout
  kernels, cur = collections.OrderedDict(), None
  for line in txt.split('\n'):
    m = re.match(r'^([0-9a-f]+) <([^>]+)>:', line)
    if m:
      cur = m.group(2)
      kernels[cur] = []
      continue
    m = re.match(r'^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):', line)
    if m and cur is not None:
      mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
      kernels[cur].append((addr, mn, classify(mn), ops))
  return kernels


def loops_of(insts):
  """Natural loops from backward branches: (start_addr, end_addr) with end the branch instruction itself."""
  addrs = [a for a, _, _, _ in insts]
  out = []
  for i, (a, mn, k, ops) in enumerate(insts):
    if mn.startswith(('s_cbranch', 's_branch')):
      m = re.match(r'(-?\d+)', ops.strip())
      if not m:
        continue
      off = int(m.group(1))
      if off >= 32768:
        off -= 65536
      tgt = a + 4 + 4 * off
      if tgt <= a:
        out.append((tgt, a))
  # merge loops sharing a header (several back edges)
  byh = {}
  for s, e in out:
    byh[s] = max(byh.get(s, e), e)
  return sorted(byh.items())


def demangle(names):
  try:
    r = subprocess.run(['c++filt'] + names, stdout=subprocess.PIPE, text=True).stdout.strip().split('\n')
    if len(r) == len(names):
      return [x.split('(')[0].replace('void ', '') for x in r]
  except OSError:
    pass
  return names


def analyse(lib, want=None, with_loops=False):
  co, sha = code_object(lib)
  tmp = f'/tmp/chm_isa_{os.getpid()}.co'
  with open(tmp, 'wb') as f:
    f.write(co)
  try:
    ks = parse(tmp)
  finally:
    os.unlink(tmp)
  names = [n for n in ks if ks[n]]
  dem = dict(zip(names, demangle(names)))
  res = {'code_object_sha256': sha, 'library': os.path.relpath(lib, ROOT), 'kernels': {},
         'issue_cost_model': {'fast': CYC_FAST, 'valu': CYC_VALU, 'trans32': CYC_TRANS32, 'f64_trans': CYC_F64_TRANS, 'fast_opcodes': list(FAST),
                              'note': 'cycles per wave64 instruction on one SIMD-32, measured by scripts/issue_cost.hip (profiles/r03/issue_cost.txt): 2 for '
                                      'the listed simple 32-bit opcodes with VGPR / constant sources, 16 for v_rcp/v_rsq/v_sqrt_f64, 8 for fp32 '
                                      'transcendentals, 4 for every other VALU instruction (all fp64 arithmetic, 64-bit moves, v_cndmask, DPP, v_readlane ...)'}}
  for n in names:
    d = dem[n]
    if want and not any(w in d for w in want):
      continue
    insts = ks[n]
    entry = {'whole_kernel': summarize(insts)}
    if with_loops:
      lps = loops_of(insts)
      ll = []
      for s, e in lps:
        body = [x for x in insts if s <= x[0] <= e]
        inner = not any((s2 > s or e2 < e) and s2 >= s and e2 <= e for s2, e2 in lps if (s2, e2) != (s, e))
        sm = summarize(body)
        sm.update({'start': hex(s), 'end': hex(e), 'innermost': inner, 'instructions': len(body)})
        ll.append(sm)
      entry['loops'] = sorted(ll, key=lambda x: -x['valu_total'])
      # the loop that stands for the kernel's dynamic mix: the largest one of at most 1600 instructions (a pass of the sample stage, an
      # item of the GW kernel); kernels without a loop are priced with their whole body
      cand = [l for l in entry['loops'] if l['instructions'] <= 1600]
      entry['hot_loop'] = cand[0] if cand else entry['whole_kernel']
    res['kernels'][d] = entry
  return res, ks, dem


def main():
  argv = sys.argv[1:]
  lib = os.path.join(ROOT, 'chimera_amd', 'lib', 'libchimera_hip.so')
  want, with_loops, jout, dump = [], False, None, None
  i = 0
  while i < len(argv):
    a = argv[i]
    if a == '--lib':
      lib = argv[i + 1]; i += 1
    elif a == '--kernel':
      want.append(argv[i + 1]); i += 1
    elif a == '--loops':
      with_loops = True
    elif a == '--json':
      jout = argv[i + 1]; i += 1
    elif a == '--dump':
      dump = argv[i + 1]; i += 1
    i += 1
  res, ks, dem = analyse(lib, want or None, with_loops)
  if dump:
    for n, d in dem.items():
      if dump in d:
        print(f'; {d}')
        for a, mn, k, ops in ks[n]:
          print(f'{a:08x}  {k:9s} {mn} {ops}')
    return
  if jout:
    with open(jout, 'w') as f:
      json.dump(res, f, indent=1, sort_keys=True)
  print('code object sha256', res['code_object_sha256'])
  for d, e in res['kernels'].items():
    w = e['whole_kernel']
    print(f"{d[:70]:70s} VALU {w['valu_total']:6d}  f64 {w.get('f64', 0):5d} trans {w.get('f64_trans', 0):3d} mov {w.get('mov', 0):4d} cnd {w.get('cndmask', 0):4d} "
          f"lane {w.get('lane', 0):4d} other {w.get('valu', 0):5d} | salu {w.get('salu', 0):5d} lds {w.get('lds', 0):4d} vmem {w.get('vmem', 0):4d} | f64 share {w['f64_share']}")
    for l in e.get('loops', []):
      print(f"    loop {l['start']}..{l['end']} {'inner' if l['innermost'] else 'outer'} insts {l['instructions']:5d} VALU {l['valu_total']:5d} f64 {l.get('f64', 0):4d} "
            f"trans {l.get('f64_trans', 0):2d} mov {l.get('mov', 0):3d} cnd {l.get('cndmask', 0):3d} lane {l.get('lane', 0):3d} other {l.get('valu', 0):4d} salu {l.get('salu', 0):4d} "
            f"lds {l.get('lds', 0):3d} vmem {l.get('vmem', 0):3d} cyc {l['issue_cycles']}")


if __name__ == '__main__':
  main()
