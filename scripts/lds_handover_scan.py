#!/usr/bin/env python3
"""Static check of the LDS hand-overs in the gfx950 code of the library (second half).

  python3 scripts/lds_handover_scan.py [--asm FILE.s] [--audit] [--write-allow tests/golden/lds_handover_allow.json] [--json OUT] [-v]

Lanes of a wave (and waves of a block) pass data to each other through LDS.  The hardware runs the LDS instructions of one wave in issue order, so
inside a wave the only party that can break a hand-over is the COMPILER, by moving an LDS access across it (round 3 saw exactly that in a build of
k_full_kde_chain).  The source therefore puts an ordering point at every hand-over: wave_sync() (fence + wave barrier: no instruction, but nothing moves
across it) or __syncthreads().  Both leave a mark in the assembly -- `; wave barrier` / `s_barrier` -- and this script walks the instruction stream of
every kernel in the compiler's assembly output and reports each place where an LDS STORE (ds_write / ds_add ...) is followed by an LDS LOAD (or a load
by a store) with no such mark in between on some control-flow path: a succession the compiler was free to reorder.  Successions that are the same
lane's own data, or accesses of two different arrays, cannot be told apart here; tests/golden/lds_handover_allow.json lists, for the kernels that hand
data over INSIDE a wave (those with a `; wave barrier`), every unmarked succession by the text of its two source lines with the reason it is harmless --
one that is not listed fails tests/test_abi_and_host.py::test_lds_handovers_sit_between_ordering_points.  The production GW kernel
(k_kde_marg_sub2<32, *, 200, false>) has none at all.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOT_MEMORY = ('ds_bpermute', 'ds_permute', 'ds_swizzle', 'ds_nop', 'ds_gws', 'ds_consume', 'ds_append', 'ds_ordered')
STORES = ('ds_write', 'ds_store', 'ds_add', 'ds_sub', 'ds_min', 'ds_max', 'ds_and', 'ds_or', 'ds_xor', 'ds_inc', 'ds_dec', 'ds_wrxchg', 'ds_cmpst', 'ds_pk_add')
LOADS = ('ds_read', 'ds_load')


def device_asm(path=None):
  if path:
    return open(path).read()
  sys.path.insert(0, ROOT)
  import __graft_entry__ as g
  flags = [f for f in g.HIP_FLAGS if f not in ('-shared', '-fPIC')]
  with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, 'chm_dev.s')
    subprocess.check_call([g.HIPCC] + flags + ['-gline-tables-only', '--cuda-device-only', '-S', '-o', out, g.SRC], cwd=ROOT, stderr=subprocess.DEVNULL)
    return open(out).read()


def _functions(asm):
  """-> (files, [(name, [lines])]) of the device functions in the assembly"""
  files = {}
  for m in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', asm, re.M):
    files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
  fns, fn, body = [], None, None
  for raw in asm.split('\n'):
    s = raw.strip()
    m = re.match(r'^(_Z\w+|k_\w+):', s)
    if m and not s.startswith('.L'):
      fn, body = m.group(1), []
      fns.append((fn, body))
      continue
    if s.startswith('.Lfunc_end'):
      fn = None
      continue
    if fn is not None:
      body.append(s)
  return files, fns


def _classify(s):
  """'store' | 'load' | 'both' | None for one instruction"""
  if not s.startswith('ds_') or s.startswith(NOT_MEMORY):
    return None
  op = s.split()[0]
  if op.startswith(LOADS):
    return 'load'
  if op.startswith(STORES):
    return 'both' if '_rtn' in op else 'store'
  return 'both'


def scan(asm):
  """-> {kernel: [ {kind, first, first_at, second, second_at} ]}: along the CONTROL FLOW of every function (basic blocks from the labels, edges from
  s_branch / s_cbranch_* and fall-through; iterated to a fixed point), each LDS access whose predecessor LDS access on some path -- with no
  `; wave barrier` / `s_barrier` in between -- is of the other kind."""
  files, fns = _functions(asm)
  out = collections.OrderedDict()
  for fn, body in fns:
    # basic blocks
    blocks, cur, order = {}, '<entry>', ['<entry>']
    blocks[cur] = []
    loc = ''
    for s in body:
      m = re.match(r'^(\.LBB\w+):', s)
      if m:
        cur = m.group(1)
        blocks[cur] = []
        order.append(cur)
        continue
      m = re.match(r'^\.loc\s+(\d+)\s+(\d+)', s)
      if m:
        loc = f"{files.get(int(m.group(1)), m.group(1))}:{m.group(2)}"
        continue
      if not s or s.startswith(('.', ';')) and not s.startswith('; wave barrier'):
        continue
      blocks[cur].append((s, loc))
    succ = {b: [] for b in order}
    for i, b in enumerate(order):
      fall = True
      for s, _ in blocks[b]:
        m = re.match(r'^s_(c?branch\w*)\s+(\.LBB\w+)', s)
        if m:
          if m.group(2) in succ:
            succ[b].append(m.group(2))
          if m.group(1) == 'branch':
            fall = False
        elif s.startswith(('s_endpgm', 's_setpc_b64')):
          fall = False
      if fall and i + 1 < len(order):
        succ[b].append(order[i + 1])
    # forward data flow: the set of "last LDS access since the last ordering point" that can reach a block's entry
    IN = {b: set() for b in order}
    found, changed = set(), True
    while changed:
      changed = False
      for b in order:
        state = set(IN[b])
        for s, at in blocks[b]:
          if s.startswith('; wave barrier') or s.startswith('s_barrier'):
            state = set()
            continue
          k = _classify(s)
          if k is None:
            continue
          for (pk, ps, pat) in state:
            if pk != k or 'both' in (pk, k) and not (pk == 'both' and k == 'both'):
              found.add((f"{pk}->{k}", ps, pat, s, at))
          state = {(k, s, at)}
        for n in succ[b]:
          if not state <= IN[n]:
            IN[n] |= state
            changed = True
    out[fn] = [dict(kind=a, first=b_, first_at=c, second=d, second_at=e) for (a, b_, c, d, e) in sorted(found)]
  return out


def wave_private_kernels(asm):
  """kernels whose code contains a wave-level ordering point: the ones that hand data over between the lanes of a wave without a block barrier"""
  has, fn = set(), None
  for raw in asm.split('\n'):
    s = raw.strip()
    m = re.match(r'^(_Z\w+|k_\w+):', s)
    if m and not s.startswith('.L'):
      fn = m.group(1)
    elif s.startswith('; wave barrier') and fn:
      has.add(fn)
  return has


_SRC = {}


def source_text(at):
  """'chm_kernels.h:1486' -> the stripped text of that line (files of chimera_amd/csrc only; headers of the toolchain by name)"""
  name, _, line = at.partition(':')
  path = os.path.join(ROOT, 'chimera_amd', 'csrc', name)
  if not os.path.exists(path):
    return f'<{name}>'
  if name not in _SRC:
    _SRC[name] = open(path).read().split('\n')
  n = int(line or 0)
  return ' '.join(_SRC[name][n - 1].split())[:110] if 0 < n <= len(_SRC[name]) else f'<{name}: no line>'


def audit_keys(asm):
  """{kernel family: sorted list of (kind, text of the first access's line, text of the second's)} over the wave-private kernels"""
  res, priv = scan(asm), wave_private_kernels(asm)
  names = demangle(list(res))
  out = collections.OrderedDict()
  for fn, tr in res.items():
    if fn not in priv:
      continue
    fam = names[fn].split('<')[0]
    out.setdefault(fam, set()).update((t['kind'], source_text(t['first_at']), source_text(t['second_at'])) for t in tr)
  return collections.OrderedDict((k, sorted(v)) for k, v in out.items())


def demangle(names):
  try:
    d = subprocess.run(['c++filt'] + list(names), stdout=subprocess.PIPE, text=True).stdout.strip().split('\n')
    return {n: x.split('(')[0].replace('void ', '') for n, x in zip(names, d)}
  except OSError:
    return {n: n for n in names}


def main():
  argv = sys.argv[1:]
  asm = device_asm(argv[argv.index('--asm') + 1] if '--asm' in argv else None)
  res = scan(asm)
  names = demangle(list(res))
  summary = collections.OrderedDict()
  for fn, tr in res.items():
    k = names[fn]
    e = summary.setdefault(k, {"unmarked": 0, "where": collections.Counter()})
    e["unmarked"] += len(tr)
    for t in tr:
      e["where"][f"{t['kind']} {t['first_at']} -> {t['second_at']}"] += 1
  for k, e in summary.items():
    if e["unmarked"] or '-v' in argv:
      print(f"{k}: {e['unmarked']} unmarked LDS store<->load transition(s)")
      for w, c in sorted(e["where"].items()):
        print(f"    {c} x {w}")
  if '--audit' in argv:                                       # the keys tests/golden/lds_handover_allow.json is checked against
    for fam, keys in audit_keys(asm).items():
      print(fam)
      for k in keys:
        print('   ', json.dumps(list(k)))
  if '--write-allow' in argv:                                 # (re)generate the allow-list; every entry's reason is then reviewed by hand
    path = argv[argv.index('--write-allow') + 1]
    try:
      known = {(e['kernel'], e['kind'], e['first'], e['second']): e['why'] for e in json.load(open(path))['allowed']}
    except (OSError, ValueError, KeyError):
      known = {}
    allowed = []
    for fam, keys in audit_keys(asm).items():
      for kind, a, b in keys:
        why = known.get((fam, kind, a, b)) or ('REVIEW: same source line' if a == b else 'REVIEW: different lines')
        allowed.append({"kernel": fam, "kind": kind, "first": a, "second": b, "why": why})
    with open(path, 'w') as f:
      json.dump({"note": "unmarked LDS store<->load successions in the kernels that hand data over inside a wave (scripts/lds_handover_scan.py), each reviewed: "
                         "see `why`.  tests/test_abi_and_host.py fails on a succession that is not listed here.", "allowed": allowed}, f, indent=1)
    print('wrote', path, len(allowed), 'entries;', sum(e['why'].startswith('REVIEW') for e in allowed), 'to review')
  if '--json' in argv:
    with open(argv[argv.index('--json') + 1], 'w') as f:
      json.dump({k: {"unmarked": e["unmarked"], "where": dict(e["where"])} for k, e in summary.items()}, f, indent=1, sort_keys=True)
  return summary


if __name__ == '__main__':
  main()
