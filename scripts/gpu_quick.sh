#!/bin/bash
# one gpurun call: GPU test suite (optional: TESTS=0 skips it), then the default bench line -> gpurun_out/quick/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/quick; mkdir -p $O
if [ "${TESTS:-1}" != "0" ]; then timeout -k 10 900 python3 -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; [ $rc -ne 0 ] && exit $rc; fi
export CHIMERA_NO_REBUILD=1
timeout -k 10 300 python3 bench.py ${BENCH_ARGS} > $O/bench.json 2> $O/bench.err; rc=$?
python3 - <<'PY'
import json
try:
  j = json.loads(open('gpurun_out/quick/bench.json').read().strip().split('\n')[-1])
  print('value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'single', j.get('single_call_ms'), 'parity', (j.get('parity_full_size') or {}).get('abs_diff'))
  print({k: round(v, 4) if isinstance(v, float) else v for k, v in j['roofline']['stage_ms'].items()})
except Exception as e:
  print('bench line not parsed', e); print(open('gpurun_out/quick/bench.err').read()[-2000:])
PY
exit $rc
