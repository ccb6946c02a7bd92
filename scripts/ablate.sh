for g in 1 2 4 8; do
  CHM_GROUPS=$g timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('groups=$g evals/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['stage_ms'].items()})"
done
