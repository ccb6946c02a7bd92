for m in full; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --nbatch 1 --events 200 --mode $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode=$m E=200 evals/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['stage_ms'].items()})"
done
