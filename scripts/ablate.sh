for d in 0; do
  CHM_SERIAL=1 CHM_DEBUG_SKIP=$d timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --nbatch 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$d', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['stage_ms'].items()})"
done
