for nb in 16; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --nbatch $nb 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nb=$nb', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['stage_ms'].items()}, d['last_log_hyper'])"
done
