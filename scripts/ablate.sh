for nb in 16 32 64; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --nbatch $nb 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nb=$nb', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['stage_ms'].items()})"
done
for nb in 16 32; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --nbatch $nb --events 125 --inj 12500 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('E=125 nb=$nb', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['stage_ms'].items()})"
done
