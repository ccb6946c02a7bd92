for nb in 1 8; do
for ev in 1000 125; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --nbatch $nb --events $ev --inj $((ev*100)) 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('E=$ev nb=$nb evals/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['stage_ms'].items()})"
done; done
