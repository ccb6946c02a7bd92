cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
for a in "--events 125 --inj 12500" "--events 250 --inj 25000" "--events 500 --inj 50000"; do
python3 bench.py --no-cpu-baseline --no-single-call --steps 30 --warmup 5 $a 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']
print('$a', 'ms/step %.4f' % j['ms_per_step'], {k: round(v,4) for k,v in s.items() if isinstance(v,float)})"
done
