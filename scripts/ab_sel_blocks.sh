cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for nbk in 1000000 32768 16384 8192 4096; do
CHM_SELF_BLOCKS=$nbk CHM_SERIAL=1 timeout -k 10 200 python3 bench.py --config C4 --no-cpu-baseline --no-single-call --steps 20 --warmup 3 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']
print('blocks=%-8s value=%.1f ms_per_step=%.4f sel=%.4f last=%r' % ('$nbk', j['value'], j['ms_per_step'], s['selection'], j['last_log_hyper']))" || exit 1
done; done
