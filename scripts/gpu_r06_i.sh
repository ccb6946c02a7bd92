#!/bin/bash
# round 6: profile set part 2 (full mode, C4, C5) + the one-draw collection again (per-kernel times restored for the --no-graph run)
export CHIMERA_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_r06; mkdir -p $O
python3 scripts/collect_profiles.py r06 --tag nbatch1 --skip-trace -- --nbatch 1 --no-graph --steps 200 --warmup 20 > $O/collect_nb1.log 2>&1; tail -4 $O/collect_nb1.log | cut -c1-600
bash scripts/gpu_profiles_r06b.sh
