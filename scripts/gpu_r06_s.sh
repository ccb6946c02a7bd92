#!/bin/bash
# round 6, gpurun call S: the shader clock during production steps (C3, 128 draws) and during the loop-free body probes
OUT=gpurun_out/r06s; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_clock.so
timeout -k 10 300 python3 scripts/clock_under_load.py --events 1000 --seconds 4 > $OUT/clock_production.json 2> $OUT/clock.err || { tail -20 $OUT/clock.err; exit 1; }
cat $OUT/clock_production.json
timeout -k 10 300 python3 scripts/clock_under_load.py --events 4 --draws 16 --inj 4000 --seconds 3 > $OUT/clock_probes.json 2>> $OUT/clock.err || { tail -20 $OUT/clock.err; exit 1; }
cat $OUT/clock_probes.json
