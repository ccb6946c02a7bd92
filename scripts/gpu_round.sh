#!/bin/bash
# One gpurun call: GPU tests, multi-rank rehearsal, profiles of the round.   scripts/gpu_round.sh r02 [tests|rehearse|profiles ...]
R=${1:-r02}; shift
STEPS=${*:-tests rehearse profiles}
OUT=gpurun_out/$R; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in $STEPS; do
  case $s in
    tests)
      timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q --durations=8 > $OUT/gpu_tests.log 2>&1; rc=$?
      tail -15 $OUT/gpu_tests.log
      [ $rc -ne 0 ] && exit $rc ;;
    rehearse)
      bash scripts/rehearse_multirank.sh > $OUT/rehearse.log 2>&1; rc=$?
      cat $OUT/rehearse.log
      [ $rc -ne 0 ] && exit $rc ;;
    profiles)
      timeout -k 10 1100 python3 scripts/collect_profiles.py $R > $OUT/profiles.log 2>&1; rc=$?
      tail -25 $OUT/profiles.log
      [ $rc -ne 0 ] && exit $rc ;;
  esac
done
exit 0
