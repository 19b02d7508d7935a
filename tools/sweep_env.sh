#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 ..." [bench_render args]   -- one timed frame per value
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  echo "== $VAR=$v"; env $VAR=$v timeout -k 10 200 python3 tools/bench_render.py "$@" 2>&1 | grep -E "Msamples|profile" || exit 1
done
