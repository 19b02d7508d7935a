#!/bin/bash
# A/B an environment switch on the C3 frame.  usage: tools/ab_env.sh VAR "values" [reps] [bench_render args]
set -u
VAR=$1; VALS=$2; REPS=${3:-3}; shift 3 || shift $#
ARGS="${*:---scene c3 --spp 256}"
for rep in $(seq $REPS); do for v in $VALS; do
  echo -n "$VAR=$v: "; env $VAR=$v timeout -k 10 200 python tools/bench_render.py $ARGS 2>&1 | tail -1 | cut -c1-62 || exit 1
done; done
