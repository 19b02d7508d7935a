#!/bin/bash
# usage: tools/sweep2.sh "ENV1=a ENV2=b" "ENV1=c" ... -- bench_render args   (one timed frame per setting)
SETS=(); while [ "$1" != "--" ]; do SETS+=("$1"); shift; done; shift
for s in "${SETS[@]}"; do
  echo "== $s"; env $s timeout -k 10 200 python3 tools/bench_render.py "$@" 2>&1 | grep -E "Msamples" | cut -c1-64 || exit 1
done
