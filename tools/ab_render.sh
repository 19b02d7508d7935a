#!/bin/bash
# Same-box A/B of tools/bin/libptmi_prev.so against the working tree's library on render scenes.  usage: tools/ab_render.sh "<bench_render args>" ...
for sc in "$@"; do
  echo "### $sc"
  for v in prev new prev new; do
    if [ $v = new ]; then L=$GRAFT_REPO_ROOT/pathtrace_amd/libptmi.so; else L=$GRAFT_REPO_ROOT/tools/bin/libptmi_prev.so; fi
    echo -n "$v: "; PTMI_LIB=$L timeout -k 10 300 python3 tools/bench_render.py $sc 2>&1 | grep Msamples | cut -c1-64 || exit 1
  done
done
