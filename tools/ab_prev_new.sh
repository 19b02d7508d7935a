#!/bin/bash
# Same-box A/B of the previous commit's library (tools/bin/libptmi_prev.so: git stash; make; cp; git stash pop; make)
# against the working tree's: headline batch kernels, C3 and the C4-class frame, twice each.  Boxes differ by ~10 %.
for v in prev new prev new; do
  if [ $v = new ]; then L=$GRAFT_REPO_ROOT/pathtrace_amd/libptmi.so; else L=$GRAFT_REPO_ROOT/tools/bin/libptmi_prev.so; fi
  echo "== $v"; PTMI_LIB=$L timeout -k 10 300 python3 bench.py --no-render --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'], d['roofline_hbm_resident']['kernel_ms'], d.get('any_hit',{}).get('value'), d['cpu_baseline'] if 'cpu_baseline' in d else '')" || exit 1
  PTMI_LIB=$L timeout -k 10 300 python3 tools/bench_render.py --scene c3 --spp 256 2>&1 | grep Msamples | cut -c1-64
  PTMI_LIB=$L timeout -k 10 300 python3 tools/bench_render.py --scene c4 --width 3840 --height 2160 --spp 32 2>&1 | grep Msamples | cut -c1-64
done
