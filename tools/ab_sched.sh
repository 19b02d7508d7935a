#!/bin/bash
# Same-box A/B of scheduler-strategy builds (tools/build_variant.sh ilp|memclause -mllvm -amdgpu-sched-strategy=...) on C3, C5 and the headline
for v in base ilp memclause base; do
  if [ $v = base ]; then L=$GRAFT_REPO_ROOT/pathtrace_amd/libptmi.so; else L=$GRAFT_REPO_ROOT/tools/bin/libptmi_$v.so; fi
  echo "== $v"
  PTMI_LIB=$L timeout -k 10 200 python3 tools/bench_render.py --scene c3 --spp 256 2>&1 | grep Msamples | cut -c1-64 || exit 1
  PTMI_LIB=$L timeout -k 10 200 python3 tools/bench_render.py --scene c5 --width 3840 --height 2160 --spp 2 2>&1 | grep Msamples | cut -c1-64 || exit 1
  PTMI_LIB=$L timeout -k 10 300 python3 bench.py --no-render --no-cpu-baseline --no-traffic --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Mrays/s; 10 M leg ms', d['roofline_hbm_resident']['kernel_ms'], '; any-hit', d.get('any_hit',{}).get('value'))" || exit 1
done
