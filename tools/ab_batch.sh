#!/bin/bash
# Same-box A/B of tools/bin/libptmi_prev.so against the working tree's library on the headline batch kernels (bench.py
# without the render legs: C2 closest-hit, the 10 M-triangle leg, any-hit; every run checks all rays against the oracle).
for v in prev new prev new; do
  if [ $v = new ]; then L=$GRAFT_REPO_ROOT/pathtrace_amd/libptmi.so; else L=$GRAFT_REPO_ROOT/tools/bin/libptmi_prev.so; fi
  echo -n "$v: "; PTMI_LIB=$L timeout -k 10 300 python3 bench.py --no-render 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Mrays/s; kernel ms', d['roofline']['kernel_ms'], '; 10 M leg ms', d['roofline_hbm_resident']['kernel_ms'], '; any-hit', d.get('any_hit',{}).get('value'), '; mismatches', d['cpu_baseline'].get('gpu_mismatches_on_sample'))" || exit 1
done
