#!/bin/bash
# Sweep of the batch kernels' chunk / refill settings on one box (bench.py headline, 10 M-triangle leg, any-hit).
for s in "PTMI_VOTE_CHUNK=128" "PTMI_VOTE_CHUNK=256" "PTMI_VOTE_CHUNK=512" "PTMI_VOTE_CHUNK=512 PTMI_VOTE_REFILL_MIN=32" "PTMI_VOTE_CHUNK=256 PTMI_VOTE_REFILL_MIN=24" "PTMI_VOTE_CHUNK=128"; do
  echo -n "$s: "; env $s timeout -k 10 200 python3 bench.py --no-render --no-cpu-baseline --no-traffic --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Mrays/s; kernel ms', d['roofline']['kernel_ms'], '; 10 M leg ms', d['roofline_hbm_resident']['kernel_ms'], '; any-hit', d.get('any_hit',{}).get('value'))" || exit 1
done
