#!/bin/bash
# sweep env-var knobs of the extend kernel: each argument is a space-separated "K=V K=V" set
for cfg in "$@"; do
  env $cfg timeout -k 10 300 python -u bench.py --steps 5 --warmup 1 --no-render --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg: Mrays/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'])" || exit 1
done
