#!/bin/bash
# sweep persistent-extend knobs: "refill:wpc" pairs
for cfg in $1; do
  r=${cfg%%:*}; w=${cfg##*:}
  PTMI_EXTEND_VARIANT=2 PTMI_REFILL_BELOW=$r PTMI_PERSIST_WAVES_PER_CU=$w timeout -k 10 300 python -u bench.py --steps 5 --warmup 1 --no-render --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('refill $r wpc $w: Mrays/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'])" || exit 1
done
