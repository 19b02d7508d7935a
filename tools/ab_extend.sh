#!/bin/bash
# A/B the extend kernel variants: parity tests + bench per variant.  usage: tools/ab_extend.sh "0 1"  (0 = one-shot kernels, 1 = vote-scheduled persistent kernels)
set -u
for v in $1; do
  echo "=== variant $v"
  PTMI_EXTEND_VARIANT=$v timeout -k 10 300 python -u -m pytest tests/test_traversal_gpu.py -x -q -m gpu --timeout 200 --timeout-method thread 2>&1 | tail -2 || exit 1
  PTMI_EXTEND_VARIANT=$v timeout -k 10 300 python -u bench.py --steps 5 --warmup 1 --no-render --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Mrays/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])" || exit 1
done
