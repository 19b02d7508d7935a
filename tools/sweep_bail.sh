#!/bin/bash
# connect straggler hand-over: parity tests at one setting, then C3 samples/s per setting.  usage: tools/sweep_bail.sh "0 4 8 12 16"
set -u
PTMI_CONNECT_BAIL=8 timeout -k 10 400 python -u -m pytest tests/test_render_gpu.py -x -q -m gpu --timeout 200 --timeout-method thread 2>&1 | tail -3 || exit 1
for v in $1; do
  echo "=== PTMI_CONNECT_BAIL=$v"
  PTMI_CONNECT_BAIL=$v timeout -k 10 200 python tools/bench_render.py --scene c3 --spp 128 2>&1 | tail -1 || exit 1
done
