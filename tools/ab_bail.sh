#!/bin/bash
set -u
for rep in 1 2 3; do for v in 0 8; do
  echo -n "bail $v: "; PTMI_CONNECT_BAIL=$v timeout -k 10 200 python tools/bench_render.py --scene c3 --spp 256 2>&1 | tail -1 | cut -c1-60 || exit 1
done; done
