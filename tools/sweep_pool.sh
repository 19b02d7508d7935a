#!/bin/bash
# pool size / chunk size sweep on C3 at 256 spp.  usage: tools/sweep_pool.sh
set -u
for cfg in "16777216 48 256" "16777216 48 1024" "33554432 96 256" "33554432 96 1024" "67108864 128 1024"; do
  set -- $cfg
  echo "=== paths $1, budget $2 GiB, chunk $3 Msamples"
  PTMI_CONNECT_BAIL=8 PTMI_POOL_PATHS=$1 PTMI_POOL_GIB=$2 PTMI_CHUNK_MSAMPLES=$3 timeout -k 10 200 python tools/bench_render.py --scene c3 --spp 256 2>&1 | tail -1 || exit 1
done
