#!/bin/bash
# usage: tools/ab_env_tiles.sh VAR "values": tools/bench_tiles.py per value
set -u
for v in $2; do echo "== $1=$v"; env $1=$v timeout -k 10 200 python tools/bench_tiles.py --spp 64 2>&1 | grep -E "whole|per tile" || exit 1; done
