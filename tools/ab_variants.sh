#!/bin/bash
# Per-kernel totals of the C3 frame for several builds of the library.  usage: tools/ab_variants.sh <outdir> <variant names...>
# ("base" = the in-tree pathtrace_amd/libptmi.so; others are tools/bin/libptmi_<name>.so from tools/build_variant.sh)
OUT=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset PTMI_LIB; else export PTMI_LIB="$GRAFT_REPO_ROOT/tools/bin/libptmi_$v.so"; fi
  echo "== $v"
  bash tools/kt_render.sh "$OUT/$v" --scene ${SCENE:-c3} --spp ${SPP:-128} ${RARGS:-} | head -4 | cut -c1-40,96-
  grep Msamples "$OUT/$v/kt.log" | tail -1 | cut -c1-60
  rm -rf "$OUT/$v/kt"
done
