#!/bin/bash
# Development: interleaved A/B of a -D switch:  tools/macro_ab.sh MACRO <command...>
# builds _build/libsmmc_hip_MACRO{0,1}.so once (the product library is not touched) and runs the
# command twice against each, alternating.
M=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L0=$($R/tools/variant_build.sh ${M}0 -D$M=0 | tail -1) || exit 1
L1=$($R/tools/variant_build.sh ${M}1 -D$M=1 | tail -1) || exit 1
for X in 0 1 0 1; do
  echo "== $M=$X"
  L=$L0; [ $X = 1 ] && L=$L1
  SMMC_LIB=$L "$@" 2>&1 | grep -v "amdgpu.ids\|DEVELOPMENT library" || exit 1
done
