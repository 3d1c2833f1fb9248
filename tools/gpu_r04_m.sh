#!/bin/bash
# Round 4, GPU pass M: launch shapes of plain streaming reads and writes (tools/ubench_shapes.hip).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04m; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 tools/ubench_shapes.hip -o /tmp/ubench_shapes || exit 1
timeout -k 10 300 /tmp/ubench_shapes > $O/ubench_shapes.txt 2>&1 || exit 1
cat $O/ubench_shapes.txt | cut -c1-140
