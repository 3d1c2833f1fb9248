#!/bin/bash
# ref_generic_kernel (the reference stream beyond 454 periods) against its workgroups per CU: 2e7 x 1000 paths
for pc in 1 2 4 8 12 16; do echo "SMMC_REF_GENERIC_BLOCKS_PER_CU=$pc"; SMMC_REF_GENERIC_BLOCKS_PER_CU=$pc python tools/bench_ref.py 1000000 20000000 2>&1 | grep "ref generic"; done
