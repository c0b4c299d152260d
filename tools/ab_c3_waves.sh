#!/bin/bash
# configs[3]-shaped throughput (4,096 x 2 MiB mixed chunks) against the number of compressor waves per CU
# (SQ_STAGE2_WAVES_PER_CU): shows that the compressible path scales with resident waves (experiment tooling).
set -e
for w in 8 6 4; do
  SQ_STAGE2_WAVES_PER_CU=$w timeout -k 10 200 python tools/prof_mixed.py 4096 128 mixed 2>/dev/null | tail -1 | sed "s/^/waves per CU $w: /"
done
