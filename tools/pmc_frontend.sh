#!/bin/bash
# Counter passes over the front-end / attention probe (separate rocprofv3 --pmc runs, one counter group each) plus one
# kernel-trace pass for the durations.   usage: tools/pmc_frontend.sh OUTDIR
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o vm -- python3 $R/tools/probes/frontend_probe.py > $OUT/kt.log 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o vm -- python3 $R/tools/probes/frontend_probe.py > $OUT/p$i.log 2>&1
done
