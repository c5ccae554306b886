#!/bin/bash
# Kernel-level rocprofv3 summaries of the BASELINE configs besides C2 (C3, C4, C5 slice, C5 whole job) (run through gpurun from the repo root):
#   tools/profile_configs.sh r02_v4      -> gpurun_out/<tag>_gqa_kernel_stats.txt
# One --kernel-trace --stats pass and one --pmc FETCH_SIZE pass per config (counter passes carry no trace domains).
set -u
TAG=${1:-r02_vX}
REPO=$(pwd)
OUT=$REPO/gpurun_out/${TAG}_gqa_kernel_stats.txt
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
: > $OUT
for C in "C3 (" "C4 two" "C5 TP" "C5 whole"; do
  CMD="python $REPO/tools/bench_configs.py --only \"$C\" --no-baseline --iters 20"
  rm -rf /tmp/pc_s /tmp/pc_f
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pc_s -o run -- python $REPO/tools/bench_configs.py --only "$C" --no-baseline --iters 20 > /tmp/pc_s.log 2>&1
  echo "# rocprofv3 --kernel-trace --stats -- $CMD   (3 warm-ups + 20 calls in each of: back to back / write-flushed / cold)" >> $OUT
  grep "^| C" /tmp/pc_s.log >> $OUT
  DB=$(find /tmp/pc_s -name '*.db' | head -1)
  python $REPO/tools/rocprof_summary.py stats $DB | grep -E "kernel  |hyd::" >> $OUT
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pc_f -o run -- python $REPO/tools/bench_configs.py --only "$C" --no-baseline --iters 20 > /tmp/pc_f.log 2>&1
  DB=$(find /tmp/pc_f -name '*.db' | head -1)
  echo "# rocprofv3 --pmc FETCH_SIZE (KB per launch; x2 on gfx950 per MI355X_MICROARCH.md)" >> $OUT
  python $REPO/tools/rocprof_summary.py pmc $DB _attn >> $OUT
  rm -rf /tmp/pc_w
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pc_w -o run -- python $REPO/tools/bench_configs.py --only "$C" --no-baseline --iters 20 > /tmp/pc_w.log 2>&1
  DB=$(find /tmp/pc_w -name '*.db' | head -1)
  echo "# rocprofv3 --pmc WRITE_SIZE (KB per launch)" >> $OUT
  python $REPO/tools/rocprof_summary.py pmc $DB _attn >> $OUT
  echo >> $OUT
done
cat $OUT
