#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02_v1 [steps] [warmup]
# writes gpurun_out/<tag>_{kernel_stats,pmc}.txt, gpurun_out/<tag>_bench.json and gpurun_out/traffic_latest.json;
# copy them into profiles/ afterwards.  Counter passes are separate runs with --pmc only (no trace domains).
# The profiled command is the driver's bench command (same --steps/--warmup, hence the same suffix schedule) with the
# legs outside the timed region switched off.
set -u
TAG=${1:-r02_vX}
STEPS=${2:-20}
WARM=${3:-5}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps $STEPS --warmup $WARM --trials 0 --no-cpu-baseline --no-protocol --no-model --no-accuracy --no-paper-sweep --no-live-traffic"
cd /tmp
rm -rf /tmp/prof_*
rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o run -- $CMD > $OUT/${TAG}_bench_profiled.json 2>/tmp/prof_stats.log
DB=$(find /tmp/prof_stats -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- $CMD"; echo "# (averages are over every launch of the run: 3 eager warm-ups per captured suffix length, $WARM warm-up steps, $STEPS timed steps -- bench.launch_schedule; mean suffix $(python -c "import sys; sys.path.insert(0, \"$REPO\"); import bench; l = bench.launch_schedule($STEPS, $WARM, 0, 128); print(round(sum(l) / len(l), 2), len(l))") = mean, launches)"; python $REPO/tools/rocprof_summary.py stats $DB; } > $OUT/${TAG}_kernel_stats.txt
{
echo "# rocprofv3 --pmc passes (separate runs of: $CMD). FETCH_SIZE/WRITE_SIZE in KB per launch; gfx950 FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM)"
for C in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  D=/tmp/prof_pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C -d $D -o run -- $CMD --kv-candidates 1 > /dev/null 2>$D.log   # (bytes per launch do not depend on where the cache sits)
  DB=$(find $D -name '*.db' | head -1)
  if [ -n "$DB" ]; then python $REPO/tools/rocprof_summary.py pmc $DB _attn; else echo "# pass '$C' produced no db: $(tail -2 $D.log | tr '\n' ' ')"; fi
done
} > $OUT/${TAG}_pmc.txt
python $REPO/tools/make_traffic.py $OUT/${TAG}_pmc.txt $TAG $STEPS $WARM > $OUT/traffic_latest.json
cd $REPO
cp $OUT/traffic_latest.json $REPO/profiles/traffic_latest.json
python bench.py --steps $STEPS --warmup $WARM --detail-out $OUT/${TAG}_bench_detail.json > $OUT/${TAG}_bench.json 2>$OUT/${TAG}_bench.err
tail -c 2500 $OUT/${TAG}_bench.json
