#!/bin/bash
# One gpurun call: kernel statistics + PMC passes (separate runs, --kernel-trace only) of the C2 bench, written under gpurun_out/<tag>/.
# usage (on the GPU box): bash tools/profile_round.sh r02_v1 [workload]
set -e
tag=${1:-r02}
wl=${2:-C2}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-batched"
SHORT="python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-batched"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $BENCH > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${wl}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- $SHORT > $out/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $out/pmc_sq1 -- $SHORT > $out/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc_sq2 -- $SHORT > $out/pmc_sq2.log 2>&1 || true
python3 tools/pmc_aggregate.py $out > $out/${wl}_pmc_summary.txt 2>&1 || true
$BENCH > $out/${wl}_bench_quick.json 2>/dev/null || true
ls $out
