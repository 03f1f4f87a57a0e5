# Evidence set for one BASELINE workload on the GPU box: bench line, rocprofv3 kernel statistics, PMC passes (traffic + SQ / MFMA counters).
# usage (through gpurun): bash tools/evidence.sh C4 r03_v1   ->  gpurun_out/r03_v1/C4_{bench.json,kernel_stats.csv,pmc.txt}
# full evidence run for one workload: tests (optional), bench line, kernel stats, PMC traffic
W=${1:-C4}; TAG=${2:-r03_v1}
mkdir -p gpurun_out/$TAG && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --workload $W --steps 100 --warmup 10 > gpurun_out/$TAG/${W}_bench.json 2> gpurun_out/$TAG/${W}_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/prof_$W -o k -- python3 bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --no-batched > gpurun_out/$TAG/prof_$W.log 2>&1
python tools/kstats.py gpurun_out/$TAG/prof_$W/k_results.db > gpurun_out/$TAG/${W}_kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/$TAG/pmc_${W}_$i -o p -- python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-batched > gpurun_out/$TAG/pmc_${W}_$i.log 2>&1 || echo "pmc pass $i failed"
done
python tools/pmc_aggregate.py gpurun_out/$TAG --prefix pmc_${W}_ | grep -v "rocclr" > gpurun_out/$TAG/${W}_pmc.txt
head -12 gpurun_out/$TAG/${W}_kernel_stats.csv
grep -E "FETCH_SIZE|WRITE_SIZE" gpurun_out/$TAG/${W}_pmc.txt
python -c "
import json;d=json.load(open('gpurun_out/$TAG/${W}_bench.json'));print(d['value'],d['ms_per_step'],d['config']['kernel_launches_per_solve'],d['roofline'])"
