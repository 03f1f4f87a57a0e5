mkdir -p gpurun_out/r3c && cd $GRAFT_REPO_ROOT
TREEQP_AMD_LIB=$GRAFT_REPO_ROOT/treeqp_amd/lib_var/dev/libtreeqp_amd.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py -m gpu -x -q -k "wide or c4 or shapes or regul or mixed_batch or levels or pruned" > gpurun_out/r3c/pytest.log 2>&1; tail -3 gpurun_out/r3c/pytest.log
for v in dev wps2; do
  TREEQP_AMD_LIB=$GRAFT_REPO_ROOT/treeqp_amd/lib_var/$v/libtreeqp_amd.so python bench.py --workload C4 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r3c/c4bench_$v.json 2> gpurun_out/r3c/c4bench_$v.err
  python -c "
import json;d=json.load(open('gpurun_out/r3c/c4bench_$v.json'));print('$v',d['value'],d['ms_per_step'],d['config']['kernel_launches_per_solve'],d['roofline']['launch_us'])"
done
for v in st2; do echo "== $v"; TREEQP_AMD_LIB=$GRAFT_REPO_ROOT/treeqp_amd/lib_var/$v/libtreeqp_amd.so python tools/wide3_stamps.py; done > gpurun_out/r3c/stamps.txt 2>&1
cat gpurun_out/r3c/stamps.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in dev wps2; do
  TREEQP_AMD_LIB=$GRAFT_REPO_ROOT/treeqp_amd/lib_var/$v/libtreeqp_amd.so rocprofv3 --kernel-trace --stats -d gpurun_out/r3c/prof_$v -o c4 -- python3 tools/prof_flat.py C4 30 > gpurun_out/r3c/prof_$v.log 2>&1
  python tools/kstats.py gpurun_out/r3c/prof_$v/c4_results.db | head -5
done
