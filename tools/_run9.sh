cd $GRAFT_REPO_ROOT
export TREEQP_AMD_LIB=$GRAFT_REPO_ROOT/treeqp_amd/lib_var/dev/libtreeqp_amd.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py -m gpu -x -q -k "wide or c4 or shapes or regul or mixed_batch or levels or pruned or three_launch or iteration_limit or nan or warm" 2>&1 | tail -3
python tools/single_trees.py pruned
python bench.py --workload C4 --steps 100 --warmup 10 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4', d['value'], d['ms_per_step'], d['roofline']['launch_us'], d['config']['kernel_launches_per_solve'])"
