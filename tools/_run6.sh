mkdir -p gpurun_out/r3h && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3h/pytest.log; tail -4 gpurun_out/r3h/pytest.log
python tools/single_trees.py > gpurun_out/r3h/single_trees.txt 2>&1; cat gpurun_out/r3h/single_trees.txt
# bench.py --gpus 2 rehearsal on ONE GPU: gloo backend, both ranks on device 0, the sharded leg through IPC-mapped slabs
TREEQP_BENCH_SHARD_ANYWAY=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --no-cpu-baseline > gpurun_out/r3h/bench_gloo2.json 2> gpurun_out/r3h/bench_gloo2.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r3h/bench_gloo2.json; tail -5 gpurun_out/r3h/bench_gloo2.err
