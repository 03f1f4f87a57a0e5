mkdir -p gpurun_out/r3h && cd $GRAFT_REPO_ROOT
TREEQP_BENCH_SHARD_ANYWAY=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --no-cpu-baseline > gpurun_out/r3h/bench_gloo2.json 2> gpurun_out/r3h/bench_gloo2.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('gpurun_out/r3h/bench_gloo2.json').read().strip().splitlines()[-1]);print(d['value'], d['n_gpus'], d.get('sharded'), d.get('sharded_ok'))"; tail -3 gpurun_out/r3h/bench_gloo2.err
