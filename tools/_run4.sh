mkdir -p gpurun_out/r3f && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3f/pytest.log; tail -4 gpurun_out/r3f/pytest.log
python bench.py --workload C4 --steps 100 --warmup 10 > gpurun_out/r3f/C4_bench.json 2> gpurun_out/r3f/C4_bench.err; tail -c 400 gpurun_out/r3f/C4_bench.err
python -c "
import json;d=json.load(open('gpurun_out/r3f/C4_bench.json'));print(d['value'],d['ms_per_step'],d['config']['kernel_launches_per_solve'],d['roofline']['launch_us'],d['roofline']['frac'])"
