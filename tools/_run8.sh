cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03_v1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_v1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_v1/pytest.log; tail -3 gpurun_out/r03_v1/pytest.log
for W in C2 C1 C3 C5; do bash tools/_run5.sh $W r03_v1 2>&1 | tail -4; done
python bench.py > gpurun_out/r03_v1/default_bench.json 2> gpurun_out/r03_v1/default_bench.err; python -c "
import json;d=json.load(open('gpurun_out/r03_v1/default_bench.json'));print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('critical_path',{}).get('achieved_over_floor'), d.get('batched',{}).get('value'))"
