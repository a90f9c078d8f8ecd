#!/bin/bash
# round 6, run 4: (1) packed-f32 accumulator chains in a FEW long-running workgroups next to the pipeline competitor
# (tools/microbench/handoff arith); (2) the cluster registration kernel re-applied on HEAD and built without the SLP
# vectoriser: 20 000 runs next to the competitor; (3) the registration-related GPU tests with the f64 arbiter;
# (4) the driver's bench command.
R=$PWD; O=$R/gpurun_out/run4; mkdir -p $O; rm -rf $O/*
python -c "import torch" 2>/dev/null
H=$R/tools/microbench/handoff
timeout 500 python tools/repro_stress.py 100000 12000 > $O/comp.txt 2>&1 &
CP=$!
sleep 20
{
echo "## next to the pipeline competitor"
timeout 100 $H arith 1500 4 3000
timeout 100 $H arith 1500 32 1000
echo "== cluster kernel on HEAD, reg.hip without the SLP vectoriser: $(timeout 300 python tools/repro_reg.py 20000 2>&1 | tail -1)"
} 2>&1 | tee $O/contention.txt
kill $CP 2>/dev/null; wait $CP 2>/dev/null
echo "## alone" | tee -a $O/contention.txt
timeout 100 $H arith 500 4 3000 | tee -a $O/contention.txt
DGR_PARITY_REPORT=$O/parity timeout 1500 python -m pytest tests/test_gpu_knn_reg.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_bench_ranks.py tests/test_gpu_register_e2e.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
cat $O/parity/refine_parity.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -5 $O/bench_default.err; python - <<P
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'], indent=0)); print(json.dumps(d['parity'], indent=0)[:3000]); print(d.get('exact_f32_leg')); print(d['stage_ms_per_batch'])
P
