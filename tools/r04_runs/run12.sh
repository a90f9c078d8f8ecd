set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/final3; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py --no-parity --steps 60 > $O/bench_s3_b4.json 2> $O/bench_s3_b4.err
timeout 300 python bench.py --no-parity --steps 60 --streams 4 > $O/bench_s4_b4.json 2> $O/bench_s4_b4.err
timeout 300 python bench.py --no-parity --steps 60 --streams 2 --pairs-per-step 6 > $O/bench_s2_b6.json 2> $O/bench_s2_b6.err
timeout 300 python bench.py --no-parity --steps 60 --streams 3 --pairs-per-step 6 > $O/bench_s3_b6.json 2> $O/bench_s3_b6.err
timeout 300 python bench.py --no-parity --steps 60 > $O/bench_s3_b4_again.json 2> $O/bench_s3_b4_again.err
for f in s3_b4 s4_b4 s2_b6 s3_b6 s3_b4_again; do python -c "
import json;d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]);print('$f',round(d['value'],1),round(d['ms_per_step'],2))"; done
