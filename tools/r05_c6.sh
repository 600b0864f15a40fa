#!/bin/bash
# round 5, call 6: no memsets / no input-raster copy in a section's steady state, bench pre-warm: full GPU suite, then timing
O=gpurun_out/r05_c6; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -25) > $O/tests.log; tail -4 $O/tests.log
for v in "k20 20 5" "k200 200 10" "k20b 20 5"; do set -- $v
  timeout 200 python bench.py --steps $2 --warmup $3 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json $1 <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'], d['config']['plan_retries(lean,resident)'], d['config']['device_prewarm']['runs'])
P
done
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prewarm-ms 0 > $O/bench_cold.json 2> $O/bench_cold.err
python - $O/bench_cold.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('no prewarm', d['value'], 'ms/step', d['ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'])
P
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OLDPWD/$O/prof_bench.json 2> $OLDPWD/$O/prof.err
cd $OLDPWD; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {} | cut -c1-160
