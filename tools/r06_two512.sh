#!/bin/bash
# Round 6: the 512-thread instantiation of k_two_run (256 VGPRs per lane, nothing spills) against the 1024-thread one: parity tests with the
# workgroup size forced, then wall clock of the two-layer BASELINE configs both ways on one box.
O=gpurun_out/r06_two512; mkdir -p $O
export TMPDIR=/tmp
for nt in 512 1024; do
(SNN_TWO_NT=$nt timeout 1200 python -m pytest tests/test_gpu_twolayer.py tests/test_gpu_rules.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize.py -m gpu -x -q --no-header -k "not dc and not conv" 2>&1 | tail -5) > $O/tests_nt$nt.log; echo "forced $nt:"; tail -2 $O/tests_nt$nt.log
done
for rep in 1 2; do
for nt in 1024 512; do
  SNN_TWO_NT=$nt timeout 600 python tools/bench_configs.py --runs 5 --only cfg3_shard,cfg3_b32,cfg5,f_hebbian --no-cpu-baseline > $O/bench_nt${nt}_$rep.jsonl 2> $O/bench_nt${nt}_$rep.err
  python - $O/bench_nt${nt}_$rep.jsonl $nt <<'P'
import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if not l.startswith("{"): continue
    d=json.loads(l)
    cfg = d["config"] if isinstance(d.get("config"), str) else d["config"]["workload"][:12]
    print("nt", sys.argv[2], cfg, d.get("value", d.get("timesteps_per_s")), "timesteps/s, us per timestep", round(1e3 * d.get("ms_per_timestep", 0), 3))
P
done
done
