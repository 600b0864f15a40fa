#!/bin/bash
# Round 6: the 1 524-case soak of the D&C resident plan (k_dc2015_async where the lean form applies) against the generic plan and inside pipelined
# sections, bit for bit -- the same seeds as round 5's (profiles/r05_soak_resident_generic_sections_mi355x.log)
O=gpurun_out/r06_soak; mkdir -p $O
export TMPDIR=/tmp
for pair in "450 0" "200 450" "550 650" "324 1200"; do
  set -- $pair
  timeout 900 python tools/r04_soak.py $1 $2 sections 2>&1 | tail -3; echo "rc=$?"
done | tee $O/soak.log
grep "cases," $O/soak.log
