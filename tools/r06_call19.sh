#!/bin/bash
O=gpurun_out/r06_c19; mkdir -p $O
export TMPDIR=/tmp
python tools/r06_conv2d_probe.py | tee $O/conv2d_probe.jsonl
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/r06_conv2d_probe.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/r06_conv2d_probe.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
acc=collections.defaultdict(list)
for path in glob.glob(R+'/gpurun_out/r06_c19/pmc*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        if 'k_conv2d' in row['Kernel_Name']:
            acc[(row['Counter_Name'], row.get('Grid_Size'))].append(float(row['Counter_Value']))
for k,v in sorted(acc.items()):
    print(k, round(sum(v)/len(v),1), len(v))
PY
