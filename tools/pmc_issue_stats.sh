cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcx
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcx/$tag -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
acc=collections.defaultdict(list)
for path in glob.glob(R+'/gpurun_out/pmcx/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        if 'k_dc2015_async' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in sorted(acc.items()):
    print(k, sum(v)/len(v), len(v))
PY
