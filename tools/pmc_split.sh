# SQ counters of the fp16x3 conv kernels on one layer (default conv3_2 at 512^2; usage: pmc_split.sh [layer] [size]): LDS conflicts / activity, MFMA busy
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=${1:-conv3_2}
SZ=${2:-512}
rm -rf $R/gpurun_out/pmcS; mkdir -p $R/gpurun_out/pmcS
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmcS -o a --output-format csv -- python $R/tools/conv_bench.py $SZ 4 $L > $R/gpurun_out/pmcS/log 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$R/gpurun_out/pmcS/a_counter_collection.csv')))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    if 'conv_split' in k or 'conv_pc' in k:
        print(k)
        for c, x in sorted(v.items()): print(f'   {c:28s} {x:16.0f}')
PY
