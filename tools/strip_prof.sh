# kernel trace of one rank's strip iterations (exchanges stubbed): what runs on the trunk's queue, where it waits
#   gpurun -- bash tools/strip_prof.sh [size, default 2896x2172] [ranks, 8] [rank, 7]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SZ=${1:-2896x2172}; N=${2:-8}; RK=${3:-7}
rm -rf $R/gpurun_out/stripprof; mkdir -p $R/gpurun_out/stripprof
STRIP_TRACE_RANK=$RK rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stripprof -o s --output-format csv -- python $R/tools/strip_bench.py $SZ $N > $R/gpurun_out/stripprof/log 2>&1
T=$(find $R/gpurun_out/stripprof -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_strip.py $T -3 > $R/gpurun_out/stripprof/timeline_rank$RK.txt 2>&1
python - $T <<'P' > $R/gpurun_out/stripprof/kernels_rank$RK.txt
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
starts = [i for i, r in enumerate(rows) if 'conv_first_fwd' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'conv_first_fwd' in r['Kernel_Name']]
it = rows[starts[-6]:starts[-1]]          # the last five complete iterations of the traced rank
c = collections.defaultdict(lambda: [0, 0.0])
for r in it:
    n = re.sub(r'^void ', '', r['Kernel_Name']).replace('st::', '').replace('(anonymous namespace)::', '').split('(')[0][:60]
    c[n][0] += 1; c[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
span = (max(int(r['End_Timestamp']) for r in it) - int(it[0]['Start_Timestamp'])) / 5e3
print(f'five iterations span {span * 5:.1f} us = {span:.1f} us per iteration (under the profiler)')
print('per iteration (5 iterations averaged): kernel | launches | us')
for n, (k, us) in sorted(c.items(), key=lambda kv: -kv[1][1]):
    print(f'{n:62s} {k / 5:6.1f} {us / 5:9.1f}')
P
find $R/gpurun_out/stripprof -name "*.csv" -size +8M -delete
