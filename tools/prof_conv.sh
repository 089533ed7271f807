# usage: bash tools/prof_conv.sh <precision> [layer]   -- rocprofv3 kernel stats of tools/conv_bench.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/convprof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/convprof -o c --output-format csv -- python $R/tools/conv_bench.py 512 $1 $2 > /dev/null 2>&1
python - <<PY
import csv,collections
d=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open('$R/gpurun_out/convprof/c_kernel_trace.csv')):
    k=r['Kernel_Name'][:90]; d[k][0]+=1; d[k][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for k,(n,t) in sorted(d.items(), key=lambda kv:-kv[1][1])[:12]:
    print(f'{t/n:9.1f} us avg x{n:4d}  {k}')
PY
