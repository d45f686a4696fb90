export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktrace -- python3 bench.py --steps 600 --warmup 600 --no-cpu-baseline > gpurun_out/ktrace.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/ktrace/**/*kernel_trace.csv',recursive=True)[0]
d=[]
for r in csv.DictReader(open(f)):
    if 'rmj4::k_step' in r['Kernel_Name']:
        d.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
d=d[600:]
d.sort()
n=len(d)
print(n, 'min',d[0],'p10',d[n//10],'p50',d[n//2],'p90',d[9*n//10],'max',d[-1],'mean',sum(d)/n)
PY
rm -rf gpurun_out/ktrace
