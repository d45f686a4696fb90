export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3-avail list 2>/dev/null | grep -i "icache\|ifetch\|SQC_INST\|DCACHE" | head -20
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/ic_tmp -- python3 bench.py --steps 30 --warmup 200 --no-cpu-baseline > gpurun_out/ic.log 2>&1
  python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/ic_tmp/**/*counter_collection.csv',recursive=True):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'rmj4::k_step' in r['Kernel_Name']:
            per[(r['Counter_Name'],r['Dispatch_Id'])]+=float(r['Counter_Value'])
    for (n,_),v in per.items(): acc[n].append(v)
for n,v in acc.items(): print(n, sum(v[-20:])/len(v[-20:]))
PY
  rm -rf gpurun_out/ic_tmp
done
