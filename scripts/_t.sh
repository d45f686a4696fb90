python3 bench.py > gpurun_out/final_bench_n1.json 2>/dev/null
python3 bench.py --mode 5 --no-cpu-baseline > gpurun_out/final_bench_3p_mode5.json 2>/dev/null
python3 bench.py --games 524288 --steps 500 --warmup 300 --no-cpu-baseline --no-extras > gpurun_out/final_bench_524288.json 2>/dev/null
python3 bench.py --games 4096 --mode 0 --no-cpu-baseline --no-extras > gpurun_out/final_bench_4096_mode0.json 2>/dev/null
for f in n1 3p_mode5 524288 4096_mode0; do python3 -c "
import json
d=json.loads(open('gpurun_out/final_bench_$f.json').readline()); print('$f', round(d['value']/1e6,1), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['traffic'])"; done
