set -e
O=gpurun_out/r3q; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -30 $O/bench_default.err; exit 1; }
cat $O/bench_default.err | grep "\[bench\]"
python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(json.dumps({k:v for k,v in d.items() if k not in ('roofline',)}, indent=1)); print(json.dumps({k:v for k,v in d['roofline'].items() if k!='kernels'}))"
