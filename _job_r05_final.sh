#!/bin/bash
R=/root/repo
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
s=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
echo "bench wall $(( $(date +%s) - s )) s"
cp $R/gpurun_out/bench_detail.json $O/ 2>/dev/null
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/final/bench_line.json'))
c=d['config']
print(d['value'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'))
for k,v in c.items():
    if ('pnpf' in k or 'shared_focal' in k) and ('batch' in k or 'ms_per' in k) or k.startswith('batch_mixed') and 'problems_per_s' in k or k.endswith('_hyp_per_s') or 'parity' in k: print(k, v)
PY
python -c "import __graft_entry__ as g; g.smoke()"
