#!/bin/bash
R=/root/repo
O=$R/gpurun_out/fg7
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
s=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
echo "bench wall $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/fg7/bench_line.json'))
c=d['config']
print(d['value'], d['roofline']['frac'])
for k,v in c.items():
    if 'pnpf' in k or 'shared_focal' in k or 'batch_mixed' in k and 'problems_per_s' in k: print(k, v)
PY
timeout 300 python scripts/focal_batch_bench.py 1024 2000 > $O/focal_batch.md 2>$O/focal_batch.err; cat $O/focal_batch.md
