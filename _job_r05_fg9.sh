#!/bin/bash
R=/root/repo
O=$R/gpurun_out/fg9
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_gpu_focal.py tests/test_zz_gpu_shared_focal.py tests/test_zz_gpu_focal_group.py -x -q 2>&1 | tail -4
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/fg9/bench_line.json'))
c=d['config']
print(d['value'], d['roofline']['frac'])
for k,v in c.items():
    if ('pnpf' in k or 'shared_focal' in k) and ('per_s' in k or 'ms_per' in k or 'identical' in k) or k.startswith('batch_mixed') and 'problems_per_s' in k: print(k, v)
PY
cp $R/gpurun_out/bench_detail.json $O/ 2>/dev/null
timeout 300 python scripts/focal_batch_bench.py 1024 2000 > $O/focal_batch.md 2>$O/focal_batch.err; cat $O/focal_batch.md
