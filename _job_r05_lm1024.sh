#!/bin/bash
R=/root/repo
O=$R/gpurun_out/lm1024
mkdir -p $O
cd $R
for rep in 1 2; do
for v in base lm1024; do
  if [ $v = base ]; then unset POSELIB_AMD_LIB; else export POSELIB_AMD_LIB=$R/scripts/exp/variants/$v/libposelib_amd.so; fi
  timeout 600 python bench.py --no-parity --no-cpu-baseline --steps 10 --warmup 3 --detail-file $O/detail_${v}_$rep.json > $O/line_${v}_$rep.json 2> $O/err_${v}_$rep.log
  python - <<PY
import json
d=json.load(open('$O/line_${v}_$rep.json')); c=d['config']
print('$v', '$rep', 'p3p %.3e'%d['value'], 'rel %.3e'%c.get('relpose_5000_hyp_per_s',0), 'fund %.3e'%c.get('fund_10000_hyp_per_s',0), 'hom %.3e'%c.get('hom_10000_hyp_per_s',0), 'batch %.0f'%c.get('batch_mixed_problems_per_s',0), 'b512 %.0f'%c.get('batch_mixed_512_problems_per_s',0))
PY
done
done
