for g in 1 0 1 0; do
  SSN_TAB_GATE=$g python bench.py --steps 20 --warmup 5 --slam-steps 0 --slam-cpu-steps 0 --cpu-steps 0 > gpurun_out/e2e_ab_$g.json 2> gpurun_out/e2e_ab.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/e2e_ab_$g.json').read().strip().splitlines()[-1]); print('gate $g', d['value'], d['value_end_to_end'], d['end_to_end']['value_plain_closures'], round(d['value_end_to_end']/d['value'],4), round(d['end_to_end']['value_plain_closures']/d['value'],4))"
done
