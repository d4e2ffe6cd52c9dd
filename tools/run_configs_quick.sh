# one bench line per configuration (no CPU baseline, no second region), for before / after comparisons
for c in 0 1 2 3 4 5; do
  timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-fast-path --repeats 3 > gpurun_out/q_$c.json 2> gpurun_out/q_$c.err || exit 1
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('config', sys.argv[2], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'])
" gpurun_out/q_$c.json $c
done
