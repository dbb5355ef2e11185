#!/bin/bash
# N decoder processes side by side on one GPU (each: bench.py --decode, its own host thread, its own HIP queues): aggregate frames/s
out=gpurun_out/r5y; mkdir -p $out
python bench.py --decode --steps 4 --warmup 1 > $out/one.json 2> $out/one.err
for n in 2 4 6; do
  pids=""
  for i in $(seq 1 $n); do python bench.py --decode --steps 6 --warmup 1 > $out/p${n}_$i.json 2> $out/p${n}_$i.err & pids="$pids $!"; done
  for p in $pids; do wait $p; done
done
python - <<'P'
import json,glob
o='gpurun_out/r5y/'
one=json.load(open(o+'one.json')); print('one process', round(one['value'],2), 'frames/s', round(one['host_cpu_ms_per_frame']), 'ms host cpu')
for n in (2,4,6):
    v=[json.load(open(f))['value'] for f in sorted(glob.glob(o+f'p{n}_*.json'))]
    print(n, 'processes: per process', [round(x,2) for x in v], 'sum', round(sum(v),2))
P
