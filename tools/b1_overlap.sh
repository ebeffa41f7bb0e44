export ADAS_BENCH_NO_PMC=1
for P in north-star c5; do
for F in "" "--no-overlap" "--no-graph" "--no-graph --no-overlap"; do
  timeout 120 python bench.py --preset $P --precision fp16 --streams 1 --micro-batch 1 --steps 200 --warmup 20 --repeats 0 --latency-steps 8 --no-extras --no-cpu-baseline $F 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$P','[$F]', d['value'], 'fps', d['ms_per_step'], 'ms/step', d['config'].get('stages'))
"
done; done
