mkdir -p gpurun_out/r05g
for p in c4 c5; do ( ADAS_BENCH_NO_PMC=1 timeout 500 python bench.py --preset $p --no-cpu-baseline --steps 20 --repeats 2 > gpurun_out/r05g/bench_$p.json 2> /dev/null ); python - gpurun_out/r05g/bench_$p.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d['config']
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['dtype'], c.get('parity_e2e_summary') or c.get('exact_mode_e2e'), c.get('frame_at_a_time'))
PY
done
