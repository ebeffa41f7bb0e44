cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py tests/test_gpu_frontend.py -q -x 2>&1 | tail -3
for s in 1 0 1 0; do echo "no_side=$s: $(ADAS_NO_SIDE=$s timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages'])")"; done
