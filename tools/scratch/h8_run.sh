cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q 2>&1 | tail -3
for m in 1 1; do echo "run: $(timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages'])")"; done
