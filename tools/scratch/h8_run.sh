cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for m in 1 1 0; do echo "mode $m: $(ADAS_HALO8=$m timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages']['lane_net_ms'], d['roofline']['kernel'], d['roofline']['achieved'])")"; done
