cd $GRAFT_REPO_ROOT
for m in 0 1 2 0 1 2; do echo "prio mode $m: $(ADAS_STREAM_PRIO=$m timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"; done
