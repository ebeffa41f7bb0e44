cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/b1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/b1.json')); r=d['roofline']; print(d['value'], r['kernel'], r['achieved'], r['frac'], r['gflop_per_launch'], r['avg_launch_us'], r['all_conv_frac'])"
