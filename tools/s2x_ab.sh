# same-box A/B of the split-precision stride-2 kernel (conv_s2p_x3_kernel): per-layer tables and the exact-mode step
mkdir -p gpurun_out/s2x
for v in 0 1; do
  export ADAS_NO_HALO_S2P_X3=$v
  python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16x3 --top 100 > gpurun_out/s2x/layers_ufld_x3_off$v.txt 2>&1
  python tools/profile_layers.py yolov8n --batch 64 --precision fp16x3 --top 100 > gpurun_out/s2x/layers_y8n_x3_off$v.txt 2>&1
  echo "== ADAS_NO_HALO_S2P_X3=$v"; head -1 gpurun_out/s2x/layers_ufld_x3_off$v.txt; grep k3s2 gpurun_out/s2x/layers_ufld_x3_off$v.txt | cut -c1-150
  head -1 gpurun_out/s2x/layers_y8n_x3_off$v.txt; grep k3s2 gpurun_out/s2x/layers_y8n_x3_off$v.txt | cut -c1-150
  ADAS_BENCH_NO_PMC=1 timeout 200 python bench.py --precision fp16x3 --no-extras --no-cpu-baseline --steps 20 --repeats 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bench fp16x3:', d['value'], 'fps', d['ms_per_step'], 'ms/step', d['config'].get('stages'), d.get('repeats'))
"
done
