cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/presets
timeout 600 python bench.py --preset c4 --no-cpu-baseline > gpurun_out/presets/c4.json 2> gpurun_out/presets/c4.err; tail -c 300 gpurun_out/presets/c4.err
timeout 600 python bench.py --preset c5 --no-cpu-baseline > gpurun_out/presets/c5.json 2> gpurun_out/presets/c5.err; tail -c 300 gpurun_out/presets/c5.err
