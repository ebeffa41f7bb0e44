cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_h8
timeout 600 bash tools/pmc_conv.sh gpurun_out/pmc_h8/l4 --hw 10 50 --cin 512 --cout 512 --batch 64 --precision fp16 > gpurun_out/pmc_h8/pmc_h8_10x50x512_b64.txt 2>&1
timeout 600 bash tools/pmc_conv.sh gpurun_out/pmc_h8/l2 --hw 40 200 --cin 128 --cout 128 --batch 64 --precision fp16 > gpurun_out/pmc_h8/pmc_h8_40x200x128_b64.txt 2>&1
find gpurun_out/pmc_h8 -name '*kernel_trace.csv' -delete; find gpurun_out/pmc_h8 -name '*counter_collection.csv' -size +2M -delete
tail -30 gpurun_out/pmc_h8/pmc_h8_10x50x512_b64.txt
