#!/bin/bash
# first GPU call of round 3: new parity tests + whole gpu suite + default bench
out=$GRAFT_REPO_ROOT/gpurun_out/r03a
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_configs.py -m gpu -x -q -s > $out/pytest_chain.log 2>&1; echo "exit $?" >> $out/pytest_chain.log )
tail -30 $out/pytest_chain.log
( timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_chain.py --deselect tests/test_gpu_configs.py > $out/pytest_rest.log 2>&1; echo "exit $?" >> $out/pytest_rest.log )
tail -8 $out/pytest_rest.log
( timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/bench.err )
tail -3 $out/bench.err
cat $out/bench.json
