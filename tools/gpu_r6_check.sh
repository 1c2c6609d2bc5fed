#!/bin/bash
# round 6: parity tests of the sweep files, then the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_beam_gpu.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r6_check_tests.txt
( timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -3 ) > gpurun_out/r6_check_bench.txt
cat gpurun_out/r6_check_tests.txt; cat gpurun_out/r6_check_bench.txt | cut -c1-3000
