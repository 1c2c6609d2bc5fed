#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_beam_gpu.py -m gpu -x -q -k "equals_the_oracle" 2>&1 | grep -E "passed|failed|assert|rror" | grep -v szhip_decompress | tail -5 ) > gpurun_out/r6_check_tests.txt
cat gpurun_out/r6_check_tests.txt
out=gpurun_out/r6_var.txt; : > $out
run() { tag="$1"; shift; env TAG="$tag" "$@" python tools/gpu_r6_calls.py 2>&1 | grep -E "median" >> $out; }
run "default"
run "rounds=0" SZ_HIP_SEG_ROUNDS=0
run "tile12" SZ_HIP_SEG_TILE_KB=12
run "m default" FIELD=m
cat $out
bash tools/gpu_r6_pmc.sh k_col_encode colenc 2>&1 | grep -E "INSTS_VALU|INSTS_SALU|INSTS_LDS|WAVE_CYCLES|ACTIVE_INST_ANY|BANK_CONFLICT|IDX_ACTIVE"
R5_DEC=0 bash tools/gpu_r6_trace.sh s chk > /dev/null 2>&1
grep -E "k_col|k_hist" gpurun_out/r6_chk_timeline.txt | tail -6
