#!/bin/bash
# round 6: the final build's GPU suite, smoke(), profile artefacts (kernel stats + PMC of the S-field bench and of the M-field calls), the default bench line, the c4 line, time lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|assert|rror" | grep -v "szhip_decompress\|Error: " | tail -12 ) > gpurun_out/r6_final_tests.txt
( timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 ) >> gpurun_out/r6_final_tests.txt
cat gpurun_out/r6_final_tests.txt
bash tools/gpu_profile_round.sh r6s python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 20 --warmup 5 > gpurun_out/r6_final_prof_s.txt 2>&1
bash tools/gpu_profile_round.sh r6m python $GRAFT_REPO_ROOT/tools/gpu_r5_mtime.py 512 m > gpurun_out/r6_final_prof_m.txt 2>&1
tail -8 gpurun_out/r6_final_prof_s.txt | cut -c1-200
( timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r6_final_bench_err.txt | tail -1 ) > gpurun_out/r6_final_bench.json
cut -c1-600 gpurun_out/r6_final_bench.json
( timeout 600 python bench.py --config c4 --steps 10 --warmup 3 2>/dev/null | tail -1 ) > gpurun_out/r6_final_bench_c4.json
cut -c1-400 gpurun_out/r6_final_bench_c4.json
python tools/gpu_r5_mtime.py 512 s,m 2>&1 | grep field > gpurun_out/r6_final_timelines.txt
R5_DEC=0 bash tools/gpu_r6_trace.sh s fin > /dev/null 2>&1
cat gpurun_out/r6_final_timelines.txt
