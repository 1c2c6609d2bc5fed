#!/bin/bash
# round 5, final build: the GPU suite, the default bench line and the C4 line, the round's profile of the timed region (kernel stats + PMC passes), the same for one
# M-field call, and one call's time line both ways for the S and the M field
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>/dev/null | grep -E "passed in|failed in| passed,| failed,|^FAILED|^ERROR" | tail -8 > gpurun_out/r5_final_gpu_tests.txt
python bench.py 2> /dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r5_final_bench_default.json
python bench.py --config c4 2> /dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r5_final_bench_c4.json
bash tools/gpu_profile_round.sh r5f > gpurun_out/r5_final_profile_round.txt 2>&1
R5_DEC=0 bash tools/gpu_profile_round.sh r5m python $GRAFT_REPO_ROOT/tools/gpu_r5_mtime.py 512 m > gpurun_out/r5_final_profile_mfield.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/gpu_r5_mtime.py 512 s,m 2>&1 | grep -E "field|header|sections|chain threads" > gpurun_out/r5_final_timelines.txt
python tools/gpu_r5_multi_cache.py 2>/dev/null | grep cache > gpurun_out/r5_final_multi_cache.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/r5_final_gpu_tests.txt
cat gpurun_out/r5_final_gpu_tests.txt; cut -c1-900 gpurun_out/r5_final_bench_default.json; echo; cut -c1-300 gpurun_out/r5_final_bench_c4.json; echo; tail -12 gpurun_out/r5_final_profile_round.txt; tail -8 gpurun_out/r5_final_profile_mfield.txt; cat gpurun_out/r5_final_timelines.txt gpurun_out/r5_final_multi_cache.txt
