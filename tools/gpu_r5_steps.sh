#!/bin/bash
# round 5: where do the timed steps of the default line go?  Five runs of the timed region; per-step times in every line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r5_steps.log
for i in 1 2 3 4 5; do
  python bench.py --timed-only --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','step_ms','single_call','phase_ms')}))" >> gpurun_out/r5_steps.log
done
cat gpurun_out/r5_steps.log
