#!/bin/bash
# development: parity diag + bench line
cd $GRAFT_REPO_ROOT
timeout 200 python tools/gpu_diag.py > gpurun_out/diag.log 2>&1; echo "identical: $(grep -c IDENTICAL gpurun_out/diag.log)  dec-bit-mismatch-0: $(grep -c 'bit mismatches vs oracle 0 ' gpurun_out/diag.log)"; grep -E "FAILURES|MISMATCH|rror" gpurun_out/diag.log | head
timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['out_bytes'], d['psnr'], d['phase_ms'], d['decompress_GBps'])"
