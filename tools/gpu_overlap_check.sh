cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_recorded.py -m gpu -q -x -k "M or reg or mean or f64 or stream_and_decode" > gpurun_out/ov_tests.log 2>&1; grep -aE "^E  |[0-9]+ passed|failed|FAILED" gpurun_out/ov_tests.log | head -8 | cut -c1-300
SZ_HIP_TIMING=1 timeout 300 python tools/gpu_mfield.py 2>&1 | grep -E "^M call [4-7]|shipped" | tail -8 | cut -c1-420
SZ_HIP_CHAIN_OVERLAP=0 timeout 300 python tools/gpu_mfield.py 2>&1 | grep -E "^M call [6-7]" | cut -c1-200
