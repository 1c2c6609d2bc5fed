#!/bin/bash
# round 6: the decompression's passes between the Huffman decode and the inverse sweep: parity, one call's device time line, per-call times both ways
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_colenc.py -m gpu -x -q 2>&1 | grep -E "passed|failed|assert|rror" | grep -v "szhip_decompress\|Error: " | tail -5 ) > gpurun_out/r6_dec_tests.txt
cat gpurun_out/r6_dec_tests.txt
python tools/gpu_r5_mtime.py 512 s 2>&1 | grep dec_it
SZ_HIP_UNPACK_TILE_KB=12 python tools/gpu_r5_mtime.py 512 s 2>&1 | grep dec_it
SZ_HIP_UNPACK_TILE_KB=48 python tools/gpu_r5_mtime.py 512 s 2>&1 | grep dec_it
bash tools/gpu_r6_trace.sh s dec > /dev/null 2>&1
sed -n '/last decompress/,$p' gpurun_out/r6_dec_timeline.txt | grep -E "k_col|k_beam|k_hdec_write"
