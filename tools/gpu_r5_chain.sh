#!/bin/bash
# round 5: the host coefficient chain on the GPU box's CPU: fast form against the reference's loop (tools/dev/chain_test.c)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
gcc -O2 -std=gnu99 -ffp-contract=off -fno-fast-math -Isz_amd/csrc tools/dev/chain_test.c sz_amd/csrc/szhost.c -o /tmp/chain_test -lm -lpthread && /tmp/chain_test > gpurun_out/r5_chain.log 2>&1
gcc -O2 -std=gnu99 -ffp-contract=off -fno-fast-math -DTHREADS_MAIN -Isz_amd/csrc tools/dev/chain_test.c sz_amd/csrc/szhost.c -o /tmp/chain_thr -lm -lpthread && /tmp/chain_thr >> gpurun_out/r5_chain.log 2>&1
lscpu | grep -i "model name" >> gpurun_out/r5_chain.log
tail -14 gpurun_out/r5_chain.log
