#!/bin/bash
# development: run the bench line for several prebuilt library variants (sz_amd/csrc/variants/*.so, built beforehand)
cd $GRAFT_REPO_ROOT
cp sz_amd/csrc/libszhip.so /tmp/libszhip.keep
for rep in 1 2; do
for so in sz_amd/csrc/variants/*.so; do
  cp $so sz_amd/csrc/libszhip.so
  echo "== $so $(timeout 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['out_bytes'], d['phase_ms']['quant'], d['phase_ms']['decompress_quant'])")"
done
done
cp /tmp/libszhip.keep sz_amd/csrc/libszhip.so
