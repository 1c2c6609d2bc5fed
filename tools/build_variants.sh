#!/bin/bash
# development: build library variants in parallel: tools/build_variants.sh name1 "-DFLAG=.. -DFLAG2=.." name2 "..." ...
cd "$(dirname "$0")/../sz_amd/csrc"
mkdir -p variants
make -s szhost.o sz_api.o sz_conf.o sz_rw.o sz_slab.o
HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-unused-value"
pids=()
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc $HIPFLAGS $flags -c szhip.hip -o variants/szhip_$name.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libszhip_$name.so variants/szhip_$name.o szhost.o sz_api.o sz_conf.o sz_rw.o sz_slab.o -ldl -lm && rm variants/szhip_$name.o && echo built $name ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
ls -la variants/
