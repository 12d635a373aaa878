#!/bin/bash
# Ablation / timeline build of the library (results may be wrong on purpose, see CMLPL_ABL in conv3x3.hip):
#   bash scripts/build_abl.sh 9    ->  cmlpl_amd/libabl9.so   (use with CMLPL_LIB=cmlpl_amd/libabl9.so)
#   ABL_FLAGS=-DCMLPL_STAMP_MIN_H=20 ABL_SUFFIX=p bash scripts/build_abl.sh 9   ->  libabl9p.so (timeline of the general path's conv1 launches at 20x20)
set -e
N=${1:-9}
cd "$(dirname "$0")/../cmlpl_amd"
HASH=$(cd .. && python3 -c "from cmlpl_amd.build_ext import source_hash; print(source_hash())")
SFX=${ABL_SUFFIX:-}
mkdir -p build_abl$N$SFX
for f in api augment conv0 conv3x3 dense dist head loss memobank ntxent optim wgrad3x3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DCMLPL_ABL=$N $ABL_FLAGS -DCMLPL_SOURCE_HASH=\"$HASH\" -c csrc/$f.hip -o build_abl$N$SFX/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libabl$N$SFX.so build_abl$N$SFX/*.o
echo cmlpl_amd/libabl$N$SFX.so
