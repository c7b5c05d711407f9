#!/bin/bash
# usage: build_sched.sh <file-stem>:<strategy> ...  -> tools/abl/libmfp_<stem>_<strategy>.so with <stem>.hip compiled under
# -mllvm -amdgpu-sched-strategy=<strategy> (everything else as the product build): A/B of hipcc's scheduling strategies.
set -e
cd "$(dirname "$0")/../../flex-dm_amd/csrc"
make -j8 >/dev/null
for v in "$@"; do
  STEM=${v%%:*}; STRAT=${v##*:}
  OBJS=$(ls *.o | grep -v "^$STEM.o\$")
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-sched-strategy=$STRAT -c $STEM.hip -o ../../tools/abl/${STEM}_$STRAT.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../tools/abl/${STEM}_$STRAT.o -o ../../tools/abl/libmfp_${STEM}_$STRAT.so
done
