#!/bin/bash
# usage: build_abl.sh <file-stem> <MACRO> n1 n2 ...  -> tools/abl/libmfp_<stem>_<n>.so with <stem>.hip compiled under -D<MACRO>=<n>
set -e
STEM=$1; MACRO=$2; shift 2
cd "$(dirname "$0")/../../flex-dm_amd/csrc"
make -j8 >/dev/null
OBJS=$(ls *.o | grep -v "^$STEM.o\$")
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -D$MACRO=$n -c $STEM.hip -o ../../tools/abl/${STEM}_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../tools/abl/${STEM}_$n.o -o ../../tools/abl/libmfp_${STEM}_$n.so
done
