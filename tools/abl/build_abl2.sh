#!/bin/bash
# usage: build_abl2.sh <file-stem> <tag> <flags...>  -> tools/abl/libmfp_<stem>_<tag>.so with <stem>.hip compiled under the extra flags
set -e
STEM=$1; TAG=$2; shift 2
cd "$(dirname "$0")/../../flex-dm_amd/csrc"
OBJS=$(ls *.o | grep -v "^$STEM.o\$")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $STEM.hip -o ../../tools/abl/${STEM}_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../tools/abl/${STEM}_$TAG.o -o ../../tools/abl/libmfp_${STEM}_$TAG.so
