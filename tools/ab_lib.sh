#!/bin/bash
# A/B libraries: libpangu_hip.so re-linked with ONE object rebuilt under extra flags (nothing else changes), for interleaved
# kernel A/Bs through PANGU_HIP_LIB (pangu-pytorch_amd/_lib.py):
#   bash tools/ab_lib.sh <tag> <file.hip> [hipcc flags..]      ->  scratch/libpangu_<tag>.so     (build container; ships with gpurun)
set -eu
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift 2
C=pangu-pytorch_amd/csrc
mkdir -p scratch
make -C $C -j4 > /dev/null
OBJS=$(ls $C/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $C/$SRC -o scratch/ab_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libpangu_$TAG.so $OBJS scratch/ab_$TAG.o -lpthread
rm -f scratch/ab_$TAG.o
ls -la scratch/libpangu_$TAG.so
