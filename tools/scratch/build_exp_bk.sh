#!/bin/bash
# Experiment builds that differ only in build_kernels.hip: tools/scratch/lib/libsebm25_<name>.so with extra -D flags,
# linked with the other objects of the regular build.  Usage: build_exp_bk.sh <name> [-DFLAG ...]
set -e
cd "$(dirname "$0")/../../search-engine_amd/csrc"
make -s
name=$1; shift
mkdir -p ../../tools/scratch/lib
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=1000000 "$@" -c build_kernels.hip -o $tmp/build_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/lib/libsebm25_$name.so abi.o $tmp/build_kernels.o query_kernels.o update_kernels.o corpus_gen.o
rm -rf $tmp
