#!/bin/bash
# Experiment builds of the library: tools/scratch/lib/libsebm25_<name>.so with extra -D flags
# (select one at run time with SEBM25_LIB=...).  Usage: build_exp.sh <name> [-DFLAG ...]
set -e
cd "$(dirname "$0")/../../search-engine_amd/csrc"
name=$1; shift
tmp=$(mktemp -d)
pids=()
for f in abi build_kernels query_kernels update_kernels corpus_gen; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=1000000 "$@" -c $f.hip -o $tmp/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "error: a source file did not compile: no library written"; rm -rf $tmp; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/lib/libsebm25_$name.so $tmp/*.o
rm -rf $tmp
