#!/bin/bash
# builds a variant of libmcq_hip.so for same-box A/B runs: scripts/mk_variant.sh NAME [git-rev | -] [extra hipcc flags]
# rev "-" = the working tree.  The library lands in scripts/_ab/libmcq_hip_NAME.so (git-ignored, travels with gpurun);
# use it with MCQ_HIP_LIB=$PWD/scripts/_ab/libmcq_hip_NAME.so python3 bench.py ...
set -e
NAME=$1; REV=${2:--}; shift; shift || true
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/scripts/_ab
if [ "$REV" = "-" ]; then SRC=$ROOT; else
  SRC=/tmp/mcq_variant_$NAME; rm -rf $SRC; mkdir -p $SRC
  git -C $ROOT archive $REV metacache-mpi_amd/csrc include | tar -x -C $SRC
fi
O=/tmp/mcq_variant_obj_$NAME; mkdir -p $O
for u in mcq_engine mcq_build; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $SRC/metacache-mpi_amd/csrc/$u.hip -o $O/$u.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/mcq_engine.o $O/mcq_build.o -o $ROOT/scripts/_ab/libmcq_hip_$NAME.so -ldl
echo built scripts/_ab/libmcq_hip_$NAME.so
