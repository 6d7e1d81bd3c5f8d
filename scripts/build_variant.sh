#!/bin/bash
# Builds a VARIANT of the HIP library for same-box A/B runs without touching the product build: one source recompiled with extra flags,
# linked with the product's other objects into scripts/_variants/libosr_<name>.so (git-ignored; travels to the GPU box).
# usage: build_variant.sh <name> <source.hip> "<flags>"      (run openset-rcnn_amd/build.py first: the other objects come from _obj/)
set -e -o pipefail
NAME=$1; SRC=$2; FLAGS=$3; OVERRIDE=$4  # optional 4th argument: another file to compile in place of csrc/<source.hip>
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/scripts/_variants
EXTRA=""
case $SRC in osr_preproc_pool.hip|osr_rpn.hip|osr_roi_align.hip|osr_det_tail.hip|osr_train_fwd.hip|osr_rpn_sparse.hip) EXTRA="-ffp-contract=off";; esac
OBJ=$ROOT/scripts/_variants/${NAME}_${SRC%.hip}.o
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $EXTRA $FLAGS -I$ROOT/openset-rcnn_amd/csrc -c ${OVERRIDE:-$ROOT/openset-rcnn_amd/csrc/$SRC} -o $OBJ
OTHERS=$(ls $ROOT/openset-rcnn_amd/_obj/*.o | grep -v "/${SRC%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scripts/_variants/libosr_${NAME}.so $OBJ $OTHERS
rm -f $OBJ
echo $ROOT/scripts/_variants/libosr_${NAME}.so
