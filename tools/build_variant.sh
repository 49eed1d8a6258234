#!/bin/bash
# A/B builds: tools/build_variant.sh <name> <source.hip> "<extra flags>"  -> bs_yolo_amd/libbsyolo_<name>.so = the shipped objects with
# <source> recompiled under the extra flags (use with BSY_LIB=bs_yolo_amd/libbsyolo_<name>.so; the file is git-ignored)
set -e
name=$1; src=$2; flags=$3
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/bs_yolo_amd/csrc/build
python3 -m bs_yolo_amd.build > /dev/null
obj=/tmp/variant_${name}_$(basename ${src%.*}).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -fno-slp-vectorize $flags -c $R/bs_yolo_amd/csrc/$src -o $obj
objs=$(ls $B/*.o | grep -v "/$(basename ${src%.*}).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/bs_yolo_amd/libbsyolo_${name}.so $objs $obj
ls -la $R/bs_yolo_amd/libbsyolo_${name}.so
