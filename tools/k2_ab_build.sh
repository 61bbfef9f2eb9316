#!/bin/bash
# tools/k2_ab_build.sh <name> [git-rev]   build the working tree (or a revision's csrc) as build_ab/<name>.so
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build_ab
SRC=$R/hinge_amd/csrc
if [ -n "${2:-}" ]; then rm -rf /tmp/k2ab_src_$1 && mkdir -p /tmp/k2ab_src_$1 && git -C $R archive $2 hinge_amd/csrc include | tar -x -C /tmp/k2ab_src_$1 && SRC=/tmp/k2ab_src_$1/hinge_amd/csrc && INC=/tmp/k2ab_src_$1/include; else INC=$R/include; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w ${3:-} -I$INC -shared -o $R/build_ab/$1.so $SRC/hinge_capi.hip && echo built build_ab/$1.so
