#!/bin/bash
# build spumoni_amd/libspumoni_gpu_<tag>.so from the working tree with extra flags (A/B builds for tools/ab.sh, tools/ms_ab.sh)
#   usage: bash tools/build_variant.sh <tag> [-DSPX_... flags]
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=/tmp/variant_$tag
rm -rf $d && mkdir -p $d/spumoni_amd && cp -r $root/spumoni_amd/csrc $d/spumoni_amd/ && cp -r $root/include $d/
rm -f $d/spumoni_amd/csrc/*.o
make -C $d/spumoni_amd/csrc -j5 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc $*" >/dev/null
cp $d/spumoni_amd/libspumoni_gpu.so $root/spumoni_amd/libspumoni_gpu_$tag.so
echo built spumoni_amd/libspumoni_gpu_$tag.so "$@"
