#!/bin/bash
# Developer A/B builds: tools/build_variant.sh NAME "-DFOO=1 -DBAR=2" -> cuda-slam_amd/variants/libmislam_NAME.so (git-ignored; select it
# with MISLAM_LIB=cuda-slam_amd/variants/libmislam_NAME.so).  The product build (cuda-slam_amd/csrc/Makefile) is untouched.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/cuda-slam_amd/variants
obj=/tmp/mislam_variant_$name
mkdir -p "$out" "$obj"
cd "$root/cuda-slam_amd/csrc"
srcs=$(sed -n 's/^SRCS *= *//p' Makefile)              # the product build's own list: a new translation unit cannot drop out
for f in ${srcs//.hip/}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I/opt/rocm/include "$@" -c $f.hip -o $obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libmislam_$name.so $obj/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo $out/libmislam_$name.so
