#!/bin/bash
# Build vyomai_amd/lib/libvyom_hip_ab.so with ONE source file taken from another git revision, for
# same-box A/B timing (devices differ by several percent):  tools/ab_lib.sh <rev> <file.hip>
# then on the box:  VY_LIB_PATH=$PWD/vyomai_amd/lib/libvyom_hip_ab.so python bench.py ...
set -e
rev=$1; f=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/vyomai_amd/csrc $tmp/include
cp $root/vyomai_amd/csrc/*.h $tmp/vyomai_amd/csrc/
cp $root/include/vyom_hip.h $tmp/include/
git -C $root show $rev:vyomai_amd/csrc/$f > $tmp/vyomai_amd/csrc/$f
git -C $root show $rev:vyomai_amd/csrc/vy_common.h > $tmp/vyomai_amd/csrc/vy_common.h
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $tmp/vyomai_amd/csrc/$f -o $tmp/ab.o
objs=""
for o in $root/vyomai_amd/lib/*.o; do
  [ "$(basename $o)" = "${f%.hip}.o" ] || objs="$objs $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/vyomai_amd/lib/libvyom_hip_ab.so $objs $tmp/ab.o -ldl
rm -rf $tmp
echo built $root/vyomai_amd/lib/libvyom_hip_ab.so
