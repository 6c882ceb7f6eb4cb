#!/bin/bash
# builds ab/<name>.so = the current objects with ONE source recompiled with extra -D flags:  tools/build_variant.sh name file.hip "-DX=1 -DY=2"
set -e
name=$1; src=$2; dst=${4:-$2}; flags=$3
mkdir -p ab /tmp/ddvar
obj=/tmp/ddvar/$name.$src.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 $flags -std=c++17 -fPIC -x hip -w -I include -c distdiff_amd/csrc/$src -o $obj
objs=""
for o in distdiff_amd/csrc/_obj/*.o; do
  if [ "$(basename $o)" == "$dst.o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$name.so $objs
echo built ab/$name.so
