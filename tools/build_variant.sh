#!/bin/bash
# dev tool: build the csrc of a git revision (default HEAD) into tools/_bin/libssie_hip_<name>.so for A/B timing
# usage: tools/build_variant.sh <name> [rev|WORK] [extra hipcc flags]
set -e
name=$1; rev=${2:-HEAD}; extra=$3
P=self-supervised-image-enhancement-network-training-with-low-light-images-only_amd
root=$(mktemp -d); tmp=$root/pkg/csrc
mkdir -p tools/_bin $tmp $root/include
if [ "$rev" = WORK ]; then cp include/ssie_hip.h $root/include/; cp $P/csrc/* $tmp/; else
git show $rev:include/ssie_hip.h > $root/include/ssie_hip.h
for f in $(git ls-tree --name-only $rev $P/csrc/); do git show $rev:$f > $tmp/$(basename $f); done; fi
objs=""
for f in $tmp/*.hip; do o=${f%.hip}.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -w $extra -c $f -o $o & objs="$objs $o"; done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libssie_hip_$name.so $objs
rm -rf $root
echo tools/_bin/libssie_hip_$name.so
