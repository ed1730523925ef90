#!/bin/bash
# Build the library of a git revision (default HEAD) next to the working tree's, for same-run A/B timings:
#   tools/build_base_lib.sh [rev]   ->  tools/ab_base/libpascohip_base.so   (git-ignored; travels with gpurun)
#   LAYER_AB_BASE=tools/ab_base/libpascohip_base.so python tools/layer_ab.py out.txt 0
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/base_src.XXXXXX)
git -C "$root" archive "$rev" pasco_amd/csrc include | tar -x -C "$tmp"
cd "$tmp/pasco_amd/csrc"
objs=""
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c "$f" -o "${f%.hip}.o" &
  objs="$objs ${f%.hip}.o"
done
wait
mkdir -p "$root/tools/ab_base"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/ab_base/libpascohip_base.so" $objs
echo "$rev -> $root/tools/ab_base/libpascohip_base.so"
rm -rf "$tmp"
