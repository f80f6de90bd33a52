#!/bin/bash
# build a bf16-flavor library variant with extra -D flags into prcv2025reid_amd/csrc/libreid_hip_<name>.so (use with REID_LIB_BF16=...)
# usage: tools/build_variant.sh <name> <flags...>
set -e -o pipefail
NAME=$1; shift
cd "$(dirname "$0")/../prcv2025reid_amd/csrc"
mkdir -p /tmp/reid_var_$NAME
rm -f /tmp/reid_var_$NAME/*.o                                # (a failed compile must not link a stale object)
pids=()
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value "$@" -c $f -o /tmp/reid_var_$NAME/${f%.hip}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "compile failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libreid_hip_$NAME.so /tmp/reid_var_$NAME/*.o
echo built libreid_hip_$NAME.so
