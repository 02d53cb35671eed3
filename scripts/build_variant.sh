#!/bin/bash
# A variant build of libsgam_hip.so for a same-box A / B:   bash scripts/build_variant.sh <name> KEY=VALUE ...
# -> ablib/<name>/libsgam_hip.so (git-ignored, travels to the GPU box; select it with SGAM_HIP_LIB).  The objects of the default build
# are copied first, so only the translation units whose flags change are recompiled.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p ablib/$name
cp -u sgam_neurips22_amd/lib/*.o sgam_neurips22_amd/lib/*.sha ablib/$name/ 2>/dev/null || true
env "$@" SGAM_LIB_DIR=$PWD/ablib/$name python -m sgam_neurips22_amd.build > /dev/null
echo "$name: $* -> ablib/$name/libsgam_hip.so"
