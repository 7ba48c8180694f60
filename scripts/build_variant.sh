#!/bin/bash
# Build scenelib2_amd/libsl2_var_<name>.so from the current sources with extra compiler flags (A/B runs: scripts/ab_libs.sh).
# usage: scripts/build_variant.sh <name> [flags...]
set -e
cd "$(dirname "$0")/../scenelib2_amd/csrc"
NAME=$1; shift
T=$(mktemp -d)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -Wno-unused-value"
OBJS=""
for s in sl2_engine sl2_frontend sl2_search sl2_ekf_update sl2_featureinit sl2_mapping sl2_ingest sl2_synth sl2_snapshot; do
  X=""; [ $s = sl2_ekf_update ] && X="-ffp-contract=fast"
  /opt/rocm/bin/hipcc $F $X "$@" -c $s.hip -o $T/$s.o &
  OBJS="$OBJS $T/$s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsl2_var_$NAME.so $OBJS -lz
rm -rf $T
echo built ../libsl2_var_$NAME.so
