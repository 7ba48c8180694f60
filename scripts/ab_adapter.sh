#!/bin/bash
# A/B of the adapter's per-frame latency: this tree's build against another build of the library AND of the example, alternated in
# one box.  The other build is a directory laid out like the repository, e.g.
#   mkdir _prev && git archive <commit> scenelib2_amd/csrc include examples | tar -x -C _prev
#   make -C _prev/scenelib2_amd/csrc ../libscenelib2_amd.so && cp examples/monoslam_adapter.cpp _prev/examples/ && make -C _prev/examples monoslam_adapter
# (the example finds its library through its RUNPATH.  Do NOT put another build in front with LD_PRELOAD: the process then
# holds two copies of the library and of its kernels, and a run that way diverged from the oracle after 90 frames.)
# Usage: scripts/ab_adapter.sh <other-build-dir> <out-dir>
OTHER=${1:-_prev}; OUT=${2:-gpurun_out/ab}
mkdir -p $OUT
make -s -C examples > /dev/null 2>&1
for i in 1 2; do
  timeout 300 python scripts/adapter_latency.py $OUT/adapter_new$i.json > /dev/null 2>&1
  ADAPTER_EXE=$PWD/$OTHER/examples/monoslam_adapter timeout 300 python scripts/adapter_latency.py $OUT/adapter_prev$i.json > /dev/null 2>&1
done
python - <<PY
import json
for n in ("new1", "prev1", "new2", "prev2"):
    a = json.load(open("$OUT/adapter_%s.json" % n))
    m = a["mapping_on_dozen_features"]
    print(n, {k: v["frame_us_median"] for k, v in a.items() if isinstance(v, dict)},
          "mapping scene, frames starting without / with a partial feature:",
          m.get("frames_starting_without_partial_feature"), m.get("frame_us_median_without_partial_feature"),
          m.get("frames_starting_with_partial_feature"), m.get("frame_us_median_with_partial_feature"), m.get("partial_features_at_frame_start"))
PY
