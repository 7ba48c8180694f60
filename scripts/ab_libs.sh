#!/bin/bash
# A/B of several builds of the library in ONE box visit (boxes differ by several per cent).
# usage: scripts/ab_libs.sh "libA.so libB.so ..." [bench args]      (names relative to scenelib2_amd/)
cd "$(dirname "$0")/.."
LIBS=$1; shift
for rep in 1 2; do
for lib in $LIBS; do
  SL2_LIB_PATH=$PWD/scenelib2_amd/$lib python bench.py --cpu-sample 0 --steps 60 --warmup 30 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-28s' % '$lib', round(d['value']), round(d['ms_per_step'],4), 'search', round(d['roofline_search']['avg_launch_ms'],4), {k:round(v['ms_per_step'],3) for k,v in list(d['kernels'].items())[:6]})"
done; done
