cd /root/repo
# A/B of two builds of the library in ONE box visit (boxes differ by several per cent): prev = libscenelib2_amd_prev.so
for rep in 1 2; do
for lib in libscenelib2_amd_prev.so libscenelib2_amd.so; do
  echo "$lib"; SL2_LIB_PATH=$PWD/scenelib2_amd/$lib python bench.py --cpu-sample 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms_per_step'],3) for k,v in list(d['kernels'].items())[:5]})"
done; done
