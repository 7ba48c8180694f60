for m in 0 1 2 4 8 15; do echo mask $m; SL2_DUMMY_MASK=$m NK=5 bash scripts/ab_probe.sh "--steps 40 --warmup 20" scenelib2_amd/libscenelib2_amd_probe.so; done
