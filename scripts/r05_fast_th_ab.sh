#!/bin/bash
# k_fast_nms tile height A/B: rebuilds vo_orb / vo_capi with -DVO_FAST_TH=<rows> on the GPU box and times the ORB batch (32 frames x 2000 features,
# and the bench's own shape: 20 frames x 1000 features)
O=gpurun_out/fastth; mkdir -p $O
cd rgbd_visualodometry_amd/csrc
for th in 16 32 48 64; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -I../../include -DVO_FAST_TH=$th -c vo_orb.hip -o build/vo_orb.o &&
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -I../../include -DVO_FAST_TH=$th -c vo_capi.hip -o build/vo_capi.o &&
  hipcc --offload-arch=gfx950 -shared -fPIC -o libvo_hip.so build/vo_orb.o build/vo_track.o build/vo_ba.o build/vo_tri.o build/vo_kf.o build/vo_capi.o || exit 2
  echo "== TH=$th" | tee -a ../../$O/ab.txt
  (cd ../.. && timeout -k 10 120 python scripts/bench_orb_only.py 2000 32 2>/dev/null | grep -E "orb_only|k_fast_nms" | tee -a $O/ab.txt && timeout -k 10 120 python scripts/bench_orb_only.py 1000 20 2>/dev/null | grep -E "orb_only|k_fast_nms" | tee -a $O/ab.txt) || exit 3
done
