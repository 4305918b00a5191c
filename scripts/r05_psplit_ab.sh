#!/bin/bash
# A/B of a compile-time constant of vo_ba.hip on ONE box: builds a variant library beside the product's and alternates (scripts/r05_psplit_ab.sh "-DPSPLIT=8")
set -e
cd rgbd_visualodometry_amd/csrc
mkdir -p build/var
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -I../../include $1 -c vo_ba.hip -o build/var/vo_ba.o
hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/libvo_hip.so build/vo_orb.o build/vo_track.o build/var/vo_ba.o build/vo_tri.o build/vo_kf.o build/vo_capi.o
cd ../..
B="--no-cpu-baseline --no-latency-mode --multi-streams= --no-roofline-pass"
for rep in 1 2 3; do
  for v in base var; do
    if [ $v = var ]; then export LD_LIBRARY_PATH=$PWD/rgbd_visualodometry_amd/csrc/build/var:$LD_LIBRARY_PATH_ORIG; else export LD_LIBRARY_PATH=$LD_LIBRARY_PATH_ORIG; fi
    r=$(python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ate_rmse_m'])")
    echo "[$v $1] 300 steps: $r"
  done
done
