#!/bin/bash
# ab_snapshot.sh NAME -- a copy of the built tree under ab_NAME/ (git-ignored, travels to the GPU box) so that a later state of the repo can be
# measured against it on ONE box in ONE gpurun call: `python ab_NAME/bench.py ...` imports its own package and loads its own libraries.
set -e
cd "$(dirname "$0")/.."
rm -rf "ab_$1"
mkdir -p "ab_$1"
tar -c --exclude=./.git --exclude=./gpurun_out --exclude='./ab_*' --exclude=./profiles --exclude=./.pytest_cache --exclude=__pycache__ --exclude='./rgbd_visualodometry_amd/csrc/build' . | tar -x -C "ab_$1"
echo "snapshot ab_$1: $(du -sh ab_$1 | cut -f1)"
