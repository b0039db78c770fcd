#!/bin/bash
# one large-path iteration on the box: determinism, the large-path GPU tests, then tools/r03_large.sh <tag>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
TAG=${1:-x}
timeout 300 python tools/det_check.py > gpurun_out/r03/det_$TAG.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q -k "large or synth or lazy" > gpurun_out/r03/t_$TAG.txt 2>&1
bash tools/r03_large.sh $TAG
