#!/bin/bash
# A/B (library vs build/ab_*.so) followed by the whole GPU test suite on the library; usage: gpu_ab5.sh tag "precisions" rounds
TAG=${1:-ab5}
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/scripts/gpu_ab4.sh "$@"
O=$R/gpurun_out/$TAG; cd $R
echo "== pytest gpu"; S=$SECONDS; timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? in $((SECONDS-S)) s"; tail -12 $O/pytest_gpu.log
