#!/bin/bash
# run one python script on the GPU box, stdout -> gpurun_out/<tag>.json
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
TAG=$1; shift
rm -rf gpurun_out/*; mkdir -p gpurun_out
timeout 1200 python "$@" > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err; echo "rc=$?"; tail -3 gpurun_out/$TAG.err; cat gpurun_out/$TAG.json
