#!/bin/bash
# one gpurun job: tools/gpu_job.sh <name> '<command>' -- runs the command from the repo root, log in gpurun_out/<name>.log
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
name=$1; shift
( eval "$@" ) > gpurun_out/$name.log 2>&1
echo "exit $?" >> gpurun_out/$name.log
tail -c 6000 gpurun_out/$name.log
