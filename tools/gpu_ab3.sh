#!/bin/bash
# A/B of variant builds on chosen roofline commands: usage gpu_ab3.sh "cmd1 cmd2 ..." name [name ...] -- tmp_libs/lib_<name>.so; cmds of tools/run_roofline_cmd.py
cd "$GRAFT_REPO_ROOT" || exit 1
CMDS=$1; shift
for rep in 1 2 3; do
  for n in "$@"; do
    for w in $CMDS; do
      echo -n "$n $w: "; CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 120 python tools/run_roofline_cmd.py $w 2>&1 | grep -o "[0-9.]* us per launch"
    done
  done
done
