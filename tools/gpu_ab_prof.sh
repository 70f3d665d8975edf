#!/bin/bash
# Same-box A/B of two builds UNDER rocprofv3 (the roofline command of bench.py: tools/run_roofline_cmd.py fused | vq): per build the
# profiler's average duration of the dominant kernel over the last 100 launches + the HIP-event figure of the same run, interleaved.
# usage (through gpurun): bash tools/gpu_ab_prof.sh nameA nameB   -> gpurun_out/ab_prof.md   (tmp_libs/lib_<name>.so)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/ab_prof; rm -rf $O; mkdir -p $O
{
echo "| build | what | rep | rocprofv3 avg of the last 100 launches (us) | HIP events, same run (us) |"
echo "|---|---|---:|---:|---:|"
for rep in 1 2 3; do
  for n in "$@"; do
    for w in fused vq; do
      d=$O/${n}_${w}_$rep
      (cd /tmp && CGIC_LIB=$GRAFT_REPO_ROOT/tmp_libs/lib_$n.so timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$d -o t -- python $GRAFT_REPO_ROOT/tools/run_roofline_cmd.py $w) > $d.log 2>&1
      db=$(find $d -name '*.db' | head -1)
      avg=$(python - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = [r for r in db.execute("select name, start, end from kernels order by start") if "vq_filter" in r[0]][-100:]
print(f"{sum(e - s for _, s, e in rows) / len(rows) / 1e3:.3f}")
PY
)
      hip=$(grep -o "[0-9.]* us per launch" $d.log | tail -1 | cut -d' ' -f1)
      echo "| $n | $w | $rep | $avg | $hip |"
      rm -rf $d
    done
  done
done
} | tee gpurun_out/ab_prof.md
