#!/bin/bash
# profiles/r03_* from the output of tools/gpu_profile_r03.sh (gpurun_out/r03p)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03p
cp $O/r03_roofline.json profiles/r03_roofline.json; cp $O/pmc_hbm.json profiles/pmc_vq.json
{
echo "# Kernel-trace summaries, round 3 (final build: prepared codebook image, funnel-shift self-synchronising decoder, 512-thread compress workgroups -- four per image, 32 KB of LDS -- and one-band merge, one-launch split decoder only)"
echo
echo "Made by \`tools/gpu_profile_r03.sh\` on one MI355X (everything below is from ONE gpurun call; \`profiles/r03_roofline.json\` holds the same numbers machine-readable)."
echo "Commands: \`rocprofv3 --kernel-trace --stats -- python bench.py --steps 96 --warmup 16 --no-report --lanes L\` for L = 1 and 4"
echo "(\`--no-report\`: only the timed loop, so that the last 96 five-launch chains of the trace are the timed steps; tools/trace_concurrency.py),"
echo "and \`rocprofv3 --kernel-trace --stats -- python tools/run_roofline_cmd.py fused|vq\` = the roofline command of bench.py on its own"
echo "(20 launches of the dominant kernel per hipGraph, HIP events around the last 5 replays)."
echo "NOTE: the kernel trace itself slows the step down and serialises part of the overlap of the four lanes (≈60 us per step under the"
echo "profiler against 34-40 us without): the --lanes 4 durations are an upper bound of what a kernel costs in flight."
echo
echo "## one batch in flight (--lanes 1): every launch waits for the previous one; latency decoder (decode_split_kernel), 4-band merge"
echo; cat $O/loop_lanes1.md; echo; echo '```'; head -9 $O/kernel_stats_lanes1.md; echo '```'
echo
echo "## four batches in flight (--lanes 4, the default): independent HIP streams on four hardware queues; throughput decoder (decode_image_kernel), one-band merge"
echo; cat $O/loop_lanes4.md; echo; echo '```'; head -9 $O/kernel_stats_lanes4.md; echo '```'
echo
echo "## the roofline command: the fused VQ + router launch alone, back to back (tools/run_roofline_cmd.py fused)"
echo; grep HIP $O/alone.log; echo; echo '```'; head -5 $O/kernel_stats_alone.md; echo '```'
echo "rocprofv3 average over the last 100 launches (the five timed replays): see r03_roofline.json \`rocprof_avg_us_alone_graph\`."
echo
echo "## the VQ kernel without the router workgroups (tools/run_roofline_cmd.py vq)"
echo; grep HIP $O/alonevq.log; echo; echo '```'; head -5 $O/kernel_stats_alone_vq.md; echo '```'
} > profiles/r03_kernel_stats.md
{
echo "# HBM traffic per launch, round 3 (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes, --kernel-trace)"
echo
echo "Command per pass: \`rocprofv3 --pmc <C> --kernel-trace -- python bench.py --steps 40 --warmup 10 --no-report --lanes 1 --no-graph --no-dist\`;"
echo "tools/pmc_summary.py: KiB x 1024; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (a 128-byte request is tallied as 64)."
echo
cat $O/pmc_hbm.md
echo
echo "Algorithmic bytes per launch at B=64 of 256x256: entropy maps 50.33 MB read (the image, once); VQ + router 4.2 MB read (latent) + 0.33 MB (entropy maps)"
echo "and 2.1 MB (int64 indices) + 4.2 MB (z_q) + 1.38 MB (int32 masks) written = 12.2 MB; counted 13.57 MB (r02: 14.85) = 1.11x.  Of the 1.4 MB"
echo "over the algorithmic bytes, the ISA of the final build accounts for ~0.5 MB: the 128-VGPR cap leaves ONE spilled dword per lane, stored once per"
echo "workgroup before the group loop (256 workgroups x 512 lanes x 4 B; round 2 spilled 4-10 registers inside the loop); the rest (loss partials,"
echo "ticket words, partial cache lines of the mask writes) was not attributed by a separate measurement.  The 54 KB prepared codebook image read by"
echo "256 workgroups is served by L2."
} > profiles/r03_pmc_hbm.md
{
echo "# SQ instruction counters, round 3"
echo
echo "Every kernel of the step, one pass each: \`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU"
echo "--kernel-trace -- python bench.py --steps 40 --warmup 10 --no-report --lanes L --no-graph --no-dist\` (counters summed over all XCDs / SEs of a dispatch, averaged over dispatches)."
echo
echo "## --lanes 1 (latency decoder, 4-band merge)"
echo
cat $O/pmc_sq.md
echo
echo "## --lanes 4 (what the timed step launches: throughput decoder decode_image_kernel, one-band merge_kernel<512>)"
echo
cat $O/pmc_sq_lanes4.md
echo
echo "## the dominant kernel alone (tools/run_roofline_cmd.py fused): matrix-core and LDS counters"
echo
cat $O/pmc_sq_vq.md
} > profiles/r03_pmc_sq.md
python -c "
import json; d=json.load(open('profiles/r03_roofline.json')); print({k: d[k] for k in ('rocprof_avg_us_alone_graph','hip_events_us_alone_graph_same_run','rocprof_avg_us_lanes1_loop','rocprof_avg_us_lanes4_loop','frac_alone_graph','frac_lanes1_loop','hbm_bytes_per_launch')})"
