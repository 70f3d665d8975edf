#!/bin/bash
# profiles/<round>_* from the output of tools/gpu_profile.sh (gpurun_out/<round>p); usage: tools/assemble_profiles.sh [r05] ["what changed this round"]
R=${1:-r05}
WHAT=${2:-"refinement band by value, refinement queues in the stand-alone router launch, one-call drivers"}
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${R}p
cp $O/${R}_roofline.json profiles/${R}_roofline.json; cp $O/pmc_hbm.json profiles/pmc_vq.json
{
echo "# Kernel-trace summaries, ${R} (${WHAT})"
echo
echo "Made by \`tools/gpu_profile.sh\` on one MI355X (everything below is from ONE gpurun call; \`profiles/${R}_roofline.json\` holds the same numbers machine-readable)."
echo "Commands: \`rocprofv3 --kernel-trace --stats -- python bench.py --steps 96 --warmup 16 --no-report --lanes L\` for L = 1 and 4"
echo "(\`--no-report\`: only the timed loop, so that the last 96 five-launch chains of the trace are the timed steps; tools/trace_concurrency.py),"
echo "and \`rocprofv3 --kernel-trace --stats -- python tools/run_roofline_cmd.py fused|vq\` = the roofline command of bench.py on its own"
echo "(20 launches of the dominant kernel per hipGraph, HIP events around the last 5 replays; the fused launch gets the pixels: the router refines)."
echo "NOTE: the kernel trace itself slows the step down and serialises part of the overlap of the four lanes (≈65 us per step under the"
echo "profiler against 34-40 us without): the --lanes 4 durations are an upper bound of what a kernel costs in flight.  Boxes of the pool differ"
echo "by ~1 us per kernel: compare numbers of ONE file, and across rounds only with that in mind (A/B runs of two builds in one call: tools/gpu_ab.sh)."
echo
echo "## one batch in flight (--lanes 1): every launch waits for the previous one; latency decoder (decode_split_kernel), 4-band merge"
echo; cat $O/loop_lanes1.md; echo; echo '```'; head -9 $O/kernel_stats_lanes1.md; echo '```'
echo
echo "## four batches in flight (--lanes 4, the default): independent HIP streams on four hardware queues; throughput decoder (decode_image_kernel), one-band merge"
echo; cat $O/loop_lanes4.md; echo; echo '```'; head -9 $O/kernel_stats_lanes4.md; echo '```'
echo
echo "## the roofline command: the fused VQ + router launch alone, back to back (tools/run_roofline_cmd.py fused)"
echo; grep HIP $O/alone.log; echo; echo '```'; head -5 $O/kernel_stats_alone.md; echo '```'
echo "rocprofv3 average over the last 100 launches (the five timed replays): see ${R}_roofline.json \`rocprof_avg_us_alone_graph\`."
echo
echo "## the VQ kernel without the router workgroups (tools/run_roofline_cmd.py vq)"
echo; grep HIP $O/alonevq.log; echo; echo '```'; head -5 $O/kernel_stats_alone_vq.md; echo '```'
} > profiles/${R}_kernel_stats.md
{
echo "# HBM traffic per launch, ${R} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes, --kernel-trace)"
echo
echo "Command per pass: \`rocprofv3 --pmc <C> --kernel-trace -- python bench.py --steps 40 --warmup 10 --no-report --lanes 1 --no-graph --no-dist\`;"
echo "tools/pmc_summary.py: KiB x 1024; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (a 128-byte request is tallied as 64)."
echo "(--lanes 1 = the latency decoder and the four-band merge: the merge's reads are four stagings of an image's streams; the timed step's"
echo "one-band merge stages once.)"
echo
cat $O/pmc_hbm.md
echo
echo "## the four-lane step's kernels (--lanes 4: decode_image_kernel, one-band merge_kernel)"
echo
cat $O/pmc_hbm_lanes4.md 2>/dev/null
echo
echo "Algorithmic bytes per launch at B=64 of 256x256: entropy maps 50.33 MB read (the image, once) + 0.6 MB written (two maps + the flat8 by-product);"
echo "VQ + router 4.2 MB read (latent) + 0.33 MB (entropy maps) + 0.26 MB (flat8) and 2.1 MB (int64 indices) + 4.2 MB (z_q) + 1.38 MB (int32 masks) written = 12.5 MB."
} > profiles/${R}_pmc_hbm.md
{
echo "# SQ counters of the dominant kernel, ${R}: the shipping fused VQ + router launch (vq_filter_router_kernel<true,false>, fp16 MFMA filter + exact fp32 resolve + router workgroups with refinement)"
echo
echo "Three passes of \`rocprofv3 --pmc <8 counters> --kernel-trace -- python tools/run_roofline_cmd.py fused\` (counters summed over all XCDs / SEs of a dispatch, averaged over the dispatches)."
echo
cat $O/pmc_sq_vq_fused.md
echo
python - <<PY
import json
d = json.load(open("$O/${R}_roofline.json")); m = dict(d.get("mfma", {}))
for k, v in json.load(open("$O/pmc_sq_vq_fused.json")).items():
    if "vq_filter_router_kernel" in k: m.update(v)
dur = d["rocprof_avg_us_alone_graph"]
cyc = dur * 1e-6 * 2.4e9
print("Derived (launch duration %.2f us by rocprofv3 = %.0f cycles at 2.4 GHz; 1024 SIMDs):" % (dur, cyc))
print("* matrix pipes busy: SQ_VALU_MFMA_BUSY_CYCLES / 1024 / cycles = **%.3f** (%.0f MFMAs per SIMD x 32 cycles = %.0f cycles = %.2f us per SIMD)" % (m["mfma_busy_frac"], m["SQ_INSTS_MFMA"] / 1024, m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024, m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / 2.4e3))
print("* %.0f %% of the matrix-busy cycles have a VALU instruction executing beside them (SQ_VALU_MFMA_COEXEC_CYCLES)" % (100 * m["SQ_VALU_MFMA_COEXEC_CYCLES"] / m["SQ_VALU_MFMA_BUSY_CYCLES"]))
print("* VALU issue: %.0f instructions per SIMD x 4 cycles = %.2f us per SIMD" % (m["SQ_INSTS_VALU"] / 1024, m["SQ_INSTS_VALU"] / 1024 * 4 / 2.4e3))
print("* waves wait (SQ_WAIT_ANY / SQ_WAVE_CYCLES): %.0f %%; LDS bank conflicts: %.0f %% of the LDS-active cycles" % (100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 100 * m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]))
print("* the two levers round 4's verdict named, priced by these counters: (a) LDS bank conflicts -- %.0f conflict cycles per CU = %.2f us of the launch's %.1f if NONE of it were hidden behind the VALU scan (LDS is index-active %.0f %% of the busy CU cycles; waves wait on an LDS instruction %.1f %% of their wait cycles: SQ_WAIT_INST_LDS / SQ_WAIT_ANY): a swizzle of the ldsA tiles is worth <= %.1f us and costs address VALU in a loop that is VALU-issue-bound at the 128-VGPR cap -- not attempted; (b) the router workgroups as the launch's tail (they end at 21.0-21.7 us, the VQ workgroups at 18.5-19.5: NOTES 10.2) -- routers on 64 CUs of their own leave the VQ scan 192 CUs: its %.1f us x 256 CU of work take %.1f us there, more than the whole launch now (priced, not built); three waves per SIMD need <= 84 VGPRs where the kernel already spills at its 128 cap (uncapped it takes 144-150; ScratchSize 20 B/lane, 247 spilled SGPRs: llvm -Rpass-analysis=kernel-resource-usage) -- the scan loop's registers would go to scratch (priced, not built); measured and slower: router waves at raised priority throughout, 256- / 128-thread router teams, 1024-thread or two VQ workgroups per CU (NOTES 10.2, 10.3f)" % (m["SQ_LDS_BANK_CONFLICT"] / 256, m["SQ_LDS_BANK_CONFLICT"] / 256 / 2.4e3, dur, 100 * m["SQ_LDS_IDX_ACTIVE"] / m["SQ_BUSY_CU_CYCLES"], 100 * m["SQ_WAIT_INST_LDS"] / m["SQ_WAIT_ANY"], m["SQ_LDS_BANK_CONFLICT"] / 256 / 2.4e3, 19.0, 19.0 * 256 / 192))
print("* fp16 MFMA work: %.2f GFLOP per launch = %.0f TFLOP/s = %.3f of the 2.5 PFLOP/s dense fp16 peak; algorithmic 2NKD = 2.147 GFLOP = %.1f TFLOP/s = %.3f of the 157.3 fp32 yardstick" % (m["SQ_INSTS_MFMA"] * 32768 / 1e9, m["SQ_INSTS_MFMA"] * 32768 / dur / 1e6, m["SQ_INSTS_MFMA"] * 32768 / dur / 1e6 / 2500, 2147.48 / dur, 2147.48 / dur / 157.3))
PY
echo
echo "## every kernel of the step (--lanes 1 / --lanes 4 eager passes: instruction counters)"
echo
echo '### --lanes 1'
cat $O/pmc_sq.md
echo
echo '### --lanes 4 (what the timed step launches: throughput decoder decode_image_kernel, one-band merge)'
cat $O/pmc_sq_lanes4.md
} > profiles/${R}_pmc_sq_vq.md
python -c "
import json; d=json.load(open('profiles/${R}_roofline.json')); print({k: d[k] for k in ('rocprof_avg_us_alone_graph','hip_events_us_alone_graph_same_run','rocprof_avg_us_lanes1_loop','rocprof_avg_us_lanes4_loop','frac_alone_graph','frac_lanes1_loop','hbm_bytes_per_launch')}); print(d['mfma']['mfma_busy_frac'])"
