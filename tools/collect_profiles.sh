#!/bin/bash
# copies the evidence set tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/ (tracked) as <prefix>_*
set -eu
T=gpurun_out/$1; P=profiles/$2
cp $T/bench_default.json ${P}_bench_default.json
cp $T/bench_inflight1.json ${P}_bench_inflight1.json
cp $T/bench_traced.json ${P}_bench_traced_inflight1.json
cp $T/trace/t_kernel_stats.csv ${P}_c3_kernel_stats.csv
cp $T/hip/h_hip_api_stats.csv ${P}_hip_api_stats.csv
cp $T/summary.txt ${P}_c3_pmc_summary.txt
cp $T/refine_pmc.json ${P}_refine_pmc.json
cp $T/refine_pmc.json profiles/refine_pmc.json   # the one bench.py quotes (hash-stamped)
cp $T/sweep_pmc.json ${P}_sweep_pmc.json
cp $T/sweep_pmc.json profiles/sweep_pmc.json     # likewise, for roofline.kernels.spatial
cp $T/valu_issue.txt ${P%_*}_valu_issue.txt
[ -s $T/rccl/t_kernel_stats.csv ] && cp $T/rccl/t_kernel_stats.csv ${P}_c4_rccl_kernel_stats.csv
[ -s $T/bench_c4_rccl.json ] && cp $T/bench_c4_rccl.json ${P}_bench_c4_rccl.json
[ -s $T/overlap_inflight.txt ] && cp $T/overlap_inflight.txt ${P}_overlap_inflight.txt
[ -s $T/concurrent_phases.txt ] && cp $T/concurrent_phases.txt ${P}_concurrent_phases.txt
for c in c1 c2 c4 c5; do [ -s $T/bench_$c.json ] && cp $T/bench_$c.json ${P}_bench_$c.json; done
ls -la ${P}_* profiles/refine_pmc.json | awk '{print $5, $9}'
