#!/bin/bash
# The evidence set of a round, on the GPU box: tools/profile_round.sh <tag>  ->  gpurun_out/<tag>/
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
# 2. kernel trace of the one-pair-at-a-time command (average launch durations must agree with its hipEvent figures)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-real-pair > $O/bench_traced.json 2> /dev/null
# 3. HIP API trace: no allocator call in a steady-state step
rocprofv3 --hip-trace --stats --output-format csv -d $O/hip -o h -- python $R/bench.py --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-real-pair > /dev/null 2>&1
cd $R
# 4. PMC passes of the same command
tools/pmc_run.sh $TAG refine,sweep,view_eval,init -- python $R/bench.py --in-flight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-real-pair > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/$TAG k_refine gpurun_out/$TAG/refine_pmc.json "bench.py --in-flight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-real-pair: C3, (1 warm-up + 1 timed) pairs x 3 iterations = 6 launches of k_refine<true,1>" > /dev/null
python tools/pmc_to_json.py gpurun_out/$TAG k_spatial_sweep gpurun_out/$TAG/sweep_pmc.json "same command: 6 launches of k_spatial_sweep<true,1>" > /dev/null
# 1. the default bench line (three pairs in flight, CPU baseline) and the one-pair-at-a-time line -- after the counter passes, so
#    that the line quotes the counters of THIS build (profiles/refine_pmc.json is refused when its source hash differs)
cp $O/refine_pmc.json profiles/refine_pmc.json
cp $O/sweep_pmc.json profiles/sweep_pmc.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --in-flight 1 --no-cpu-baseline > $O/bench_inflight1.json 2> /dev/null
# 5. the other BASELINE.json configs (3 pairs in flight; C5 includes post-processing; C4 goes through batch.run_batch) -- for reference, not the headline
for C in C1 C2 C4 C5; do
  S=6; [ $C = C5 ] && S=3
  python bench.py --config $C --no-cpu-baseline --steps $S --warmup 2 > $O/bench_$(echo $C | tr A-Z a-z).json 2> /dev/null
done
# BASELINE.json configs[3] at its stated size: 200 pairs held by rank 0, dispatched by batch.run_batch, default pairs in flight
python bench.py --config C4 --steps 200 --warmup 3 --no-cpu-baseline --no-real-pair > $O/bench_c4_200.json 2> /dev/null
./tools/ubench/valu_issue > $O/valu_issue.txt 2>&1
# 6. the RCCL code path on this one GPU: a process group of world size 1, C4 dispatched by run_batch's broadcast / scatter / gather
cd /tmp
CSPM_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rccl -o t -- python $R/bench.py --config C4 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_c4_rccl.json 2> /dev/null
# 7. how the pairs in flight overlap
rocprofv3 --kernel-trace --output-format csv -d $O/overlap -o t -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-real-pair --no-kernel-timing > /dev/null 2>&1
cd $R
python tools/overlap_timeline.py $O/overlap/t_kernel_trace.csv > $O/overlap_inflight.txt
python tools/concurrent_phases.py C3 > $O/concurrent_phases.txt 2>&1
grep -i "nccl\|rccl" $O/rccl/t_kernel_stats.csv | cut -c1-160
python tools/bench_brief.py default < $O/bench_default.json
python tools/bench_brief.py inflight1 < $O/bench_inflight1.json
head -4 $O/trace/t_kernel_stats.csv | cut -c1-150
grep -i "malloc\|free" $O/hip/h_hip_api_stats.csv | cut -c1-150
