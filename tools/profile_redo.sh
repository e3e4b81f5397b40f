# the trace and counter parts of tools/profile_round.sh only:  tools/profile_redo.sh <tag>
set -u
TAG=$1; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace $O/pmc* $O/overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-real-pair > $O/bench_traced.json 2> /dev/null
cd $R
tools/pmc_run.sh $TAG refine,sweep,view_eval,init -- python $R/bench.py --in-flight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-real-pair > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/$TAG k_refine gpurun_out/$TAG/refine_pmc.json "bench.py --in-flight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-real-pair: C3, (1 warm-up + 1 timed) pairs x 3 iterations = 6 launches of k_refine<true,1>" > /dev/null
python tools/pmc_to_json.py gpurun_out/$TAG k_spatial_sweep gpurun_out/$TAG/sweep_pmc.json "same command: 6 launches of k_spatial_sweep<true,1>" > /dev/null
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/overlap -o t -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-real-pair --no-kernel-timing > /dev/null 2>&1
cd $R
python tools/overlap_timeline.py $O/overlap/t_kernel_trace.csv > $O/overlap_inflight.txt
head -5 $O/trace/t_kernel_stats.csv | cut -c1-150
grep -E "launches_profiled|valu_winstr_per_launch|lds_bank|launch_ns" $O/refine_pmc.json $O/sweep_pmc.json
