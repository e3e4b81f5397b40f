O=gpurun_out/r06_corun; mkdir -p $O
python tools/corun_probe.py C3 5 > $O/default.txt 2>&1
for v in prio0 sleep8 prio0_sleep8; do CSPM_LIB=$PWD/crossscalepatchmatch_amd/libv_$v.so python tools/corun_probe.py C3 5 > $O/$v.txt 2>&1; done
CSPM_TABLE_VOLUMES=0 python tools/corun_probe.py C3 5 > $O/computed_tables.txt 2>&1
CSPM_SWEEP_WG=1 python tools/corun_probe.py C3 5 > $O/sweep_wg1.txt 2>&1
CSPM_SWEEP_WG=3 python tools/corun_probe.py C3 5 > $O/sweep_wg3.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_WAVES --output-format csv -d /tmp/pmcx -o p -- python $GRAFT_REPO_ROOT/tools/corun_probe.py C3 3 > $GRAFT_REPO_ROOT/$O/under_pmc.txt 2>&1)
cat $O/*.txt
python -m pytest tests/test_gpu_bench_ranks.py -m gpu -x -q 2>&1 | tail -5
python bench.py --no-cpu-baseline > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err; python tools/bench_brief.py a < gpurun_out/r06_bench_a.json
