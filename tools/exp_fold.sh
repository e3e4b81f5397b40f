O=gpurun_out/r06_fold; mkdir -p $O
python -m pytest tests/test_gpu_patchmatch.py tests/test_gpu_adversarial.py -m gpu -x -q 2>&1 | tail -3
for f in 0 1; do CSPM_SWEEP_FOLD=$f python tools/time_phases.py C3 3 | tail -1; done
for f in 0 1; do CSPM_SWEEP_FOLD=$f python tools/corun_probe.py C3 5 > $O/corun_fold$f.txt 2>&1; cat $O/corun_fold$f.txt; done
CSPM_SWEEP_FOLD=1 CSPM_SWEEP_WG=3 python tools/corun_probe.py C3 5 > $O/corun_fold1_wg3.txt 2>&1; cat $O/corun_fold1_wg3.txt
for f in 0 1; do for wg in 1 2 3; do
  CSPM_SWEEP_FOLD=$f CSPM_SWEEP_WG=$wg python bench.py --no-cpu-baseline --no-real-pair --steps 12 --warmup 3 > $O/bench_fold${f}_wg$wg.json 2>/dev/null
  python tools/bench_brief.py fold${f}_wg$wg < $O/bench_fold${f}_wg$wg.json | cut -c1-90
done; done
CSPM_SWEEP_FOLD=1 python bench.py --no-cpu-baseline --no-real-pair --in-flight 1 > $O/bench_fold1_inflight1.json 2>/dev/null; python tools/bench_brief.py fold1_if1 < $O/bench_fold1_inflight1.json | cut -c1-90
CSPM_SWEEP_FOLD=1 CSPM_SWEEP_WG=2 python bench.py --no-cpu-baseline --no-real-pair --in-flight 2 --steps 12 --warmup 2 > $O/bench_fold1_inflight2.json 2>/dev/null; python tools/bench_brief.py fold1_if2 < $O/bench_fold1_inflight2.json | cut -c1-90
