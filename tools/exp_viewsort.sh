O=gpurun_out/r06_viewsort; mkdir -p $O
python -m pytest tests/test_gpu_patchmatch.py tests/test_gpu_fullsize.py -m gpu -x -q -k "view or ties or adversarial" 2>&1 | tail -4
python -m pytest tests/test_gpu_adversarial.py -m gpu -x -q 2>&1 | tail -3
for vs in 0 1; do
  CSPM_VIEW_SORT=$vs python tools/row_stats.py C3 2>&1 | grep -E "view propagation|view [0-9] +view" > $O/row_stats_c3_sort$vs.txt
  CSPM_VIEW_SORT=$vs python tools/row_stats.py real 2>&1 | grep -E "view propagation|view [0-9] +view" > $O/row_stats_real_sort$vs.txt
  CSPM_VIEW_SORT=$vs python tools/time_phases.py C3 3 > $O/time_phases_sort$vs.txt 2>&1
done
grep "level 0" $O/row_stats_c3_sort0.txt; grep "level 0" $O/row_stats_c3_sort1.txt; grep "level 0" $O/row_stats_real_sort0.txt; grep "level 0" $O/row_stats_real_sort1.txt
tail -3 $O/time_phases_sort0.txt; tail -3 $O/time_phases_sort1.txt
python bench.py --no-cpu-baseline > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err; python - <<'PY'
import json; d=json.load(open("gpurun_out/r06_bench_b.json")); print(d["value"], d["ms_per_step"], d["sweep_fallbacks"], d["volume_fallbacks"], d["table_volumes_active"], {k:round(v["avg_launch_ms"],2) for k,v in d["roofline"]["kernels"].items()}, d.get("real_pair_bad2"))
PY
