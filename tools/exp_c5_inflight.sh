O=gpurun_out/r06_c5; mkdir -p $O
for cfg in "1 2" "0 3" "1 3" "1 4" "0 2"; do set -- $cfg
  CSPM_SWEEP_FOLD=$1 CSPM_SWEEP_WG=$2 python bench.py --config C5 --no-cpu-baseline --no-real-pair --steps 3 --warmup 2 > $O/bench_c5_fold$1_wg$2.json 2>/dev/null
  python tools/bench_brief.py "C5 fold=$1 wg=$2" < $O/bench_c5_fold$1_wg$2.json | cut -c1-80 | tee -a $O/bench_c5_matrix.txt
done
