O=gpurun_out/r06_wpl; mkdir -p $O
L=$PWD/crossscalepatchmatch_amd/libv_wpl2_minw2.so
python tools/time_phases.py C3 3 | tail -1 | sed "s/^/wpl=1 wg=2: /" | tee -a $O/time_phases.txt
for wg in 1 2 3; do CSPM_LIB=$L CSPM_SWEEP_WG=$wg python tools/time_phases.py C3 3 | tail -1 | sed "s/^/wpl=2 wg=$wg: /" | tee -a $O/time_phases.txt; done
for wg in 1 2; do CSPM_LIB=$L CSPM_SWEEP_FOLD=0 CSPM_SWEEP_WG=$wg python bench.py --no-cpu-baseline --no-real-pair --steps 20 --warmup 5 2>/dev/null | python tools/bench_brief.py "wpl=2 wg=$wg if2" | cut -c1-60 | tee -a $O/bench.txt; done
CSPM_LIB=$L CSPM_SWEEP_FOLD=0 CSPM_SWEEP_WG=1 python bench.py --no-cpu-baseline --no-real-pair --steps 20 --warmup 5 --in-flight 3 2>/dev/null | python tools/bench_brief.py "wpl=2 wg=1 if3" | cut -c1-60 | tee -a $O/bench.txt
CSPM_LIB=$L CSPM_SWEEP_WG=2 python tools/corun_probe.py C3 5 | tee -a $O/corun.txt
