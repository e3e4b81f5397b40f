O=gpurun_out/r06_c5; mkdir -p $O
for f in 0 1; do for wg in 2 3 4 5; do
  CSPM_SWEEP_FOLD=$f CSPM_SWEEP_WG=$wg python tools/time_phases.py C5 1 2>&1 | tail -1 | sed "s/^/fold=$f wg=$wg /" | tee -a $O/time_phases_c5.txt
done; done
