#!/bin/bash
# throughput vs sweep workgroups per CU, pairs in flight and wave priority of the sweep kernel (GPU box)
for lib in libcspm_hip.so libcspm_prio0.so; do for wg in 1 2 3; do for n in 3 4; do
  CSPM_LIB=$PWD/crossscalepatchmatch_amd/$lib CSPM_SWEEP_WG=$wg python bench.py --no-cpu-baseline --in-flight $n --steps 9 --warmup 3 2>/dev/null | python tools/bench_brief.py "$lib sweep_wg=$wg"
done; done; done
