#!/bin/bash
# throughput vs sweep workgroups per CU and pairs in flight (GPU box)
for wg in 1 2 3; do for n in 2 3 4; do
  CSPM_SWEEP_WG=$wg python bench.py --no-cpu-baseline --in-flight $n --steps 9 --warmup 3 2>/dev/null | python tools/bench_brief.py "sweep_wg=$wg"
done; done
