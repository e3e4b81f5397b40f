#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
R=$PWD
O=$R/gpurun_out/r04h5; mkdir -p $O
{
python tools/time_phases.py C3 3
for rep in 1 2 3; do for F in 2 3; do
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --in-flight $F | python tools/bench_brief.py rep$rep
done; done
} > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt
