#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r04e; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 ) > $O/pytest.txt 2>&1
tail -30 $O/pytest.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
CSPM_BENCH_FORCE_DIST=1 python bench.py --config C4 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c4_rccl.json 2> $O/bench_c4_rccl.err; python tools/bench_brief.py c4rccl < $O/bench_c4_rccl.json; tail -3 $O/bench_c4_rccl.err
