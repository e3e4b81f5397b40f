#!/bin/bash
# time_phases for every alternative build crossscalepatchmatch_amd/libv_*.so (tuning experiments; GPU box)
for lib in crossscalepatchmatch_amd/libv_*.so; do
  echo "$lib: $(CSPM_LIB=$PWD/$lib python tools/time_phases.py ${1:-C3} ${2:-1} 2>&1 | tail -1)"
done
