#!/bin/bash
# Round 6, job 11: two call chains side by side with the second one started a fraction of a call later (job 4 started them together: every stage met
# the same stage of the other chain — both bound by the same unit — and the pair took 1.92 x one call).
TAG=r06_j11
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 600 python profiles/r06_overlap_probe.py c2 20 2>/dev/null | tail -2 | tee $OUT/${TAG}_overlap_c2.txt
timeout 600 python profiles/r06_overlap_probe.py c3part 10 2>/dev/null | tail -2 | tee $OUT/${TAG}_overlap_c3part.txt
