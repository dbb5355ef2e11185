#!/bin/bash
# usage (on the GPU box): bash tools/pmc_sq.sh <tag>
# One counter pass (SQ wave-state counters) over 1 frame of tools/run_frame.py; no tracing options next to --pmc.
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_$1_sq -- python3 $R/tools/run_frame.py 16 1 1 > $R/gpurun_out/pmc_$1_sq.log 2>&1
tail -3 $R/gpurun_out/pmc_$1_sq.log | cut -c1-200
python3 $R/tools/pmc_sq.py $1
