#!/bin/bash
# final validation of round 6: build, random encode -> decode sweep, the whole GPU suite, the bench line as the driver runs it
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6z; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python tools/roundtrip_sweep.py 36 > $O/roundtrip_sweep.txt 2>&1; tail -3 $O/roundtrip_sweep.txt
bash tools/r6_check.sh r6z all
