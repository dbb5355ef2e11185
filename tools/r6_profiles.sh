#!/bin/bash
# end of round 6: kernel stats, PMC traffic and SQ wave states of the EHEM L16-m frame and of the OctAttention L14 frame (tools/r5_profiles.sh), the
# decoder's kernel stats, and the CPU port timed on a WHOLE frame next to its bounded sample on the same box
cd "${GRAFT_REPO_ROOT:?}" || exit 1
python scp_amd/build.py > /dev/null 2>&1
bash tools/r5_profiles.sh r6 ehem octattn
O=$GRAFT_REPO_ROOT/gpurun_out/r6
cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py --steps 6 --warmup 2 --no-legs --no-strict-leg --cpu-baseline full > $O/r6_bench_cpu_full.log 2>&1; grep '^{' $O/r6_bench_cpu_full.log | tail -1 > $O/r6_bench_cpu_full.json
timeout 600 python3 bench.py --steps 6 --warmup 2 --no-legs --no-strict-leg > $O/r6_bench_cpu_sampled.log 2>&1; grep '^{' $O/r6_bench_cpu_sampled.log | tail -1 > $O/r6_bench_cpu_sampled.json
python3 - <<PY
import json
for n in ("full", "sampled"):
    try:
        z = json.loads(open("$O/r6_bench_cpu_%s.json" % n).read())["cpu_baseline"]
        print(n, z["seconds_per_frame"], z.get("full_window_cost_s"), z.get("full_window_s"), z.get("sampled_over_full_frame_recorded"))
    except Exception as e: print(n, "failed", e)
PY
ls $O
