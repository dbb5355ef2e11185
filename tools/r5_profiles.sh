#!/bin/bash
# usage (on the GPU box): bash tools/r5_profiles.sh <tag> [what...]   what: ehem octattn stageg bench (default: all)
# Everything the round's profiles/ entries come from, into gpurun_out/<tag>/ : rocprofv3 kernel stats, PMC traffic (two separate --pmc
# passes, no tracing options beside them), SQ wave states - for the EHEM L16-m frame AND for the OctAttention L14 --cylin frame -, the
# stage-G pass tables (L12 x1 / x4, L16-m, F17-m) and the five bench lines.
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; T=$1; shift; W=${@:-ehem octattn stageg bench}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() {   # name, script args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$name -- python3 "$@" > $O/ks_$name.log 2>&1
  cp $O/ks_$name/*/*_kernel_stats.csv $O/${T}_${name}_kernel_stats.csv 2>/dev/null; rm -rf $O/ks_$name
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${name}_$c -- python3 "$@" > $O/pmc_${name}_$c.log 2>&1
  done
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_${name}_sq -- python3 "$@" > $O/pmc_${name}_sq.log 2>&1
  python3 $R/tools/r4_pmc_summary.py $O $T $name
}
for w in $W; do
  case $w in
    ehem) prof ehem_L16m_frame $R/tools/run_frame.py 16 1 3 ;;
    octattn) prof octattn_L14_frame $R/tools/run_octattn.py 14 1 ;;
    stageg)
      for cfg in "1 20 12 0 0 L12_x1" "4 20 12 0 0 L12_x4" "16 10 12 0 0 L12_x16" "1 20 16 1 0 L16m_x1" "1 20 17 1 1 F17m_x1"; do
        set -- $cfg
        rocprofv3 --kernel-trace --stats --output-format csv -d $O/g_$6 -- python3 $R/tools/run_geom_batch.py $1 $2 $3 $4 $5 > $O/g_$6.log 2>&1
        cp $O/g_$6/*/*_kernel_stats.csv $O/${T}_stageg_$6.csv 2>/dev/null; rm -rf $O/g_$6
        python3 $R/tools/run_geom_batch.py $1 $2 $3 $4 $5 > $O/g_$6_plain.log 2>&1      # un-profiled wall time
        tail -2 $O/g_$6.log | cut -c1-200; tail -1 $O/g_$6_plain.log | cut -c1-200
      done ;;
    bench)
      cd $R
      timeout 1800 python3 bench.py > $O/${T}_bench.log 2>&1; grep '^{' $O/${T}_bench.log | tail -1 > $O/${T}_bench.json; cut -c1-200 $O/${T}_bench.json
      timeout 1500 python3 bench.py --all-configs --no-cpu-baseline --out-dir $O --tag $T > $O/${T}_all_configs.log 2>&1; cut -c1-160 $O/${T}_all_configs.log
      timeout 300 python3 bench.py --decode > $O/${T}_decode.log 2>&1; grep '^{' $O/${T}_decode.log | tail -1 > $O/${T}_bench_decode.json
      cd /tmp ;;
  esac
done
