"""Summarise tools/pmc_sq.sh: per kernel, wave-state fractions (SQ_* counters count quad-cycles; MFMA busy counts cycles)."""
import csv, glob, sys, collections, os
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = max(glob.glob(f"{root}/gpurun_out/pmc_{tag}_sq/*/*counter_collection.csv"), key=os.path.getmtime)   # newest run
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    acc[k]["n"] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:12]
print(f"{'kernel':44s} {'wave_cyc':>10s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'valu':>6s} {'mfma_busy/busy':>14s}")
for k, v in rows:
    wc = max(v["SQ_WAVE_CYCLES"], 1.0)
    print(f"{k:44s} {wc:10.3e} {v['SQ_WAIT_ANY']/wc:8.2f} {v['SQ_WAIT_INST_ANY']/wc:9.2f} {v['SQ_ACTIVE_INST_ANY']/wc:7.2f} "
          f"{v['SQ_ACTIVE_INST_VALU']/wc:6.2f} {v['SQ_VALU_MFMA_BUSY_CYCLES']/max(v['SQ_BUSY_CYCLES'],1.0):14.3f}")
