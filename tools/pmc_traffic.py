"""Summarise the two PMC passes of tools/pmc_traffic.sh into profiles/<tag>_pmc_traffic.json (HBM bytes per launch per kernel).
FETCH_SIZE is doubled (gfx950 reports half of the bytes of wide coalesced reads, MI355X_MICROARCH.md); both counters are in KB."""
import csv, glob, json, os, sys, collections
tag = sys.argv[1]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob(f"{ROOT}/gpurun_out/pmc_{tag}_{c}/*/*counter_collection.csv"), key=os.path.getmtime)   # newest run
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    tot[c] = acc
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of tools/run_frame.py 16 1 2 (2 frames, L16 "
               "mullevel); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); units KB -> bytes",
       "per_kernel": {}}
for k in tot["FETCH_SIZE"]:
    n = tot["FETCH_SIZE"][k][0]
    fb = 2.0 * 1024 * tot["FETCH_SIZE"][k][1]
    wb = 1024.0 * tot["WRITE_SIZE"].get(k, [0, 0.0])[1]
    out["per_kernel"][k] = {"launches": n, "hbm_bytes_per_launch": (fb + wb) / n, "fetch_bytes_corrected": fb, "write_bytes": wb}
g = [v for k, v in out["per_kernel"].items() if "gemm_split_kernel" in k]
out["gemm_split_all_variants"] = {"launches": sum(v["launches"] for v in g),
                                  "hbm_bytes_per_launch": sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in g) / max(1, sum(v["launches"] for v in g))}
for name in ("rc_post_attn_kernel", "rc_ln_linear_kernel"):
    g2 = [v for k, v in out["per_kernel"].items() if name in k]
    if g2:
        out[name] = {"launches": sum(v["launches"] for v in g2),
                     "hbm_bytes_per_launch": sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in g2) / max(1, sum(v["launches"] for v in g2))}
frames = 2
out["frame_hbm_bytes"] = sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in out["per_kernel"].values()) / frames
big = dict(sorted(out["per_kernel"].items(), key=lambda kv: -(kv[1]["fetch_bytes_corrected"] + kv[1]["write_bytes"]))[:14])
out["per_kernel"] = big
json.dump(out, open(f"{ROOT}/profiles/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["gemm_split_all_variants"]), "frame HBM traffic %.1f GB" % (out["frame_hbm_bytes"] / 1e9))
for k, v in big.items():
    print(f"{k[:60]:60s} {v['launches']:5d} {v['hbm_bytes_per_launch']/1e6:10.1f} MB/launch")
