"""The bench line's live launch brackets against a rocprofv3 --kernel-trace --stats summary of the same box:
python tools/compare_roofline.py <bench.json> <frame_kernel_stats.csv> [frames in the trace] > profiles/<tag>_roofline_vs_rocprof.md
(`roofline.frac` must be within +-5 % of the rocprof summary for every entry of `roofline_kernels`.)"""
import csv, json, sys
b = json.load(open(sys.argv[1]))
rows = list(csv.DictReader(open(sys.argv[2])))
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 4
match = {"rc_post_attn_kernel": ["rc_post_attn_kernel"], "swin_attn_planes_kernel": ["swin_attn_planes_kernel"], "rc_ln_linear_kernel": ["rc_ln_linear_kernel"],
         "knn_f16x3_wg256_kernel": ["knn_f16x3_wg256_kernel"], "gemm_split_kernel": ["gemm_split_kernel"], "rc_edge_mlp_kernel": ["rc_edge_mlp_kernel"],
         "rc_merge_kernel": ["rc_merge_kernel"], "edge_gather_max_kernel": ["edge_gather_max_kernel"], "gemm_f32_kernel": ["gemm_f32_kernel"],
         "knn_mfma_kernel<2,16> (positions)": ["knn_mfma_kernel<2, 16"], "split_rows_kernel": ["split_rows_kernel"], "cdf_kernel": ["cdf_kernel"],
         "gemm_bf16x3_kernel": ["gemm_bf16x3_kernel"], "oa_attn_f16x3_kernel": ["oa_attn_f16x3_kernel"], "layernorm_*_kernel": ["layernorm_add_kernel", "layernorm_rows_kernel"]}
print(f"# Live launch brackets of `{sys.argv[1]}` (HIP events inside libscp_hip.so, per-launch minimum of 3 frames, un-profiled) against")
print(f"# `{sys.argv[2]}` (rocprofv3 --kernel-trace --stats of tools/run_frame.py on the same box, average over {nf} frames, profiled).")
print("# A profiled run is 2 - 3 % slower than an un-profiled one on this pool (MI355X_MICROARCH.md, DVFS give-back item 2).\n")
print("| kernel | launches / frame | live avg us | rocprof avg us | live / rocprof | live ms / frame | rocprof ms / frame |")
print("|---|---|---|---|---|---|---|")
for name, e in b["roofline_kernels"].items():
    pats = match.get(name)
    if not pats:
        continue
    sel = [r for r in rows if any(p in r["Name"] for p in pats)]
    if not sel:
        continue
    calls = sum(int(r["Calls"]) for r in sel)
    tot = sum(float(r["TotalDurationNs"]) for r in sel)
    avg = tot / calls / 1e3
    print(f"| `{name}` | {e['launches_per_frame']} ({calls / nf:.1f}) | {e['avg_launch_us']:.1f} | {avg:.1f} | {e['avg_launch_us'] / avg:.3f} | {e['total_ms_per_frame']:.2f} | {tot / nf / 1e6:.2f} |")
r = b["roofline"]
print(f"\nDominant kernel: live {r['avg_launch_us']:.1f} us per launch -> {r['achieved']:.1f} TFLOP/s algorithmic = {r['frac']:.3f} of {r['peak']:.1f}; self-check "
      f"`valid` = {r['valid']} ({r['kernel_ms_sum']:.1f} ms of event-timed kernels <= 1.02 x {r['kernel_ms_bound']:.1f} ms model stage wall).")
