"""Per-kernel achieved HBM GB/s of stage G (transform -> sort -> occupancy -> context) from a rocprofv3 kernel-stats CSV of
tools/run_frame.py 16 1 <frames>: python tools/stage_g_table.py <kernel_stats.csv> <frames> > profiles/<tag>_stage_g_passes.md
Bytes per call are the algorithmic reads + writes of the kernel at the L16 --spher --mullevel workload (3 shells x 120 000 points,
577 515 nodes)."""
import csv, sys
f, nf = sys.argv[1], int(sys.argv[2])
# optional: points per frame, shells per frame, nodes per frame, frames per scp_geom_build call (round 3: batched L12 frames)
P, S, N, FB = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (120000, 3, 577515, 1)
K = P * S * FB          # keys per build
N = N * FB              # nodes per build
nf = nf / FB            # builds
B = {   # kernel name fragment -> (bytes per call, what moves)
    "front_transform_kernel": (24 * P * FB, "xyz f32 in, (rho, phi, theta | z) f32 out, every frame of the build, each point once"),
    "front_key_kernel": (12 * K + 8 * K, "transformed point in (L2: a point is read once per shell), 64-bit key out + the sort's first digit histogram"),
    "ctx_ehem_all_kernel": (37 * N, "ancestor gathers in, 12 B context + 12 B position + 1 B coded symbol out, every segment"),
    "transform_kernel": (24 * P, "xyz f32 in, (rho, phi, theta) f32 out, one shell of one frame (scp_quantize path)"),
    "quantize_kernel": (24 * P, "f32 in, int32 out, one shell of one frame (scp_quantize path)"),
    "seg_minmax_kernel": (12 * K, "int32 coordinates in"),
    "morton_key_kernel": (20 * K, "int32 x3 in, 64-bit key out, all shells"),
    "radix_hist_kernel": (8 * K, "keys in (digit histogram)"),
    "radix_scatter_kernel": (16 * K, "keys in, keys out (stable scatter)"),
    "tree_count_kernel": (8 * K, "sorted keys in"),
    "tree_segrank_kernel": (8 * K, "sorted keys in"),
    "tree_write_kernel": (8 * K + 22 * N, "sorted keys in, node tables out"),
    "tree_occ_kernel": (10 * N, "octant / parent in, occupancy out"),
    "ctx_ehem_kernel(": (int(37 * N / (S * FB)), "ancestor gathers in, 12 B context + 12 B position + 1 B symbol out, one segment"),
}
rows = list(csv.DictReader(open(f)))
print("| kernel | calls / frame | avg µs | algorithmic MB / call | achieved GB/s | moves |")
print("|---|---|---|---|---|---|")
tot_t = tot_b = 0.0
for r in rows:
    for k, (b, what) in B.items():
        if r["Name"].startswith(k):
            us = float(r["AverageNs"]) / 1e3
            calls = int(r["Calls"]) / nf
            print(f"| `{k}` | {calls:.0f} | {us:.1f} | {b / 1e6:.2f} | {b / us / 1e3:.0f} | {what} |")
            tot_t += us * calls
            tot_b += b * calls
scan = [r for r in rows if "radix_scan_kernel" in r["Name"]]
if scan:
    us = float(scan[0]["AverageNs"]) / 1e3
    calls = int(scan[0]["Calls"]) / nf
    print(f"| `radix_scan_kernel` | {calls:.0f} | {us:.1f} | (256 x blocks counters) | - | exclusive scan of the digit counters: latency-bound, not bandwidth |")
    tot_t += us * calls
print()
unit = "build of %d frames" % FB if FB > 1 else "frame"
print(f"Sum over the listed kernels: {tot_t / 1e3:.2f} ms of kernel time per {unit}, {tot_b / 1e6:.0f} MB moved = {tot_b / tot_t / 1e3:.0f} GB/s "
      f"({100 * tot_b / tot_t / 1e3 / 8000:.1f} % of 8 TB/s).  SURVEY.md 8d's algorithmic figure for the whole stage is 12 P + 25 N = "
      f"{(12 * P * FB + 25 * N) / 1e6:.1f} MB per {unit}: bytes moved / algorithmic = {tot_b / (12 * P * FB + 25 * N):.1f}.")
