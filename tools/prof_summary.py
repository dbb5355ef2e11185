import csv, glob, sys
tag = sys.argv[1]; nf = int(sys.argv[2]) if len(sys.argv) > 2 else 4; top = int(sys.argv[3]) if len(sys.argv) > 3 else 16
f = glob.glob(f'{ROOT}/gpurun_out/prof_{tag}/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
print(sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / nf, 'ms/frame total kernel time')
for r in rows[:top]:
    print(f"{r['Name'][:64]:64s} {int(r['Calls'])/nf:7.1f} {float(r['TotalDurationNs'])/1e6/nf:8.2f} ms/frame avg {float(r['AverageNs'])/1e3:8.1f} us")
