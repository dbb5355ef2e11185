"""The drop-in CLI timed like bench.py: N synthetic KITTI-layout .bin files through `encode_mullevel.py --spher --lidar_level 16`
(file reads and parsing included), frames/s from the per-frame `time(s)` lines (completion intervals of the pipelined path) after
the first three frames, next to `bench.py`'s number from the same box.   python tools/bench_cli.py [frames]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from scp_amd.synth import synth_frame, write_kitti_bin
n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
with tempfile.TemporaryDirectory() as tmp:
    seq = os.path.join(tmp, "seq00"); os.makedirs(seq)
    for i in range(n):
        write_kitti_bin(os.path.join(seq, f"{i:06d}.bin"), synth_frame(i))
    cmd = [sys.executable, os.path.join(ROOT, "encode_mullevel.py"), "--test_files", os.path.join(seq, "*.bin"), "--type", "kitti", "--lidar_level", "16",
           "--spher", "--random_weights", "0", "--out_dir", os.path.join(tmp, "out")]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
    assert r.returncode == 0, r.stderr[-2000:]
    times = [float(l.split(":")[1]) for l in r.stdout.splitlines() if l.startswith("time(s)")]
    assert len(times) == n
    steady = times[3:]
    cli_fps = len(steady) / sum(steady)
b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline", "none", "--steps", str(n - 3), "--warmup", "3"], capture_output=True, text=True)
bench = json.loads(b.stdout.strip().splitlines()[-1])
out = dict(frames=n, cli_frames_per_s=cli_fps, cli_ms_per_frame=1e3 / cli_fps, bench_frames_per_s=bench["value"], bench_ms_per_frame=bench["ms_per_step"],
           cli_over_bench=cli_fps / bench["value"], note="CLI: .bin files read and parsed on a reader thread, frames pipelined three deep, .bin/.dat/.scp.json written; "
           "bench.py: frames resident in HBM, four in flight")
print(json.dumps(out))
