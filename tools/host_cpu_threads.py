"""Where does a rank's host CPU go?  Runs the bench loop (pipelined encode, L16-m) and prints CPU seconds per OS thread of the process
(/proc/self/task/*/stat: utime + stime), before / after, per frame.  python tools/host_cpu_threads.py [frames]"""
import os, sys, time, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.synth import synth_frame

def snap():
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open(f"/proc/self/task/{tid}/stat").read()
            name = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(f[11]) + int(f[12])) / tck)
        except Exception:
            pass
    return out

strict = "--strict" in sys.argv          # the strict-identity front end (host_ints on one prefetch thread, as bench.py's second leg)
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
nf = int(argv[0]) if argv else 24
dev = torch.device("cuda:0")
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
enc = FrameEncoder(model, "kitti", 16, spher=True, mullevel=True, device=dev)
for a_ in sys.argv:
    if a_.startswith("--rc-grid="):         # persistent workgroups of the row-chain launches (default: one per CU)
        from scp_amd import native
        native.lib().scp_rc_set_grid(int(a_.split("=")[1]))
frames_h = [synth_frame(i) for i in range(nf + 2)]
frames = [torch.from_numpy(f).to(dev) for f in frames_h]
from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(max_workers=1)
futs, nxt = {}, 2
def ints_of(i):
    global nxt
    if not strict:
        return None
    while nxt < nf + 2 and nxt <= i + 2:
        futs[nxt] = pool.submit(enc.host_ints, frames_h[nxt]); nxt += 1
    return futs.pop(i).result()
for i in range(2):
    enc.finish(enc.encode_async(frames[i]))
torch.cuda.synchronize()
prof = None
if "--cprofile" in sys.argv:
    import cProfile
    prof = cProfile.Profile(); prof.enable()
a = snap(); t0 = time.perf_counter(); tt0 = time.thread_time()
pending = []
front = "--front" in sys.argv            # FrameEncoder.front_async one frame ahead (what bench.py does)
nxt_f = enc.front_async(frames[2], ints=ints_of(2)) if front else None
for i in range(2, nf + 2):
    if front:
        cur_f, nxt_f = nxt_f, (enc.front_async(frames[i + 1], ints=ints_of(i + 1)) if i + 1 < nf + 2 else None)
        pending.append(enc.encode_async(frames[i], front=cur_f))
    else:
        pending.append(enc.encode_async(frames[i], ints=ints_of(i)))
    if len(pending) > 4:
        enc.finish(pending.pop(0))
for h in pending:
    enc.finish(h)
torch.cuda.synchronize()
dt = time.perf_counter() - t0; tt = time.thread_time() - tt0
if prof is not None:
    import pstats
    prof.disable(); pstats.Stats(prof).sort_stats("tottime").print_stats(18)
b = snap()
print(f"{nf / dt:.2f} frames/s, {1e3 * dt / nf:.1f} ms per frame; launch thread (this one, tid {threading.get_native_id()}): {1e3 * tt / nf:.1f} ms CPU per frame")
rows = sorted(((b[t][1] - a.get(t, (None, 0.0))[1], t, b[t][0]) for t in b), reverse=True)
for d, t, name in rows[:12]:
    print(f"  tid {t:8d} {name:20s} {1e3 * d / nf:8.1f} ms CPU per frame")
print(f"  total {1e3 * sum(r[0] for r in rows) / nf:.1f} ms CPU per frame over {len(rows)} threads")
