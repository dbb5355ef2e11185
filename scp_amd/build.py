"""Build recipe for libscp_hip.so (hipcc, gfx950 only, in-tree so the .so travels with the snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libscp_hip.so")
SOURCES = ["api.cpp", "geom.hip", "sort_u64.hip", "cdf.hip", "rangecoder.cpp", "legacy_octree.cpp",
           "knn.hip", "edge.hip", "attn.hip", "octattn.hip", "octattn_f16.hip", "octattn_embed.hip", "gemm.hip", "gemm_split.hip", "rowchain.hip", "fused.hip", "metrics.hip", "plan.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-fvisibility=hidden", "-x", "hip"]


# bit-exact kernels: numpy / torch evaluate a*a + b*b as two roundings, so FMA contraction must be off there
# (HIP's __fmul_rn / __fadd_rn are plain operators and do NOT stop the contraction).
EXTRA = {"geom.hip": ["-ffp-contract=off"], "knn.hip": ["-ffp-contract=off"], "cdf.hip": ["-ffp-contract=off"],
         "edge.hip": ["-ffp-contract=off"], "metrics.hip": ["-ffp-contract=off"], "octattn_embed.hip": ["-ffp-contract=off"]}
if os.environ.get("SCP_ATTN_DEFS"):    # experiment builds of csrc/attn.hip (e.g. SCP_ATTN_DEFS="-DPNS=3")
    EXTRA["attn.hip"] = os.environ["SCP_ATTN_DEFS"].split()
if os.environ.get("SCP_KNN_DEFS"):     # experiment builds of csrc/knn.hip (e.g. SCP_KNN_DEFS="-DKNN_INSERT_CHAIN")
    EXTRA["knn.hip"] = EXTRA["knn.hip"] + os.environ["SCP_KNN_DEFS"].split()
if os.environ.get("SCP_RC_DEFS"):      # experiment builds of csrc/rowchain.hip (e.g. SCP_RC_DEFS="-DRC_WAIT0")
    EXTRA["rowchain.hip"] = os.environ["SCP_RC_DEFS"].split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "scp.h")]
    objs = []
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(CSRC, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [HIPCC] + FLAGS + EXTRA.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        # the command line is part of an object's identity (SCP_RC_DEFS probe builds must not survive into the next plain build)
        stamp, line = o + ".cmd", " ".join(cmd)
        same_cmd = os.path.exists(stamp) and open(stamp).read() == line
        if force or not same_cmd or _stale(o, [s] + hdrs):
            # the stamp names the command an object was BUILT with: it is removed (with the object) before the compile starts and
            # written only once hipcc has succeeded, so an interrupted or failed probe build can never be linked by the next plain build
            for stale in (stamp, o):
                if os.path.exists(stale):
                    os.remove(stale)
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((s, subprocess.Popen(cmd), stamp, line))
    bad = []
    for s, p, stamp, line in procs:
        if p.wait() != 0:
            bad.append(s)
            continue
        with open(stamp, "w") as f:
            f.write(line)
    if bad:
        raise RuntimeError("hipcc failed for: " + ", ".join(bad))
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
