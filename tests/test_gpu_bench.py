"""bench.py as the driver runs it: the N-rank line produced by `bench.py --gpus N` itself, on a one-GPU box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=1500):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` (no torchrun around it) starts two ranks as a child launcher before touching the GPU; here both
    ranks share the box's one GPU (SCP_FORCE_DEVICE=0) and reduce over gloo.  The line says n_gpus 2, the whole-job rate counts both
    ranks' frames, and the frame every rank encodes with the shared seed gives byte-identical streams."""
    out = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--config", "ehem-L12-s"],
               {"SCP_FORCE_DEVICE": "0", "SCP_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["rccl_world"] == 2 and out["scaling"] == "weak"
    assert out["steps"] == 4 and abs(out["value"] - 2 * 4 / (out["ms_per_step"] * 4e-3)) < 1e-6 * out["value"]
    r = out["ranks"]
    assert len(r["per_rank"]) == 2 and {x["rank"] for x in r["per_rank"]} == {0, 1}
    assert r["shared_frame_streams_identical"]
    assert r["fps_min"] > 0 and r["host_cpu_ms_per_frame_max"] > 0
    assert "roofline" in out and out["roofline"]["frac"] > 0


def test_bench_single_rank_line_has_the_contract_fields():
    out = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--config", "ehem-L12-s"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in out
    assert out["n_gpus"] == 1 and out["rccl_world"] == 1 and "ranks" not in out
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "valid", "kernel_ms_sum", "kernel_ms_bound"} <= set(out["roofline"])
    # the self-check of the live roofline (VERDICT r3 item 1 iv): the event-timed kernels of a frame cannot take longer than the synchronised
    # wall time of the same frame's model stage - if they do, the brackets measured host time and the line says so
    r = out["roofline"]
    assert r["valid"] is True and 0 < r["kernel_ms_sum"] <= 1.02 * r["kernel_ms_bound"]
    assert 0 < r["frac"] < 1 and r["traffic"] is None              # PMC traffic is quoted for the configuration it was measured on only
    ks = out["roofline_kernels"]
    assert "rc_post_attn_kernel" in ks and "swin_attn_planes_kernel" in ks and "knn_f16x3_wg256_kernel" in ks
    assert abs(sum(k["total_ms_per_frame"] for n, k in ks.items() if n not in ("cdf_kernel", "stage_G_front_and_context_kernels")) - r["kernel_ms_sum"]) < 1e-6 * r["kernel_ms_sum"]
    for k in ks.values():
        assert k["launches_per_frame"] >= 1 and k["avg_launch_us"] > 0 and ("frac" not in k or 0 < k["frac"] < 1)
    f = out["roofline_frame"]
    assert f["flops"] > 1e12 and 0 < f["frac_mfma"] < 1 and f["attention_flops"] < f["flops"]
    # round 5: the headline runs in strict-identity mode (the reference's own float -> integer arithmetic, 0 differing points asserted in the run),
    # the device transform is the second leg
    assert "strict identity" in out["config"]["transform"] and out["strict_identity_verified"] is True and out["strict_identity_fps"] == out["value"]
    s = out["device_transform"]
    assert s["fps"] > 0 and out["device_transform_fps"] == s["fps"] and abs(s["bpp_mean"] - out["bpp_mean"]) < 0.05 * out["bpp_mean"]
    tp = out["transform_parity"]
    assert tp["host_transform"] == [0] and tp["device_transform"][0] > 0       # L12 --spher: the device transform misses 853 of the reference's points
    assert out["config"]["rank_cores"] >= 1
    k = out["roofline_kernels"]["edge_gather_max_kernel"]
    assert k["bound"] == "hbm" and 0 < k["frac"] < 1
    assert "configs" not in out                                                 # side legs belong to the headline configuration only


def test_bench_device_transform_headline_and_strict_leg():
    out = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--config", "ehem-L12-s", "--device-transform"])
    assert out["config"]["transform"] == "device" and out["strict_identity"]["fps"] == out["strict_identity_fps"] and out["device_transform_fps"] == out["value"]
    tp = out["transform_parity"]
    assert tp["host_transform"] == [0] and tp["device_transform"][0] > 0       # L12 --spher: the device transform misses 853 of the reference's points


def test_bench_side_legs_fill_configs_decode_and_cli():
    """The default run of the headline configuration adds the other four BASELINE.json workloads, the decoder and the drop-in CLI as child
    processes behind the headline: five `configs` entries, `decode`, `cli`, `cli_over_bench`."""
    out = _run(["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-strict-leg", "--leg-steps", "2", "--leg-warmup", "1"], timeout=2400)
    assert out["strict_identity_verified"] is True and out["transform_parity"]["host_transform"] == [0, 0, 0]
    assert set(out["configs"]) == {"ehem-L16-m", "ehem-L12-s", "ehem-F17-m", "octattn-L12-spher", "octattn-L14-cylin"}, out.get("legs_error")
    for name, c in out["configs"].items():
        assert "error" not in c, (name, c)
        assert c["fps"] > 0 and c["ms_per_step"] > 0 and 0 < c["roofline"]["frac"] < 1 and c["roofline"]["valid"] is True, (name, c)
    assert out["configs"]["ehem-L12-s"]["steps"] == 32 and out["configs"]["ehem-L12-s"]["strict_identity_verified"] is True
    assert out["configs"]["ehem-L16-m"]["fps"] == out["value"]
    d = out["decode"]
    assert d["decoded_occupancy_equals_encoded"] is True and d["fps"] > 0
    c = out["cli"]
    assert c["streams_written"] == c["files"] == 9 and c["files_timed"] == 4 and c["fps"] > 0 and abs(out["cli_over_bench"] - c["fps"] / out["value"]) < 1e-9
    for n in (2, 4):                                                            # independent decoder processes sharing the GPU, timed loops started together
        dn = out[f"decode_{n}_procs"]
        assert "error" not in dn, dn
        assert dn["decoded_occupancy_equals_encoded"] is True and dn["processes"] == n and dn["frames"] == 4 * n and len(dn["per_process_fps"]) == n
        assert dn["fps"] > 0 and dn["start_skew_ms"] < 500.0
    assert out["decode_2_procs"]["fps"] > 1.2 * d["fps"]                        # two streams decode faster than one (measured: 1.7 x)


def test_bench_decode_mode_times_the_decoder_and_checks_the_round_trip():
    out = _run(["--decode", "--steps", "1", "--warmup", "1", "--config", "ehem-L12-s"])
    assert out["decoded_occupancy_equals_encoded"] is True and out["value"] > 0 and "decode" in out["metric"]
    assert out["config"]["phase2_launch_sequences_per_frame"] >= 10


def test_four_ranks_on_one_gpu_keep_the_aggregate_rate():
    """The only proxy for host-side contention a one-GPU box offers: four ranks - four launch threads, four coder
    pools, four HIP runtimes - share ONE GPU (SCP_FORCE_DEVICE=0, gloo).  The GPU is the bottleneck either way, so the aggregate rate must
    stay at the one-rank rate; it falls when the ranks' host sides get in each other's way (cores, allocator locks, the GIL of one
    process is not shared here).  Each rank also reports its host CPU per frame: about 80 ms (1.2 cores) since round 4."""
    common = ["--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-strict-leg", "--no-legs", "--config", "ehem-L16-m"]
    one = _run(common)
    four = _run(["--gpus", "4"] + common, {"SCP_FORCE_DEVICE": "0", "SCP_DIST_BACKEND": "gloo"})
    assert four["n_gpus"] == 4 and four["ranks"]["shared_frame_streams_identical"]
    print(f"1 rank {one['value']:.2f} frames/s ({one['host_cpu_ms_per_frame']:.0f} ms host CPU per frame); 4 ranks on the same GPU {four['value']:.2f} frames/s "
          f"aggregate, per rank {four['ranks']['fps_min']:.2f} .. {four['ranks']['fps_max']:.2f}, host CPU per frame {four['ranks']['host_cpu_ms_per_frame_max']:.0f} ms")
    assert four["value"] >= 0.9 * one["value"], (four["value"], one["value"])
    assert one["host_cpu_ms_per_frame"] < 120.0


def test_eight_ranks_on_one_gpu_keep_the_aggregate_rate():
    """the host side of an 8-GPU node, on the one GPU a builder has - eight ranks (eight launch threads, eight coder pools,
    eight HIP runtimes, eight reader / prefetch threads) share ONE MI355X (SCP_FORCE_DEVICE=0, gloo), 24 timed frames each.  Measured
    (profiles/r6_eight_ranks_one_gpu.json): the aggregate rate is 0.86 - 0.92 of the one-rank rate - eight processes' hardware queues are
    time-sliced on one GPU, and a rank that waits for a device it shares with seven others spins in the runtime's synchronisation (400 - 560 ms of
    host CPU per frame = its wall time per frame; ONE rank alone: 17 ms of host CPU per 64 ms frame).  Neither effect exists with one device per
    rank; what the measurement does bound is the host: eight ranks' launch threads, coder pools and allocators run side by side on one box without
    starving the GPU (>= 0.75 of the one-rank rate asserted), every rank produces the byte stream of the shared frame, and a lone rank needs a quarter
    of a core.  Writes gpurun_out/eight_ranks_one_gpu.json."""
    common = ["--steps", "24", "--warmup", "4", "--no-cpu-baseline", "--no-strict-leg", "--no-legs", "--config", "ehem-L16-m"]
    one = _run(common)
    eight = _run(["--gpus", "8"] + common, {"SCP_FORCE_DEVICE": "0", "SCP_DIST_BACKEND": "gloo"})
    assert eight["n_gpus"] == 8 and eight["ranks"]["shared_frame_streams_identical"] and len(eight["ranks"]["per_rank"]) == 8
    rec = dict(one_rank_fps=one["value"], one_rank_host_cpu_ms_per_frame=one["host_cpu_ms_per_frame"], eight_ranks_aggregate_fps=eight["value"],
               ratio=eight["value"] / one["value"], per_rank=eight["ranks"]["per_rank"], host_cores=os.cpu_count(),
               note="eight ranks of bench.py --gpus 8 on ONE MI355X (SCP_FORCE_DEVICE=0, gloo): per-rank frames/s and host CPU-ms per frame")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "eight_ranks_one_gpu.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(f"1 rank {one['value']:.2f} frames/s; 8 ranks on the same GPU {eight['value']:.2f} aggregate ({rec['ratio']:.3f} x), per rank "
          f"{eight['ranks']['fps_min']:.2f} .. {eight['ranks']['fps_max']:.2f}, host CPU per frame {eight['ranks']['host_cpu_ms_per_frame_min']:.0f} .. "
          f"{eight['ranks']['host_cpu_ms_per_frame_max']:.0f} ms")
    assert eight["value"] >= 0.75 * one["value"], (eight["value"], one["value"])
    assert one["host_cpu_ms_per_frame"] < 90.0
