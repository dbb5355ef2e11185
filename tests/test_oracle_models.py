"""The floating-point oracle (oracle/models_ref.py) against logits produced by the reference modules, and the
product modules' state_dict layout against the reference's (drop-in checkpoint loading)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from cfgs import ehem_cfg, octattn_cfg
from conftest import GOLDEN, golden

TOL = 2e-5   # CPU fp32 vs CPU fp32: only op-fusion level differences are expected


@pytest.fixture(scope="module")
def ehem_sd():
    from scp_amd.models import EHEM
    from scp_amd.weights import fill_weights
    m = EHEM(ehem_cfg())
    fill_weights(m, 0)
    return m.state_dict()


@pytest.fixture(scope="module")
def oct_sd():
    from scp_amd.models import OctAttention
    from scp_amd.weights import fill_weights
    m = OctAttention(octattn_cfg())
    fill_weights(m, 0)
    return m.state_dict()


def test_state_dict_layout_matches_reference(ehem_sd, oct_sd):
    keys = json.load(open(os.path.join(GOLDEN, "state_keys.json")))
    for name, sd in (("EHEM", ehem_sd), ("OctAttention", oct_sd)):
        want = {k: (tuple(s), d) for k, s, d in keys[name]}
        got = {k: (tuple(v.shape), str(v.dtype)) for k, v in sd.items()}
        assert got == want, (set(want) ^ set(got))
    # the relative_position_index buffer is data, not a parameter: it must hold i - j + 511
    idx = ehem_sd["swin_self_transformer.layers.0.blocks.0.attention.self.relative_position_index"]
    assert idx[3, 5] == 509 and idx[511, 0] == 1022 and idx[0, 511] == 0


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "logits_ehem_*.npz"))))
def test_ehem_oracle_vs_reference_logits(ehem_sd, name):
    from oracle import models_ref
    z = golden(name)
    data = torch.from_numpy(z["data"].astype(np.int64))
    pos = torch.from_numpy(z["pos"])
    if data.dim() == 3:
        data, pos = data[None], pos[None]
    with torch.no_grad():
        o1, o2 = models_ref.ehem_forward(ehem_sd, data, pos)
    if "out1_sub" in z:
        st = int(z["stride"])
        assert np.abs(o1[0, ::st].numpy() - z["out1_sub"]).max() < TOL
        assert np.abs(o2[0, ::st].numpy() - z["out2_sub"]).max() < TOL
    else:
        w1, w2 = z["out1"], z["out2"]
        if w1.ndim == 2:
            w1, w2 = w1[None], w2[None]
        assert o1.shape == w1.shape and o2.shape == w2.shape
        assert np.abs(o1.numpy() - w1).max() < TOL
        if w2.size:
            assert np.abs(o2.numpy() - w2).max() < TOL


@pytest.mark.parametrize("name", ["tiefree_ehem_c600", "tiefree_ehem_c2049"])
def test_ehem_oracle_vs_reference_tiefree(ehem_sd, name):
    """Random float positions (no exactly tied neighbour distances): the oracle must pick the reference's neighbour SETS in all
    three searches (dgcnn.py:10-45) and reproduce every logit row."""
    from oracle import models_ref
    z = golden(name)
    data = torch.from_numpy(z["data"].astype(np.int64))[None]
    pos = torch.from_numpy(z["pos"])[None]
    seen = []
    models_ref.KNN_OVERRIDE = lambda x, k: (seen.append(models_ref.knn_default(x, k)) or seen[-1])
    try:
        with torch.no_grad():
            o1, o2 = models_ref.ehem_forward(ehem_sd, data, pos)
    finally:
        models_ref.KNN_OVERRIDE = None
    assert len(seen) == 3
    for i, idx in enumerate(seen):
        assert np.array_equal(np.sort(idx[0].numpy().astype(np.int16), axis=1), z[f"knn{i}"]), i
    st = int(z["stride"])
    assert np.abs(o1[0, ::st].numpy() - z["out1"]).max() < TOL
    assert np.abs(o2[0, ::st].numpy() - z["out2"]).max() < TOL


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "logits_octattn_*.npz"))))
def test_octattn_oracle_vs_reference_logits(oct_sd, name):
    from oracle import models_ref
    z = golden(name)
    with torch.no_grad():
        o = models_ref.octattn_forward(oct_sd, torch.from_numpy(z["data"].astype(np.int64))[None], torch.from_numpy(z["pos"])[None])
    assert np.abs(o[0].numpy() - z["out"]).max() < TOL


def _layer_sd(prefix_free_module, seed):
    from scp_amd.weights import fill_weights
    fill_weights(prefix_free_module, seed)
    return prefix_free_module.state_dict()


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "swin_*_s*_L*.npz"))))
def test_swin_layer_oracle_vs_reference(name):
    from oracle import models_ref
    from scp_amd.models.ehem import SwinLayer
    z = golden(name)
    _, kind, s, L = name.split("_")
    shift, L = int(s[1:]), int(L[1:])
    sd = {"x." + k: v for k, v in _layer_sd(SwinLayer(), int(z["wseed"])).items()}
    rng = np.random.default_rng(int(z["x_seed"]))
    x = rng.standard_normal((1, L, 256), dtype=np.float32)
    q = rng.standard_normal((1, L, 256), dtype=np.float32)
    assert np.array_equal(x[0, :2], z["x_head"])
    with torch.no_grad():
        y = models_ref.swin_layer(sd, "x", torch.from_numpy(x), L, shift, torch.from_numpy(q) if kind == "cross" else None)
    assert np.abs(y[0, ::int(z["stride"])].numpy() - z["y"]).max() < TOL


@pytest.mark.parametrize("L", [2, 3, 513])
def test_patch_merge_oracle_vs_reference(L):
    from oracle import models_ref
    from scp_amd.models.ehem import SwinPatchMerging
    z = golden(f"swin_merge_L{L}")
    sd = {"m." + k: v for k, v in _layer_sd(SwinPatchMerging(), int(z["wseed"])).items()}
    x = np.random.default_rng(int(z["x_seed"])).standard_normal((1, L, 256), dtype=np.float32)
    with torch.no_grad():
        y = models_ref.patch_merge(sd, "m", torch.from_numpy(x), L)
    assert np.abs(y[0].numpy() - z["y"]).max() < TOL


@pytest.mark.parametrize("name,level,mul", [("e2e_ehem_spher_L12", 12, False), ("e2e_ehem_mul_spher_L14", 14, True)])
def test_cpu_encode_path_vs_reference_driver(ehem_sd, name, level, mul):
    """oracle/cpu_encode.py (the CPU baseline bench.py times) is the reference's encode flow: same node count and a stream of the
    reference driver's length (compress_ehem, encode.py:85-160 / encode_mullevel.py:88-154) on the e2e fixtures' frame."""
    from oracle import cpu_encode
    z = golden(name)
    r = cpu_encode.encode_frame(z["xyz"], ehem_sd, level, mullevel=mul, mode="spher")
    assert r["n_nodes"] == int(z["n_nodes"]) and r["rows_coded"] == r["n_nodes"]
    assert abs(r["bits"] - 8 * len(z["bytes"])) <= 16
    assert set(r["stage_s"]) == {"quantise_octree_records", "context", "model", "cdf_rangecoder"}
