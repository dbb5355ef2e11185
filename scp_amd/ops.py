"""Dense-layer helpers used by the context models.

Default path: the hand-written bf16x3 MFMA GEMM (csrc/gemm.hip, fp32-class accuracy, fused bias / activation / residual
epilogue) through the C ABI.  Layers the split does not cover (K % 4 != 0 or K < 32) and every layer that feeds a kNN search run
on the exact fp32 MFMA kernel (scp_linear_f32): k-ordered FMA chains whose result does not depend on the batch, so the encoder's
one packed launch and the decoder's per-window launches agree bit for bit.  There is ONE backend: every layer goes through the C ABI;
a tensor that is not on the device raises ScpError (the fp32 library bracket the tests compare against lives in tests/).
"""
import weakref

import torch

from . import native

MODE = "bf16x3"      # the `gemm=` field of the stream's numeric-profile string (native.numeric_profile); a constant since round 4
_ACT = {None: native.ACT_NONE, "leaky": native.ACT_LEAKY, "gelu": native.ACT_GELU, "relu": native.ACT_RELU}
_cache = {}


def _split(w):
    """bf16 hi/lo planes of a weight, cached per tensor OBJECT (evicted when the tensor dies or is modified in place)."""
    key = id(w)
    ent = _cache.get(key)
    ep = getattr(_TLS, "epoch", 0)
    if ep and ent is not None and ent[3] == ep and ent[2]() is w:
        return ent[0]
    ver = (w._version, w.data_ptr(), w.device)
    if ent is None or ent[1] != ver or ent[2]() is not w:
        sw = native.SplitWeight(w)
        _cache[key] = (sw, ver, weakref.ref(w, lambda _r, k=key: _cache.pop(k, None)), ep)
        return sw
    if ep:
        _cache[key] = (ent[0], ent[1], ent[2], ep)
    return ent[0]


_cache16 = {}


def _split16(w):
    """row-scaled f16 hi/lo planes of a weight (native.SplitWeightF16), cached like _split."""
    key = id(w)
    ent = _cache16.get(key)
    ver = (w._version, w.data_ptr(), w.device)
    if ent is None or ent[1] != ver or ent[2]() is not w:
        sw = native.SplitWeightF16(w)
        _cache16[key] = (sw, ver, weakref.ref(w, lambda _r, k=key: _cache16.pop(k, None)))
        return sw
    return ent[0]


_TLS = __import__("threading").local()
_EPOCHS = __import__("itertools").count(1)


def frozen_epoch():
    """The calling thread's frozen-weights epoch (0 = none): see `frozen_weights`."""
    return getattr(_TLS, "epoch", 0)


class frozen_weights:
    """`with ops.frozen_weights():` - inside the block (one frame of a FrameEncoder / FrameDecoder) the parameters do not change, so a derived
    weight is validated against its sources ONCE, at its first use, instead of at every launch: the decoder runs ~1 400 Swin blocks per frame and
    the validation (twelve (data_ptr, _version, device) triples per block, plus the nn.Module attribute walks to reach them) was a sixth of its
    host time.  Outside the block every use validates, as before (`test_weights_replaced_after_a_forward_take_effect`).  Per thread; re-entrant."""

    def __enter__(self):
        self.prev = getattr(_TLS, "epoch", 0)
        if not self.prev:
            _TLS.epoch = next(_EPOCHS)
        return self

    def __exit__(self, *exc):
        _TLS.epoch = self.prev
        return False


def derived(owner, name, sources, build):
    """A tensor (tuple) derived from parameters - fused qkv weights, conv / BN folds, weight slabs - cached on the owning module
    and rebuilt whenever a source changed: its storage, device or in-place version (`load_state_dict`, `fill_weights` and
    optimiser steps all bump `_version`).  The bf16 splits further down are keyed on the derived tensors, so they follow.
    Inside `frozen_weights` an entry validated in this epoch is returned as it is."""
    store = owner.__dict__.setdefault("_scp_derived", {})
    ent = store.get(name)
    ep = getattr(_TLS, "epoch", 0)
    if ep and ent is not None and ent[2] == ep:
        return ent[1]
    key = tuple((t.data_ptr(), t._version, t.device) for t in sources)
    if ent is None or ent[0] != key:
        ent = (key, build(), ep)
        store[name] = ent
        native.note_cache_fill()
    elif ep:
        ent = (ent[0], ent[1], ep)
        store[name] = ent
    return ent[1]


def clear_cache():
    _cache.clear()
    _cache16.clear()


def linear(x, w, b=None, act=None, residual=None, exact=False, precise=False, scales=None):
    """act(x @ w.T + b) + residual.  exact=True keeps plain fp32 (used where the result feeds a kNN search); precise=True uses
    the f16x3 kernel (22-bit operands: fp32-chain accuracy at the MFMA rate; OctAttention; `scales`: native.RowScales of x shared
    between layers that read the same rows)."""
    K = w.shape[1]
    if not x.is_cuda:
        raise native.ScpError("ops.linear: device tensor expected (the SCP hot path has no CPU or library-GEMM backend)")
    if precise and not exact and K % 4 == 0 and K >= 32:
        return native.linear_f16x3(x, _split16(w), b, _ACT[act], residual, scales=scales)
    if not exact and not precise and K % 4 == 0 and K >= 32:
        return native.linear_bf16x3(x, _split(w), b, _ACT[act], residual)
    # exact fp32 MFMA kernel: k-ordered FMA chains, results independent of how many rows share the launch
    y = native.linear_f32(x, w, b, _ACT[act])
    return y if residual is None else y + residual


def linear_s(a, w, b=None, act=None, residual=None, want="f32", out=None, out_split=None, res_map=None, res_first=False):
    """The same layer on a pre-split activation (native.SplitAct): operands go global -> LDS by LDS-DMA (csrc/gemm_split.hip).
    want: "f32" -> fp32 [M, N] (optionally into `out`, which may be a column slice), "split" -> SplitAct for the next layer."""
    N = w.shape[0]
    if want == "f32" and out is None and N % 4:
        # rows must be 16-byte aligned for the vector epilogue: pad the row stride, hand back the [M, N] view
        buf = torch.empty((a.M, -(-N // 4) * 4), dtype=torch.float32, device=a.t.device)
        return native.linear_split(a, _split(w), b, _ACT[act], residual, out=buf, want="f32", res_map=res_map, res_first=res_first)[:, :N]
    return native.linear_split(a, _split(w), b, _ACT[act], residual, out=out, out_split=out_split, want=want, res_map=res_map,
                               res_first=res_first)


def leaky_mlp3_s(seq, a, want="f32", out=None, out_split=None):
    """leaky_mlp3 on a SplitAct: the two hidden activations stay in the split format, the last layer writes `want`."""
    a = linear_s(a, seq[0].weight, seq[0].bias, act="leaky", want="split")
    a = linear_s(a, seq[2].weight, seq[2].bias, act="leaky", want="split")
    return linear_s(a, seq[4].weight, seq[4].bias, want=want, out=out, out_split=out_split)


def split_cat(parts, device=None):
    """SplitAct of torch.cat(parts, 1) without materialising the fp32 concatenation (column offsets must be multiples of 8)."""
    K = sum(p.shape[1] for p in parts)
    out = native.SplitAct.empty(parts[0].shape[0], K, parts[0].device)
    c = 0
    for p_ in parts:
        assert c % 8 == 0 and p_.shape[1] % 4 == 0
        native.split_rows(p_, out=out.cols(c, c + p_.shape[1]))
        c += p_.shape[1]
    if out.t.shape[2] > K:
        out.t[:, :, K:].zero_()
    return out


def layer_norm(x, ln):
    """LayerNorm over the last axis (256 or 512 channels) on the in-tree kernel (csrc/fused.hip: one wavefront per row)."""
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    return native.layernorm_rows(x2, ln.weight, ln.bias, eps=ln.eps).reshape(x.shape)


def gelu_linear(x, w, b):
    return linear(x, w, b, act="gelu")


def leaky_mlp3(seq, x, slope=0.01, exact=False):
    """nn.Sequential(Linear, LeakyReLU, Linear, LeakyReLU, Linear) as stored under keys .0 / .2 / .4"""
    x = linear(x, seq[0].weight, seq[0].bias, act="leaky", exact=exact)
    x = linear(x, seq[2].weight, seq[2].bias, act="leaky", exact=exact)
    return linear(x, seq[4].weight, seq[4].bias, exact=exact)
