"""ctypes binding of libscp_hip.so (include/scp.h).

PyTorch appears here only as the owner of device memory and of the HIP stream; every call below goes
through the C ABI with raw device pointers.  There is NO fallback: if the library is missing or a
call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libscp_hip.so")
MAX_DEPTH = 21

CART, SPHER, CYLIN = 0, 1, 2
POS_MINMAX, POS_MINMAX_MUL, POS_POW2 = 0, 1, 2
_ERR = {-1: "SCP_EINVAL", -2: "SCP_ENOMEM", -3: "SCP_ESMALL", -4: "SCP_EHIP", -5: "SCP_ESTATE"}


class ScpError(RuntimeError):
    pass


class QuantInfo(C.Structure):
    _fields_ = [("bin_num", C.c_double), ("qs", C.c_double * 3), ("offset", C.c_double * 3),
                ("max_coord", C.c_int32), ("min_coord", C.c_int32)]


class Segment(C.Structure):
    _fields_ = [("point_begin", C.c_int64), ("point_count", C.c_int64), ("path_len", C.c_int32),
                ("path_bits", C.c_int32), ("drop_last", C.c_int32), ("reserved", C.c_int32)]


class SegmentInfo(C.Structure):
    _fields_ = [("depth", C.c_int32), ("max_coord", C.c_int32), ("n_leaves", C.c_int64), ("n_nodes", C.c_int64),
                ("node_base", C.c_int64), ("level_count", C.c_int64 * (MAX_DEPTH + 1))]


_lib = None
_vp = C.c_void_p


ABI_VERSION = 220      # include/scp.h: SCP_ABI_VERSION


def lib():
    """Load libscp_hip.so; raises if it has not been built (python -m scp_amd.build / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScpError(f"{LIB_PATH} is missing: run `python scp_amd/build.py` (hipcc, gfx950). "
                       "The SCP hot path has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    i32, i64 = C.c_int32, C.c_int64
    sig = {
        "scp_version": (C.c_int, []),
        "scp_geo_edge_mlps": (C.c_int, [_vp, i64, _vp, i64, _vp, i64, _vp, _vp, _vp, i64, i32, _vp]),
        "scp_mlp3_rows": (C.c_int, [_vp, i64, i64, _vp, _vp, _vp, _vp, i64, _vp, i32, i32, _vp]),
        "scp_swin_merge": (C.c_int, [_vp, i64, i64, _vp, _vp, _vp, _vp, C.c_float, _vp, i64, i32, _vp]),
        "scp_swin_ln_qkv": (C.c_int, [_vp, i64, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, i64, _vp, i64, i32, i32, _vp]),
        "scp_swin_kv_planes": (C.c_int, [_vp, _vp, i64, i64, _vp, _vp, _vp, _vp, _vp]),
        "scp_swin_attention_packed_planes": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, i32, i32, i32, _vp, _vp, _vp, i64, _vp, _vp]),
        "scp_split_rows_f16": (C.c_int, [_vp, i64, i32, i32, _vp, _vp, i64, _vp, _vp, _vp]),
        "scp_linear_split_f16": (C.c_int, [_vp, _vp, i64, _vp, _vp, _vp, _vp, i32, i32, _vp, _vp, i64, _vp, i64, i32, i32, i32, i32, i32, _vp]),
        "scp_last_hip_error": (C.c_int, []),
        "scp_device_count": (C.c_int, []),
        "scp_device_name": (C.c_int, [C.c_char_p, C.c_int]),
        "scp_quantize": (C.c_int, [_vp, i64, i32, C.c_double, C.c_double, _vp, _vp, C.POINTER(QuantInfo), _vp]),
        "scp_geom_create": (C.c_int, [C.POINTER(_vp)]),
        "scp_geom_destroy": (C.c_int, [_vp]),
        "scp_geom_build": (C.c_int, [_vp, _vp, i64, C.POINTER(Segment), i32, C.POINTER(SegmentInfo), _vp]),
        "scp_geom_build_xyz": (C.c_int, [_vp, _vp, _vp, i32, i32, _vp, i32, C.c_double, C.POINTER(Segment), _vp, C.POINTER(QuantInfo),
                                         C.POINTER(SegmentInfo), _vp]),
        "scp_geom_context_ehem_all": (C.c_int, [_vp, i32, i32, i32, _vp, _vp, _vp, _vp, _vp]),
        "scp_geom_emit_nodes": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
        "scp_geom_emit_leaves": (C.c_int, [_vp, i32, _vp, _vp]),
        "scp_geom_krecords_i64": (C.c_int, [_vp, i32, _vp, _vp]),
        "scp_geom_context_ehem": (C.c_int, [_vp, i32, i32, i32, _vp, _vp, _vp, _vp, _vp]),
        "scp_geom_context_octattn": (C.c_int, [_vp, i32, _vp, _vp, _vp, _vp]),
        "scp_knn_topk": (C.c_int, [_vp, i32, i32, i32, i32, _vp, _vp]),
        "scp_knn_topk_packed": (C.c_int, [_vp, _vp, i32, i32, _vp, _vp]),
        "scp_knn_topk_packed_bounded": (C.c_int, [_vp, _vp, i32, i32, _vp, _vp, _vp]),
        "scp_swin_attention_packed": (C.c_int, [_vp, _vp, _vp, _vp, _vp, i32, i32, i32, i32, _vp, _vp]),
        "scp_swin_attention_packed_split": (C.c_int, [_vp, _vp, _vp, _vp, _vp, i32, i32, i32, i32, _vp, _vp, i64, _vp]),
        "scp_set_attention_mode": (C.c_int, [i32]),
        "scp_set_attention_variant": (C.c_int, [i32]),
        "scp_get_attention_variant": (C.c_int, []),
        "scp_ctx_create": (C.c_int, [C.POINTER(_vp)]),
        "scp_ctx_destroy": (C.c_int, [_vp]),
        "scp_ctx_set": (C.c_int, [_vp, i32, i32]),
        "scp_ctx_get": (C.c_int, [_vp, i32]),
        "scp_ctx_make_current": (C.c_int, [_vp]),
        "scp_rc_debug_buffer": (C.c_int, [_vp]),
        "scp_rc_set_wide": (C.c_int, [i32]),
        "scp_prof_enable": (C.c_int, [i32]),
        "scp_prof_count": (C.c_int, []),
        "scp_prof_read": (C.c_int, [i32, _vp, _vp, _vp]),
        "scp_set_knn_mode": (C.c_int, [i32]),
        "scp_row_scale_f16": (C.c_int, [_vp, i64, i32, i32, _vp, _vp, _vp]),
        "scp_row_scale_from_max": (C.c_int, [_vp, i32, _vp, _vp, _vp]),
        "scp_linear_split_f16_max": (C.c_int, [_vp, _vp, i64, _vp, _vp, _vp, _vp, i32, i32, _vp, _vp, i64, _vp, i64, i32, i32, i32, i32, i32, _vp, _vp, i32, i32, i32, _vp]),
        "scp_octattn_attention_f16x3_vmax": (C.c_int, [_vp, _vp, _vp, _vp, _vp, i64, i32, i32, i32, i32, _vp, _vp, _vp, i64, _vp, _vp]),
        "scp_linear_f16x3_scaled": (C.c_int, [_vp, i64, _vp, _vp, _vp, i32, _vp, _vp, i64, _vp, i64, i32, i32, i32, i32, _vp, _vp, _vp]),
        "scp_layernorm_add": (C.c_int, [_vp, _vp, i64, i32, _vp, _vp, C.c_float, _vp, _vp]),
        "scp_layernorm_add_split_f16": (C.c_int, [_vp, _vp, i64, i32, _vp, _vp, C.c_float, _vp, _vp, _vp, i64, _vp, _vp, _vp]),
        "scp_set_knn_workgroup": (C.c_int, [i32]),
        "scp_knn_debug_buffer": (C.c_int, [_vp]),
        "scp_nn_sqdist_f64": (C.c_int, [_vp, i64, _vp, i64, _vp, _vp]),
        "scp_edge_gather_max_ld": (C.c_int, [_vp, i64, _vp, i64, _vp, _vp, _vp, i32, i32, i32, i32, _vp, i32, _vp]),
        "scp_embed_gather": (C.c_int, [_vp, _vp, _vp, i64, _vp, _vp, _vp, _vp, _vp, _vp, i64, _vp]),
        "scp_packed_plan_sizes": (C.c_int, [_vp, i32, _vp]),
        "scp_packed_plan": (C.c_int, [_vp, i32, _vp, _vp, i32, _vp]),
        "scp_decode_expand": (C.c_int, [_vp, _vp, _vp, _vp, _vp, i64, i32, i32, i32, i32, i32, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
        "scp_edge_gather_max": (C.c_int, [_vp, _vp, _vp, _vp, _vp, i32, i32, i32, i32, _vp, i32, _vp]),
        "scp_swin_attention": (C.c_int, [_vp, _vp, _vp, _vp, i32, i32, i32, i32, i32, _vp, _vp]),
        "scp_octattn_attention": (C.c_int, [_vp, _vp, _vp, _vp, _vp, i32, i32, i32, i32, _vp, _vp, _vp]),
        "scp_octattn_f16x3_ws_bytes": (C.c_int64, [i32, i32, i32]),
        "scp_octattn_embed": (C.c_int, [_vp, _vp, i64, i32, _vp, i32, _vp, i32, i32, _vp, i32, _vp, _vp, i32, _vp, i32, _vp, _vp, _vp, i64, _vp, _vp, _vp]),
        "scp_octattn_attention_f16x3": (C.c_int, [_vp, _vp, _vp, _vp, _vp, i64, i32, i32, i32, i32, _vp, _vp, _vp, i64, _vp]),
        "scp_split_weight_bf16": (C.c_int, [_vp, i32, i32, i32, i32, _vp, _vp, _vp]),
        "scp_linear_bf16x3": (C.c_int, [_vp, i64, _vp, _vp, i32, _vp, _vp, i64, _vp, i64, i32, i32, i32, i32, _vp]),
        "scp_split_weight_f16": (C.c_int, [_vp, i32, i32, i32, i32, _vp, _vp, _vp, _vp]),
        "scp_linear_f16x3": (C.c_int, [_vp, i64, _vp, _vp, _vp, i32, _vp, _vp, i64, _vp, i64, i32, i32, i32, i32, _vp, _vp]),
        "scp_linear_split": (C.c_int, [_vp, _vp, i64, _vp, _vp, i32, i32, _vp, _vp, i64, _vp, i64, _vp, _vp, i64, i32, i32, i32, i32, i32, _vp]),
        "scp_tile_weight_bf16": (C.c_int, [_vp, i32, i32, _vp, _vp]),
        "scp_swin_ln_linear": (C.c_int, [_vp, i64, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, i64, i32, i32, _vp]),
        "scp_swin_post_attn": (C.c_int, [_vp, _vp, i64, _vp, i64, _vp, _vp, _vp, _vp, C.c_float, _vp, i64, i32, _vp, i32, _vp]),
        "scp_swin_post_attn_weight_bytes": (C.c_int64, []),
        "scp_gelu_prescale": (C.c_double, []),
        "scp_split_rows": (C.c_int, [_vp, i64, i64, _vp, i32, _vp, _vp, i64, i64, _vp]),
        "scp_linear_split_scatter": (C.c_int, [_vp, _vp, i64, _vp, _vp, i32, i32, _vp, _vp, _vp, i64, i32, i32, i32, i32, i32, _vp]),
        "scp_linear_split_gather": (C.c_int, [_vp, _vp, i64, _vp, _vp, i32, i32, _vp, _vp, i64, _vp, _vp, i64, _vp, _vp, i64, i32, i32, i32, i32, i32, _vp]),
        "scp_linear_split_hier2": (C.c_int, [_vp, _vp, i64, i32, _vp, _vp, i64, i64, _vp, _vp, _vp, _vp, i32, _vp, _vp, _vp, i64, _vp, _vp, _vp, i64, i32, i32, i32, _vp]),
        "scp_layernorm_rows": (C.c_int, [_vp, i64, i64, _vp, _vp, i32, _vp, _vp, _vp, C.c_float, _vp, i64, i64, _vp]),
        "scp_layernorm_rows_split": (C.c_int, [_vp, i64, i64, _vp, _vp, i32, _vp, _vp, _vp, C.c_float, _vp, _vp, i64, i64, _vp]),
        "scp_gather_rows": (C.c_int, [_vp, i64, _vp, i32, _vp, i64, i64, _vp]),
        "scp_linear_f32": (C.c_int, [_vp, i64, _vp, _vp, _vp, i64, i32, i32, i32, i32, _vp]),
        "scp_softmax_cdf": (C.c_int, [_vp, i64, i64, i32, _vp, _vp, _vp, _vp, _vp]),
        "scp_pmf_cdf": (C.c_int, [_vp, i64, i32, _vp, _vp, _vp, _vp]),
        "scp_ac_encode_cdf": (C.c_int, [_vp, _vp, i64, i32, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "scp_ac_encode_lohi": (C.c_int, [_vp, i64, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "scp_ac_dec_new": (C.c_int, [C.POINTER(_vp), _vp, C.c_size_t, i32]),
        "scp_ac_dec_next": (C.c_int, [_vp, _vp]),
        "scp_ac_dec_run": (C.c_int, [_vp, _vp, i64, _vp]),
        "scp_ac_dec_free": (C.c_int, [_vp]),
    }
    for name, (res, args) in sig.items():
        if hasattr(L, name):
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
    if L.scp_version() != ABI_VERSION:
        raise ScpError(f"{LIB_PATH} implements ABI {L.scp_version()}, this binding is written for {ABI_VERSION} (include/scp.h: "
                       "SCP_ABI_VERSION; the weight-plane layout differs between versions): rebuild with `python scp_amd/build.py`")
    _lib = L
    return L


def _check(rc, what):
    if rc < 0:
        extra = f" (hipError {lib().scp_last_hip_error()})" if rc == -4 else ""
        raise ScpError(f"{what} failed: {_ERR.get(rc, rc)}{extra}")
    return rc


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The calling thread's current HIP stream as an integer handle.  torch.cuda.current_stream() builds a Stream object through four
    Python frames - 8 us per call, 71 ms per decoded frame (8 900 launches); the raw accessor takes 0.3 us."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return _RAW_STREAM(_RAW_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _dev(t, dtype=None):
    if not t.is_cuda:
        raise ScpError("device tensor expected (the SCP hot path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise ScpError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ScpError("contiguous tensor expected")
    return t.data_ptr()


def _opt(t):
    return None if t is None else _dev(t)


# ------------------------------------------------------------------------------------------------- launch brackets (scp_debug.h)
PROF_TAGS = {1: "post_attn", 2: "ln_linear", 3: "attention", 4: "knn_feat", 5: "knn_pos", 6: "gemm_split", 7: "edge_mlp", 8: "merge",
             9: "edge_gather", 10: "cdf", 11: "gemm_f32", 12: "gemm_rows", 13: "split_rows", 14: "layernorm", 15: "oa_attention", 16: "geom",
             17: "other", 18: "mlp3"}


class launch_profile:
    """`with native.launch_profile() as p: ...` - while inside, every bracketed C-ABI launch records a hipEvent pair around its kernel
    (include/scp_debug.h: scp_prof_*; the events are recorded in C next to the launch, no Python between them).  `p.records()` waits for
    the launches and returns [(tag name, milliseconds, algorithmic work)] in launch order.  A measurement hook, not a product path."""

    def __enter__(self):
        _check(lib().scp_prof_enable(1), "scp_prof_enable")
        return self

    def __exit__(self, *exc):
        lib().scp_prof_enable(0)
        return False

    def records(self):
        n = _check(lib().scp_prof_count(), "scp_prof_count")
        tags, ms, work = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.float64)
        if n:
            n = _check(lib().scp_prof_read(n, tags.ctypes.data, ms.ctypes.data, work.ctypes.data), "scp_prof_read")
        return [(PROF_TAGS.get(int(t), str(int(t))), float(m), float(w)) for t, m, w in zip(tags[:n], ms[:n], work[:n])]


# ------------------------------------------------------------------------------------------------- stage G1
def quantize(xyz, mode, qs, cart_offset=-200.0, want_transformed=False):
    """xyz cuda float32 [n,3] -> (q cuda int32 [n,3], QuantInfo, transformed or None)."""
    n = xyz.shape[0]
    q = torch.empty((n, 3), dtype=torch.int32, device=xyz.device)
    tr = torch.empty((n, 3), dtype=torch.float32, device=xyz.device) if want_transformed else None
    info = QuantInfo()
    rc = lib().scp_quantize(_dev(xyz, torch.float32), n, mode, float(qs), float(cart_offset), _dev(q), _opt(tr),
                            C.byref(info), _stream())
    _check(rc, "scp_quantize")
    return q, info, tr


# ------------------------------------------------------------------------------------------------- stage G2/G3
class Geom:
    """Owns one scp_geom workspace (reusable across frames)."""

    def __init__(self):
        self._h = _vp()
        _check(lib().scp_geom_create(C.byref(self._h)), "scp_geom_create")
        self.info = []
        self.device = None

    def close(self):
        if self._h:
            lib().scp_geom_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self, q, segments):
        """q cuda int32 [n,3]; segments: list of (begin, count, path(list of bits) or None, drop_last)."""
        nseg = len(segments)
        segs = (Segment * nseg)()
        for i, (b, c, path, drop) in enumerate(segments):
            path = list(path or [])
            bits = 0
            for p in path:
                bits = (bits << 1) | int(p)
            segs[i] = Segment(int(b), int(c), len(path), bits, int(bool(drop)), 0)
        infos = (SegmentInfo * nseg)()
        rc = lib().scp_geom_build(self._h, _dev(q, torch.int32), q.shape[0], segs, nseg, infos, _stream())
        _check(rc, "scp_geom_build")
        self.info = list(infos)
        self.segments = segments
        self.device = q.device
        self.total_nodes = sum(i.n_nodes for i in self.info)
        return self.info

    def build_xyz(self, frames, mode, qs_list, cart_offset, shells, want_q=False):
        """Stage G1 + G2 in one launch sequence (scp_geom_build_xyz): frames = list of cuda float32 [n,3] tensors, shells = list of
        (path bits or None, drop_last) - one tree per (frame, shell), segment index = frame * len(shells) + shell.  Every point is
        transformed once; -> list of QuantInfo per segment (and the int32 [sum n, 3] integers of all segments when want_q)."""
        nf, ns = len(frames), len(shells)
        if len(qs_list) != ns or nf == 0:
            raise ScpError("build_xyz: one step per shell expected")
        ptrs = (C.c_void_p * nf)(*[_dev(f, torch.float32) for f in frames])
        cnt = (C.c_int64 * nf)(*[int(f.shape[0]) for f in frames])
        qs = (C.c_double * ns)(*[float(q) for q in qs_list])
        segs = (Segment * ns)()
        for i, (path, drop) in enumerate(shells):
            path = list(path or [])
            bits = 0
            for p in path:
                bits = (bits << 1) | int(p)
            segs[i] = Segment(0, 0, len(path), bits, int(bool(drop)), 0)
        nseg = nf * ns
        qinfo, infos = (QuantInfo * nseg)(), (SegmentInfo * nseg)()
        q = torch.empty((sum(int(f.shape[0]) for f in frames) * ns, 3), dtype=torch.int32, device=frames[0].device) if want_q else None
        rc = lib().scp_geom_build_xyz(self._h, ptrs, cnt, nf, mode, qs, ns, float(cart_offset), segs, _opt(q), qinfo, infos, _stream())
        _check(rc, "scp_geom_build_xyz")
        self.info = list(infos)
        self.segments = [(0, int(f.shape[0]), path, drop) for f in frames for (path, drop) in shells]
        self.device = frames[0].device
        self.total_nodes = sum(i.n_nodes for i in self.info)
        return (list(qinfo), q) if want_q else list(qinfo)

    def context_ehem_all(self, pos_mode, lidar_level, context_size):
        """Context tables of EVERY segment of the last build in one launch, rows back to back: -> (ctx u8 [R,12], pos f32 [R,3], the coded
        symbols IN CODING ORDER u8 [R], (min, max) per level int64 [sum of depths, 2])."""
        R = sum(self.rows(s) for s in range(len(self.info)))
        dev = self.device
        ctx = torch.empty((R, 12), dtype=torch.uint8, device=dev)
        pos = torch.empty((R, 3), dtype=torch.float32, device=dev)
        sym = torch.empty(R, dtype=torch.uint8, device=dev)
        mm = torch.empty((sum(i.depth for i in self.info), 2), dtype=torch.int64, device=dev)
        _check(lib().scp_geom_context_ehem_all(self._h, pos_mode, lidar_level, context_size, _dev(ctx), _dev(pos), _dev(sym), _dev(mm), _stream()),
               "scp_geom_context_ehem_all")
        return ctx, pos, sym, mm

    def level_counts(self, seg):
        i = self.info[seg]
        return [int(i.level_count[l]) for l in range(i.depth)]

    def nodes(self, want=("occ", "level", "octant", "parent", "pos")):
        N, dev = self.total_nodes, self.device
        out = {}
        if "occ" in want:
            out["occ"] = torch.empty(N, dtype=torch.uint8, device=dev)
        if "level" in want:
            out["level"] = torch.empty(N, dtype=torch.uint8, device=dev)
        if "octant" in want:
            out["octant"] = torch.empty(N, dtype=torch.uint8, device=dev)
        if "parent" in want:
            out["parent"] = torch.empty(N, dtype=torch.int32, device=dev)
        if "pos" in want:
            out["pos"] = torch.empty((N, 3), dtype=torch.int32, device=dev)
        rc = lib().scp_geom_emit_nodes(self._h, _opt(out.get("occ")), _opt(out.get("level")), _opt(out.get("octant")),
                                       _opt(out.get("parent")), _opt(out.get("pos")), _stream())
        _check(rc, "scp_geom_emit_nodes")
        return out

    def leaves(self, seg):
        pts = torch.empty((self.info[seg].n_leaves, 3), dtype=torch.int32, device=self.device)
        _check(lib().scp_geom_emit_leaves(self._h, seg, _dev(pts), _stream()), "scp_geom_emit_leaves")
        return pts

    def rows(self, seg):
        return int(self.info[seg].n_nodes) - int(bool(self.segments[seg][3]))

    def krecords(self, seg):
        out = torch.empty((self.rows(seg), 4, 6), dtype=torch.int64, device=self.device)
        _check(lib().scp_geom_krecords_i64(self._h, seg, _dev(out), _stream()), "scp_geom_krecords_i64")
        return out

    def context_ehem(self, seg, pos_mode, lidar_level):
        r, dev = self.rows(seg), self.device
        ctx = torch.empty((r, 12), dtype=torch.uint8, device=dev)
        pos = torch.empty((r, 3), dtype=torch.float32, device=dev)
        sym = torch.empty(r, dtype=torch.uint8, device=dev)
        mm = torch.empty((self.info[seg].depth, 2), dtype=torch.int64, device=dev)
        rc = lib().scp_geom_context_ehem(self._h, seg, pos_mode, lidar_level, _dev(ctx), _dev(pos), _dev(sym), _dev(mm),
                                         _stream())
        _check(rc, "scp_geom_context_ehem")
        return ctx, pos, sym, mm

    def context_octattn(self, seg):
        r, dev = self.rows(seg), self.device
        ctx = torch.empty((r, 12), dtype=torch.uint8, device=dev)
        pos = torch.empty((r, 4, 3), dtype=torch.float32, device=dev)
        sym = torch.empty(r, dtype=torch.uint8, device=dev)
        _check(lib().scp_geom_context_octattn(self._h, seg, _dev(ctx), _dev(pos), _dev(sym), _stream()),
               "scp_geom_context_octattn")
        return ctx, pos, sym


# ------------------------------------------------------------------------------------------------- stage M
def knn_topk(x, k):
    """x cuda f32 [B,n,C] -> idx int32 [B,n,k]."""
    B, n, Cc = x.shape
    idx = torch.empty((B, n, k), dtype=torch.int32, device=x.device)
    _check(lib().scp_knn_topk(_dev(x, torch.float32), B, n, Cc, k, _dev(idx), _stream()), "scp_knn_topk")
    return idx


def knn_topk_packed(x, ctab, thr0=None):
    """x cuda f32 [T,C] (T % 512 == 0), ctab cuda int32 [T/512, 2] = (sequence base row, real length) -> idx int32 [T,20] global.
    thr0: optional f32 [T] a-priori pruning bound per row (a value of 2 x.y - |x|^2 - |y|^2 that >= 20 candidates reach)."""
    T, Cc = x.shape
    idx = torch.zeros((T, 20), dtype=torch.int32, device=x.device)
    if thr0 is None:
        _check(lib().scp_knn_topk_packed(_dev(x, torch.float32), _dev(ctab, torch.int32), T, Cc, _dev(idx), _stream()), "scp_knn_topk_packed")
    else:
        _check(lib().scp_knn_topk_packed_bounded(_dev(x, torch.float32), _dev(ctab, torch.int32), T, Cc, _dev(thr0, torch.float32), _dev(idx),
                                                 _stream()), "scp_knn_topk_packed_bounded")
    return idx


def swin_attention_packed(q, k, v, bias_table, wtab, shift, split=False):
    """q,k,v cuda f32 [T,256] views (unit channel stride), wtab int32 [T/512, 2] = (sequence base row, padded length).
    split=True returns the output as a SplitAct (operand of linear_split) instead of fp32."""
    T = q.shape[0]
    if split:
        o = SplitAct.empty(T, 256, q.device)
        rc = lib().scp_swin_attention_packed_split(q.data_ptr(), k.data_ptr(), v.data_ptr(), _dev(bias_table, torch.float32),
                                                   _dev(wtab, torch.int32), T // 512, shift, q.stride(0), k.stride(0), o.t[0].data_ptr(),
                                                   o.t[1].data_ptr(), o.t.stride(1), _stream())
        _check(rc, "scp_swin_attention_packed_split")
        return o
    out = torch.empty((T, 256), dtype=torch.float32, device=q.device)
    rc = lib().scp_swin_attention_packed(q.data_ptr(), k.data_ptr(), v.data_ptr(), _dev(bias_table, torch.float32),
                                         _dev(wtab, torch.int32), T // 512, shift, q.stride(0), k.stride(0), _dev(out), _stream())
    _check(rc, "scp_swin_attention_packed")
    return out


class KvPlanes:
    """k, v fp32 [T, 256] views (T % 512 == 0 rows of the packed layout) as the bf16 hi / lo planes of the plane-fed window attention
    (csrc/attn.hip: swin_attn_planes_kernel): K [T, 256] with swizzled 16-byte chunks, V^T per (32-token block, head) as the linear image
    of the kernel's LDS tile.  `t`: bfloat16 [4, T, 256] = K hi, K lo, V^T hi, V^T lo."""

    __slots__ = ("t",)

    def __init__(self, k=None, v=None, t=None):
        if t is None:
            T = k.shape[0]
            t = torch.empty((4, T, 256), dtype=torch.bfloat16, device=k.device)
            _check(lib().scp_swin_kv_planes(k.data_ptr(), v.data_ptr(), k.stride(0), T, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                            t[3].data_ptr(), _stream()), "scp_swin_kv_planes")
        self.t = t


def swin_attention_packed_planes(q, kv, bias_table, wtab, shift, split=False, valid=None):
    """swin_attention_packed with K / V as KvPlanes: operands staged by LDS-DMA, no conversion in the kernel; identical bits.
    valid (fp32 [T] / [T, 1], optional): query tiles of nothing but window padding are skipped, their output rows stay unwritten."""
    T = q.shape[0]
    p = kv.t
    vp = None if valid is None else _dev(valid, torch.float32)
    if split:
        o = SplitAct.empty(T, 256, q.device)
        rc = lib().scp_swin_attention_packed_planes(q.data_ptr(), p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr(), p[3].data_ptr(),
                                                    _dev(bias_table, torch.float32), _dev(wtab, torch.int32), T // 512, shift, q.stride(0), None,
                                                    o.t[0].data_ptr(), o.t[1].data_ptr(), o.t.stride(1), vp, _stream())
        _check(rc, "scp_swin_attention_packed_planes")
        return o
    out = torch.empty((T, 256), dtype=torch.float32, device=q.device)
    rc = lib().scp_swin_attention_packed_planes(q.data_ptr(), p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr(), p[3].data_ptr(),
                                                _dev(bias_table, torch.float32), _dev(wtab, torch.int32), T // 512, shift, q.stride(0), _dev(out), None,
                                                None, 0, vp, _stream())
    _check(rc, "scp_swin_attention_packed_planes")
    return out


def nn_sqdist(a, b):
    """a [na,3], b [nb,3] float64 device tensors -> float64 [na]: squared distance of every a to its nearest b (exhaustive)."""
    a = _dev_f64(a)
    b = _dev_f64(b)
    out = torch.empty((a.shape[0],), dtype=torch.float64, device=a.device)
    _check(lib().scp_nn_sqdist_f64(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(), _stream()), "scp_nn_sqdist_f64")
    return out


def _dev_f64(t):
    if not t.is_cuda:
        raise ScpError("device tensor required (there is no CPU path in the product)")
    return t.to(torch.float64).contiguous()


def embed_gather(ctx, pos, inmap, occ_enc, level_enc, octant_enc):
    """ctx uint8 [T,12], pos f32 [T,3], inmap int64 [rows] -> (x f32 [rows,80], pos f32 [rows,3], occ_self int64 [rows])."""
    rows = inmap.shape[0]
    dev = ctx.device
    x = torch.empty((rows, 80), dtype=torch.float32, device=dev)
    p = torch.empty((rows, 3), dtype=torch.float32, device=dev)
    o = torch.empty((rows,), dtype=torch.int64, device=dev)
    _check(lib().scp_embed_gather(_dev(ctx, torch.uint8), _dev(pos, torch.float32), _dev(inmap, torch.int64), ctx.shape[0], _dev(occ_enc),
                                  _dev(level_enc), _dev(octant_enc), x.data_ptr(), p.data_ptr(), o.data_ptr(), rows, _stream()),
           "scp_embed_gather")
    return x, p, o


_POPC = {}


def decode_expand(sym, pos, anc, octant, L, shift, lv_next, lv_clamp, polar, mn, den):
    """Decoder: children of one decoded level + the next level's model inputs, one launch (csrc/plan.hip: scp_decode_expand).
    sym int64 [n] (-1 = unknown), pos int32 [n,3], anc uint8 [n,9], octant uint8 [n] ->
    (occ8 uint8 [n], cpos int32 [m,3], canc uint8 [m,9], coct uint8 [m], cctx uint8 [m,12], cposn float32 [m,3]); one host sync (m)."""
    dev, n = sym.device, sym.shape[0]
    tab = _POPC.get(dev)
    if tab is None:
        tab = _POPC[dev] = torch.tensor([bin(v).count("1") for v in range(256)], dtype=torch.int64, device=dev)
    cum = torch.cumsum(tab[sym + 1], 0)
    m = int(cum[-1])
    occ8 = torch.empty(n, dtype=torch.uint8, device=dev)
    cpos = torch.empty((m, 3), dtype=torch.int32, device=dev)
    canc = torch.empty((m, 9), dtype=torch.uint8, device=dev)
    coct = torch.empty(m, dtype=torch.uint8, device=dev)
    cctx = torch.empty((m, 12), dtype=torch.uint8, device=dev)
    cposn = torch.empty((m, 3), dtype=torch.float32, device=dev)
    if m == 0:
        # no parent has a child (a multi-level shell whose last level holds only the dropped node): the C entry rejects the null
        # pointers of empty outputs, and there is nothing to launch - the occupancy bytes of unknown symbols are 0
        return (sym + 1).clamp_(min=0).to(torch.uint8), cpos, canc, coct, cctx, cposn
    _check(lib().scp_decode_expand(_dev(sym, torch.int64), cum.data_ptr(), _dev(pos, torch.int32), _dev(anc, torch.uint8), _dev(octant, torch.uint8), n,
                                   int(L), int(shift), int(lv_next), int(lv_clamp), 1 if polar else 0, float(mn), float(den), cpos.data_ptr(), canc.data_ptr(),
                                   coct.data_ptr(), cctx.data_ptr(), cposn.data_ptr(), occ8.data_ptr(), _stream()), "scp_decode_expand")
    return occ8, cpos, canc, coct, cctx, cposn


def packed_plan(lengths, device):
    """All index maps of the packed forward for one list of window lengths, ONE kernel launch (csrc/plan.hip).
    Returns (rows[11], dict of device tensors keyed like models/packed.py: PackedPlan.d)."""
    import numpy as np
    c = np.ascontiguousarray(np.asarray(lengths, np.int64))
    W = int(c.shape[0])
    rows = np.zeros(11, np.int64)
    _check(lib().scp_packed_plan_sizes(c.ctypes.data, W, rows.ctypes.data), "scp_packed_plan_sizes")
    r = [int(x) for x in rows]
    i64, i32, f32 = torch.int64, torch.int32, torch.float32
    spec = [("inmap", i64, (r[0],)), ("a1map", i64, (r[5],)), ("a2map", i64, (r[5],)), ("even_rows", i64, (r[9],)), ("odd_rows", i64, (r[10],)),
            ("even_dst", i64, (r[9],)), ("odd_dst", i64, (r[10],))]
    for s_ in range(4):
        spec += [(f"sme{s_}", i64, (r[s_ + 1],)), (f"smo{s_}", i64, (r[s_ + 1],))]
    for s_ in range(3):
        spec += [(f"cme{s_}", i64, (r[6 + s_],)), (f"cmo{s_}", i64, (r[6 + s_],))]
    spec += [(f"sc{s_}", i64, (r[0],)) for s_ in range(1, 5)] + [(f"cc{s_}", i64, (r[5],)) for s_ in range(1, 4)]
    spec += [(f"tab{l}", i32, (r[l] // 512, 2)) for l in range(9)] + [("knn_tab", i32, (r[0] // 512, 2))]
    spec += [(f"valid{l}", f32, (r[l], 1)) for l in range(9)]
    spec += [(f"sp{s_}", i64, (r[s_],)) for s_ in range(4)] + [(f"cp{s_}", i64, (r[5 + s_],)) for s_ in range(3)]
    spec += [("even_out", i64, (r[5],)), ("odd_out", i64, (r[5],))]
    t = {name: torch.empty(shape, dtype=dt, device=device) for name, dt, shape in spec}
    ptrs = (C.c_void_p * len(spec))(*[t[name].data_ptr() for name, _, _ in spec])
    scratch = torch.empty((36 * W,), dtype=torch.int64, device=device)
    _check(lib().scp_packed_plan(c.ctypes.data, W, scratch.data_ptr(), ptrs, len(spec), _stream()), "scp_packed_plan")
    d = dict(inmap=t["inmap"], a1map=t["a1map"], a2map=t["a2map"], even_rows=t["even_rows"], odd_rows=t["odd_rows"],
             even_dst=t["even_dst"], odd_dst=t["odd_dst"],
             self_merge=[(t[f"sme{s_}"], t[f"smo{s_}"]) for s_ in range(4)], cross_merge=[(t[f"cme{s_}"], t[f"cmo{s_}"]) for s_ in range(3)],
             self_concat=[t[f"sc{s_}"] for s_ in range(1, 5)], cross_concat=[t[f"cc{s_}"] for s_ in range(1, 4)],
             self_tab=[t[f"tab{l}"] for l in range(5)], cross_tab=[t[f"tab{l}"] for l in range(5, 9)], knn_tab=t["knn_tab"],
             self_valid=[t[f"valid{l}"] for l in range(5)], cross_valid=[t[f"valid{l}"] for l in range(5, 9)],
             self_parent=[t[f"sp{s_}"] for s_ in range(4)], cross_parent=[t[f"cp{s_}"] for s_ in range(3)],
             even_out=t["even_out"], odd_out=t["odd_out"])
    d["self_tiles"], d["cross_tiles"] = real_tiles(c, device)
    return r, d


def real_tiles(lengths, device, n_self=5, n_cross=4):
    """Per Swin stage of the packed layout the 128-row tiles that hold at least one real row (int32 device tensors, ascending): host
    arithmetic on the list of window lengths (a window's real rows are a prefix of its 512-aligned run), ONE host -> device copy.  The
    row-chain block kernel walks this list instead of every tile; tiles of nothing but window padding are 4.7 % of an L16-m frame."""
    import numpy as np
    c = np.asarray(lengths, np.int64)
    e = c + (c & 1)

    def lists(L, n):
        out = []
        for _ in range(n):
            Lp = -(-L // 512) * 512
            base = np.cumsum(Lp) - Lp
            nt = -(-L // 128)
            first = np.repeat(base // 128, nt)
            within = np.arange(int(nt.sum())) - np.repeat(np.cumsum(nt) - nt, nt)
            out.append((first + within).astype(np.int32))
            L = (L + 1) // 2
        return out
    parts = lists(e, n_self) + lists(e // 2, n_cross)
    flat = torch.from_numpy(np.concatenate(parts)).to(device, non_blocking=True)
    cuts = np.cumsum([0] + [len(p) for p in parts])
    views = [flat[cuts[i]:cuts[i + 1]] for i in range(len(parts))]
    return views[:n_self], views[n_self:]


def edge_gather_max_rows(u, v, idx, scale, shift, out=None):
    """packed form: u, v f32 [n, C'] VIEWS with unit channel stride (any row stride), idx int32 [n, k] -> out f32 [n, C'] (`out`: an [n, C'] view with
    unit channel stride, e.g. the leading columns of the next search's feature buffer: no concatenation afterwards)."""
    n, Co = u.shape
    if out is None:
        out = torch.empty((n, Co), dtype=torch.float32, device=u.device)
    elif out.shape != (n, Co) or out.stride(1) != 1 or out.dtype != torch.float32 or out.stride(0) % 4 or out.data_ptr() % 16:
        raise ScpError("edge_gather_max_rows: out must be a float32 [n, C'] view with unit channel stride and 16-byte aligned rows")
    _check(lib().scp_edge_gather_max_ld(u.data_ptr(), u.stride(0), v.data_ptr(), v.stride(0), _dev(idx, torch.int32), _dev(scale), _dev(shift),
                                        1, n, Co, idx.shape[1], out.data_ptr(), out.stride(0), _stream()), "scp_edge_gather_max_ld")
    return out


_MODES = {"knn": "f32" if os.environ.get("SCP_KNN", "")[:1] == "f" else "f16x3",
          "attn": "f32" if os.environ.get("SCP_ATTN", "")[:1] == "f" else "bf16x3"}


def set_knn_mode(f16x3):
    """PROCESS default (test hook, include/scp_debug.h; an encoder uses a NumericProfile): True = f16x3 distances for the 144-/192-feature
    searches, False = exact fp32 MFMA chain."""
    _check(lib().scp_set_knn_mode(1 if f16x3 else 0), "scp_set_knn_mode")
    _MODES["knn"] = "f16x3" if f16x3 else "f32"


def set_knn_workgroup(shape):
    """256 (default): 256-query workgroups on the XCD-affine schedule, a barrier per group of tiles; 257 / 258: smaller groups; +16: outward
    sweep; 128: the 128-query kernel.  Same results."""
    _check(lib().scp_set_knn_workgroup(int(shape)), "scp_set_knn_workgroup")


class NumericProfile:
    """The numeric profile of ONE encoder / decoder (an scp_ctx of include/scp.h): which arithmetic the kernels with a choice use -
    feature kNN searches f16x3 (default) or the exact fp32 chain, window attention bf16x3 (default) or fp32 MFMA.  It belongs to the
    handle, not to the process: FrameEncoder / FrameDecoder make theirs current (for the calling thread) around their launches, so two
    encoders with different profiles in one process, or in different threads, each get their own arithmetic and their own streams."""

    def __init__(self, knn_f16x3=True, attention_bf16x3=True):
        self._h = _vp()
        _check(lib().scp_ctx_create(C.byref(self._h)), "scp_ctx_create")
        self.knn_f16x3, self.attention_bf16x3 = bool(knn_f16x3), bool(attention_bf16x3)
        _check(lib().scp_ctx_set(self._h, 1, int(self.knn_f16x3)), "scp_ctx_set")
        _check(lib().scp_ctx_set(self._h, 2, int(self.attention_bf16x3)), "scp_ctx_set")

    def describe(self, model_name):
        return numeric_profile(model_name, self)

    def __del__(self):
        # a profile that is current somewhere is referenced by that `use_profile` object, so this only runs for profiles nobody uses
        try:
            if self._h:
                lib().scp_ctx_destroy(self._h)
        except Exception:
            pass


_TLS = __import__("threading").local()


def current_profile():
    """The NumericProfile current for this thread (None = the process default)."""
    return getattr(_TLS, "profile", None)


class use_profile:
    """`with native.use_profile(p):` - p (a NumericProfile or None) is the calling thread's current profile inside the block."""

    def __init__(self, profile):
        self.profile = profile

    def __enter__(self):
        self.prev = current_profile()
        _check(lib().scp_ctx_make_current(self.profile._h if self.profile is not None else None), "scp_ctx_make_current")
        _TLS.profile = self.profile           # only once the C side has switched: the two thread-local views never disagree
        return self.profile

    def __exit__(self, *exc):
        lib().scp_ctx_make_current(self.prev._h if self.prev is not None else None)
        _TLS.profile = self.prev
        return False


def attention_bf16x3():
    """Whether the window attention of the calling thread runs its bf16x3 form (its current profile, or the process default)."""
    p = current_profile()
    return (_MODES["attn"] == "bf16x3") if p is None else p.attention_bf16x3


def numeric_profile(model_name, profile="current"):
    """The arithmetic variants that decide the logits' last bits - hence the integer CDFs a decoder must reproduce - as a string.  The
    encoder writes it into its side-info file and the decoder refuses a stream coded under another profile.  `profile`: a
    NumericProfile, None (process default) or "current" (this thread's).  The remaining fields are constants since round 4 (the
    superseded launch forms they once named are gone).  The generation prefix (ehem/N) moves whenever a kernel's last bits do, and the
    decoder compares the whole string: streams of an older generation are refused by design, not decoded to garbage."""
    from . import ops
    from .models import packed
    if model_name == "OctAttention":
        return f"octattn/1:gemm={ops.MODE},attn={OCTATTN_MODE}"
    if profile == "current":
        profile = current_profile()
    knn = _MODES["knn"] if profile is None else ("f16x3" if profile.knn_f16x3 else "f32")
    attn = _MODES["attn"] if profile is None else ("bf16x3" if profile.attention_bf16x3 else "f32")
    # ehem/3: patch merging and the geometry generator's edge MLPs on row-chain kernels (other last bits than ehem/2)
    # ehem/4: the bf16x3 attention sweeps against the fixed reference 0 with the bias as the products' start value (csrc/attn.hip: attn_tile)
    # ehem/5: GELU as max(y, 0) - |y| exp(-beta y^2) / P4(|y|) (csrc/scp_internal.h) instead of the degree-12 erf polynomial
    # ehem/6: stages 0 and 1 of the layers over concat_states in one launch (the stage-1 product starts the accumulators: gemm_hier2_kernel), the two
    #         256-wide heads as one row-chain launch each (bias as the accumulators' start value: rc_mlp3_kernel)
    L = lib()
    attnv = "" if L.scp_get_attention_variant() == 1 else f",attnv={L.scp_get_attention_variant()}"     # process-wide test bracket (scp_debug.h)
    concat = ("hier2" if packed.FUSE2 else "hier") if packed.HIER else "direct"
    heads = "" if packed.CHAIN_HEADS else ",heads=split"
    return f"ehem/6:gemm={ops.MODE},knn={knn},attn={attn},concat={concat},swin=rowchain{heads}{attnv}"


def edge_gather_max(u, v, idx, scale, shift, out=None):
    """u,v [B,n,C'] f32, idx [B,n,k] i32, scale/shift [C'] -> out [B,n,C'] (may be a column slice of a wider tensor)."""
    B, n, Co = u.shape
    k = idx.shape[2]
    if out is None:
        out = torch.empty((B, n, Co), dtype=torch.float32, device=u.device)
    stride = out.stride(1)
    if out.stride(2) != 1 or out.stride(0) != n * stride:
        raise ScpError("edge_gather_max: out must be row-major with unit channel stride")
    rc = lib().scp_edge_gather_max(_dev(u, torch.float32), _dev(v, torch.float32), _dev(idx, torch.int32), _dev(scale),
                                   _dev(shift), B, n, Co, k, out.data_ptr(), stride, _stream())
    _check(rc, "scp_edge_gather_max")
    return out


def set_attention_mode(bf16x3=True):
    """PROCESS default (test hook, include/scp_debug.h; an encoder uses a NumericProfile): True = QK^T / PV as bf16x3 splits on bf16 MFMA,
    False = plain fp32 MFMA."""
    _check(lib().scp_set_attention_mode(1 if bf16x3 else 0), "scp_set_attention_mode")
    _MODES["attn"] = "bf16x3" if bf16x3 else "f32"


def swin_attention(q, k, v, bias_table, shift):
    """q,k,v cuda f32 [B,Lp,256] views (unit channel stride, Lp % 512 == 0), bias_table [1023,4] -> out [B,Lp,256]."""
    B, Lp, _ = q.shape
    for t in (q, k, v):
        if t.stride(2) != 1 or t.stride(0) != Lp * t.stride(1) or not t.is_cuda or t.dtype != torch.float32:
            raise ScpError("swin_attention: operands must be [B,Lp,256] views with unit channel stride")
    if k.stride(1) != v.stride(1):
        raise ScpError("swin_attention: k and v must share a row stride")
    out = torch.empty((B, Lp, 256), dtype=torch.float32, device=q.device)
    rc = lib().scp_swin_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), _dev(bias_table, torch.float32), B, Lp, shift,
                                  q.stride(1), k.stride(1), _dev(out), _stream())
    _check(rc, "scp_swin_attention")
    return out


ACT_NONE, ACT_LEAKY, ACT_GELU, ACT_RELU = 0, 1, 2, 3

# Lazily built device-side caches (split weights, fused parameter tensors) are created on whatever stream is current.  Every
# creation site bumps this counter; FrameEncoder.encode_async synchronises once after a call during which it moved, so that a
# frame enqueued later on ANOTHER stream (lane) never reads a cache entry whose creating kernels are still in flight.
CACHE_FILLS = 0


def note_cache_fill():
    global CACHE_FILLS
    CACHE_FILLS += 1



# SCP_WTILE=0: the dense kernels take row-major weight planes (A/B bracket).  Read ONCE, here - csrc/gemm.hip and gemm_split.hip read
# it once too (a static), so a value changed after the first launch could only make the two sides disagree.
WTILE = os.environ.get("SCP_WTILE", "1")[:1] != "0"


def _tile_in_place_of(hi, lo, Npad, Kpad):
    """The tiled images (scp_tile_weight_bf16) of two row-major 16-bit planes, on the current stream."""
    th, tl = torch.empty_like(hi), torch.empty_like(lo)
    for src, dst in ((hi, th), (lo, tl)):
        _check(lib().scp_tile_weight_bf16(src.data_ptr(), Npad, Kpad, dst.data_ptr(), _stream()), "scp_tile_weight_bf16")
    return th, tl


class SplitWeight:
    """bf16 hi/lo planes of a Linear weight [N,K], zero-padded to [Npad,Kpad], in the layout the dense kernels stream (built once per
    weight, split AND tiled in the constructor - one cache fill, on one stream; the row-major intermediate is not kept).
    `.hi / .lo`: the planes; `.tiled_layout`: whether they are tiled (always, unless SCP_WTILE=0)."""

    def __init__(self, w, tiled=None):
        N, K = w.shape
        self.N, self.K = N, K
        self.Npad, self.Kpad = -(-N // 256) * 256, -(-K // 32) * 32
        self.hi = torch.empty((self.Npad, self.Kpad), dtype=torch.bfloat16, device=w.device)
        self.lo = torch.empty_like(self.hi)
        wc = w.detach().contiguous().float()
        _check(lib().scp_split_weight_bf16(_dev(wc), N, K, self.Npad, self.Kpad, _dev(self.hi), _dev(self.lo), _stream()),
               "scp_split_weight_bf16")
        self.tiled_layout = WTILE if tiled is None else bool(tiled)
        if self.tiled_layout:
            self.hi, self.lo = _tile_in_place_of(self.hi, self.lo, self.Npad, self.Kpad)
        note_cache_fill()

    def tiled(self):
        """(hi, lo) as the dense kernels stream them: tiled (_tile_planes) by default, row-major under SCP_WTILE=0.  A weight built with
        the other layout than the library reads (the `tiled=` override exists for the layout tests) is refused, not multiplied."""
        if self.tiled_layout != WTILE:
            raise ScpError("SplitWeight: planes are %s but the library reads %s planes (SCP_WTILE)" % (
                "tiled" if self.tiled_layout else "row-major", "tiled" if WTILE else "row-major"))
        return (self.hi, self.lo)


def linear_bf16x3(x, sw, bias=None, act=ACT_NONE, residual=None, out=None):
    """x [..., K] fp32 (unit last stride, uniform row stride) -> [..., N]; out may be a column slice of a wider buffer."""
    K, N = sw.K, sw.N
    lead = x.shape[:-1]
    x2 = x.reshape(-1, K) if x.is_contiguous() else x
    if x2.dim() != 2:
        x2 = x.contiguous().reshape(-1, K)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
        o2 = out
    else:
        o2 = out.reshape(-1, N) if out.is_contiguous() else out
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, N) if residual.is_contiguous() else residual.contiguous().reshape(-1, N)
    wt = _tiled_pair(sw)
    rc = lib().scp_linear_bf16x3(x2.data_ptr(), x2.stride(0), wt[0].data_ptr(), wt[1].data_ptr(), sw.Kpad, _opt(bias),
                                 None if r2 is None else r2.data_ptr(), 0 if r2 is None else r2.stride(0), o2.data_ptr(),
                                 o2.stride(0), M, N, K, act, _stream())
    _check(rc, "scp_linear_bf16x3")
    return out.reshape(*lead, N) if out.dim() == 2 and len(lead) != 1 else out


def _tiled_pair(sw):
    if sw.tiled_layout != WTILE:
        raise ScpError("weight planes are not in the layout the library reads (SCP_WTILE)")
    return (sw.hi, sw.lo)


def _tiled_planes_always(sw):
    """(hi, lo) of a SplitWeight in the tiled layout whatever SCP_WTILE says (the row-chain kernels have no row-major mode)."""
    if sw.tiled_layout:
        return sw.hi, sw.lo
    return _tile_in_place_of(sw.hi, sw.lo, sw.Npad, sw.Kpad)


def _tile_planes(t):
    """bf16 plane [Npad, Kpad] -> the tiled layout the LDS-DMA kernels read: block (16-row group rb, 32-element k-slab ks) = 1 KiB =
    the LDS image of one DMA instruction, [r][p][8] with the logical 16-byte chunk p ^ ((r >> 2) & 3) of row r at position p (the
    read swizzle of csrc/gemm_split.hip), blocks ordered [rb][ks]."""
    N, K = t.shape
    v = t.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 1, 3, 4)                    # [rb, ks, r, q, e]
    r = torch.arange(16, device=t.device)
    q = (torch.arange(4, device=t.device)[None, :] ^ ((r[:, None] >> 2) & 3))        # [r, p] -> source chunk
    idx = q[None, None, :, :, None].expand(N // 16, K // 32, 16, 4, 8)
    return torch.gather(v, 3, idx).contiguous()


class SplitWeightF16:
    """Row-scaled f16 hi/lo planes of a Linear weight [N,K] (scp_split_weight_f16), zero-padded to [Npad,Kpad]."""

    def __init__(self, w):
        N, K = w.shape
        self.N, self.K = N, K
        self.Npad, self.Kpad = -(-N // 256) * 256, -(-K // 32) * 32      # 256: the row tile of scp_linear_split_f16 (128 suffices for scp_linear_f16x3)
        self.hi = torch.empty((self.Npad, self.Kpad), dtype=torch.float16, device=w.device)
        self.lo = torch.empty_like(self.hi)
        self.inv_scale = torch.empty((self.Npad,), dtype=torch.float32, device=w.device)
        wc = w.detach().contiguous().float()
        _check(lib().scp_split_weight_f16(_dev(wc), N, K, self.Npad, self.Kpad, _dev(self.hi), _dev(self.lo), _dev(self.inv_scale),
                                          _stream()), "scp_split_weight_f16")
        self.tiled_layout = WTILE
        if self.tiled_layout:       # the tiling is a permutation of 16-bit elements: the same kernel serves f16 planes
            self.hi, self.lo = _tile_in_place_of(self.hi, self.lo, self.Npad, self.Kpad)
        note_cache_fill()


class RowScales:
    """Power-of-two row scales of an fp32 activation [M, K] (scale, 1 / scale) for the f16x3 layers: computed once, shared by every
    layer that reads these rows (`rows(a, b)` = the scales of the row range [a, b))."""

    __slots__ = ("sc", "isc", "x")

    def __init__(self, x2=None, sc=None, isc=None):
        if x2 is not None:
            M, K = x2.shape
            ws = torch.empty((2, M), dtype=torch.float32, device=x2.device)
            _check(lib().scp_row_scale_f16(x2.data_ptr(), x2.stride(0), M, K, ws[0].data_ptr(), ws[1].data_ptr(), _stream()), "scp_row_scale_f16")
            sc, isc = ws[0], ws[1]
        self.sc, self.isc, self.x = sc, isc, x2

    def rows(self, a, b):
        return RowScales(None, self.sc[a:b], self.isc[a:b])

    @staticmethod
    def from_max(row_max):
        """The scales of rows whose maxima a producing kernel's epilogue took (linear_split_f16(..., row_max=...)): int32 [M] bit patterns of max |row|."""
        M = row_max.shape[0]
        ws = torch.empty((2, M), dtype=torch.float32, device=row_max.device)
        _check(lib().scp_row_scale_from_max(_dev(row_max, torch.int32), M, ws[0].data_ptr(), ws[1].data_ptr(), _stream()), "scp_row_scale_from_max")
        return RowScales(None, ws[0], ws[1])


class SplitActF16:
    """An fp32 activation [M, K] as the f16x3 kernels' operand, made ONCE (scp_split_rows_f16): power-of-two row scales (sc, isc) and
    the IEEE-half planes hi / lo [M, ld] of the scaled rows (ld = K rounded up to 32, padding zero).  `rows(a, b)`: the same for the row
    range [a, b) (views)."""

    __slots__ = ("hi", "lo", "sc", "isc", "K")

    def __init__(self, x2=None, parts=None):
        if x2 is not None:
            M, K = x2.shape
            ld = -(-K // 32) * 32
            pl = torch.empty((2, M, ld), dtype=torch.float16, device=x2.device)
            ws = torch.empty((2, M), dtype=torch.float32, device=x2.device)
            _check(lib().scp_split_rows_f16(x2.data_ptr(), x2.stride(0), M, K, pl[0].data_ptr(), pl[1].data_ptr(), ld, ws[0].data_ptr(), ws[1].data_ptr(),
                                            _stream()), "scp_split_rows_f16")
            parts = (pl[0], pl[1], ws[0], ws[1], K)
        self.hi, self.lo, self.sc, self.isc, self.K = parts

    @property
    def M(self):
        return self.hi.shape[0]

    def rows(self, a, b):
        return SplitActF16(parts=(self.hi[a:b], self.lo[a:b], self.sc[a:b], self.isc[a:b], self.K))


def linear_split_f16(a, sw, bias=None, act=ACT_NONE, residual=None, cfg=0, out=None, row_max=None, col_max=None):
    """act(A W^T + bias) + residual on pre-split f16 planes (SplitActF16 x SplitWeightF16) -> fp32 [M, N]; bit-identical to linear_f16x3
    on the fp32 rows the planes were made from (csrc/gemm_split.hip, F16 instantiation: operands by LDS-DMA, no conversion in the tile).
    row_max: ZEROED int32 [M], receives the bit patterns of max |out[m, :]|; col_max = (ZEROED int32 [1], col_lo, col_hi, rows): max |out[:rows,
    col_lo:col_hi]| - taken in the epilogue (scp_linear_split_f16_max), the scales of the next f16x3 layer without a pass over `out`."""
    if not sw.tiled_layout and WTILE:
        raise ScpError("linear_split_f16: weight planes are not tiled")
    M, N = a.M, sw.N
    if a.K != sw.K:
        raise ScpError("linear_split_f16: K mismatch")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.hi.device)
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, N)
        if r2.stride(1) != 1:
            r2 = r2.contiguous()
    if row_max is not None or col_max is not None:
        if row_max is not None and (row_max.dtype != torch.int32 or row_max.shape[0] != M or not row_max.is_contiguous()):
            raise ScpError("linear_split_f16: row_max must be a contiguous int32 [M]")
        cm, lo, hi, rows = col_max if col_max is not None else (None, 0, 0, 0)
        rc = lib().scp_linear_split_f16_max(a.hi.data_ptr(), a.lo.data_ptr(), a.hi.stride(0), a.isc.data_ptr(), sw.hi.data_ptr(), sw.lo.data_ptr(),
                                            sw.inv_scale.data_ptr(), sw.Npad, sw.Kpad, _opt(bias), None if r2 is None else r2.data_ptr(),
                                            0 if r2 is None else r2.stride(0), out.data_ptr(), out.stride(0), M, N, sw.K, act, cfg,
                                            None if row_max is None else row_max.data_ptr(), None if cm is None else _dev(cm, torch.int32), int(lo), int(hi), int(rows),
                                            _stream())
        _check(rc, "scp_linear_split_f16_max")
        return out
    rc = lib().scp_linear_split_f16(a.hi.data_ptr(), a.lo.data_ptr(), a.hi.stride(0), a.isc.data_ptr(), sw.hi.data_ptr(), sw.lo.data_ptr(),
                                    sw.inv_scale.data_ptr(), sw.Npad, sw.Kpad, _opt(bias), None if r2 is None else r2.data_ptr(),
                                    0 if r2 is None else r2.stride(0), out.data_ptr(), out.stride(0), M, N, sw.K, act, cfg, _stream())
    _check(rc, "scp_linear_split_f16")
    return out


def _rows_f16x3(x, K):
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or (x2.stride(0) & 3) or (x2.data_ptr() & 15):
        x2 = x2.contiguous()
    return x2


def linear_f16x3(x, sw, bias=None, act=ACT_NONE, residual=None, scales=None):
    """x [..., K] fp32 -> act(x @ W.T + bias) + residual, [..., N] fp32, on the f16x3 kernel (22-bit operands, row scaled).
    scales: RowScales of exactly these rows (else they are computed here)."""
    K, N = sw.K, sw.N
    lead = x.shape[:-1]
    x2 = _rows_f16x3(x, K)
    M = x2.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, N)
        if r2.stride(1) != 1:
            r2 = r2.contiguous()
    if scales is not None:
        if scales.sc.shape[0] != M:
            raise ScpError("linear_f16x3: row scales of another row count")
        rc = lib().scp_linear_f16x3_scaled(x2.data_ptr(), x2.stride(0), _tiled_pair(sw)[0].data_ptr(), _tiled_pair(sw)[1].data_ptr(), sw.inv_scale.data_ptr(), sw.Kpad,
                                           _opt(bias), None if r2 is None else r2.data_ptr(), 0 if r2 is None else r2.stride(0), out.data_ptr(),
                                           out.stride(0), M, N, K, act, scales.sc.data_ptr(), scales.isc.data_ptr(), _stream())
        _check(rc, "scp_linear_f16x3_scaled")
        return out.reshape(*lead, N)
    ws = torch.empty((2 * M,), dtype=torch.float32, device=x.device)
    rc = lib().scp_linear_f16x3(x2.data_ptr(), x2.stride(0), _tiled_pair(sw)[0].data_ptr(), _tiled_pair(sw)[1].data_ptr(), sw.inv_scale.data_ptr(), sw.Kpad,
                                _opt(bias), None if r2 is None else r2.data_ptr(), 0 if r2 is None else r2.stride(0), out.data_ptr(),
                                out.stride(0), M, N, K, act, ws.data_ptr(), _stream())
    _check(rc, "scp_linear_f16x3")
    return out.reshape(*lead, N)


class SplitAct:
    """An activation [M, K] as bf16 planes hi = bf16(x), lo = bf16(x - hi): t is bfloat16 [2, M, ld] (ld = K rounded up to 32,
    padding columns zero) or a column slice of such a buffer.  This is the operand format of scp_linear_split."""

    __slots__ = ("t", "K")

    def __init__(self, t, K):
        self.t, self.K = t, K

    @staticmethod
    def empty(M, K, device):
        return SplitAct(torch.empty((2, M, -(-K // 32) * 32), dtype=torch.bfloat16, device=device), K)

    @property
    def M(self):
        return self.t.shape[1]

    def cols(self, c0, c1):
        """view of columns [c0, c1) (c0 % 8 == 0)"""
        return SplitAct(self.t[:, :, c0:c1], c1 - c0)

    def float(self):
        return self.t[0, :, :self.K].float() + self.t[1, :, :self.K].float()


def linear_split_scatter(a, sw, bias, out_map, table, act=ACT_NONE, cfg=0):
    """table[out_map[m], :N] = act(a[m] . W^T + bias) for out_map[m] >= 0: fp32 rows written straight to their final positions.
    table: fp32 [rows, ld >= N] with unit column stride (a row-offset view selects the chunk)."""
    t = a.t
    N = sw.N
    if N % 4 and table.stride(0) >= -(-N // 4) * 4 and act == ACT_NONE:
        # ragged width (255 symbols): write the padding columns of the table row as well (zero weights -> just the padded bias) so
        # that every wave takes the 16-byte store path instead of the element-wise edge path
        Np = -(-N // 4) * 4
        if bias is not None:
            pb = getattr(sw, "_bias_pad", None)
            ver = (bias._version, bias.data_ptr())
            if pb is None or pb[0] is not bias or pb[2] != ver:
                pb = (bias, torch.cat((bias.detach().float(), torch.zeros(Np - N, dtype=torch.float32, device=bias.device))).contiguous(), ver)
                sw._bias_pad = pb
                note_cache_fill()
            bias = pb[1]
        N = Np
    _check(lib().scp_linear_split_scatter(t[0].data_ptr(), t[1].data_ptr(), t.stride(1), sw.tiled()[0].data_ptr(), sw.tiled()[1].data_ptr(), sw.Npad, sw.Kpad,
                                          _opt(bias), _dev(out_map, torch.int64), table.data_ptr(), table.stride(0), a.M, N, sw.K, act, cfg,
                                          _stream()), "scp_linear_split_scatter")


def split_rows(x, idx=None, out=None):
    """fp32 rows [n, C] -> SplitAct; idx int64 [rows] gathers rows (idx == n -> zero row)."""
    Cc = x.shape[1]
    rows = x.shape[0] if idx is None else idx.shape[0]
    if out is None:
        out = SplitAct.empty(rows, Cc, x.device)
    if x.stride(1) != 1:
        x = x.contiguous()
    t = out.t
    _check(lib().scp_split_rows(x.data_ptr(), x.stride(0), x.shape[0], _opt(idx), Cc, t[0].data_ptr(), t[1].data_ptr(), t.stride(1), rows,
                                _stream()), "scp_split_rows")
    return out


def linear_split(a, sw, bias=None, act=ACT_NONE, residual=None, out=None, out_split=None, want="f32", cfg=0, res_map=None, res_first=False):
    """a SplitAct [M, K] -> act(a @ W.T + bias) + residual as fp32 [M, N] (want "f32"), as SplitAct (want "split") or both
    (want "both": returns (fp32, SplitAct)).  out / out_split may be column slices of wider buffers."""
    M, N, K = a.M, sw.N, sw.K
    if a.K != K:
        raise ScpError("linear_split: K mismatch %d vs %d" % (a.K, K))
    t = a.t
    c = o = None
    if want in ("f32", "both"):
        c = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=t.device)
    if want in ("split", "both"):
        o = out_split if out_split is not None else SplitAct.empty(M, N, t.device)
    r2 = residual
    if res_map is not None or res_first:
        # out[m] = act(a[m] . W^T + bias + residual[res_map[m]]): gathered residual, added before the activation
        rc = lib().scp_linear_split_gather(t[0].data_ptr(), t[1].data_ptr(), t.stride(1), sw.tiled()[0].data_ptr(), sw.tiled()[1].data_ptr(), sw.Npad, sw.Kpad,
                                           _opt(bias), None if r2 is None else r2.data_ptr(), 0 if r2 is None else r2.stride(0),
                                           None if res_map is None else _dev(res_map, torch.int64),
                                           None if c is None else c.data_ptr(), 0 if c is None else c.stride(0),
                                           None if o is None else o.t[0].data_ptr(), None if o is None else o.t[1].data_ptr(),
                                           0 if o is None else o.t.stride(1), M, N, K, act, cfg, _stream())
        _check(rc, "scp_linear_split_gather")
        return c if want == "f32" else (o if want == "split" else (c, o))
    rc = lib().scp_linear_split(t[0].data_ptr(), t[1].data_ptr(), t.stride(1), sw.tiled()[0].data_ptr(), sw.tiled()[1].data_ptr(), sw.Npad, sw.Kpad, _opt(bias),
                                None if r2 is None else r2.data_ptr(), 0 if r2 is None else r2.stride(0),
                                None if c is None else c.data_ptr(), 0 if c is None else c.stride(0),
                                None if o is None else o.t[0].data_ptr(), None if o is None else o.t[1].data_ptr(),
                                0 if o is None else o.t.stride(1), M, N, K, act, cfg, _stream())
    _check(rc, "scp_linear_split")
    return c if want == "f32" else (o if want == "split" else (c, o))


def linear_split_hier2(a0, sw0, a1, sw1, parent, bias=None, act=ACT_NONE, residual=None, res_map=None, out_split=None):
    """act(a0 . W0^T + a1[parent] . W1^T + bias + residual[res_map]) -> SplitAct [M, N]: the two finest stages of a layer over concat_states
    (ehem.py:75-86) in ONE launch (csrc/gemm_split.hip: gemm_hier2_kernel) - the stage-1 product is computed for the tile's own parents and
    stays in the accumulators, no fp32 partial sum of it exists.  a0: SplitAct [M, K0] (stage-0 rows, M % 256 == 0), a1: SplitAct [M1, 256]
    (stage-1 rows), parent: int64 [M] (stage-1 row of every stage-0 row; consecutive token pairs of a window share consecutive parents),
    residual: fp32 [*, N] rows gathered through res_map (int64 [M]) - the coarser stages' partial sum."""
    if not WTILE:
        raise ScpError("linear_split_hier2 reads tiled weight planes (SCP_WTILE=0 is not supported by it)")
    M, N = a0.M, sw0.N
    if a0.K != sw0.K or a1.K != sw1.K or sw1.K != 256 or sw1.N != N or sw1.Npad != sw0.Npad or parent.shape[0] != M:
        raise ScpError("linear_split_hier2: operand shapes do not match")
    o = out_split if out_split is not None else SplitAct.empty(M, N, a0.t.device)
    t0, t1 = a0.t, a1.t
    rc = lib().scp_linear_split_hier2(t0[0].data_ptr(), t0[1].data_ptr(), t0.stride(1), sw0.Kpad, t1[0].data_ptr(), t1[1].data_ptr(), t1.stride(1), a1.M,
                                      sw0.tiled()[0].data_ptr(), sw0.tiled()[1].data_ptr(), sw1.tiled()[0].data_ptr(), sw1.tiled()[1].data_ptr(), sw0.Npad,
                                      _dev(parent, torch.int64), _opt(bias), None if residual is None else residual.data_ptr(),
                                      0 if residual is None else residual.stride(0), None if (residual is None or res_map is None) else _dev(res_map, torch.int64),
                                      o.t[0].data_ptr(), o.t[1].data_ptr(), o.t.stride(1), M, N, act, _stream())
    _check(rc, "scp_linear_split_hier2")
    return o


class LnFoldedWeight:
    """A Linear that follows a LayerNorm, with the LayerNorm's affine folded in (built once per weight):
    LN(x) . W^T + b = n(x) . (W diag(gamma))^T + (b + W beta), n = the normalised row.  `sw` = split + tiled planes of W diag(gamma),
    `wbeta` = W beta (kept apart from b: rows a window pads AFTER LayerNorm get b alone, swin_transformer.py:638-641)."""

    def __init__(self, w, gamma, beta):
        w64 = w.detach().double()
        self.sw = SplitWeight((w64 * gamma.detach().double()[None, :]).float())
        self.planes = _tiled_planes_always(self.sw)
        self.wbeta = (w64 @ beta.detach().double()).float().contiguous()
        self.N, self.K = self.sw.N, self.sw.K
        note_cache_fill()


def swin_ln_linear(x, fw, bias, eps=1e-5, valid=None, out=None):
    """out = valid * LayerNorm(x) . W^T + bias in one launch (csrc/rowchain.hip): x fp32 [M, 256] rows (unit channel stride),
    fw = LnFoldedWeight, valid fp32 [M] / [M, 1] or None.  -> fp32 [M, N]."""
    M = x.shape[0]
    if x.shape[1] != 256 or fw.K != 256 or x.stride(1) != 1:
        raise ScpError("swin_ln_linear: 256-channel rows expected")
    if out is None:
        out = torch.empty((M, fw.N), dtype=torch.float32, device=x.device)
    t = fw.planes
    rc = lib().scp_swin_ln_linear(x.data_ptr(), x.stride(0), None if valid is None else _dev(valid, torch.float32), t[0].data_ptr(), t[1].data_ptr(),
                                  _opt(bias), _dev(fw.wbeta), float(eps), out.data_ptr(), out.stride(0), M, fw.N, _stream())
    _check(rc, "scp_swin_ln_linear")
    return out


def swin_ln_qkv(x, fw, bias, eps=1e-5, valid=None):
    """swin_ln_linear for a query | key | value (fw.N = 768) or key | value (512) projection whose keys and values leave as the planes of the
    plane-fed window attention: -> (q fp32 [M, 256] or None, KvPlanes).  M % 512 == 0 (rows of the packed layout)."""
    M = x.shape[0]
    if x.shape[1] != 256 or fw.K != 256 or x.stride(1) != 1 or fw.N not in (512, 768) or (M & 127):
        raise ScpError("swin_ln_qkv: 256-channel rows in multiples of 128, N = 768 or 512 expected")
    q = torch.empty((M, 256), dtype=torch.float32, device=x.device) if fw.N == 768 else None
    pl = torch.empty((4, M, 256), dtype=torch.bfloat16, device=x.device)
    t = fw.planes
    rc = lib().scp_swin_ln_qkv(x.data_ptr(), x.stride(0), None if valid is None else _dev(valid, torch.float32), t[0].data_ptr(), t[1].data_ptr(),
                               _opt(bias), _dev(fw.wbeta), float(eps), None if q is None else q.data_ptr(), 256, pl.data_ptr(), M, M, fw.N, _stream())
    _check(rc, "scp_swin_ln_qkv")
    return q, KvPlanes(t=pl)


class EdgeMlpWeights:
    """The two edge MLPs of the geometry feature generator (Sequential(Linear, LeakyReLU, Linear, LeakyReLU, Linear) each: 448 -> 256 -> 256
    -> 256 and 512 -> 256 -> 256 -> 128) in the form scp_geo_edge_mlps streams: eight tiled [256, 256] matrices in a
    scp_swin_post_attn_weight_bytes() buffer (see csrc/rowchain.hip: rc_edge_mlp_kernel) and the six biases back to back."""

    def __init__(self, mlp1, mlp2):
        w11, w12, w13 = (mlp1[i].weight.detach().float() for i in (0, 2, 4))
        w21, w22, w23 = (mlp2[i].weight.detach().float() for i in (0, 2, 4))
        if tuple(w11.shape) != (256, 448) or tuple(w12.shape) != (256, 256) or tuple(w13.shape) != (256, 256) or tuple(w21.shape) != (256, 512) or \
                tuple(w22.shape) != (256, 256) or tuple(w23.shape) != (128, 256):
            raise ScpError("EdgeMlpWeights: unexpected layer shapes")
        dev = w11.device
        P = rc_perm16(256, dev)
        z = torch.zeros((256, 64), dtype=torch.float32, device=dev)
        mats = [w11[:, :256], torch.cat((w11[:, 256:], z), 1), w12[:, P], w13[:, P], w21[:, 256:][:, P], w21[:, :256], w22[:, P],
                torch.cat((w23[:, P], torch.zeros((128, 256), dtype=torch.float32, device=dev)), 0)]
        nbytes = lib().scp_swin_post_attn_weight_bytes()
        buf = torch.zeros((nbytes // 2,), dtype=torch.bfloat16, device=dev)
        half = nbytes // 4
        for m, w in enumerate(mats):
            hi, lo = _tiled_planes_always(SplitWeight(w.contiguous()))
            buf[m * 65536:(m + 1) * 65536] = hi.reshape(-1)[:65536]
            buf[half + m * 65536:half + (m + 1) * 65536] = lo.reshape(-1)[:65536]
        self.packed = buf
        self.bias = torch.cat([mlp1[i].bias.detach().float() for i in (0, 2, 4)] + [mlp2[i].bias.detach().float() for i in (0, 2, 4)]).contiguous()
        note_cache_fill()


def geo_edge_mlps(pos1, pos2, pos3, ew, out):
    """out[:, :128] = edge_mlp2(cat(pos3, edge_mlp1(cat(pos1, pos2, pos3)))) in one launch; pos1 / pos2 / pos3 fp32 [M, 64 | 128 | 256] (unit channel
    stride), out fp32 [M, >= 128] (may be a column slice of a wider buffer)."""
    M = pos1.shape[0]
    if pos1.shape[1] != 64 or pos2.shape[1] != 128 or pos3.shape[1] != 256 or pos1.stride(1) != 1 or pos2.stride(1) != 1 or pos3.stride(1) != 1 or out.stride(1) != 1:
        raise ScpError("geo_edge_mlps: [M, 64], [M, 128], [M, 256] rows expected")
    rc = lib().scp_geo_edge_mlps(pos1.data_ptr(), pos1.stride(0), pos2.data_ptr(), pos2.stride(0), pos3.data_ptr(), pos3.stride(0), ew.packed.data_ptr(),
                                 ew.bias.data_ptr(), out.data_ptr(), out.stride(0), M, _stream())
    _check(rc, "scp_geo_edge_mlps")
    return out


class Mlp3Weights:
    """A three-layer head Sequential(Linear, LeakyReLU, Linear, LeakyReLU, Linear) with 256 inputs and layers of <= 256 outputs in the form
    scp_mlp3_rows streams (csrc/rowchain.hip: rc_mlp3_kernel): three tiled [256, 256] matrices (unused rows / columns zero; layers 2, 3 with their
    columns in accumulator order) in a scp_swin_post_attn_weight_bytes() buffer, the three biases zero padded to 256 back to back."""

    def __init__(self, seq):
        ws = [seq[i].weight.detach().float() for i in (0, 2, 4)]
        bs = [seq[i].bias.detach().float() for i in (0, 2, 4)]
        if ws[0].shape[1] != 256 or any(w.shape[0] > 256 or w.shape[1] > 256 for w in ws) or ws[1].shape[1] != ws[0].shape[0] or ws[2].shape[1] != ws[1].shape[0]:
            raise ScpError("Mlp3Weights: a 256 -> (<= 256) -> (<= 256) -> (<= 256) head expected")
        dev = ws[0].device
        P = rc_perm16(256, dev)
        nbytes = lib().scp_swin_post_attn_weight_bytes()
        buf = torch.zeros((nbytes // 2,), dtype=torch.bfloat16, device=dev)
        half = nbytes // 4
        bias = torch.zeros((3, 256), dtype=torch.float32, device=dev)
        for m, (w, b) in enumerate(zip(ws, bs)):
            full = torch.zeros((256, 256), dtype=torch.float32, device=dev)
            full[:w.shape[0], :w.shape[1]] = w
            if m:
                full = full[:, P]
            hi, lo = _tiled_planes_always(SplitWeight(full.contiguous()))
            buf[m * 65536:(m + 1) * 65536] = hi.reshape(-1)[:65536]
            buf[half + m * 65536:half + (m + 1) * 65536] = lo.reshape(-1)[:65536]
            bias[m, :b.shape[0]] = b
        self.packed, self.bias, self.N = buf, bias.reshape(-1).contiguous(), int(ws[2].shape[0])
        note_cache_fill()


def mlp3_rows(x, mw, out, in_map=None, out_map=None, ncols=None):
    """out[out_map[m] or m, :ncols] = head(x[in_map[m] or m]) in one launch (scp_mlp3_rows): x fp32 [n, 256] rows (unit channel stride), mw = Mlp3Weights,
    out fp32 rows with unit column stride and 16-byte aligned rows (a column-offset view of a wider buffer is fine); in_map / out_map int64 (rows with
    out_map < 0 are dropped); ncols = columns written (default: the head's outputs rounded up to a multiple of 4 - padding columns receive 0)."""
    M = x.shape[0] if in_map is None else in_map.shape[0]
    if out_map is not None and out_map.shape[0] != M:
        raise ScpError("mlp3_rows: out_map must have one entry per row")
    if x.shape[1] != 256 or x.stride(1) != 1 or out.stride(1) != 1:
        raise ScpError("mlp3_rows: [n, 256] rows with unit channel stride expected")
    N = -(-mw.N // 4) * 4 if ncols is None else int(ncols)
    if out_map is None and out.shape[0] < M:
        raise ScpError("mlp3_rows: output has too few rows")
    _check(lib().scp_mlp3_rows(x.data_ptr(), x.stride(0), x.shape[0], None if in_map is None else _dev(in_map, torch.int64), mw.packed.data_ptr(), mw.bias.data_ptr(),
                               out.data_ptr(), out.stride(0), None if out_map is None else _dev(out_map, torch.int64), M, N, _stream()), "scp_mlp3_rows")
    return out


class MergeWeights:
    """SwinPatchMerging's LayerNorm(512) + reduction [256, 512] in the form scp_swin_merge streams (built once per layer): W' = W
    diag(gamma) as two [256, 256] halves, tiled planes at bytes 0 / 131072 of a scp_swin_post_attn_weight_bytes() buffer (lo planes behind
    its first half), and wbeta = W beta."""

    def __init__(self, w, gamma, beta):
        w64 = w.detach().double()
        wf = (w64 * gamma.detach().double()[None, :]).float()
        self.wbeta = (w64 @ beta.detach().double()).float().contiguous()
        nbytes = lib().scp_swin_post_attn_weight_bytes()
        buf = torch.zeros((nbytes // 2,), dtype=torch.bfloat16, device=w.device)
        half = nbytes // 4                                           # bf16 elements per plane region
        for kh in range(2):
            hi, lo = _tiled_planes_always(SplitWeight(wf[:, 256 * kh:256 * (kh + 1)].contiguous()))
            buf[kh * 65536:(kh + 1) * 65536] = hi.reshape(-1)[:65536]
            buf[half + kh * 65536:half + (kh + 1) * 65536] = lo.reshape(-1)[:65536]
        self.packed = buf
        note_cache_fill()


def swin_merge(x, ia, ib, mw, eps=1e-5):
    """LayerNorm(cat(x[ia], x[ib])) . W^T for the M = len(ia) merged rows of a patch-merging layer in one launch (csrc/rowchain.hip);
    an index equal to x.shape[0] stands for a row of zeros.  x fp32 [n, 256] rows (unit channel stride) -> fp32 [M, 256]."""
    M = ia.shape[0]
    if x.shape[1] != 256 or x.stride(1) != 1:
        raise ScpError("swin_merge: 256-channel rows expected")
    out = torch.empty((M, 256), dtype=torch.float32, device=x.device)
    rc = lib().scp_swin_merge(x.data_ptr(), x.stride(0), x.shape[0], _dev(ia, torch.int64), _dev(ib, torch.int64), mw.packed.data_ptr(), _dev(mw.wbeta),
                              float(eps), out.data_ptr(), out.stride(0), M, _stream())
    _check(rc, "scp_swin_merge")
    return out


def rc_perm16(n, device):
    """Column permutation of the row-chain kernels' chained operands (csrc/rowchain.hip): inside every group of 16, columns 4-7 and
    8-11 change places - the order in which an MFMA accumulator holds a row's channels, read as the next product's B fragment."""
    p = torch.arange(n, device=device)
    q = p & 15
    return (p & ~15) | (q & 3) | ((q & 4) << 1) | ((q & 8) >> 1)


class PostAttnWeights:
    """Weights of a Swin block's post-attention half in the form scp_swin_post_attn streams (built once per block): proj as it is;
    fc1 with layernorm_after's affine folded in (W1 diag(gamma), b1 + W1 beta) and its K axis in accumulator order; fc2 with its
    hidden axis in accumulator order.  The kernel evaluates GELU in the variable s y (s = scp_gelu_prescale(), csrc/scp_internal.h): fc1
    (weight and bias) carries the factor s, fc2's weight 1 / s."""

    def __init__(self, wp, bp, gamma, beta, w1, b1, w2, b2):
        dev = wp.device
        perm_k, perm_h = rc_perm16(256, dev), rc_perm16(w1.shape[0], dev)
        w1d = w1.detach().double()
        s = float(lib().scp_gelu_prescale())
        swp = SplitWeight(wp.detach().float().contiguous())
        sw1 = SplitWeight((w1d * gamma.detach().double()[None, :] * s).float()[:, perm_k].contiguous())
        sw2 = SplitWeight((w2.detach().double() / s).float()[:, perm_h].contiguous())
        # one buffer for the kernel's single buffer resource: the tiled planes proj hi | fc1 hi | fc2 hi | proj lo | fc1 lo | fc2 lo
        tiles = [_tiled_planes_always(w) for w in (swp, sw1, sw2)]
        self.packed = torch.cat([t[0].reshape(-1) for t in tiles] + [t[1].reshape(-1) for t in tiles]).contiguous()
        if self.packed.numel() * 2 != lib().scp_swin_post_attn_weight_bytes():
            raise ScpError("swin_post_attn: 256 -> 1024 -> 256 blocks only")
        self.bp = bp.detach().float().contiguous()
        self.b1 = ((b1.detach().double() + w1d @ beta.detach().double()) * s).float().contiguous()
        self.b2 = b2.detach().float().contiguous()
        note_cache_fill()


def swin_post_attn(o, x, pw, eps=1e-5, out=None, tiles=None):
    """x + proj(o) -> LayerNorm -> fc1 -> GELU -> fc2 -> + residual, one launch (csrc/rowchain.hip).  o: SplitAct [M, 256] (the attention
    output planes), x: fp32 [M, 256] residual stream, pw: PostAttnWeights.  -> fp32 [M, 256] (`out` may be x itself).
    tiles (int32 device tensor, optional): the 128-row tiles to process (ascending); the others keep what `out` held (use out=x)."""
    M = x.shape[0]
    if o.K != 256 or o.M != M or x.shape[1] != 256 or x.stride(1) != 1:
        raise ScpError("swin_post_attn: [M, 256] operands expected")
    if out is None:
        out = torch.empty((M, 256), dtype=torch.float32, device=x.device)
    t = o.t
    rc = lib().scp_swin_post_attn(t[0].data_ptr(), t[1].data_ptr(), t.stride(1), x.data_ptr(), x.stride(0), pw.packed.data_ptr(), _dev(pw.bp),
                                  _dev(pw.b1), _dev(pw.b2), float(eps), out.data_ptr(), out.stride(0), M,
                                  None if tiles is None else _dev(tiles, torch.int32), 0 if tiles is None else int(tiles.shape[0]), _stream())
    _check(rc, "scp_swin_post_attn")
    return out


def layernorm_rows(x, gamma, beta, eps=1e-5, valid=None, ia=None, ib=None, out=None, split=False):
    """x [n,256|512...] fp32 rows (unit channel stride).  Plain LN (ia None), LN of gathered rows (ia, C = 256) or of
    cat(x[ia], x[ib]) (C = 512).  valid: fp32 [rows] or [rows,1] multiplier.  Returns [rows, C] fp32, or a SplitAct (split=True)."""
    Cc = gamma.shape[0]
    rows = x.shape[0] if ia is None else ia.shape[0]
    if split:
        o = SplitAct.empty(rows, Cc, x.device)
        rc = lib().scp_layernorm_rows_split(x.data_ptr(), x.stride(0), x.shape[0], None if ia is None else _dev(ia, torch.int64),
                                            None if ib is None else _dev(ib, torch.int64), Cc, _dev(gamma), _dev(beta),
                                            None if valid is None else _dev(valid, torch.float32), float(eps), o.t[0].data_ptr(),
                                            o.t[1].data_ptr(), o.t.stride(1), rows, _stream())
        _check(rc, "scp_layernorm_rows_split")
        return o
    if out is None:
        out = torch.empty((rows, Cc), dtype=torch.float32, device=x.device)
    rc = lib().scp_layernorm_rows(x.data_ptr(), x.stride(0), x.shape[0], None if ia is None else _dev(ia, torch.int64),
                                  None if ib is None else _dev(ib, torch.int64), Cc, _dev(gamma), _dev(beta),
                                  None if valid is None else _dev(valid, torch.float32), float(eps), out.data_ptr(), out.stride(0),
                                  rows, _stream())
    _check(rc, "scp_layernorm_rows")
    return out


def layernorm_add(a, b, gamma, beta, eps=1e-5, planes=False):
    """LayerNorm(a + b) over the last axis (contiguous fp32 tensors of one shape; b may be None) in one kernel.  planes=True: returns
    (out, SplitActF16 of out's rows) - the f16x3 operand of the dense layers that read `out`, written in the same pass."""
    Cc = a.shape[-1]
    if not a.is_contiguous() or (b is not None and (not b.is_contiguous() or b.shape != a.shape)):
        raise ScpError("layernorm_add: contiguous operands of one shape expected")
    out = torch.empty_like(a)
    if planes:
        M, ld = a.numel() // Cc, -(-Cc // 32) * 32
        pl = torch.empty((2, M, ld), dtype=torch.float16, device=a.device)
        ws = torch.empty((2, M), dtype=torch.float32, device=a.device)
        _check(lib().scp_layernorm_add_split_f16(_dev(a, torch.float32), None if b is None else _dev(b, torch.float32), M, Cc, _dev(gamma), _dev(beta),
                                                 float(eps), out.data_ptr(), pl[0].data_ptr(), pl[1].data_ptr(), ld, ws[0].data_ptr(), ws[1].data_ptr(),
                                                 _stream()), "scp_layernorm_add_split_f16")
        return out, SplitActF16(parts=(pl[0], pl[1], ws[0], ws[1], Cc))
    _check(lib().scp_layernorm_add(_dev(a, torch.float32), None if b is None else _dev(b, torch.float32), a.numel() // Cc, Cc, _dev(gamma), _dev(beta),
                                   float(eps), out.data_ptr(), _stream()), "scp_layernorm_add")
    return out


def gather_rows(src, idx, out):
    """out[r, :C] = src[idx[r], :C]; `out` may be a column slice of a wider buffer (C = out.shape[1])."""
    Cc = out.shape[1]
    rc = lib().scp_gather_rows(src.data_ptr(), src.stride(0), _dev(idx, torch.int64), Cc, out.data_ptr(), out.stride(0), idx.shape[0],
                               _stream())
    _check(rc, "scp_gather_rows")
    return out


def linear_f32(x, w, bias=None, act=ACT_NONE):
    """exact fp32, batch-invariant dense layer: x [..., K] (unit last stride) @ w[N,K]^T (+bias, act) -> [..., N]."""
    K = w.shape[1]
    N = w.shape[0]
    lead = x.shape[:-1]
    x2 = x.reshape(-1, K) if x.is_contiguous() else x.contiguous().reshape(-1, K)
    M = x2.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    wc = w if w.is_contiguous() else w.contiguous()
    rc = lib().scp_linear_f32(x2.data_ptr(), x2.stride(0), _dev(wc.detach(), torch.float32), _opt(bias), out.data_ptr(), N, M, N, K, act,
                              _stream())
    _check(rc, "scp_linear_f32")
    return out.reshape(*lead, N)


OCTATTN_MODE = os.environ.get("SCP_OCTATTN", "f16x3")   # "f32": the fp32 MFMA kernel for head width 150 as well


def octattn_attention(q_u, k, k_u, v, v_u, heads, out=None, out_u=None, vmax=None):
    """Dual-stream causal attention (models/attention_model.py:58-95).  Head width 150 (the reference configuration) runs on the
    f16x3 kernel (22-bit operands on f16 MFMA, csrc/octattn_f16.hip); SCP_OCTATTN=f32 or any other width: the fp32 kernels.
    k, k_u, v, v_u may be column slices of one key | value projection output (unit channel stride, one common row stride)."""
    B, c, D = q_u.shape
    if out is None:
        out, out_u = torch.empty_like(q_u), torch.empty_like(q_u)
    elif not (out.is_contiguous() and out_u.is_contiguous() and out.shape == q_u.shape and out_u.shape == q_u.shape):
        raise ScpError("octattn_attention: out / out_u must be contiguous [B, c, D] float32")
    hd = D // heads
    ld = k.stride(-2)
    strided_ok = all(t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1 and t.stride(-2) == ld and (t.dim() == 2 or t.stride(0) == c * ld)
                     for t in (k, k_u, v, v_u))
    if OCTATTN_MODE == "f16x3" and hd == 150 and (D & 3) == 0 and strided_ok:
        nb = lib().scp_octattn_f16x3_ws_bytes(B, c, heads)
        ws = torch.empty((nb + 1024,), dtype=torch.uint8, device=q_u.device)
        off = (-ws.data_ptr()) % 1024
        if vmax is not None:    # max |v| taken by the epilogue of the projection that wrote v (int32 [1]: the bit pattern of the float)
            rc = lib().scp_octattn_attention_f16x3_vmax(_dev(q_u), k.data_ptr(), k_u.data_ptr(), v.data_ptr(), v_u.data_ptr(), ld, B, c, heads, hd, _dev(out),
                                                        _dev(out_u), ws.data_ptr() + off, nb, _dev(vmax, torch.int32), _stream())
            _check(rc, "scp_octattn_attention_f16x3_vmax")
            return out, out_u
        rc = lib().scp_octattn_attention_f16x3(_dev(q_u), k.data_ptr(), k_u.data_ptr(), v.data_ptr(), v_u.data_ptr(), ld, B, c, heads, hd, _dev(out),
                                               _dev(out_u), ws.data_ptr() + off, nb, _stream())
        _check(rc, "scp_octattn_attention_f16x3")
        return out, out_u
    k, k_u, v, v_u = (t.contiguous() for t in (k, k_u, v, v_u))
    rc = lib().scp_octattn_attention(_dev(q_u), _dev(k), _dev(k_u), _dev(v), _dev(v_u), B, c, heads, hd,
                                     _dev(out), _dev(out_u), _stream())
    _check(rc, "scp_octattn_attention")
    return out, out_u


def octattn_embed(ctx, pos, c, occ_enc, level_enc, octant_enc, pos_w, pos_b, pe, level_cap, max_level):
    """OctAttention's input stage in one launch (csrc/octattn_embed.hip): ctx uint8 [n,12], pos f32 [n,4,3] (n = B * c rows) ->
    (E f32 [2, n, D] both streams, SplitActF16 of E's 2 n rows)."""
    n = ctx.shape[0]
    d_occ, d_lvl, d_oct = occ_enc.shape[1], level_enc.shape[1], octant_enc.shape[1]
    d_pos = 0 if pos_w is None else pos_w.shape[0]
    D = 4 * (d_occ + d_lvl + d_oct + d_pos)
    ld = -(-D // 32) * 32
    dev = ctx.device
    E = torch.empty((2, n, D), dtype=torch.float32, device=dev)
    pl = torch.empty((2, 2 * n, ld), dtype=torch.float16, device=dev)
    ws = torch.empty((2, 2 * n), dtype=torch.float32, device=dev)
    rc = lib().scp_octattn_embed(_dev(ctx, torch.uint8), _dev(pos, torch.float32), n, c, _dev(occ_enc, torch.float32), d_occ, _dev(level_enc, torch.float32),
                                 d_lvl, max_level, _dev(octant_enc, torch.float32), d_oct, _opt(pos_w), _opt(pos_b), d_pos, _dev(pe, torch.float32),
                                 level_cap, E.data_ptr(), pl[0].data_ptr(), pl[1].data_ptr(), ld, ws[0].data_ptr(), ws[1].data_ptr(), _stream())
    _check(rc, "scp_octattn_embed")
    return E, SplitActF16(parts=(pl[0], pl[1], ws[0], ws[1], D))


# ------------------------------------------------------------------------------------------------- stage C
def softmax_cdf(logits, sym=None, want_pmf=False, want_lohi=True, want_cdf=False):
    """logits cuda f32 [n,nsym] (row stride allowed) -> dict(pmf, lohi, cdf)."""
    n, nsym = logits.shape
    if logits.stride(1) != 1:
        raise ScpError("softmax_cdf: unit column stride expected")
    dev = logits.device
    pmf = torch.empty((n, nsym), dtype=torch.float32, device=dev) if want_pmf else None
    lohi = torch.empty(n, dtype=torch.int32, device=dev) if (want_lohi and sym is not None) else None
    cdf = torch.empty((n, nsym + 1), dtype=torch.int16, device=dev) if want_cdf else None
    rc = lib().scp_softmax_cdf(logits.data_ptr(), logits.stride(0), n, nsym, _opt(sym), _opt(pmf), _opt(lohi), _opt(cdf),
                               _stream())
    _check(rc, "scp_softmax_cdf")
    return dict(pmf=pmf, lohi=lohi, cdf=cdf)


def pmf_cdf(pmf, sym=None, want_cdf=False):
    n, nsym = pmf.shape
    dev = pmf.device
    lohi = torch.empty(n, dtype=torch.int32, device=dev) if sym is not None else None
    cdf = torch.empty((n, nsym + 1), dtype=torch.int16, device=dev) if want_cdf else None
    rc = lib().scp_pmf_cdf(_dev(pmf, torch.float32), n, nsym, _opt(sym), _opt(lohi), _opt(cdf), _stream())
    _check(rc, "scp_pmf_cdf")
    return dict(lohi=lohi, cdf=cdf)


# ------------------------------------------------------------------------------------------------- range coder (host)
def ac_encode_cdf(cdf, sym):
    """cdf uint16/int16 numpy [n,Lp], sym int16 numpy [n] -> bytes."""
    cdf = np.ascontiguousarray(cdf).view(np.uint16)
    sym = np.ascontiguousarray(sym, np.int16)
    cap = len(sym) * 4 + 64
    while True:
        out = np.empty(cap, np.uint8)
        n = C.c_size_t(0)
        rc = lib().scp_ac_encode_cdf(cdf.ctypes.data, sym.ctypes.data, len(sym), cdf.shape[1], out.ctypes.data, cap,
                                     C.byref(n))
        if rc == -3:
            cap *= 4
            continue
        _check(rc, "scp_ac_encode_cdf")
        return out[: n.value].tobytes()


def ac_encode_lohi(lohi):
    """lohi uint32/int32 numpy [n] (low 16 = c_low, high 16 = c_high, 0 => 65536) -> bytes."""
    lohi = np.ascontiguousarray(lohi).view(np.uint32)
    cap = len(lohi) * 4 + 64
    while True:
        out = np.empty(cap, np.uint8)
        n = C.c_size_t(0)
        rc = lib().scp_ac_encode_lohi(lohi.ctypes.data, len(lohi), out.ctypes.data, cap, C.byref(n))
        if rc == -3:
            cap *= 4
            continue
        _check(rc, "scp_ac_encode_lohi")
        return out[: n.value].tobytes()


class AcDecoder:
    def __init__(self, byte_stream, Lp=256):
        self._buf = np.frombuffer(byte_stream, np.uint8).copy()
        self._h = _vp()
        _check(lib().scp_ac_dec_new(C.byref(self._h), self._buf.ctypes.data, len(self._buf), Lp), "scp_ac_dec_new")

    def next(self, cdf_row):
        row = np.ascontiguousarray(cdf_row).view(np.uint16)
        return _check(lib().scp_ac_dec_next(self._h, row.ctypes.data), "scp_ac_dec_next")

    def run(self, cdf):
        """cdf uint16/int16 numpy [n,Lp] -> int16 numpy [n] (n consecutive symbols)."""
        cdf = np.ascontiguousarray(cdf).view(np.uint16)
        out = np.empty(cdf.shape[0], np.int16)
        _check(lib().scp_ac_dec_run(self._h, cdf.ctypes.data, cdf.shape[0], out.ctypes.data), "scp_ac_dec_run")
        return out

    def __del__(self):
        try:
            if self._h:
                lib().scp_ac_dec_free(self._h)
        except Exception:
            pass
