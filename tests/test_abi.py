"""C-ABI library: loads without a GPU, exports every symbol include/scp.h declares; host-side range coder parity."""
import hashlib
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden


def declared_symbols(header="scp.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    return re.findall(r"SCP_API\s+[A-Za-z_0-9 \*]+?\b([A-Za-z_0-9]+)\s*\(", src)


def test_library_exports_every_declared_symbol():
    """Both headers: the drop-in boundary (scp.h) and the test / measurement hooks (scp_debug.h); the hooks and the process-global
    setters are NOT in the public header."""
    import ctypes
    from scp_amd import native
    L = native.lib()
    names, dbg = declared_symbols(), declared_symbols("scp_debug.h")
    assert len(names) >= 30 and len(dbg) >= 6
    missing = [n for n in names + dbg if not hasattr(L, n)]
    assert not missing, missing
    assert not (set(names) & set(dbg))
    for n in ("scp_set_knn_mode", "scp_set_attention_mode", "scp_set_knn_workgroup", "scp_knn_debug_buffer", "scp_rc_debug_buffer", "scp_prof_enable", "scp_prof_count", "scp_prof_read"):
        assert n in dbg and n not in names
    for n in ("scp_ctx_create", "scp_ctx_set", "scp_ctx_make_current", "scp_swin_ln_linear", "scp_swin_post_attn", "scp_swin_post_attn_weight_bytes"):
        assert n in names
    import re
    hdr = open(os.path.join(ROOT, 'include', 'scp.h')).read()
    assert L.scp_version() == int(re.search(r'#define SCP_ABI_VERSION (\d+)', hdr).group(1)) == native.ABI_VERSION
    assert isinstance(L, ctypes.CDLL)


def _tile(base, n):
    return np.tile(base, (-(-n // len(base)), 1))[:n]


def test_range_coder_matches_reference_streams(orc):
    from scp_amd import native
    z = golden("ac_streams")
    for n in (1, 2, 1000):
        pdf = _tile(z[f"n{n}_pdfbase"], n)
        cdf = orc.pmf_to_cdf(pdf)
        sym = z[f"n{n}_sym"]
        want = z[f"n{n}_bytes"].tobytes()
        assert native.ac_encode_cdf(cdf, sym) == want
        s = sym.astype(np.int64)
        lo = cdf[np.arange(n), s].astype(np.uint32)
        hi = np.where(s == 254, 0, cdf[np.arange(n), np.minimum(s + 1, 255)]).astype(np.uint32)
        assert native.ac_encode_lohi(lo | (hi << 16)) == want
    n = 100000
    pdf = _tile(z[f"n{n}_pdfbase"], n)
    cdf = orc.pmf_to_cdf(pdf)
    bs = native.ac_encode_cdf(cdf, z[f"n{n}_sym"])
    assert hashlib.sha256(bs).hexdigest() == str(z[f"n{n}_sha"])
    dec = native.AcDecoder(bs)
    assert [dec.next(cdf[i]) for i in range(4000)] == z[f"n{n}_sym"][:4000].tolist()
    pdf = np.tile(z["pend_pdf_row"], (len(z["pend_sym"]), 1))
    assert native.ac_encode_cdf(orc.pmf_to_cdf(pdf), z["pend_sym"]) == z["pend_bytes"].tobytes()


def test_range_coder_equals_oracle_coder_on_adversarial_tables(orc):
    """The product coder renormalises with count-leading-zeros bursts; the oracle (a restatement of numpyAc_backend.cpp:245-323)
    shifts bit by bit.  Random CDF tables from flat to extremely peaked - long runs of shared leading bits, long underflow
    (pending) runs, the 0x10000 top symbol - must give identical streams."""
    from scp_amd import native
    rng = np.random.default_rng(7)
    for sharp in (0.0, 2.0, 8.0, 40.0):
        n = 20000
        logits = rng.standard_normal((n, 255)) * sharp
        if sharp >= 8.0:
            logits[:, 127] += 3 * sharp                      # mass piles up around the middle of the range: underflow runs
        p = np.exp(logits - logits.max(1, keepdims=True))
        pdf = (p / p.sum(1, keepdims=True)).astype(np.float32)
        cdf = orc.pmf_to_cdf(pdf)
        cum = np.cumsum(pdf.astype(np.float64), 1)
        sym = np.minimum((rng.random((n, 1)) * cum[:, -1:] > cum).sum(1), 254).astype(np.int16)   # sampled from the model
        sym[::97] = 254                                      # the top symbol (c_high = 0x10000)
        sym[5::101] = rng.integers(0, 255, len(sym[5::101]))  # and improbable ones
        want = orc.ac_encode(cdf.view(np.uint16), sym)
        assert native.ac_encode_cdf(cdf, sym) == want, sharp


def test_range_coder_argument_errors():
    from scp_amd import native
    import ctypes as C
    L = native.lib()
    n = C.c_size_t(0)
    out = np.zeros(16, np.uint8)
    cdf = np.zeros((1, 256), np.uint16)
    sym = np.array([300], np.int16)   # out of range symbol
    assert L.scp_ac_encode_cdf(cdf.ctypes.data, sym.ctypes.data, 1, 256, out.ctypes.data, 16, C.byref(n)) == -1
    assert L.scp_ac_encode_cdf(None, sym.ctypes.data, 1, 256, out.ctypes.data, 16, C.byref(n)) == -1
    # too-small output buffer
    cdf = np.tile(np.arange(256, dtype=np.uint16) * 255, (64, 1))
    sym = np.full(64, 3, np.int16)
    assert L.scp_ac_encode_cdf(cdf.ctypes.data, sym.ctypes.data, 64, 256, out.ctypes.data, 2, C.byref(n)) == -3


def test_model_kernel_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument checks come before any HIP call: NULL pointers, wrong widths and misaligned strides return SCP_EINVAL (-1) on a
    machine without a GPU - nothing is launched, nothing falls back to a CPU path."""
    from scp_amd import native
    L = native.lib()
    z, one = None, 4096            # NULL and a fake (never dereferenced) non-NULL address
    # the launch brackets of scp_debug.h work without a GPU as far as they can: nothing recorded, nothing read
    assert L.scp_prof_enable(0) == 0 and L.scp_prof_count() == 0 and L.scp_prof_read(0, z, z, z) == 0 and L.scp_prof_read(4, z, z, z) == -1
    # f16x3 dense layer: K % 4, missing workspace
    assert L.scp_linear_f16x3(one, 600, one, one, one, 608, z, z, 0, one, 600, 8, 600, 600, 0, z, z) == -1
    assert L.scp_linear_f16x3(one, 602, one, one, one, 608, z, z, 0, one, 600, 8, 600, 602, 0, one, z) == -1
    assert L.scp_split_weight_f16(z, 600, 600, 640, 608, one, one, one, z) == -1
    assert L.scp_split_weight_f16(one, 600, 600, 600, 608, one, one, one, z) == -1                               # Npad % 128
    # f16x3 OctAttention: head width other than 150, window longer than 1024, workspace too small / misaligned
    assert L.scp_octattn_f16x3_ws_bytes(0, 1024, 4) == -1
    need = L.scp_octattn_f16x3_ws_bytes(2, 1024, 4)
    assert need > 2 * 32 * 4 * (22528 + 25600)
    assert L.scp_octattn_attention_f16x3(one, one, one, one, one, 600, 2, 1024, 4, 128, one, one, 4096, need, z) == -1
    assert L.scp_octattn_attention_f16x3(one, one, one, one, one, 600, 2, 2048, 4, 150, one, one, 4096, need, z) == -1
    assert L.scp_octattn_attention_f16x3(one, one, one, one, one, 600, 2, 1024, 4, 150, one, one, 4096, need - 1, z) == -1
    assert L.scp_octattn_attention_f16x3(one, one, one, one, one, 600, 2, 1024, 4, 150, one, one, 4097, need, z) == -1
    assert L.scp_octattn_attention_f16x3(one, one, one, one, one, 598, 2, 1024, 4, 150, one, one, 4096, need, z) == -1     # k / v row stride below the row width


def test_numeric_profile_contexts_are_per_handle_and_per_thread():
    """scp_ctx: the numeric profile belongs to a handle; the current context is per host thread; bad keys / values are refused.  (No
    GPU call: only the host-side state.)"""
    import ctypes as C
    import threading
    from scp_amd import native
    L = native.lib()
    a, b = C.c_void_p(), C.c_void_p()
    assert L.scp_ctx_create(C.byref(a)) == 0 and L.scp_ctx_create(C.byref(b)) == 0
    assert L.scp_ctx_get(a, 1) == 1 and L.scp_ctx_get(a, 2) == 1
    assert L.scp_ctx_set(a, 1, 0) == 0 and L.scp_ctx_get(a, 1) == 0 and L.scp_ctx_get(b, 1) == 1       # b untouched
    assert L.scp_ctx_set(a, 7, 1) == -1 and L.scp_ctx_set(a, 1, 2) == -1 and L.scp_ctx_set(None, 1, 1) == -1 and L.scp_ctx_create(None) == -1
    p1 = native.NumericProfile(knn_f16x3=False)
    p2 = native.NumericProfile()
    assert "knn=f32" in p1.describe("EHEM") and "knn=f16x3" in p2.describe("EHEM")
    seen = {}

    def worker():
        seen["other thread"] = native.current_profile()
        with native.use_profile(p2):
            seen["other thread inside"] = native.current_profile()
    with native.use_profile(p1):
        assert native.current_profile() is p1
        t = threading.Thread(target=worker); t.start(); t.join()
        with native.use_profile(p2):
            assert native.current_profile() is p2
        assert native.current_profile() is p1
    assert native.current_profile() is None and seen["other thread"] is None and seen["other thread inside"] is p2
    assert L.scp_ctx_destroy(a) == 0 and L.scp_ctx_destroy(b) == 0 and L.scp_ctx_destroy(None) == -1


def test_row_chain_entry_points_reject_bad_arguments_without_a_gpu():
    from scp_amd import native
    L = native.lib()
    z, one = None, 4096
    assert L.scp_swin_post_attn_weight_bytes() == 2 * (256 * 512 + 1024 * 512 + 256 * 2048)
    assert L.scp_swin_ln_linear(z, 256, z, one, one, z, z, 1e-5, one, 768, 10, 768, z) == -1              # x NULL
    assert L.scp_swin_ln_linear(one, 256, z, one, one, z, z, 1e-5, one, 768, 10, 700, z) == -1            # N % 128
    assert L.scp_swin_ln_linear(one, 250, z, one, one, z, z, 1e-5, one, 768, 10, 768, z) == -1            # ldx < 256
    assert L.scp_swin_ln_linear(one, 256, z, one, one, z, z, 1e-5, one, 512, 10, 768, z) == -1            # ldo < N
    assert L.scp_swin_ln_linear(one, 256, z, one, one, z, z, 1e-5, one, 768, 0, 768, z) == -1             # M == 0
    assert L.scp_swin_post_attn(one, one, 256, one, 256, z, one, one, one, 1e-5, one, 256, 10, z, 0, z) == -1    # W NULL
    assert L.scp_swin_post_attn(one, one, 250, one, 256, one, one, one, one, 1e-5, one, 256, 10, z, 0, z) == -1  # ldo_in
    assert L.scp_swin_post_attn(one, one, 256, one, 256, one, one, one, one, 1e-5, one, 254, 10, z, 0, z) == -1  # ldc
    assert L.scp_swin_post_attn(one, one, 256, one, 256, one, one, one, one, 1e-5, one, 256, -1, z, 0, z) == -1  # M
    assert L.scp_swin_post_attn(one, one, 256, one, 256, one, one, one, one, 1e-5, one, 256, 300, one, 4, z) == -1  # more listed tiles than the rows hold


def test_range_coder_round_trips_under_address_and_ub_sanitizers(tmp_path):
    """csrc/rangecoder.cpp (host code: burst renormalisation, 64-bit bit window, AVX2 symbol search with its verified fallback) built with
    -fsanitize=address,undefined on the CPU and driven through the C ABI: 200 random streams (peaky / flat / tail-heavy tables, 1 - 5 000
    symbols, the stream handed over in an exact-size buffer so that a read past its end is caught), scp_ac_dec_run and scp_ac_dec_next mixed."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "ac_sanitize"
    cmd = ["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "src", "ac_sanitize.cpp"), os.path.join(ROOT, "scp_amd", "csrc", "rangecoder.cpp"), "-o", str(exe)]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if b.returncode != 0 and "sanitize" in b.stdout:
        pytest.skip("this g++ has no sanitizer runtime")
    assert b.returncode == 0, b.stdout[-2000:]
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "200 round trips ok" in r.stdout, r.stdout[-3000:]
