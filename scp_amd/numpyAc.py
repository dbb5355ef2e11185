"""numpyAc-compatible arithmetic coder API (drop-in for numpyAc/numpyAc.py:123-169).

`arithmeticCoding().encode(pdf, sym, binfile)` and `arithmeticDeCoding(byte_stream, sysNum, symDim, binfile)` keep the
reference's names and argument order.  The PMF -> integer CDF conversion (numpyAc.py:109-114,80-107) runs on the MI355X
(csrc/cdf.hip, bit-exact with numpy's serial float32 cumsum) and the 32-bit range coder is the C++ one in
csrc/rangecoder.cpp - there is no JIT build at import time and no torch extension.
"""
import numpy as np
import torch

from . import native

PRECISION = 16  # numpyAc.py:8


def _device():
    if not torch.cuda.is_available():
        raise native.ScpError("numpyAc on this stack evaluates CDFs on the MI355X: no GPU visible")
    return torch.device("cuda", torch.cuda.current_device())


def pmf_to_cdf_int(pdf):
    """float32 [N,L] (numpy or device tensor) -> uint16 numpy [N,L+1]; == _convert_to_int_and_normalize(pdf_convert_...(pdf))."""
    t = torch.from_numpy(np.ascontiguousarray(pdf, np.float32)).to(_device()) if isinstance(pdf, np.ndarray) else pdf
    return native.pmf_cdf(t.contiguous(), None, want_cdf=True)["cdf"].cpu().numpy().view(np.uint16)


class arithmeticCoding:
    def __init__(self):
        self.binfile = None
        self.sysNum = None
        self.byte_stream = None

    def encode(self, pdf, sym, binfile=None):
        assert pdf.shape[0] == sym.shape[0]
        assert pdf.ndim == 2 and sym.ndim == 1
        self.sysNum = sym.shape[0]
        sym = np.asarray(sym)
        if sym.min(initial=0) < 0 or sym.max(initial=0) > pdf.shape[1] - 1:
            raise ValueError(f"sym.max() == {sym.max()}, should be <=Lp - 1.!")          # numpyAc.py:36-39
        dev = _device()
        t = torch.from_numpy(np.ascontiguousarray(pdf, np.float32)).to(dev) if isinstance(pdf, np.ndarray) else pdf
        s = torch.from_numpy(sym.astype(np.uint8)).to(dev)
        lohi = native.pmf_cdf(t.contiguous(), s)["lohi"].cpu().numpy()
        self.byte_stream = native.ac_encode_lohi(lohi)
        real_bits = len(self.byte_stream) * 8
        if binfile is not None:
            with open(binfile, "wb") as fout:
                fout.write(self.byte_stream)
        return self.byte_stream, real_bits


class arithmeticDeCoding:
    """byte_stream / sysNum / symDim / binfile as in numpyAc.py:139-154."""

    def __init__(self, byte_stream, sysNum, symDim, binfile=None):
        if binfile is not None:
            with open(binfile, "rb") as fin:
                byte_stream = fin.read()
        self.byte_stream = byte_stream
        self.decoder = native.AcDecoder(byte_stream, symDim + 1)

    def decode(self, pdf):
        return self.decoder.next(pmf_to_cdf_int(pdf)[0])

    def decode_ehem(self, pdf):
        return [self.decoder.next(r) for r in pmf_to_cdf_int(pdf)]
