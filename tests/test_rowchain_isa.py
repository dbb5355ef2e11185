"""csrc/rowchain.hip keeps weight fragments in flight behind hand-counted waits (inline-asm LDS reads): the generated gfx950 code must
never touch a fragment register between the read that fills it and the wait that retires it (tools/scan_rowchain_isa.py).  Runs on the
build machine: hipcc cross-compiles to assembly without a GPU."""
import os
import re
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (shutil.which(HIPCC) or os.path.exists(HIPCC)), reason="hipcc not available")
def test_no_fragment_register_is_touched_before_its_wait(tmp_path):
    src = os.path.join(ROOT, "scp_amd", "csrc", "rowchain.hip")
    asm = tmp_path / "rowchain.s"
    r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", src, "-o", str(asm)],
                       capture_output=True, text=True, cwd=os.path.join(ROOT, "scp_amd", "csrc"))
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text()
    kernels = re.findall(r"^(_Z\w*rc_\w+kernel\w*):", text, re.M)
    assert len(kernels) >= 2, kernels
    lines = text.splitlines()
    for k in kernels:
        start = next(i for i, l in enumerate(lines) if l.startswith(k + ":"))
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        body = tmp_path / (k + ".s")
        body.write_text("\n".join(lines[start:end + 1]))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_rowchain_isa.py"), str(body)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        m = re.search(r"hazards: (\d+) of (\d+) instructions", out.stdout)
        assert m and int(m.group(2)) > 1000, out.stdout[-500:]
        assert int(m.group(1)) == 0, (k, out.stdout[-1500:])
        # the reads the scan is about are there, and so are the counted waits
        assert "ds_read_b128" in body.read_text() and "lgkmcnt(3)" in body.read_text()
