"""CPU: the library's host code under AddressSanitizer and UBSan (tools/hostcheck): the OBJ reader on hand-written, malformed
and 3000 random inputs, the plane generator, the PPM writers, the band layout and the argument checks of the entry points
that need a GPU. The device code is compiled as usual; nothing runs on a GPU."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("hipcc is not installed")
    env = dict(os.environ, RT_HOSTCHECK_TMP=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "hostcheck", "run.sh")], capture_output=True, text=True, timeout=900, env=env)
    out = r.stdout + r.stderr
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
    assert r.returncode == 0 and "host check: 0 failure(s)" in out, out[-4000:]
