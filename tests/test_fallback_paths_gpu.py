"""The shift operators have several kernel families behind one entry point (LDS-DMA streaming: plane-group / column-walk /
tile and the 2-D twins; column; per-plane generic).  The default dispatch exercises the streaming and column kernels; this
re-runs the parity suites in a subprocess on each setting of the library's one switch, RK_SHIFT_KERNELS
(rk_common.hpp), so the fallbacks a production box would land on (a buffer that is not 16-byte aligned, a shape no
streaming kernel takes) stay bit-exact too."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [
    {"RK_SHIFT_KERNELS": "column"},                # no LDS-DMA streaming kernels
    {"RK_SHIFT_KERNELS": "generic"},               # per-plane generic kernels only
    {"RK_FORCE_GENERIC": "1"},                     # older spelling of the same
    {"RK_SLAB14": "1"},                            # 14x14 planes on the slab kernels instead of the tile kernels
], ids=["column", "generic", "force-generic", "slab14"])
def test_parity_suite_on_fallback_kernels(env):
    e = dict(os.environ, **env)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-k", "bit_exact or shift_grad",
           os.path.join(ROOT, "tests", "test_parity_3d.py"), os.path.join(ROOT, "tests", "test_parity_2d.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
