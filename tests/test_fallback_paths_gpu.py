"""The 3-D operator has four kernel families behind one entry point (LDS-DMA streaming, register-staged
streaming, column, per-plane generic).  The default dispatch exercises the first and the column kernels; this
re-runs a slice of the parity suite in a subprocess with the faster families switched off, so the fallbacks a
production box would land on (odd alignment, RK_* overrides) stay bit-exact too."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [
    {"RK_DMA": "0", "RK_DMA_BWD": "0"},            # register-staged streaming kernels (rk3d_stream.hpp)
    {"RK_DMA": "0", "RK_DMA_BWD": "0", "RK_COLUMN": "0", "RK_DMA2D": "0", "RK_COLUMN2D": "0"},   # + no column / 2-D streaming kernels
    {"RK_FORCE_GENERIC": "1"},                     # per-plane generic kernels only
], ids=["register-staged", "no-column-no-2d-streaming", "generic-only"])
def test_parity_suite_on_fallback_kernels(env):
    e = dict(os.environ, **env)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-k", "bit_exact or shift_grad",
           os.path.join(ROOT, "tests", "test_parity_3d.py"), os.path.join(ROOT, "tests", "test_parity_2d.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
