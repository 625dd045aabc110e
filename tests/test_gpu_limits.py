"""The ABI's size limit exercised for real (include/gridhip.h, "Limits"): one call of 2^31 - 256 visibilities."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_largest_call_the_abi_takes():
    """tools/max_size_check.py: n = 2^31 - 256 on the headline shape, device-resident (about 180 GB of HBM) - the grid's
    analytic checksum, errors == 0, nothing dropped, the tap-reusing path, degrid2's first and last 10^6 predictions
    against calls of those visibilities alone, and n + 1 refused.  No oracle grids 2 x 10^9 visibilities; the properties
    checked do not depend on the size."""
    import torch
    torch.cuda.empty_cache()  # (what earlier tests of this process left in torch's cache)
    free, _ = torch.cuda.mem_get_info(0)
    if free < 200 * 2**30:
        pytest.skip(f"needs ~180 GB of free HBM, {free / 2**30:.0f} GiB free")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "max_size_check.py")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert "max size ok" in out.stdout
