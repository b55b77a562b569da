"""A fixed-seed slice of the randomised parity fuzzers (tests/fuzz/*.py: HIP path vs the oracles on random shapes and configurations)
inside the suite the driver runs -- the full runs (1,000+ cases) are `python tests/fuzz/<name>.py [n] [seed]` on the GPU box.
Each fuzzer is a program of its own (it prints every failing case and exits non-zero): run as a child process, one after the other."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("script,n", [("fuzz_parity.py", 100),      # MaskCBAM / MaskECA: every launch-geometry branch, mask kinds, conv sizes
                                      ("fuzz_head.py", 60),         # MGAMaskHead: channel / hidden widths off the MFMA tile sizes, odd H*W, wide rows, half I/O
                                      ("fuzz_pyramid.py", 25),      # grouped pyramid calls vs per-level calls
                                      ("fuzz_rows.py", 25)])        # segmentation loss (both modes, resized targets) and the ProbMaskGater launch
def test_fuzzer_slice_is_clean(built_lib, script, n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", script), str(n), "11"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, f"{script}: rc={r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-1500:]}"
