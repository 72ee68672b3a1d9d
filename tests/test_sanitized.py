"""The host builds of the headers the device code shares with its checkers (csrc/trt_filter.h, trt_lightgrid.h, trt_raygrid.h:
table builders, cell look-ups, list packing -- index arithmetic all of it) under the address and undefined-behaviour sanitizers:
the conservativeness tests of tests/test_filter.py, test_lightgrid.py and test_raygrid.py run once more in a child interpreter
whose checkers are compiled with -fsanitize=address,undefined (tests/support.py: TRT_TEST_SANITIZE).  The frames-of-real-scenes
cases stay with the ordinary run (they are the slow ones and add no new code path).  The sanitized runs of the host C
(csrc/host/) and of the oracle are in test_host.py and test_oracle_golden.py."""
import os
import subprocess
import sys

import pytest

import support as T


def test_table_headers_under_address_and_undefined_behaviour_sanitizers():
    if T.SANITIZE:
        pytest.skip("already inside the sanitized run")
    runtimes = [subprocess.run(["gcc", "-print-file-name=" + n], capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(r) and os.path.exists(r) for r in runtimes):
        pytest.skip("no sanitizer runtimes here")
    env = dict(os.environ, TRT_TEST_SANITIZE="1", LD_PRELOAD=":".join(runtimes), ASAN_OPTIONS="detect_leaks=0")  # the interpreter's own leaks are not ours
    here = os.path.dirname(os.path.abspath(__file__))
    run = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu", "-k", "not real_frames"] +
                         [os.path.join(here, f) for f in ("test_filter.py", "test_lightgrid.py", "test_raygrid.py")],
                         env=env, capture_output=True, text=True, timeout=1500, cwd=T.ROOT)
    assert run.returncode == 0 and " passed" in run.stdout, run.stdout[-3000:] + run.stderr[-3000:]
