"""The C ABI from a process without Python or torch: tests/native/capi_driver (plain C++, built by
`__graft_entry__.build()`) calls every entry point of include/marl_amd.h and the argument checks'
failure paths, then steps the headline batch as one handle and as two shards on two streams from one
hipGraph (same rows from both)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_entry_point_from_plain_cpp():
    exe = os.path.join(ROOT, "tests", "native", "capi_driver")
    assert os.path.exists(exe), "tests/native/capi_driver is not built: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "capi host pass: ok" in r.stdout
    assert "two shards on two streams" in r.stdout
    print(r.stdout)
