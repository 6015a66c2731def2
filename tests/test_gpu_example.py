"""examples/stream_lowercolorado.py end to end on the GPU: the run-set loop as a stream, the reference's stream-output files per run
set, the last run set cross-checked through compute_nhd_routing_v02."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_example_runs_and_agrees_with_the_drop_in(tmp_path):
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "stream_lowercolorado.py"), "--days", "3", "--out", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "final state bit-identical: True" in r.stdout
    files = sorted(os.listdir(out))
    assert len(files) == 3 and all(f.startswith("troute_output_") and f.endswith(".nc") for f in files), files
