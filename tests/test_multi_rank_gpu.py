"""The RCCL side of the N > 1 path as far as one GPU can host it: a one-rank nccl process group doing exactly the
calls bench.py makes (tests/nccl_single_rank.py), in a child process so that the group does not outlive the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_nccl_calls_of_the_bench_with_one_rank():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(here, "nccl_single_rank.py")], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "nccl ok" in p.stdout, (p.stdout[-500:], p.stderr[-1500:])
