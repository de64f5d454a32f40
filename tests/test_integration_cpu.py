"""The drop-in boundary, checked on the CPU: integration/apply_to_kajo.sh applied to a scratch copy of the two reference
files it edits, and the backend it installs compiled (-fsyntax-only) against the REFERENCE's own headers -- including the
declarations of renderer/Preview.h:15-24 (non-virtual processEvents/update, private constructor, static create), streamed
from the reference file with only its SDL/GL #include lines replaced. Needs /root/reference (absent on the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "renderer")), reason="reference tree not present")
def test_refcheck_backend_compiles_against_reference_headers():
    p = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "kajo_amd", "host"), "refcheck"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "compile against the reference's Scene.h / Image.h / Scheduler.h / Preview.h declarations" in p.stdout, p.stdout


def test_backend_source_uses_only_the_reference_preview_interface():
    # hip::Scheduler may call nothing on Preview beyond what renderer/Preview.h:23-24 declares
    import re
    src = open(os.path.join(ROOT, "kajo_amd", "host", "HipScheduler.cpp")).read()
    calls = set(re.findall(r"preview->(\w+)\(", src))
    assert calls <= {"processEvents", "update"}, calls
    assert "PassBudgetPreview" not in src and "setPassBudget" not in src
