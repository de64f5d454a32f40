import os
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)



def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    d = os.path.join(ROOT, "tests", "golden")

    class G:
        def __getattr__(self, name):
            z = np.load(os.path.join(d, name + ".npz"), allow_pickle=False)
            setattr(self, name, z)
            return z

    return G()


@pytest.fixture(scope="session")
def scenes(golden):
    from kajo_amd.scene import Scene

    z = golden.scenes
    keys = sorted({k.split("/")[0] for k in z.files if "/" in k})
    return {k: Scene.from_npz(z, k + "/", k) for k in keys}
