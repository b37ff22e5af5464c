import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Host library + oracle are plain g++ builds; make sure they exist.  The HIP
    library is built by __graft_entry__.build() (hipcc) and is NOT rebuilt here."""
    from goblin_amd import build
    build.build_host()
    import oracle_binding
    oracle_binding.build_oracle()


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    gdir = os.path.join(REPO, "tests", "golden")
    with open(os.path.join(gdir, "manifest.json")) as f:
        manifest = json.load(f)

    def load(name):
        return manifest[name], dict(np.load(os.path.join(gdir, name + ".npz")))
    load.names = sorted(manifest)
    load.manifest = manifest
    return load
