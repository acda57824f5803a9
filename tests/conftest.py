import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libopus_ref.so (built here from /root/reference)")


def pytest_collection_modifyitems(config, items):
    import reflib
    if reflib.available():
        return
    skip = pytest.mark.skip(reason="oracle/_ref/libopus_ref.so not built")
    for item in items:
        if "ref" in item.keywords:
            item.add_marker(skip)
