import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the reference build oracle/_ref (only where /root/reference exists)")


def pytest_collection_modifyitems(config, items):
    from tests import helpers
    skip_ref = pytest.mark.skip(reason="oracle/_ref/librdref.so not built (reference sources unavailable)")
    for item in items:
        if "ref" in item.keywords and not helpers.have_ref():
            item.add_marker(skip_ref)
