import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timing: the outcome depends on how long something takes; collected LAST (see below)")


def pytest_collection_modifyitems(config, items):
    """Tests marked `timing` (stop requests, deadlines) run after every other test, whatever their file is called: with `pytest -x`
    a timing test that fails must not hide the parity tests behind it (GPUTEST_r04: 51 of 84 GPU tests were never executed)."""
    items.sort(key=lambda it: 1 if it.get_closest_marker("timing") else 0)   # stable: the order inside each group is kept


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session")
def fixture_corridor():
    with open(os.path.join(ROOT, "tests", "golden", "fixture_corridor.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def known_answers():
    with open(os.path.join(ROOT, "tests", "golden", "known_answers.json")) as f:
        return json.load(f)
