import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
