import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_table():
    """The bundled synthetic 1127-entry returns table (percent units)."""
    vals = []
    with open(os.path.join(ROOT, "data", "SP500_monthly_returns.csv")) as f:
        header = f.readline().strip().split(",")
        col = header.index("returns")
        for line in f:
            cell = line.rstrip("\n").split(",")[col]
            if cell != "":
                vals.append(np.float32(cell))
    return np.array(vals, dtype=np.float32)


@pytest.fixture(scope="session")
def table():
    t = load_table()
    assert t.size == 1127
    return t


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
