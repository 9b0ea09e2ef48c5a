import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref_vectors():
    return np.load(os.path.join(GOLDEN, "ref_vectors.npz"))


def load_xlsx_csv(name):
    """-> (class names, boxes [n,8] f64, conf [n], angle [n]) from the committed extract of Output/<name>.xlsx."""
    rows = [l.rstrip("\n").split(",") for l in open(os.path.join(GOLDEN, f"xlsx_{name}.csv"))][1:]
    names = [r[0] for r in rows]
    arr = np.array([[float(v) for v in r[1:]] for r in rows], np.float64)
    return names, arr[:, :8], arr[:, 8], arr[:, 9]


CLASS_IDS = {"Landslide 1": 0, "Strike": 1, "Spring 1": 2, "Minepit 1": 3, "Hillside": 4, "Feuchte": 5, "Torf": 6,
             "Bergsturz": 7, "Landslide 2": 8, "Spring 2": 9, "Spring 3": 10, "Minepit 2": 11}
