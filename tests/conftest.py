import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def had_table(golden_dir):
    """K -> (K, K) int8 +-1 matrices and n -> K dispatch captured from the reference."""
    import numpy as np

    t = np.load(os.path.join(golden_dir, "hadk_table.npz"))
    mats = {}
    for k in (12, 20, 28, 36, 40, 52, 60, 108, 140, 156, 172):
        bits = np.unpackbits(t[f"had{k}"])[: k * k].reshape(k, k).astype(np.int8)
        mats[k] = bits * 2 - 1
    def words(h):
        K = h.shape[0]
        wpr = (K + 31) // 32
        bits = np.zeros((K, wpr * 32), dtype=np.uint8)
        bits[:, :K] = h > 0
        return np.packbits(bits.reshape(K, wpr, 32), axis=-1, bitorder="little").view("<u4").reshape(K, wpr).view(np.int32)

    return {"mats": mats, "packed": {k: t[f"had{k}"] for k in mats},
            "words": {k: words(m) for k, m in mats.items()},
            "n2k": dict(zip(t["n"].tolist(), t["K"].tolist()))}
