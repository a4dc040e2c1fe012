import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


@pytest.fixture(scope="session")
def model_text():
    return (ROOT / "gnn-mwvc_amd" / "data" / "mwvc_model.txt").read_text()


@pytest.fixture(scope="session")
def oracle_model(model_text):
    from oracle import oracle_py
    return oracle_py.OracleModel(model_text)


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
