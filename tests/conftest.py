import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def pytest_report_header(config):
    """Which prebuilt binaries (built in the container that has the reference sources, carried along git-ignored)
    this run will find — the GPU CLI test fails without oracle/_ref/GNN_VC_hip, the link-level drop-in tests are
    skipped without the reference sources."""
    names = ["oracle/_ref/GNN_VC_hip", "oracle/_ref/ref_layers.so", "oracle/_ref/GNN_VC_dropin", "oracle/_ref/ref_parse.so",
             "oracle/liboracle.so", "gnn-mwvc_amd/libgnnvc_hip.so", "gnn-mwvc_amd/libgnnvc_host.so"]
    found = [n for n in names if (ROOT / n).exists()]
    missing = [n for n in names if n not in found]
    return [f"prebuilt binaries found: {', '.join(found) or 'none'}", f"prebuilt binaries missing: {', '.join(missing) or 'none'}"]


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
