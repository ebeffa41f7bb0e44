import importlib, os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """The product package directory has a hyphen in its name; import it by path
    and alias it as `adas_amd`."""
    if "adas_amd" in sys.modules:
        return sys.modules["adas_amd"]
    pkg = importlib.import_module("vehicle-cv-adas_amd")
    sys.modules["adas_amd"] = pkg
    return pkg


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
