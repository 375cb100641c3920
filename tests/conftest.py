import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
KARATE = os.path.join(GOLDEN, "karate.txt")
TESTGRAPH = os.path.join(GOLDEN, "testgraph.txt")


# environment variables that may be present without steering a kernel, a planner or a host path
_NEUTRAL = {"SRW_SKIP_FULL_SIZE", "SRW_TIMING", "SRW_CHECK_QUICK"}


def steering_switches_set():
    """SRW_* variables of tools/SWITCHES.md that are set in this process's environment (other than the neutral ones)."""
    return sorted(k for k in os.environ if k.startswith("SRW_") and k not in _NEUTRAL)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The suite describes the DEFAULTS: a session that starts with an experiment switch in its environment is refused (tests that
    need one set it themselves, with monkeypatch, and take it away again)."""
    bad = steering_switches_set()
    if bad:
        pytest.exit("steering switches are set in the environment: %s (tools/SWITCHES.md) — unset them" % ", ".join(bad), returncode=3)


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session", autouse=True)
def _torch_cuda_first():
    """torch ships its own HIP runtime next to the system one libstellar_rw.so links; initialise torch's device
    context BEFORE the product library launches its first kernel (the order bench.py uses), otherwise a later
    torch.cuda init in the same process can report "No HIP GPUs are available" (seen on the round-2 GPU boxes)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield
