import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_ctx():
    """One libptmi context on GPU 0 for the whole session."""
    if not _have_gpu():
        pytest.skip("no GPU")
    from pathtrace_amd import Context
    # The integrator runs its vote-scheduled traversal kernels on launches large enough to feed every resident wave and
    # one-shot kernels below that; test scenes are small, so the shared context forces the vote-scheduled kernels for
    # every launch size (the parity tests then check them against the oracle), and
    # test_traversal_kernel_forms_agree_bit_for_bit ties the one-shot forms and the default mix to them bit for bit.
    old = os.environ.get("PTMI_RENDER_VOTE_MIN")
    os.environ["PTMI_RENDER_VOTE_MIN"] = "0"
    try:
        ctx = Context(0)
    finally:
        if old is None:
            del os.environ["PTMI_RENDER_VOTE_MIN"]
        else:
            os.environ["PTMI_RENDER_VOTE_MIN"] = old
    yield ctx
    ctx.close()
