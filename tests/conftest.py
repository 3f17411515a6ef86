import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # should the process die of a signal inside native code (pytest captures stderr and loses it then), the library
    # leaves its native backtrace here
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    os.environ.setdefault("ATLITE_HIP_BACKTRACE", str(ROOT / "gpurun_out" / "fatal_backtrace.log"))
    # the shared library is a build artefact (git-ignored): build it on a fresh checkout
    lib = ROOT / "atlite_amd" / "lib" / "libatlite_hip.so"
    if not lib.exists() and os.environ.get("ATLITE_HIP_LIB") is None:
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def ctx():
    """HIP context on cuda:0; fails loudly if the extension or the GPU is missing."""
    from atlite_amd.device import Context

    return Context(int(os.environ.get("ATLITE_HIP_DEVICE", "0")))


@pytest.fixture(scope="module", autouse=True)
def _finalize_module_garbage(request):
    """After every test module: collect cyclic garbage NOW (contexts, device groups, communicators, plans and readers whose
    finalizers free device memory and destroy streams) and let the device drain - so that such finalizers run at a module
    boundary, on an idle device, instead of at whatever later allocation happens to trigger the cycle collector."""
    yield
    import gc

    gc.collect()
    if "torch" in sys.modules:
        torch = sys.modules["torch"]
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    gc.collect()
