import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    """min(affinity, cgroup CPU quota): the GPU box shows 256 logical CPUs but caps the container at 16; a torch thread pool
    sized by os.cpu_count() makes the CPU oracle ~100x slower there."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock assertions on an idle MI355X (run explicitly with -m perf; never part of -m gpu)")
    import torch
    torch.set_num_threads(_usable_cores())


@pytest.fixture(scope="session")
def vq16_sd():
    from shapeformer_amd import weights as W
    return W.make_state_dict(W.vqdif_spec(16))


@pytest.fixture(scope="session")
def vq16_sd_t(vq16_sd):
    from oracle import vqdif_oracle as O
    return O.to_torch_sd(vq16_sd)


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
