import os
import sys
from pathlib import Path

import numpy as np
import pytest

# Test frames are tiny; in the product "auto" hands launches under 70 Mpx to the low-latency vector kernels.
# The parity tests want the LDS-window tile kernels under "auto", so the boundary is moved to 0 for the test
# processes (and the workers they start); test_small_jobs_take_the_low_latency_kernels covers the routing.
import os  # noqa: E402
os.environ.setdefault("LUTR_SMALL_JOB_MPX", "0")

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    from oracle import binding
    binding.load()
    return binding


@pytest.fixture(scope="session")
def cube_dir(tmp_path_factory):
    """Generated .cube files shared by the whole session."""
    from lut_renderer_amd import cube
    d = tmp_path_factory.mktemp("cubes")
    cube.write_cube(d / "identity_33.cube", cube.identity_lattice(33), title="identity 33")
    cube.write_cube(d / "identity_17.cube", cube.identity_lattice(17))
    cube.write_cube(d / "log709_33.cube", cube.log709_lattice(33), title="log709 33")
    cube.write_cube(d / "log709_65.cube", cube.log709_lattice(65), title="log709 65")
    rng = np.random.default_rng(7)
    cube.write_cube(d / "random_9.cube", rng.uniform(-0.2, 1.2, size=(9, 9, 9, 3)).astype(np.float32))
    cube.write_cube(d / "random_2.cube", rng.uniform(0.0, 1.0, size=(2, 2, 2, 3)).astype(np.float32))
    cube.write_cube(d / "domain_2.cube", cube.log709_lattice(17), domain_min=(0, 0, 0), domain_max=(2, 2, 2))
    return d


@pytest.fixture(scope="session")
def engine():
    from lut_renderer_amd.engine import LutEngine
    eng = LutEngine(0)
    yield eng
    eng.close()
