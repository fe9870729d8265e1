"""Shared fixtures. Tests marked `gpu` need a real MI355X; everything else runs on CPU."""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_pkg():
    """Import the package directory `dart-ray_amd/` under the importable name `dartray_amd`."""
    if "dartray_amd" in sys.modules:
        return sys.modules["dartray_amd"]
    path = os.path.join(ROOT, "dart-ray_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("dartray_amd", path,
                                                  submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["dartray_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


# No torch.cuda.init() here any more. Round 2 initialised torch first in a session fixture because its lazy initialisation,
# reached late from a test that aliases device buffers, failed with "No HIP GPUs are available". The cause is now known
# (tools/repro_torch_late_init.py): the torch wheel bundles its own HIP/HSA runtime, and a process that maps
# /opt/rocm's copy first (the library loaded before torch) ends up with two runtimes, the second of which sees no device.
# host.lib() imports torch before it loads the library, so the product path is protected, not just the tests
# (tests/test_abi.py::test_one_hip_runtime_in_the_process).


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def grids(pkg):
    return pkg.grids


GOLDEN_CASES = ["cube65", "slab1801", "blobA", "blobB", "blobC", "blobD", "blobE", "trusti32"]


@pytest.fixture(scope="session")
def golden():
    """name -> (dict of golden arrays, path of the case file materialised in a temp dir)."""
    import tempfile
    tmp = tempfile.mkdtemp(prefix="dartray_golden_")
    cache = {}

    def get(name):
        if name not in cache:
            z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
            path = os.path.join(tmp, name + ".bin")
            z["case_bin"].tofile(path)
            cache[name] = (z, path)
        return cache[name]

    return get


def _relelem(a, b, floor):
    s = np.abs(b).max()
    m = np.abs(b) > floor * s
    return float((np.abs(a[m] - b[m]) / np.abs(b[m])).max()) if (s > 0 and m.any()) else 0.0


def golden_scaspe_check(z, prefix, il, arr, rtol, label, rtol_elem=None, floor=1e-12):
    """Compare one scaspe array with what the fixture holds for it (full, or sums + samples): relative to the
    array maximum (rtol) and, when rtol_elem is given, element by element for everything above floor * maximum."""
    key = f"{prefix}{il}_full"
    if key in z:
        ref = z[key]
        scale = np.abs(ref).max()
        assert np.abs(arr - ref).max() <= rtol * scale, f"{label} {prefix}{il}"
        if rtol_elem is not None:
            assert _relelem(arr, ref, floor) <= rtol_elem, f"{label} {prefix}{il} element-wise"
        return
    cs, ps = z[f"{prefix}{il}_cellsum"], z[f"{prefix}{il}_pixsum"]
    assert np.abs(arr.sum(axis=1) - cs).max() <= rtol * np.abs(cs).max(), f"{label} {prefix}{il} cellsum"
    assert np.abs(arr.sum(axis=0) - ps).max() <= rtol * np.abs(ps).max(), f"{label} {prefix}{il} pixsum"
    idx, smp = z[f"{prefix}{il}_sample_idx"], z[f"{prefix}{il}_sample"]
    if idx.size:
        assert np.abs(arr[idx] - smp).max() <= rtol * np.abs(smp).max(), f"{label} {prefix}{il} samples"
        if rtol_elem is not None:
            assert _relelem(arr[idx], smp, floor) <= rtol_elem, f"{label} {prefix}{il} samples element-wise"
            assert _relelem(arr.sum(axis=1), cs, floor) <= rtol_elem, f"{label} {prefix}{il} cellsum element-wise"
