import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build the product library (and the C oracle) when the tree was checked out without built artefacts.
    Building is not a fallback: the tests still run the HIP path through the C ABI or fail."""
    import shutil
    import subprocess
    so = os.path.join(ROOT, "lidar_slam_amd", "lib", "liblslam_hip.so")
    hostcheck = os.path.join(ROOT, "lidar_slam_amd", "lib", "liblslam_hostcheck.so")
    pyhost = os.path.join(ROOT, "lidar_slam_amd", "lib", "_lslam_pyhost.so")
    if (not os.path.exists(so) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"))) or not os.path.exists(hostcheck) or not os.path.exists(pyhost):
        subprocess.run(["make", "-C", os.path.join(ROOT, "lidar_slam_amd", "csrc"), "-j4"], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "liboracle_icp.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "csrc")], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def split(cat, off):
    return [cat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


# stated parity tolerance of a relative pose vs the float64 reference
TOL_TRANS = 5e-5      # metres
TOL_ROT = 5e-6        # radians (checked on the rotation-matrix entries)
