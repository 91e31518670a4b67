import json
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


def load_gv(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    inp = json.loads(str(d["inputs"]))
    return dict(x=d["x"], rows=d["rows"], row_idx=d["row_idx"], inputs=inp)


def start_vector(inp):
    return np.concatenate([inp["s"], inp["s_ang"], np.ravel(inp["ee"]), inp["s_vel"], inp["s_ang_vel"]])


@pytest.fixture(scope="session")
def cfg():
    from qtos_amd.config import PlannerConfig
    return PlannerConfig.reference_compat()


@pytest.fixture(scope="session")
def oracle(cfg):
    from oracle.oracle import Oracle, oracle_dict
    return Oracle(oracle_dict(cfg))


@pytest.fixture(scope="session")
def gv1():
    return load_gv("gv1")


@pytest.fixture(scope="session")
def gv2():
    return load_gv("gv2")


def oracle_problem(O, inp):
    return O.problem(inp["s"], inp["s_ang"], inp["ee"], inp["g"], inp["s_vel"], inp["s_ang_vel"], inp["t0"])


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if needed) and load the C-ABI library; CPU tests only check that it loads."""
    import subprocess
    from qtos_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.dirname(capi.LIB_PATH), "-s"])
    return capi.load()
