"""pytest config: registers the ``gpu`` marker and puts the product package dir on sys.path.

``-m "not gpu"`` runs here on CPU (oracle vs golden fixtures, host logic, C-ABI symbol
check, gloo world_size-2 sharding); ``-m gpu`` runs on one MI355X and calls the HIP path
through the C-ABI, checked against the oracle and the committed fixtures.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "msra-practice-project_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP path through the C-ABI)")
    # Diagnostic builds only (tools/diag_build.sh, e.g. a sin range-reduction variant being priced against the parity
    # records): MI_DIAG_LIB=gpurun_tools/libmirender_sin2.so runs the suite against that library instead of the
    # product's.  Never set by the driver; the records then carry the library's name.
    diag = os.environ.get("MI_DIAG_LIB")
    if diag:
        from mirender import _lib
        _lib.LIB_PATH = os.path.join(ROOT, diag)
        print(f"[conftest] DIAGNOSTIC LIBRARY {_lib.LIB_PATH}", flush=True)


def load_golden(name):
    """Load a committed fixture (plain arrays only; allow_pickle stays False)."""
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def pytest_sessionfinish(session, exitstatus):
    """Every parity check leaves a record (oracle/parity.py): achieved error, gate, which bound was active.  The GPU
    session writes them under gpurun_out/ (merged back by gpurun); the committed copy is profiles/r04_parity.json (one per round)."""
    from oracle import parity
    out = os.environ.get("MI_PARITY_JSON", os.path.join(ROOT, "gpurun_out", "r04_parity.json"))
    try:
        parity.write_records(out)
    except Exception as e:      # noqa: BLE001  (bookkeeping must never turn a green run red or hide a red one)
        print(f"parity records not written: {e!r}")
