import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The ctypes-bound HIP library; GPU tests fail loudly if it is not built/loadable."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test collected on a machine without a GPU"
    from reftr_amd import hip as H
    H.lib()
    return H


@pytest.fixture(autouse=True)
def _release_captures_between_tests(request):
    """Dead CapturedTrainStep objects (hipGraphs + their private memory pools) are released BETWEEN tests, with the device idle, instead
    of whenever the cyclic collector happens to run (round 6: collections inside a later test's capture / replay brought the suite down).
    (The kernel-level files never capture: skipped there, they are 450 of the suite's tests.)"""
    yield
    if os.path.basename(str(request.node.fspath)) in ("test_gemm_gpu.py", "test_ops_gpu.py", "test_seg_ops_gpu.py", "test_post_gpu.py", "test_abi.py"):
        return
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass


# ---- measured-vs-gate table: parity tests call `record_parity(test, quantity, measured, gate)`; the rows are
# printed as ONE table at the end of the run (GPU log: profiles/*_gpu_tests.log), so the distance to every gate is visible in one place.
_PARITY_ROWS = []


def record_parity(test, quantity, measured, gate):
    _PARITY_ROWS.append((test, quantity, float(measured), float(gate)))


@pytest.fixture()
def parity_table():
    return record_parity


_NOTES = []


@pytest.fixture()
def suite_note():
    """Lines a test wants in the run's terminal summary (e.g. the summary line of a child pytest process)."""
    return _NOTES.append


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    for line in _NOTES:
        terminalreporter.write_line(line)
    if not _PARITY_ROWS:
        return
    tr = terminalreporter
    tr.write_sep("=", "parity: measured vs gate (north_star states 1e-3 rel on bf16 logits; see README 'Parity, honestly')")
    tr.write_line(f"{'test':58s} {'quantity':64s} {'measured':>10s} {'gate':>10s} {'measured/gate':>14s}")
    for t, q, m, g in _PARITY_ROWS:
        tr.write_line(f"{t[:58]:58s} {q[:64]:64s} {m:10.3e} {g:10.3e} {m / g if g else float('nan'):14.2f}")
