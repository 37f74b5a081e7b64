import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


def _heartbeat(path, period=60.0):
    """A line per minute into gpurun_out/ while the GPU tests run: the CPU oracle of the full-size parity tests works
    for minutes without printing, and a GPU box kills a command that is silent for 7 minutes."""
    import threading
    import time

    def beat():
        t0 = time.time()
        while True:
            time.sleep(period)
            try:
                with open(path, 'a') as f:
                    f.write('pytest alive, %.0f s\n' % (time.time() - t0))
            except OSError:
                return
    threading.Thread(target=beat, daemon=True).start()


def pytest_sessionstart(session):
    try:
        import torch
        if not torch.cuda.is_available():
            return
        out = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        _heartbeat(os.path.join(out, 'pytest_heartbeat.log'))
    except Exception:
        pass
