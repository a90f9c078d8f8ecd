import os
import sys

import numpy as np
import pytest
import torch

# the oracle is many tiny CPU ops: four threads run the suite as fast as eight (3 min on 8 cores) and do not collapse when
# something else is using the machine (torch's OpenMP threads spin: with a build running next to the suite at 8 threads
# a 17-s test took over 15 min)
torch.set_num_threads(max(1, min(4, os.cpu_count() or 1)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


def rot_angle_deg(Ra, Rb):
    c = (np.trace(np.asarray(Ra, np.float64).T @ np.asarray(Rb, np.float64)) - 1) / 2
    return float(np.degrees(np.arccos(np.clip(c, -1, 1))))
