import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


IDENTITY = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]


def pose_matrix(tx=0.0, ty=0.0, tz=0.0, yaw=0.0, pitch=0.0):
    """camera2world = translate * rotateY(yaw) * rotateX(pitch), glm column-major float[16]."""
    import numpy as np
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    ry = np.array([[cy, 0, sy, 0], [0, 1, 0, 0], [-sy, 0, cy, 0], [0, 0, 0, 1]], np.float64)
    rx = np.array([[1, 0, 0, 0], [0, cp, -sp, 0], [0, sp, cp, 0], [0, 0, 0, 1]], np.float64)
    t = np.eye(4)
    t[:3, 3] = [tx, ty, tz]
    m = (t @ ry @ rx).astype(np.float32)
    return m.T.reshape(-1).tolist()


@pytest.fixture(scope="session")
def micro():
    from vulkanrenderer_amd import scenes
    return scenes.micro_scene()


@pytest.fixture(scope="session")
def micro_oracle(micro):
    from oracle import oracle
    return oracle.OracleScene(micro)
