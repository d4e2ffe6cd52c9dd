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


class DeviceBuffer:
    """A hipMalloc'ed buffer for tests that hand device pointers to the C ABI, allocated through the HIP runtime the
    renderer library itself runs on (importing another framework for this could bring a second ROCm installation
    into the process)."""

    def __init__(self, nbytes):
        import ctypes as C
        from vulkanrenderer_amd import renderer
        renderer.lib()                                 # loads the runtime; the name below then resolves to that copy
        self._hip = C.CDLL("libamdhip64.so.7")
        self._hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self._hip.hipFree.argtypes = [C.c_void_p]
        self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.nbytes = nbytes
        p = C.c_void_p()
        assert self._hip.hipMalloc(C.byref(p), nbytes) == 0
        self.ptr = p.value

    def zero(self):
        assert self._hip.hipMemset(self.ptr, 0, self.nbytes) == 0
        assert self._hip.hipDeviceSynchronize() == 0

    def numpy(self, shape, dtype):
        import numpy as np
        out = np.zeros(shape, dtype)
        assert out.nbytes == self.nbytes
        assert self._hip.hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2) == 0   # hipMemcpyDeviceToHost
        return out

    def free(self):
        if self.ptr:
            self._hip.hipFree(self.ptr)
            self.ptr = None
