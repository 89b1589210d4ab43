"""TEST INFRASTRUCTURE: build + open the host simulation of the C ABI (tests/hostsim/)."""
import ctypes
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "hostsim", "libi2v_hostsim.so")
SRCS = [os.path.join(HERE, "hostsim", "hostsim_backend.cpp"),
        os.path.join(HERE, "..", "image-to-video-i2v-attack_amd", "csrc", "i2v_engine.cpp"),
        os.path.join(HERE, "..", "image-to-video-i2v-attack_amd", "csrc", "i2v_params.h"),
        os.path.join(HERE, "..", "image-to-video-i2v-attack_amd", "csrc", "i2v_kernels.h"),
        os.path.join(HERE, "..", "include", "i2v_hip.h")]
_engine = None


def hostsim_engine():
    """Engine bound to the host simulation (CPU tensors).  Planner tests only."""
    global _engine
    if _engine is None:
        stale = not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in SRCS)
        if stale:
            r = subprocess.run([os.path.join(HERE, "hostsim", "build.sh")], capture_output=True, text=True)
            if r.returncode != 0:
                pytest.fail("hostsim build failed:\n" + r.stderr)
        from i2v_amd import lib
        from i2v_amd.engine import Engine
        capi = lib.bind(ctypes.CDLL(SO))
        assert capi.i2v_backend() == b"hostsim"
        _engine = Engine("cpu", capi=capi)
    return _engine
