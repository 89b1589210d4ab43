import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(ROOT, "image-to-video-i2v-attack_amd")
for p in (PKG_ROOT, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# no ImageNet checkpoints exist offline: the suites run on the seeded synthetic initialiser, which the product only
# uses when asked to (i2v_amd/weights.py)
os.environ.setdefault("I2V_SYNTHETIC_WEIGHTS", "1")
os.environ.setdefault("I2V_QUIET_WEIGHTS", "1")
# the CLIs pin their process to its rank's CPU cores (i2v_amd/affinity.py); tests call their main() IN this process, sometimes with
# WORLD_SIZE set -- the test process itself must keep all its cores (the subprocess tests that check the pinning ask for it)
os.environ["I2V_PIN_CPUS"] = "0"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library(request):
    """GPU runs need the in-tree gfx950 library; build it when the snapshot arrived without it."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        import __graft_entry__ as ge
        if not os.path.exists(ge.LIB):
            ge.build()


@pytest.fixture
def experimental_build():
    """Tests of the kernels that are NOT in the product library (fused 3x3 -> pointwise pair, split-bf16 loop, conv_pw_stream,
    conv_stem64_halo: `csrc/i2v_conv_exp.hip`, compiled only with -DI2V_EXPERIMENTAL) run when the loaded library carries them --
    `python __graft_entry__.py --experimental`, then `I2V_LIB=.../libi2v_hip_exp.so pytest -m gpu` -- and skip on the default build."""
    from i2v_amd import lib
    if lib.load().i2v_backend_stat(b"experimental") != 1:
        pytest.skip("the loaded libi2v_hip.so is the product build (no -DI2V_EXPERIMENTAL): this kernel is not in it")
