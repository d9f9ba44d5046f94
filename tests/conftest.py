import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _intkeys(d):
    return {(int(k) if isinstance(k, str) and k.lstrip("-").isdigit() else k): v for k, v in d.items()}


def load_golden_model(name):
    """Returns (cfg dict, state dict, record dict of torch tensors) for tests/golden/model_<name>.npz."""
    z = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    cfg = json.loads(bytes(z["cfg"]).decode(), object_hook=_intkeys)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    rec = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("sd/") and k not in ("cfg", "meta")}
    return cfg, sd, rec


MODEL_VARIANTS = ["ps_fused", "ps_unfused", "ph_fused", "ph_unfused", "ps_fused_ewmul", "ps_fused_untied",
                  "ps_fused_unidir", "ps_fused_layernorm", "ps_fused_res32_odd"]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- kernel backends ---------------------------------------------------------------------------------------------
# "emu": the kernel sources compiled for the host emulator (tests/emu), CPU tensors -- runs in the build container.
# "hip": the real gfx950 library on cuda:0 -- the parity tests proper (pytest -m gpu on the MI355X box).
@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    from caduceus_amd import _lib
    if request.param == "emu":
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        from build_emu import build_emu
        _lib.use_library_for_testing(build_emu())
        yield "emu", torch.device("cpu")
        _lib.use_library_for_testing(None)
    else:
        _lib.use_library_for_testing(None)
        assert torch.cuda.is_available(), "gpu-marked test without a GPU"
        assert _lib.is_device_build(), "libcaduceus_hip.so must be the gfx950 build"
        yield "hip", torch.device("cuda:0")
