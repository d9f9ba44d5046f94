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
