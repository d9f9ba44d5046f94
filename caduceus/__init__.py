"""`caduceus`: the reference's import name for the MI355X-native engine.

`/root/reference/src/utils/registry.py:29` resolves the dotted string `caduceus.modeling_caduceus.CaduceusForMaskedLM`,
`configs/model/caduceus.yaml:4` the Hydra `_target_` `caduceus.configuration_caduceus.CaduceusConfig`, and hub checkpoints
name `caduceus.*` modules in their auto_map.  With this repository on `sys.path` (instead of the reference's own
`caduceus/` directory) those strings load `caduceus_amd` UNCHANGED: the submodules below are not copies, they ARE the
`caduceus_amd` modules (registered under both names in `sys.modules`), so `isinstance` checks, pickles and state dicts
agree whichever name a caller used.
"""
import importlib
import sys

import caduceus_amd as _impl

for _name in ("configuration_caduceus", "modeling_caduceus", "modeling_rcps", "tokenization_caduceus"):
    _mod = importlib.import_module("caduceus_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod

# same public names as /root/reference/caduceus/__init__.py:5-7 (+ the rest of caduceus_amd's surface)
from caduceus_amd import *  # noqa: E402,F401,F403
from caduceus_amd import __all__ as _all  # noqa: E402

__all__ = list(_all)
register_auto_classes = _impl.register_auto_classes
