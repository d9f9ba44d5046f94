"""Builds tuning variants of the library next to the default one (caduceus_amd/libcaduceus_hip_<name>.so) for same-box A/B
runs on the GPU box:  python tools/build_variants.py name=DEF1,DEF2=3 [name2=...]   (an empty define list = default flags)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import _build  # noqa: E402

for spec in sys.argv[1:]:
    name, _, defs = spec.partition("=")
    defines = tuple(d for d in defs.split(",") if d)
    out = os.path.join(_build.HERE, f"libcaduceus_hip_{name}.so")
    _build.build_hip(force=True, verbose=False, defines=defines, out=out)
    print("built", out, defines)
