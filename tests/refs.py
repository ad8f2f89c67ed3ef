"""Test-side names of the checkers and of the public-API binding (see oracle/bindings.py and linne_amd/api.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from bindings import (MAX_CH, MAX_LAYERS, MAX_PARAMS, ORACLE_SO, PRESET_LAYERS, REF_SO, ChannelTap, EncodeParameter,  # noqa: E402,F401
                      FrameTap, Oracle, Reference, fnv1a64, oracle_available, reference_available)
from linne_amd.api import (LinneApi, _planar_ptrs, _RefDecoderConfig, _RefEncodeParameter, _RefEncoderConfig,  # noqa: E402,F401
                           _RefHeader)
