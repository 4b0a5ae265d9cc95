"""Drop-in import surface of the reference (`from models import IQ`, reference models/__init__.py:1): the MI355X-native IQ."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
import bltvqg_amd  # noqa: E402,F401
from bltvqg_amd.iq import IQ  # noqa: E402

__all__ = ["IQ"]
