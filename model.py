"""Drop-in for the reference's `model.py` module: `from model import LowLightEnhance` resolves to the MI355X HIP path."""
import ssie as _ssie

_ssie.load()
from ssie_amd.model import FusedAdam, LowLightEnhance  # noqa: E402,F401

__all__ = ["LowLightEnhance", "FusedAdam"]
