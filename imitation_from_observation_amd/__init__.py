"""MI355X-native context-translation encoder/decoder ("translator") of imitation_from_observation.

The compute path is libctxtrans.so (hand-written HIP for gfx950, C ABI in include/ctxtrans.h);
this package is the thin Python host the reference's rllab reward hook / training script talk to.
"""
from .translator import CtxError, Translator  # noqa: F401

__all__ = ["Translator", "CtxError"]
