"""Test-only stand-in for OpenAI `clip` (absent, and its weights are unobtainable offline).

The reference calls clip.load("ViT-B/32") once and encode_text(tokenize(prompts)) to get a
(T,512) tensor (MP_HSIR.py:512-515).  This shim returns the seeded surrogate from
tests/golden/detfill.py instead, so goldens and tests agree on the injected tensor.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from golden.detfill import surrogate_clip_prompt  # noqa: E402


class _TextOnlyModel:
    def encode_text(self, tokens):
        return surrogate_clip_prompt(len(tokens))


def load(name, device="cpu"):
    return _TextOnlyModel(), None


def tokenize(prompts):
    return list(prompts)
