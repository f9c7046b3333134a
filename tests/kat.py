"""Helpers shared by the KAT tests: the reference's known answers (tests/golden)."""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    with open(os.path.join(_HERE, "golden", "reference_kats.json")) as f:
        return json.load(f)


def words(hexes):
    return np.array([int(h, 16) for h in hexes], dtype=np.uint32).view(np.int32)
