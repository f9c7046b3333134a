"""Integer helper of the Python layer: a domain value (x, alpha) of up to 128 bits travels to the native side as two
unsigned 64-bit halves, as in the reference binding (fss_crypto/_int.py:4-8, _csrc/binding_common.cuh:39-46).
Pinned to the reference's own function by tests/golden/validation_contract.json."""

_MASK64 = (1 << 64) - 1


def split_uint128(value):
    """(low 64 bits, next 64 bits) of a Python integer; bits above 127 are dropped."""
    return value & _MASK64, (value >> 64) & _MASK64
