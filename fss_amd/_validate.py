"""Argument checks for the Python classes.

The behaviour to reproduce is the reference binding's error contract (fss_crypto/_validate.py:16-108 in
myl7/fss): which exception type each mistake raises and the wording its tests match by regular expression
(test/test_validation.py:20-177).  Everything here is table-driven: one enumerated-choice check, one tensor
shape/dtype check, one integer-range check, one device check; the public `validate_*` names are the entry
points the scheme classes call."""
from numbers import Integral

import torch

# name of the argument -> the values the reference accepts for it
_CHOICES = {
    "group": ("bytes", "uint"),
    "prg": ("chacha", "aes128_mmo"),
    "pred": ("lt", "gt"),
    "scheme": ("dpf", "dcf"),
}
# name of the tensor argument -> its fixed shape (cws depends on in_bits and is handled separately)
_SHAPES = {"s0": (4,), "s0s": (2, 4), "beta": (4,)}


def _choice(name, value):
    allowed = _CHOICES[name]
    if value not in allowed:
        raise ValueError(f"{name} must be one of {allowed}, got {value!r}")


def _int32_tensor(name, t, shape):
    shape = tuple(shape)
    if t.dtype != torch.int32 or tuple(t.shape) != shape:
        raise TypeError(f"{name} must be a {shape} int32 tensor, got shape {tuple(t.shape)} dtype {t.dtype}")


def _in_range(name, value, lo, hi_exclusive, what):
    if not isinstance(value, Integral) or isinstance(value, bool):
        raise TypeError(f"{name} must be an integer, got {type(value).__name__}")
    if not (lo <= value < hi_exclusive):
        raise ValueError(f"{name} must be in {what}, got {value}")


def validate_in_bits(in_bits):
    if in_bits < 1 or in_bits > 128:
        raise ValueError(f"in_bits must be between 1 and 128, got {in_bits}")


def validate_group(group):
    _choice("group", group)


def validate_prg(prg, scheme):
    _choice("scheme", scheme)
    _choice("prg", prg)


def validate_pred(pred):
    _choice("pred", pred)


def validate_party(party):
    if party != 0 and party != 1:
        raise ValueError(f"party must be 0 or 1, got {party}")


def validate_s0(s0):
    _int32_tensor("s0", s0, _SHAPES["s0"])


def validate_s0s(s0s):
    _int32_tensor("s0s", s0s, _SHAPES["s0s"])


def validate_beta(beta):
    _int32_tensor("beta", beta, _SHAPES["beta"])


def validate_cws(cws, in_bits, slots=None):
    _int32_tensor("cws", cws, (in_bits + 1 if slots is None else slots, 8))


def validate_domain_value(name, value, in_bits):
    _in_range(name, value, 0, 1 << in_bits, f"[0, 2^{in_bits})")


def validate_alpha(alpha, in_bits):
    validate_domain_value("alpha", alpha, in_bits)


def validate_device_match(*tensors):
    seen = sorted({str(t.device) for t in tensors})
    if len(seen) > 1:
        raise RuntimeError("expected all tensors to be on the same device, but found at least two devices, "
                           + ", ".join(seen) + "!")


def validate_cpu_only(*tensors, fn_name=""):
    who = f"{fn_name} expects" if fn_name else "expected"
    for t in tensors:
        if t.device.type != "cpu":
            raise RuntimeError(f"{who} all tensors to be on cpu, but found tensor on {t.device}")
