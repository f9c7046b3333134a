"""Argument validation with the error contract of the reference's Python binding
(fss_crypto/_validate.py:16-108; the message shapes are what test/test_validation.py
matches by regex)."""
from numbers import Integral

import torch

VALID_GROUPS = ("bytes", "uint")
VALID_PRGS = ("chacha", "aes128_mmo")
VALID_PREDS = ("lt", "gt")


def validate_in_bits(in_bits):
    if not (1 <= in_bits <= 128):
        raise ValueError(f"in_bits must be between 1 and 128, got {in_bits}")


def validate_group(group):
    if group not in VALID_GROUPS:
        raise ValueError(f"group must be one of {VALID_GROUPS}, got {group!r}")


def validate_prg(prg, scheme):
    if scheme not in ("dpf", "dcf"):
        raise ValueError(f"scheme must be one of ('dpf', 'dcf'), got {scheme!r}")
    if prg not in VALID_PRGS:
        raise ValueError(f"prg must be one of {VALID_PRGS}, got {prg!r}")


def validate_pred(pred):
    if pred not in VALID_PREDS:
        raise ValueError(f"pred must be one of {VALID_PREDS}, got {pred!r}")


def validate_party(party):
    if party not in (0, 1):
        raise ValueError(f"party must be 0 or 1, got {party}")


def _shape_dtype(name, t, shape):
    if tuple(t.shape) != tuple(shape) or t.dtype != torch.int32:
        raise TypeError(f"{name} must be a {tuple(shape)} int32 tensor, got shape {tuple(t.shape)} dtype {t.dtype}")


def validate_s0(s0):
    _shape_dtype("s0", s0, (4,))


def validate_s0s(s0s):
    _shape_dtype("s0s", s0s, (2, 4))


def validate_beta(beta):
    _shape_dtype("beta", beta, (4,))


def validate_cws(cws, in_bits, slots=None):
    _shape_dtype("cws", cws, ((in_bits + 1) if slots is None else slots, 8))


def validate_domain_value(name, value, in_bits):
    if isinstance(value, bool) or not isinstance(value, Integral):
        raise TypeError(f"{name} must be an integer, got {type(value).__name__}")
    if value < 0 or value >= (1 << in_bits):
        raise ValueError(f"{name} must be in [0, 2^{in_bits}), got {value}")


def validate_alpha(alpha, in_bits):
    validate_domain_value("alpha", alpha, in_bits)


def validate_device_match(*tensors):
    devices = {t.device for t in tensors}
    if len(devices) > 1:
        dev_list = ", ".join(str(d) for d in sorted(devices, key=str))
        raise RuntimeError(
            f"expected all tensors to be on the same device, but found at least two devices, {dev_list}!")


def validate_cpu_only(*tensors, fn_name=""):
    for t in tensors:
        if t.device.type != "cpu":
            prefix = f"{fn_name} expects" if fn_name else "expected"
            raise RuntimeError(f"{prefix} all tensors to be on cpu, but found tensor on {t.device}")
