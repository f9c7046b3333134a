#!/usr/bin/env python3
"""Generate tests/golden/validation_contract.json FROM THE REFERENCE ITSELF (build container only).

Imports /root/reference/fss_crypto/_validate.py and _int.py by file path -- pure Python; nothing of the reference's
JIT / native path is imported, and no reference source is copied anywhere -- drives every validator over the case table
below and records, per case, the exception type and exact message it raises, or the value it returns.  The committed
JSON is data (inputs and expected outcomes); tests/test_validation.py::test_contract_* replays it against
fss_amd/_validate.py and fss_amd/_int.py on any box (the reference does not exist on the GPU box).

The case table covers every good value and every wrong shape / dtype / range / type / device that the reference's own
test/test_validation.py:20-177 names, plus the boundaries around them and the 128-bit splits of _int.py:4-8.  Devices
other than cpu are represented by torch's "meta" device, which needs no GPU and takes the same code path
(t.device / t.device.type) the validators read.

usage: python3 tools/gen_validation_fixtures.py [--reference /root/reference]
"""
import argparse
import importlib.util
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "validation_contract.json")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def T(shape, dtype="int32", device="cpu"):
    return {"__tensor__": {"shape": list(shape), "dtype": dtype, "device": device}}


def decode(a):
    """JSON argument -> Python value (shared with tests/test_validation.py through the fixture's own description)."""
    if isinstance(a, dict) and "__tensor__" in a:
        d = a["__tensor__"]
        return torch.zeros(tuple(d["shape"]), dtype=getattr(torch, d["dtype"]), device=d["device"])
    if isinstance(a, dict) and "__int__" in a:
        return int(a["__int__"])          # integers beyond 2^53 travel as decimal strings
    if isinstance(a, dict) and "__float__" in a:
        return float(a["__float__"])
    return a


def I(v):
    return {"__int__": str(v)}


def cases():
    c = []
    add = lambda fn, *args, **kw: c.append({"fn": fn, "args": list(args), "kwargs": kw})
    for v in (1, 2, 64, 127, 128, 0, 129, -1, 1000):
        add("validate_in_bits", v)
    for v in ("bytes", "uint", "invalid", "Bytes", "", "u32"):
        add("validate_group", v)
    for prg in ("chacha", "aes128_mmo", "invalid", "aes", ""):
        for scheme in ("dpf", "dcf", "invalid", "vdpf"):
            add("validate_prg", prg, scheme)
    for v in ("lt", "gt", "eq", "le", "LT", ""):
        add("validate_pred", v)
    for v in (0, 1, 2, -1, True, False, 3):
        add("validate_party", v)
    shapes = {"validate_s0": [(4,), (2, 4), (3,), (4, 1), ()],
              "validate_s0s": [(2, 4), (3, 4), (2, 3), (4,), (8,), (1, 2, 4)],
              "validate_beta": [(4,), (3,), (2, 4), (5,)]}
    for fn, shs in shapes.items():
        for sh in shs:
            for dt in ("int32", "float32", "int64", "uint8"):
                add(fn, T(sh, dt))
        add(fn, T(shs[0], "int32", "meta"))     # the device is not this validator's business
    for in_bits in (1, 16, 20, 64, 128):
        for rows in (in_bits + 1, in_bits, in_bits + 2, 1):
            add("validate_cws", T((rows, 8)), in_bits)
        add("validate_cws", T((in_bits + 1, 4)), in_bits)
        add("validate_cws", T((in_bits + 1, 8), "float32"), in_bits)
        add("validate_cws", T((in_bits + 1, 8), "int64"), in_bits)
        add("validate_cws", T((8, in_bits + 1)), in_bits)
    for in_bits in (1, 8, 20, 32, 64, 100, 128):
        top = 1 << in_bits
        for v in (0, 1, top - 1, top, top + 1, -1, -top):
            add("validate_alpha", I(v), in_bits)
            add("validate_domain_value", "x", I(v), in_bits)
    for bad in (True, False, {"__float__": "1.5"}, {"__float__": "0.0"}, "7", None):
        add("validate_alpha", bad, 20)
        add("validate_domain_value", "x", bad, 20)
        add("validate_domain_value", "alpha", bad, 20)
    add("validate_domain_value", "y", I(5), 3)
    add("validate_domain_value", "y", I(8), 3)
    cpu, meta = T((4,)), T((4,), "int32", "meta")
    add("validate_device_match", cpu)
    add("validate_device_match", cpu, cpu)
    add("validate_device_match", cpu, cpu, cpu)
    add("validate_device_match", meta, meta)
    add("validate_device_match", cpu, meta)
    add("validate_device_match", meta, cpu)
    add("validate_device_match", cpu, cpu, meta)
    add("validate_device_match")
    for fn_name in ("", "eval_all", "gen"):
        add("validate_cpu_only", cpu, fn_name=fn_name)
        add("validate_cpu_only", cpu, cpu, fn_name=fn_name)
        add("validate_cpu_only", meta, fn_name=fn_name)
        add("validate_cpu_only", cpu, meta, fn_name=fn_name)
    add("validate_cpu_only", cpu)
    add("validate_cpu_only", meta)
    for v in (0, 1, (1 << 64) - 1, 1 << 64, (1 << 64) + 5, (1 << 127) + 12345, (1 << 128) - 1, 1 << 128,
              0x0123456789ABCDEF_FEDCBA9876543210, (1 << 130) + 3):
        add("split_uint128", I(v))
    return c


def outcome(fn, args, kwargs):
    try:
        r = fn(*[decode(a) for a in args], **kwargs)
    except Exception as e:  # noqa: BLE001 -- the exception IS the recorded behaviour
        return {"raises": type(e).__name__, "message": str(e)}
    if isinstance(r, tuple):
        return {"returns": [I(v) if isinstance(v, int) else v for v in r]}
    return {"returns": r}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    a = ap.parse_args()
    val = load(os.path.join(a.reference, "fss_crypto", "_validate.py"), "ref_validate")
    rint = load(os.path.join(a.reference, "fss_crypto", "_int.py"), "ref_int")
    out = []
    for c in cases():
        mod = rint if c["fn"] == "split_uint128" else val
        out.append(dict(c, expect=outcome(getattr(mod, c["fn"]), c["args"], c["kwargs"])))
    doc = {
        "_provenance": "generated by tools/gen_validation_fixtures.py from the reference's own fss_crypto/_validate.py and "
                       "fss_crypto/_int.py (imported by path in the build container; pure Python, no native path). "
                       "Each case: function, JSON-encoded arguments, and the exception type + exact message the "
                       "REFERENCE raised, or the value it returned.",
        "torch": torch.__version__, "python": sys.version.split()[0],
        "encoding": {"__tensor__": "torch.zeros(shape, dtype, device); device 'meta' stands for any non-cpu device",
                     "__int__": "decimal string of a Python int", "__float__": "decimal string of a Python float"},
        "cases": out,
    }
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=0)
        f.write("\n")
    raised = sum("raises" in c["expect"] for c in out)
    print(f"{OUT}: {len(out)} cases ({raised} raise, {len(out) - raised} return)")


if __name__ == "__main__":
    main()
