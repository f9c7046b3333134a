"""Constructor / argument validation of the Python mirror: same error types and message
shapes as the reference binding (test/test_validation.py:20-177, fss_crypto/_validate.py:16-108).
Runs without a GPU: validation happens before the engine is touched."""
import pytest
import torch

import fss_amd


def test_in_bits_range():
    for bad in (0, 129, -1):
        with pytest.raises(ValueError, match="in_bits must be between 1 and 128"):
            fss_amd.Dpf(bad)
    assert fss_amd.Dpf(100, "uint")._g == fss_amd.U128 and fss_amd.Dpf(128).in_bits == 128


def test_group_prg_pred():
    with pytest.raises(ValueError, match="group must be one of"):
        fss_amd.Dpf(16, group="floats")
    with pytest.raises(ValueError, match="prg must be one of"):
        fss_amd.Dpf(16, prg="sha")
    assert fss_amd.Dcf(16, prg="chacha").prg == "chacha"
    with pytest.raises(ValueError, match="pred must be one of"):
        fss_amd.Dcf(16, pred="le")
    assert fss_amd.Dcf(16, pred="gt").pred == "gt"
    assert fss_amd.Dpf(16, group="uint")._g == fss_amd.U32
    assert fss_amd.Dpf(33, group="uint")._g == fss_amd.U64


@pytest.fixture
def dpf():
    return fss_amd.Dpf(16, "bytes")


def _s0s():
    return torch.zeros((2, 4), dtype=torch.int32)


def test_gen_validation(dpf):
    beta = torch.zeros(4, dtype=torch.int32)
    with pytest.raises(TypeError, match=r"s0s must be a \(2, 4\) int32 tensor"):
        dpf.gen(torch.zeros((2, 3), dtype=torch.int32), 1, beta)
    with pytest.raises(TypeError, match=r"s0s must be a \(2, 4\) int32 tensor"):
        dpf.gen(torch.zeros((2, 4), dtype=torch.int64), 1, beta)
    with pytest.raises(ValueError, match=r"alpha must be in \[0, 2\^16\)"):
        dpf.gen(_s0s(), 1 << 16, beta)
    with pytest.raises(ValueError, match="alpha must be in"):
        dpf.gen(_s0s(), -1, beta)
    with pytest.raises(TypeError, match="alpha must be an integer"):
        dpf.gen(_s0s(), 1.5, beta)
    with pytest.raises(TypeError, match="alpha must be an integer"):
        dpf.gen(_s0s(), True, beta)
    with pytest.raises(TypeError, match=r"beta must be a \(4,\) int32 tensor"):
        dpf.gen(_s0s(), 1, torch.zeros(3, dtype=torch.int32))


def test_eval_validation(dpf):
    s0 = torch.zeros(4, dtype=torch.int32)
    cws = torch.zeros((17, 8), dtype=torch.int32)
    with pytest.raises(ValueError, match="party must be 0 or 1"):
        dpf.eval(2, s0, cws, 1)
    with pytest.raises(TypeError, match=r"s0 must be a \(4,\) int32 tensor"):
        dpf.eval(0, torch.zeros(5, dtype=torch.int32), cws, 1)
    with pytest.raises(TypeError, match=r"cws must be a \(17, 8\) int32 tensor"):
        dpf.eval(0, s0, torch.zeros((16, 8), dtype=torch.int32), 1)
    with pytest.raises(ValueError, match=r"x must be in \[0, 2\^16\)"):
        dpf.eval(0, s0, cws, 1 << 16)
    with pytest.raises(TypeError, match="x must be an integer"):
        dpf.eval(0, s0, cws, "7")
    with pytest.raises(ValueError, match="party must be 0 or 1"):
        dpf.eval_all(-1, s0, cws)


def test_half_tree_cws_has_in_bits_slots():
    h = fss_amd.HalfTreeDpf(8, "uint")
    assert h.slots == 8 and fss_amd.Dpf(8).slots == 9
    with pytest.raises(TypeError, match=r"cws must be a \(8, 8\) int32 tensor"):
        h.eval(0, torch.zeros(4, dtype=torch.int32), torch.zeros((9, 8), dtype=torch.int32),
               torch.zeros(4, dtype=torch.int32), 1)


# ---------------------------------------------------------------- the reference's own behaviour, replayed
# tests/golden/validation_contract.json was produced by tools/gen_validation_fixtures.py from the reference's
# fss_crypto/_validate.py and _int.py themselves: per case the exception type and exact message the reference raised,
# or the value it returned.  fss_amd must reproduce every entry.
def _contract():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "validation_contract.json")
    return json.load(open(path))["cases"]


def _decode(a):
    if isinstance(a, dict) and "__tensor__" in a:
        d = a["__tensor__"]
        return torch.zeros(tuple(d["shape"]), dtype=getattr(torch, d["dtype"]), device=d["device"])
    if isinstance(a, dict) and "__int__" in a:
        return int(a["__int__"])
    if isinstance(a, dict) and "__float__" in a:
        return float(a["__float__"])
    return a


@pytest.mark.parametrize("case", _contract(), ids=[f"{i}-{c['fn']}" for i, c in enumerate(_contract())])
def test_contract_matches_reference(case):
    from fss_amd import _int, _validate
    fn = getattr(_int if case["fn"] == "split_uint128" else _validate, case["fn"])
    args, kwargs, want = [_decode(a) for a in case["args"]], case["kwargs"], case["expect"]
    if "raises" in want:
        with pytest.raises(Exception) as ei:
            fn(*args, **kwargs)
        assert type(ei.value).__name__ == want["raises"], (type(ei.value).__name__, str(ei.value))
        assert str(ei.value) == want["message"]
    else:
        got = fn(*args, **kwargs)
        if isinstance(got, tuple):
            got = [v for v in got]
            assert got == [_decode(v) for v in want["returns"]]
        else:
            assert got == want["returns"]


def test_contract_covers_every_validator():
    from fss_amd import _validate
    fns = {c["fn"] for c in _contract()}
    public = {n for n in dir(_validate) if n.startswith("validate_")}
    assert public <= fns, public - fns
    assert "split_uint128" in fns


def test_reference_module_layout():
    """The reference's module names resolve (its own tests import `fss_crypto._validate`, test/test_validation.py:3-17; callers
    may import `fss_crypto.dpf.Dpf` / `fss_crypto.dcf.Dcf`, fss_crypto/__init__.py:3-4) and are this package's objects."""
    import importlib
    import fss_amd._validate as ours
    ref = importlib.import_module("fss_crypto._validate")
    for name in ("validate_alpha", "validate_beta", "validate_cpu_only", "validate_cws", "validate_device_match", "validate_domain_value",
                 "validate_group", "validate_in_bits", "validate_party", "validate_pred", "validate_prg", "validate_s0", "validate_s0s"):
        assert getattr(ref, name) is getattr(ours, name)
    assert importlib.import_module("fss_crypto._int").split_uint128((1 << 64) + 5) == (5, 1)
    import fss_crypto
    assert importlib.import_module("fss_crypto.dpf").Dpf is fss_crypto.Dpf
    assert importlib.import_module("fss_crypto.dcf").Dcf is fss_crypto.Dcf
