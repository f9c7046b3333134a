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
