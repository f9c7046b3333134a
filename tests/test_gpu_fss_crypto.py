"""The reference's Python integration tests restated against the `fss_crypto` drop-in
(test/test_dpf_integration.py:21-65, test/test_dcf_integration.py:21-74): constructor defaults, tensor shapes,
dtypes and devices, the out-of-domain error, CPU and GPU tensors, plus reconstruction of the shares (the
reference's Python tests stop at shapes because its ChaCha nonce is random per process)."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture
def s0s():
    return torch.randint(-2**31, 2**31, (2, 4), dtype=torch.int32)


@pytest.fixture
def beta():
    return torch.tensor([0, 0, 0, 604], dtype=torch.int32)


@pytest.mark.parametrize("scheme", ["dpf", "dcf"])
@pytest.mark.parametrize("prg", ["chacha", "aes128_mmo"])
def test_reference_binding_surface(scheme, prg, s0s, beta):
    import fss_crypto
    f = (fss_crypto.Dpf(in_bits=16, group="bytes", prg=prg) if scheme == "dpf"
         else fss_crypto.Dcf(in_bits=16, group="bytes", prg=prg, pred="lt"))
    assert (fss_crypto.Dpf(in_bits=16).prg, fss_crypto.Dcf(in_bits=16).prg, fss_crypto.Dcf(in_bits=16).pred) == \
        ("chacha", "chacha", "lt")
    cws = f.gen(s0s, alpha=107, beta=beta)
    assert cws.shape == (17, 8) and cws.dtype == torch.int32 and cws.device.type == "cpu"
    out = f.eval(party=0, s0=s0s[0], cws=cws, x=50)
    assert out.shape == (4,) and out.dtype == torch.int32 and out.device.type == "cpu"
    with pytest.raises(ValueError, match="x must be"):
        f.eval(party=0, s0=s0s[0], cws=cws, x=2**16)
    out = f.eval(party=0, s0=s0s[0].to("cuda"), cws=cws.to("cuda"), x=50)
    assert out.shape == (4,) and out.dtype == torch.int32 and out.device.type == "cuda"
    ys = [f.eval_all(party=b, s0=s0s[b], cws=cws) for b in (0, 1)]
    assert ys[0].shape == (2**16, 4) and ys[0].dtype == torch.int32 and ys[0].device.type == "cpu"
    # shares reconstruct (bytes group: XOR); beta is clamped to an even last word
    want = torch.zeros((2**16, 4), dtype=torch.int32)
    clamped = beta.clone()
    clamped[3] &= ~1
    if scheme == "dpf":
        want[107] = clamped
    else:
        want[:107] = clamped
    assert torch.equal(ys[0] ^ ys[1], want)
    for x in (0, 106, 107, 108, 65535):
        assert torch.equal(f.eval(0, s0s[0], cws, x) ^ f.eval(1, s0s[1], cws, x), want[x])
