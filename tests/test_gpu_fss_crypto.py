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


# The same checks under the reference's own class and test names (test/test_dpf_integration.py:21-65,
# test/test_dcf_integration.py:21-74), so that the two lists can be read side by side; written here, fixtures as there
# (in_bits = 16, group "bytes", the default PRG "chacha", alpha = 107).
@pytest.fixture
def dpf():
    import fss_crypto
    return fss_crypto.Dpf(in_bits=16, group="bytes", prg="chacha")


@pytest.fixture
def dcf():
    import fss_crypto
    return fss_crypto.Dcf(in_bits=16, group="bytes", prg="chacha", pred="lt")


def _is_int32(t, shape, device):
    return tuple(t.shape) == shape and t.dtype == torch.int32 and t.device.type == device


class _GenShape:
    def test_cws_shape(self, f, s0s, beta):
        assert _is_int32(f.gen(s0s, alpha=107, beta=beta), (17, 8), "cpu")

    def test_cws_device_is_cpu(self, f, s0s, beta):
        assert f.gen(s0s, alpha=107, beta=beta).device.type == "cpu"


class _EvalShape:
    def test_output_shape_cpu(self, f, s0s, beta):
        cws = f.gen(s0s, alpha=107, beta=beta)
        assert _is_int32(f.eval(party=0, s0=s0s[0], cws=cws, x=50), (4,), "cpu")

    def test_rejects_out_of_domain_x(self, f, s0s, beta):
        cws = f.gen(s0s, alpha=107, beta=beta)
        with pytest.raises(ValueError, match="x must be"):
            f.eval(party=0, s0=s0s[0], cws=cws, x=2**16)

    def test_output_shape_cuda(self, f, s0s, beta):
        cws = f.gen(s0s, alpha=107, beta=beta)
        assert _is_int32(f.eval(party=0, s0=s0s[0].to("cuda"), cws=cws.to("cuda"), x=50), (4,), "cuda")


class _EvalAllShape:
    def test_output_shape(self, f, s0s, beta):
        cws = f.gen(s0s, alpha=107, beta=beta)
        assert _is_int32(f.eval_all(party=0, s0=s0s[0], cws=cws), (2**16, 4), "cpu")


class _OnDpf:
    @pytest.fixture
    def f(self, dpf):
        return dpf


class _OnDcf:
    @pytest.fixture
    def f(self, dcf):
        return dcf


class TestDpfGenShape(_OnDpf, _GenShape):
    pass


class TestDpfEvalShape(_OnDpf, _EvalShape):
    pass


class TestDpfEvalAllShape(_OnDpf, _EvalAllShape):
    pass


class TestDcfGenShape(_OnDcf, _GenShape):
    pass


class TestDcfEvalShape(_OnDcf, _EvalShape):
    pass


class TestDcfEvalAllShape(_OnDcf, _EvalAllShape):
    pass


class TestDcfPredVariant:
    def test_gt_variant(self, s0s, beta):
        import fss_crypto
        dcf_gt = fss_crypto.Dcf(in_bits=16, group="bytes", prg="chacha", pred="gt")
        cws = dcf_gt.gen(s0s, alpha=107, beta=beta)
        assert _is_int32(cws, (17, 8), "cpu")
        # ... and it is the other predicate: beta above alpha, nothing at or below it
        want = beta.clone()
        want[3] &= ~1
        for x, hit in ((50, False), (107, False), (108, True), (65535, True)):
            got = dcf_gt.eval(0, s0s[0], cws, x) ^ dcf_gt.eval(1, s0s[1], cws, x)
            assert torch.equal(got, want if hit else torch.zeros(4, dtype=torch.int32))
