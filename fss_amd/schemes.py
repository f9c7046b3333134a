"""Host-side mirror of the reference's scheme interfaces for the evaluation path.

`Dpf` and `Dcf` keep the constructor / method surface and the error behaviour of
the reference's Python binding (fss_crypto/dpf.py:23-109, fss_crypto/dcf.py:24-113):
gen / eval / eval_all with the same tensor shapes.  Differences, all deliberate:
  * prg="aes128_mmo" (default here; the headline path) runs on the GPU, where the reference rejects it
    (fss_crypto/dpf.py:82-87); prg="chacha" uses a fixed engine nonce instead of the reference
    binding's per-process random one;
  * CPU tensors are accepted everywhere and results come back on the input's
    device, but the arithmetic always runs in the HIP kernels -- no CPU path;
  * batched methods are added: eval_points (one key, many x), relayout +
    eval_keys (independent keys, as fss::gpu::*EvalPointGpu), gen_batch, and
    eval_all on a sub-block [first_leaf, first_leaf + nleaves) for domain sharding.
`HalfTreeDpf` and `GrottoDcf` mirror the C++ classes (include/fss/half_tree_dpf.cuh,
include/fss/grotto_dcf.cuh), which the reference does not bind to Python.
"""
import torch

from . import engine as E
from ._int import split_uint128
from ._validate import (validate_alpha, validate_beta, validate_cpu_only, validate_cws, validate_device_match,
                        validate_domain_value, validate_group, validate_in_bits, validate_party, validate_pred,
                        validate_prg, validate_s0, validate_s0s)

_MAX_IN_BITS = 128  # kernels take x as uint32 / uint64 / two uint64 (__uint128_t)


def _xs_tensor(values, in_bits, device):
    """Python ints / tensor -> int32 (in_bits <= 32) or int64 bit patterns."""
    if isinstance(values, torch.Tensor):
        return values
    def s64(v):
        return v - (1 << 64) if v >= (1 << 63) else v
    if in_bits <= 32:
        return torch.tensor([v - (1 << 32) if v >= (1 << 31) else v for v in values], dtype=torch.int32, device=device)
    if in_bits <= 64:
        return torch.tensor([s64(v) for v in values], dtype=torch.int64, device=device)
    return torch.tensor([[s64(h) for h in split_uint128(v)] for v in values], dtype=torch.int64, device=device)


def _x_tensor(x, in_bits, device):
    """One evaluation point for the per-point `eval` methods, created on the engine's device by a fill (no
    host-to-device copy, which would synchronise) for in_bits <= 64."""
    if in_bits > 64 or torch.device(device).type != "cuda":
        return _xs_tensor([x], in_bits, "cpu")
    if in_bits <= 32:
        return torch.full((1,), x - (1 << 32) if x >= (1 << 31) else x, dtype=torch.int32, device=device)
    return torch.full((1,), x - (1 << 64) if x >= (1 << 63) else x, dtype=torch.int64, device=device)


def _check_group(group):
    """'bytes' / 'uint' as the reference; 'u8' .. 'u128' name one Uint<T> explicitly."""
    if group not in E.GROUP_IDS:
        validate_group(group)


class _Base:
    scheme = None
    _slots_extra = 1  # cws has in_bits + 1 slots

    def __init__(self, in_bits, group, prg, engine=None):
        validate_in_bits(in_bits)
        _check_group(group)
        if prg not in ("chacha", "aes128_mmo"):
            raise ValueError(f"prg must be one of ('chacha', 'aes128_mmo'), got {prg!r}")
        if in_bits > _MAX_IN_BITS:
            raise NotImplementedError(f"fss_amd kernels cover in_bits <= {_MAX_IN_BITS}, got {in_bits}")
        self.in_bits, self.group, self.prg = in_bits, group, prg
        self._g = E.resolve_group(group, in_bits)
        self._engine = engine

    @property
    def engine(self):
        if self._engine is None:
            self._engine = E.default_engine(prg=self.prg)
        return self._engine

    @property
    def slots(self):
        return self.in_bits + self._slots_extra

    def _back(self, t, like):
        return t if like.device.type == "cuda" else t.cpu()

    # -- batched, all schemes --------------------------------------------------
    def relayout(self, cws):
        """int32[nkeys, slots, 8] key-major -> level-major device arrays."""
        return self.engine.relayout(self.scheme, self.in_bits, cws)


class Dpf(_Base):
    """2-party distributed point function (fss::Dpf, include/fss/dpf.cuh:61-304)."""
    scheme = E.DPF

    def __init__(self, in_bits, group="bytes", prg="aes128_mmo", engine=None):
        validate_in_bits(in_bits)
        _check_group(group)
        validate_prg(prg, "dpf")
        super().__init__(in_bits, group, prg, engine)

    def gen(self, s0s, alpha, beta):
        validate_s0s(s0s)
        validate_alpha(alpha, self.in_bits)
        validate_beta(beta)
        validate_cpu_only(s0s, beta, fn_name="gen")
        cws, _ = self.engine.gen(self.scheme, self.in_bits, self._g, s0s.view(1, 2, 4),
                                 _xs_tensor([alpha], self.in_bits, "cpu"), beta.view(1, 4))
        return cws[0].cpu()

    def gen_batch(self, s0s, alphas, betas):
        """s0s int32[n,2,4], alphas int32/int64[n], betas int32[n,4] -> device int32[n, in_bits+1, 8]."""
        return self.engine.gen(self.scheme, self.in_bits, self._g, s0s, alphas, betas)[0]

    def eval(self, party, s0, cws, x):
        validate_party(party)
        validate_s0(s0)
        validate_cws(cws, self.in_bits)
        validate_device_match(s0, cws)
        validate_domain_value("x", x, self.in_bits)
        ys = self.engine.eval_points(self.scheme, self.in_bits, self._g, party, s0, cws,
                                     _x_tensor(x, self.in_bits, self.engine.device))
        return self._back(ys[0], s0)

    def eval_points(self, party, s0, cws, xs):
        """One key, many points: xs int32/int64 bit patterns [n] -> int32[n,4] (device of xs).  From 2^18 points on
        the engine expands the key's top levels once per call (the shared-prefix plan, FSS_AMD_POINTS_AUTO: identical
        outputs, 2.9 x at 2^24 points); `engine.set_points_plan("plain")` walks every point from the root."""
        validate_party(party)
        validate_s0(s0)
        validate_cws(cws, self.in_bits)
        return self._back(self.engine.eval_points(self.scheme, self.in_bits, self._g, party, s0, cws, xs), xs)

    def eval_keys(self, party, seeds, keys, xs):
        validate_party(party)
        return self.engine.eval_keys(keys, self._g, party, seeds, xs)

    def eval_all(self, party, s0, cws, first_leaf=0, nleaves=None):
        validate_party(party)
        validate_s0(s0)
        validate_cws(cws, self.in_bits)
        validate_device_match(s0, cws)
        ys = self.engine.eval_all(self.scheme, self.in_bits, self._g, party, s0, cws, first_leaf=first_leaf,
                                  nleaves=nleaves)
        return self._back(ys, s0)


class Dcf(_Base):
    """2-party distributed comparison function (fss::Dcf, include/fss/dcf.cuh:74-388)."""
    scheme = E.DCF

    def __init__(self, in_bits, group="bytes", prg="aes128_mmo", pred="lt", engine=None):
        validate_in_bits(in_bits)
        _check_group(group)
        validate_prg(prg, "dcf")
        validate_pred(pred)
        super().__init__(in_bits, group, prg, engine)
        self.pred = pred
        self._pred = 0 if pred == "lt" else 1

    def gen(self, s0s, alpha, beta):
        validate_s0s(s0s)
        validate_alpha(alpha, self.in_bits)
        validate_beta(beta)
        validate_cpu_only(s0s, beta, fn_name="gen")
        cws, _ = self.engine.gen(self.scheme, self.in_bits, self._g, s0s.view(1, 2, 4),
                                 _xs_tensor([alpha], self.in_bits, "cpu"), beta.view(1, 4), pred=self._pred)
        return cws[0].cpu()

    def gen_batch(self, s0s, alphas, betas):
        return self.engine.gen(self.scheme, self.in_bits, self._g, s0s, alphas, betas, pred=self._pred)[0]

    eval = Dpf.eval
    eval_points = Dpf.eval_points
    eval_keys = Dpf.eval_keys
    eval_all = Dpf.eval_all


class HalfTreeDpf(_Base):
    """Half-Tree DPF (fss::HalfTreeDpf, include/fss/half_tree_dpf.cuh:39-357); cws has in_bits slots."""
    scheme = E.HT
    _slots_extra = 0

    def __init__(self, in_bits, group="bytes", prg="aes128_mmo", engine=None):
        validate_in_bits(in_bits)
        _check_group(group)
        super().__init__(in_bits, group, prg, engine)

    def gen(self, s0s, alpha, beta):
        validate_s0s(s0s)
        validate_alpha(alpha, self.in_bits)
        validate_beta(beta)
        cws, ocws = self.engine.gen(self.scheme, self.in_bits, self._g, s0s.view(1, 2, 4),
                                    _xs_tensor([alpha], self.in_bits, "cpu"), beta.view(1, 4))
        return cws[0].cpu(), ocws[0].cpu()

    def gen_batch(self, s0s, alphas, betas):
        return self.engine.gen(self.scheme, self.in_bits, self._g, s0s, alphas, betas)

    def eval(self, party, s0, cws, ocw, x):
        validate_party(party)
        validate_s0(s0)
        validate_cws(cws, self.in_bits, slots=self.in_bits)
        validate_domain_value("x", x, self.in_bits)
        ys = self.engine.eval_points(self.scheme, self.in_bits, self._g, party, s0, cws,
                                     _x_tensor(x, self.in_bits, self.engine.device), ocw=ocw)
        return self._back(ys[0], s0)

    def eval_points(self, party, s0, cws, ocw, xs):
        validate_party(party)
        return self._back(self.engine.eval_points(self.scheme, self.in_bits, self._g, party, s0, cws, xs, ocw=ocw), xs)

    def eval_keys(self, party, seeds, keys, ocws, xs):
        validate_party(party)
        return self.engine.eval_keys(keys, self._g, party, seeds, xs, ocws=ocws)

    def eval_all(self, party, s0, cws, ocw, first_leaf=0, nleaves=None):
        validate_party(party)
        ys = self.engine.eval_all(self.scheme, self.in_bits, self._g, party, s0, cws, ocw=ocw, first_leaf=first_leaf,
                                  nleaves=nleaves)
        return self._back(ys, s0)


class GrottoDcf(_Base):
    """Grotto DCF (fss::GrottoDcf, include/fss/grotto_dcf.cuh:45-241): DPF keys over Bytes with beta = 0;
    outputs are XOR shares (one byte, 0/1) of 1[alpha <= x]."""
    scheme = E.GROTTO

    def __init__(self, in_bits, prg="aes128_mmo", engine=None):
        validate_in_bits(in_bits)
        super().__init__(in_bits, "bytes", prg, engine)

    def gen(self, s0s, alpha):
        validate_s0s(s0s)
        validate_alpha(alpha, self.in_bits)
        beta = torch.zeros((1, 4), dtype=torch.int32)
        cws, _ = self.engine.gen(E.DPF, self.in_bits, E.BYTES, s0s.view(1, 2, 4),
                                 _xs_tensor([alpha], self.in_bits, "cpu"), beta)
        return cws[0].cpu()

    def gen_batch(self, s0s, alphas):
        betas = torch.zeros((alphas.numel(), 4), dtype=torch.int32)
        return self.engine.gen(E.DPF, self.in_bits, E.BYTES, s0s, alphas, betas)[0]

    def eval_all(self, party, s0, cws, first_leaf=0, nleaves=None):
        """= GrottoDcf::EvalAll (grotto_dcf.cuh:151-163); uint8[2^in_bits]."""
        validate_party(party)
        ys = self.engine.eval_all(self.scheme, self.in_bits, E.BYTES, party, s0, cws, first_leaf=first_leaf,
                                  nleaves=nleaves)
        return self._back(ys, s0)

    def preprocess(self, party, s0, cws):
        """= GrottoDcf::Preprocess (grotto_dcf.cuh:94-104): the ParityTree table, uint8[2^(in_bits+1) - 1]."""
        validate_party(party)
        validate_s0(s0)
        validate_cws(cws, self.in_bits)
        return self.engine.grotto_preprocess(self.in_bits, party, s0, cws)

    def eval(self, pt, xs):
        """= GrottoDcf::Eval (grotto_dcf.cuh:116-135) for a batch of x: uint8[n] shares of 1[alpha <= x]."""
        return self.engine.grotto_eval_table(self.in_bits, pt, xs)

    def eval_keys_pathwalk(self, party, seeds, keys, xs):
        """Path-walk shares for independent keys, one point each (any in_bits <= 64; the reference has no evaluator
        for domains too large for a table, grotto_dcf.cuh:117).  The two parties' shares XOR to 1[alpha <= x], but a
        party's share is NOT the one `eval` / `eval_all` (= the reference's GrottoDcf::Eval / EvalAll) give for the
        same key and x: they differ by a party-independent mask.  BOTH parties must use this method for a given
        (key, x); mixing it with `eval` or `eval_all` on the other side reconstructs a wrong bit half of the time."""
        validate_party(party)
        return self.engine.eval_keys(keys, E.BYTES, party, seeds, xs)

    eval_keys = eval_keys_pathwalk  # older name; same incompatibility caveat


class Vdpf(_Base):
    """2-party verifiable DPF (fss::Vdpf, include/fss/vdpf.cuh:57-404) with the BLAKE3 hashes
    (fss::hash::Blake3 as both XorHash and Hash; key = the engine's hash IV, Engine.set_hash_iv).
    A key is (cws int32[in_bits, 8], cs int32[4, 4], ocw int32[4]) plus the party's seed."""
    scheme = E.VDPF
    _slots_extra = 0

    def __init__(self, in_bits, group="bytes", prg="aes128_mmo", engine=None):
        super().__init__(in_bits, group, prg, engine)

    def gen(self, s0s, alpha, beta):
        """-> (cws, cs, ocw) on the CPU; raises if the reference's Gen would ask for fresh seeds (vdpf.cuh:177)."""
        validate_s0s(s0s)
        validate_alpha(alpha, self.in_bits)
        validate_beta(beta)
        cws, cs, ocws, retry = self.engine.vdpf_gen(self.in_bits, self._g, s0s.view(1, 2, 4),
                                                    _xs_tensor([alpha], self.in_bits, "cpu"), beta.view(1, 4))
        if int(retry[0]) != 0:
            raise RuntimeError("Vdpf.gen: t0 == t1 at the last level, resample s0s and retry")
        return cws[0].cpu(), cs[0].cpu(), ocws[0].cpu()

    def gen_batch(self, s0s, alphas, betas, level_major=False):
        """-> (cws | LevelMajorKeys, cs int32[n,4,4], ocws int32[n,4], retry int32[n]) on the device."""
        return self.engine.vdpf_gen(self.in_bits, self._g, s0s, alphas, betas, level_major=level_major)

    def relayout(self, cws):
        return self.engine.vdpf_relayout(self.in_bits, cws)

    def eval(self, party, s0, cws, cs, ocw, x):
        """-> (y int32[4], pi_tilde int32[4,4])   (Vdpf::Eval, vdpf.cuh:204-257)."""
        validate_party(party)
        validate_s0(s0)
        validate_domain_value("x", x, self.in_bits)
        ys, pis = self.engine.vdpf_eval_points(self.in_bits, self._g, party, s0, cws, cs, ocw,
                                               _x_tensor(x, self.in_bits, self.engine.device))
        return self._back(ys[0], s0), self._back(pis[0], s0)

    def eval_points(self, party, s0, cws, cs, ocw, xs):
        validate_party(party)
        validate_s0(s0)
        ys, pis = self.engine.vdpf_eval_points(self.in_bits, self._g, party, s0, cws, cs, ocw, xs)
        return self._back(ys, xs), self._back(pis, xs)

    def eval_keys(self, party, seeds, keys, cs, ocws, xs):
        validate_party(party)
        return self.engine.vdpf_eval_keys(keys, self._g, party, seeds, cs, ocws, xs)

    def prove(self, pi_tildes, cs):
        """Vdpf::Prove (vdpf.cuh:268-278) over all hashes of one key."""
        return self._back(self.engine.vdpf_prove(pi_tildes, cs), pi_tildes)

    def prove_keys(self, pi_tildes, cs):
        return self.engine.vdpf_prove_keys(pi_tildes, cs)

    @staticmethod
    def verify(pi0, pi1):
        """Vdpf::Verify (vdpf.cuh:287-292)."""
        return bool(torch.equal(pi0.cpu(), pi1.cpu()))
