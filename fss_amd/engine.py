"""Engine: one libfss_amd context (AES keys + Half-Tree hash key) on one GPU, with
torch tensors as device memory.  Everything here is plumbing around the C ABI
(include/fss_amd.h); the arithmetic runs in the HIP kernels only."""
import ctypes as C

import numpy as np
import torch

from . import _lib

# Output groups (enum fss_amd_group).
BYTES, U8, U16, U32, U64, U128 = 0, 1, 2, 3, 4, 5
GROUP_IDS = {"bytes": BYTES, "u8": U8, "u16": U16, "u32": U32, "u64": U64, "u128": U128}
DPF, DCF, HT, GROTTO = 0, 1, 2, 3
VDPF = 4  # host-side tag only (LevelMajorKeys of a verifiable DPF); the C ABI has its own fss_amd_vdpf_* calls

# AES keys / hash key of the reference's benches and samples (src/bench_cpu.cu:37-40, :509).
DEFAULT_AES_KEYS = (
    bytes(range(1, 17)),
    bytes(range(16, 0, -1)),
    bytes([1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8]),
    bytes([8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1]),
)
DEFAULT_HASH_KEY = (0x12345678, 0x9ABCDEF0, 0x13572468, 0x2468ACE0)
# Nonce of the reference's C++ tests (src/dcf_test.cu:35).  The reference's Python binding draws a random
# process-global nonce instead (fss_crypto/_csrc/binding_common.cuh:13-37), so its ChaCha outputs are not
# reproducible across processes; here the nonce is an explicit, fixed engine parameter.
DEFAULT_CHACHA_NONCE = (0x12345678, 0x9ABCDEF0)
PRG_AES128_MMO, PRG_CHACHA = 0, 1
PRG_IDS = {"aes128_mmo": PRG_AES128_MMO, "chacha": PRG_CHACHA}


def resolve_group(group, in_bits):
    """'bytes' | 'uint' (sized by in_bits as fss_crypto/_jit.py:76-87) | 'u8'..'u128' | int id."""
    if isinstance(group, int):
        return group
    if group == "uint":
        return U32 if in_bits <= 32 else (U64 if in_bits <= 64 else U128)
    return GROUP_IDS[group]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class LevelMajorKeys:
    """Level-major device layout of a batch of keys (include/fss_amd.h, relayout)."""

    def __init__(self, scheme, in_bits, nkeys, cw_s, cw_v=None, tr_bits=None, out_cw=None, extra=None):
        self.scheme, self.in_bits, self.nkeys = scheme, in_bits, nkeys
        self.cw_s, self.cw_v, self.tr_bits, self.out_cw, self.extra = cw_s, cw_v, tr_bits, out_cw, extra


class Engine:
    def __init__(self, device=None, aes_keys=DEFAULT_AES_KEYS, hash_key=DEFAULT_HASH_KEY, prg="aes128_mmo",
                 chacha_nonce=DEFAULT_CHACHA_NONCE, chacha_rounds=20, lib=None):
        L = lib if lib is not None else _lib.lib()  # lib: another build of the library (bench.py: _lib.diag_lib())
        if not torch.cuda.is_available():
            raise RuntimeError("fss_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        keys = b"".join(bytes(k) for k in aes_keys)
        if len(keys) != 64:
            raise ValueError("aes_keys must be four 16-byte keys")
        hk = (C.c_int32 * 4)(*[int(np.uint32(v).astype(np.int32)) for v in hash_key])
        h = C.c_void_p()
        _lib.check(L.fss_amd_ctx_create(C.byref(h), self.device.index, C.c_char_p(keys), hk), "fss_amd_ctx_create")
        self._h, self._L = h, L
        self.prg = prg
        # The C ABI's contexts start FSS_AMD_POINTS_PLAIN (nothing allocated inside an eval call, as the reference's
        # launchers); this layer allocates its outputs per call anyway and opts in to the shared-prefix plan.
        _lib.check(L.fss_amd_ctx_set_points_plan(h, 0), "fss_amd_ctx_set_points_plan")
        self.points_plan = "auto"
        if prg not in PRG_IDS:
            raise ValueError(f"prg must be one of {tuple(PRG_IDS)}, got {prg!r}")
        if prg == "chacha":
            nn = (C.c_int32 * 2)(*[int(np.uint32(v).astype(np.int32)) for v in chacha_nonce])
            _lib.check(L.fss_amd_ctx_set_prg(h, PRG_CHACHA, nn, chacha_rounds), "fss_amd_ctx_set_prg")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.fss_amd_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, dtype=None):
        t = t if dtype is None or t.dtype == dtype else t.to(dtype)
        return t.to(self.device).contiguous()

    @staticmethod
    def _need(t, numel, what):
        """The C ABI takes raw pointers: refuse tensors that are too small before a kernel reads past them."""
        if t is None or t.numel() < numel:
            have = "None" if t is None else f"{tuple(t.shape)}"
            raise ValueError(f"{what}: expected at least {numel} elements, got {have}")
        return t

    def _out(self, out, nbytes, what, dtypes=(torch.int32,)):
        """A caller-supplied output tensor is handed to the kernels as a raw pointer: it must live on this engine's
        device, be contiguous, have the element type the kernel writes and hold at least `nbytes` bytes."""
        if not isinstance(out, torch.Tensor):
            raise TypeError(f"{what}: expected a torch.Tensor, got {type(out).__name__}")
        if out.device != self.device:
            raise ValueError(f"{what}: must be on {self.device}, got {out.device}")
        if not out.is_contiguous():
            raise ValueError(f"{what}: must be contiguous")
        if out.dtype not in dtypes:
            raise ValueError(f"{what}: dtype must be {' or '.join(str(d) for d in dtypes)}, got {out.dtype}")
        have = out.element_size() * out.numel()
        if have < nbytes:
            raise ValueError(f"{what}: expected at least {nbytes} bytes, got {have} ({tuple(out.shape)} {out.dtype})")
        return out

    @staticmethod
    def _slots(scheme, in_bits):
        return in_bits if scheme in (HT, VDPF) else in_bits + 1

    @staticmethod
    def _xbytes(xs, in_bits):
        if xs.dtype in (torch.uint8, torch.int16):  # In = uint8_t / uint16_t (bit patterns)
            w = 1 if xs.dtype == torch.uint8 else 2
            if in_bits > 8 * w:
                raise ValueError(f"{xs.dtype} xs only cover in_bits <= {8 * w}")
            return w
        if xs.dtype == torch.int32:
            if in_bits > 32:
                raise ValueError("int32 xs only cover in_bits <= 32; pass int64")
            return 4
        if xs.dtype == torch.int64 and xs.dim() == 2 and xs.shape[-1] == 2:
            return 16  # (lo, hi) pairs = __uint128_t
        if xs.dtype == torch.int64:
            if in_bits > 64:
                raise ValueError("in_bits > 64 needs xs of shape (n, 2): (lo, hi) int64 bit patterns")
            return 8
        raise TypeError(f"xs must be uint8, int16, int32 or int64 bit patterns (or (n, 2) int64 pairs), got {xs.dtype}")

    def set_points_plan(self, plan):
        """'auto' (this class's default; the C ABI's own default is 'plain'): eval_points takes the shared-prefix plan for
        batches of >= 2^18 points of one key (same outputs, about a third of the AES work at 2^24 points; the node table
        lives in a per-stream buffer of the context, allocated stream-ordered; release_stream() before dropping a side
        stream).  'plain': every point walks from the root, nothing is allocated -- bench.py's own engine is set to it."""
        if plan not in ("auto", "plain"):
            raise ValueError(f"plan must be 'auto' or 'plain', got {plan!r}")
        _lib.check(self._L.fss_amd_ctx_set_points_plan(self._h, 0 if plan == "auto" else 1), "fss_amd_ctx_set_points_plan")
        self.points_plan = plan
        return self

    def release_stream(self, stream=None):
        """Free what the context keeps for `stream` (default: the current stream): the node table of the shared-prefix
        plan (up to 2^26 x 16 / 32 B, grown by eval_points under the 'auto' plan), its scratch buffer and its
        work-distribution counters.  Waits for the device.  Call before dropping a side stream that evaluated large
        batches; the context's destruction frees everything anyway."""
        st = C.c_void_p((stream or torch.cuda.current_stream(self.device)).cuda_stream)
        _lib.check(self._L.fss_amd_ctx_scratch_release(self._h, st), "fss_amd_ctx_scratch_release")

    def last_kernel(self):
        g, b = C.c_int(0), C.c_int(0)
        name = self._L.fss_amd_last_kernel(self._h, C.byref(g), C.byref(b)).decode()
        return name, g.value, b.value

    # ------------------------------------------------------------ PRG
    def prg_gen(self, mul, seeds):
        seeds = self._dev(seeds, torch.int32).view(-1, 4)
        out = torch.empty((seeds.shape[0], mul, 4), dtype=torch.int32, device=self.device)
        _lib.check(self._L.fss_amd_prg_gen(self._h, mul, _ptr(seeds), seeds.shape[0], _ptr(out), self._stream()),
                   "fss_amd_prg_gen")
        return out

    # ------------------------------------------------------------ one key, many points
    def eval_points(self, scheme, in_bits, group, party, seed, cws, xs, ocw=None, out=None):
        seed, cws, xs = self._dev(seed, torch.int32), self._dev(cws, torch.int32), self._dev(xs)
        xb = self._xbytes(xs, in_bits)
        n = xs.shape[0] if xb == 16 else xs.numel()
        self._need(seed, 4, "seed"); self._need(cws, 8 * self._slots(scheme, in_bits), "cws")
        ys = self._out(out, 16 * n, "out") if out is not None else torch.empty((n, 4), dtype=torch.int32, device=self.device)
        L, st = self._L, self._stream()
        if scheme == DPF:
            rc = L.fss_amd_dpf_eval_points(self._h, in_bits, group, party, _ptr(seed), _ptr(cws), _ptr(xs), xb, n,
                                           _ptr(ys), st)
        elif scheme == DCF:
            rc = L.fss_amd_dcf_eval_points(self._h, in_bits, group, party, _ptr(seed), _ptr(cws), _ptr(xs), xb, n,
                                           _ptr(ys), st)
        elif scheme == HT:
            ocw = self._need(self._dev(self._need(ocw, 4, "ocw"), torch.int32), 4, "ocw")
            rc = L.fss_amd_ht_eval_points(self._h, in_bits, group, party, _ptr(seed), _ptr(cws), _ptr(ocw), _ptr(xs),
                                          xb, n, _ptr(ys), st)
        else:
            raise ValueError("scheme")
        _lib.check(rc, "eval_points")
        return ys

    def eval_points_host(self, scheme, in_bits, group, party, seed, cws, xs, ocw=None, out=None):
        """eval_points on a batch that lives in PINNED host memory (the PCIe-inclusive path,
        fss_amd_eval_points_host): the walk reads x from and writes y to the host tensors themselves, one launch, no
        staging copies.  xs: a pinned CPU tensor; out: a pinned CPU int32[n, 4] tensor (allocated pinned when None).
        The call is asynchronous on the current stream: synchronise before reading `out`."""
        seed, cws = self._dev(seed, torch.int32), self._dev(cws, torch.int32)
        if xs.device.type != "cpu" or not (xs.is_pinned() or xs.numel() == 0) or not xs.is_contiguous():
            raise ValueError("eval_points_host: xs must be a contiguous pinned CPU tensor (tensor.pin_memory())")
        xb = self._xbytes(xs, in_bits)
        n = xs.shape[0] if xb == 16 else xs.numel()
        self._need(seed, 4, "seed"); self._need(cws, 8 * self._slots(scheme, in_bits), "cws")
        if out is None:
            out = torch.empty((n, 4), dtype=torch.int32, pin_memory=True)
        if (not isinstance(out, torch.Tensor) or out.device.type != "cpu" or not (out.is_pinned() or n == 0) or not out.is_contiguous()
                or out.dtype != torch.int32 or out.numel() < 4 * n):
            raise ValueError(f"eval_points_host: out must be a contiguous pinned CPU int32 tensor of at least {4 * n} elements")
        if scheme not in (DPF, DCF, HT):
            raise ValueError("scheme")
        if scheme == HT:
            ocw = self._need(self._dev(self._need(ocw, 4, "ocw"), torch.int32), 4, "ocw")
        with torch.cuda.device(self.device):
            rc = self._L.fss_amd_eval_points_host(self._h, scheme, in_bits, group, party, _ptr(seed), _ptr(cws),
                                                  _ptr(ocw) if scheme == HT else None, C.c_void_p(xs.data_ptr()), xb, n,
                                                  C.c_void_p(out.data_ptr()), self._stream())
        _lib.check(rc, "eval_points_host")
        return out

    def eval_points_shared(self, scheme, in_bits, group, party, seed, cws, xs, ocw=None, out=None, prefix_bits=None):
        """eval_points with the shared top of the tree expanded once per launch (same outputs).
        prefix_bits=None: the library's own choice for this batch (fss_amd_points_prefix_bits: floor(log2 n) - 1, capped
        by in_bits and 26), also for batches below the size from which eval_points would pick this plan by itself."""
        seed, cws, xs = self._dev(seed, torch.int32), self._dev(cws, torch.int32), self._dev(xs)
        xb = self._xbytes(xs, in_bits)
        if xb == 16:
            raise ValueError("eval_points_shared covers in_bits <= 64: pass int32 / int64 xs, not (n, 2) pairs")
        n = xs.numel()
        cap = min(in_bits - (1 if scheme == HT else 0), 26)
        if prefix_bits is None:
            prefix_bits = self._L.fss_amd_points_prefix_bits(scheme, in_bits, n) or min(cap, max(n, 4).bit_length() - 2)
        if prefix_bits < 1 or cap < 1:  # nothing to share (e.g. a 1-level Half-Tree key): plain walk
            return self.eval_points(scheme, in_bits, group, party, seed, cws, xs, ocw=ocw, out=out)
        L, st = self._L, self._stream()
        ws_bytes = L.fss_amd_eval_points_workspace(scheme, in_bits, prefix_bits)
        if ws_bytes == 0:
            raise ValueError(f"eval_points_shared: unsupported prefix_bits={prefix_bits} for in_bits={in_bits}")
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=self.device)
        self._need(seed, 4, "seed"); self._need(cws, 8 * self._slots(scheme, in_bits), "cws")
        ys = self._out(out, 16 * n, "out") if out is not None else torch.empty((n, 4), dtype=torch.int32, device=self.device)
        if scheme == DPF:
            rc = L.fss_amd_dpf_eval_points_shared(self._h, in_bits, group, party, _ptr(seed), _ptr(cws), _ptr(xs), xb, n,
                                                  _ptr(ys), prefix_bits, _ptr(ws), ws_bytes, st)
        elif scheme == DCF:
            rc = L.fss_amd_dcf_eval_points_shared(self._h, in_bits, group, party, _ptr(seed), _ptr(cws), _ptr(xs), xb, n,
                                                  _ptr(ys), prefix_bits, _ptr(ws), ws_bytes, st)
        elif scheme == HT:
            ocw = self._dev(self._need(ocw, 4, "ocw"), torch.int32)
            rc = L.fss_amd_ht_eval_points_shared(self._h, in_bits, group, party, _ptr(seed), _ptr(cws), _ptr(ocw),
                                                 _ptr(xs), xb, n, _ptr(ys), prefix_bits, _ptr(ws), ws_bytes, st)
        else:
            raise ValueError("scheme")
        _lib.check(rc, "eval_points_shared")
        ws.record_stream(torch.cuda.current_stream(self.device))
        return ys

    # ------------------------------------------------------------ relayout + independent keys
    def relayout(self, scheme, in_bits, cws):
        """cws: int32[nkeys, slots, 8] key-major raw Cw slots -> LevelMajorKeys."""
        cws = self._dev(cws, torch.int32)
        if cws.dim() != 3 or cws.shape[1] != self._slots(scheme, in_bits) or cws.shape[2] != 8:
            raise ValueError(f"relayout: cws must be int32[nkeys, {self._slots(scheme, in_bits)}, 8], got {tuple(cws.shape)}")
        nkeys = cws.shape[0]
        dev, L, st = self.device, self._L, self._stream()
        cw_s = torch.empty((in_bits, nkeys, 4), dtype=torch.int32, device=dev)
        if scheme in (DPF, GROTTO):
            tr = torch.empty(((in_bits + 63) // 64 * nkeys,), dtype=torch.int64, device=dev)
            oc = torch.empty((nkeys, 4), dtype=torch.int32, device=dev)
            _lib.check(L.fss_amd_dpf_relayout(self._h, in_bits, _ptr(cws), nkeys, _ptr(cw_s), _ptr(tr), _ptr(oc), st),
                       "dpf_relayout")
            return LevelMajorKeys(scheme, in_bits, nkeys, cw_s, tr_bits=tr, out_cw=oc)
        if scheme == DCF:
            cw_v = torch.empty((in_bits, nkeys, 4), dtype=torch.int32, device=dev)
            oc = torch.empty((nkeys, 4), dtype=torch.int32, device=dev)
            _lib.check(L.fss_amd_dcf_relayout(self._h, in_bits, _ptr(cws), nkeys, _ptr(cw_s), _ptr(cw_v), _ptr(oc), st),
                       "dcf_relayout")
            return LevelMajorKeys(scheme, in_bits, nkeys, cw_s, cw_v=cw_v, out_cw=oc)
        ex = torch.empty((nkeys,), dtype=torch.int32, device=dev)
        _lib.check(L.fss_amd_ht_relayout(self._h, in_bits, _ptr(cws), nkeys, _ptr(cw_s), _ptr(ex), st), "ht_relayout")
        return LevelMajorKeys(scheme, in_bits, nkeys, cw_s, extra=ex)

    def key_major(self, keys):
        """LevelMajorKeys -> int32[nkeys, slots, 8] key-major raw Cw slots (the inverse of relayout)."""
        n, nb = keys.nkeys, keys.in_bits
        scheme = DPF if keys.scheme == GROTTO else keys.scheme
        self._check_keys(keys)
        cws = torch.empty((n, self._slots(scheme, nb), 8), dtype=torch.int32, device=self.device)
        L, st = self._L, self._stream()
        if scheme == DPF:
            rc = L.fss_amd_dpf_key_major(self._h, nb, _ptr(keys.cw_s), _ptr(keys.tr_bits), _ptr(keys.out_cw), n, _ptr(cws), st)
        elif scheme == DCF:
            rc = L.fss_amd_dcf_key_major(self._h, nb, _ptr(keys.cw_s), _ptr(keys.cw_v), _ptr(keys.out_cw), n, _ptr(cws), st)
        elif scheme == HT:
            rc = L.fss_amd_ht_key_major(self._h, nb, _ptr(keys.cw_s), _ptr(keys.extra), n, _ptr(cws), st)
        else:
            raise ValueError("key_major: DPF, DCF, Half-Tree or Grotto keys")
        _lib.check(rc, "key_major")
        return cws

    def _check_keys(self, keys):
        """Sizes of the level-major arrays of a LevelMajorKeys batch (they reach the kernels as raw pointers)."""
        n, nb = keys.nkeys, keys.in_bits
        self._need(keys.cw_s, 4 * nb * n, "keys.cw_s")
        if keys.scheme == DCF:
            self._need(keys.cw_v, 4 * nb * n, "keys.cw_v")
        if keys.scheme in (DPF, GROTTO, VDPF):
            self._need(keys.tr_bits, (nb + 63) // 64 * n, "keys.tr_bits")
        if keys.scheme in (DPF, DCF, GROTTO):
            self._need(keys.out_cw, 4 * n, "keys.out_cw")
        if keys.scheme == HT:
            self._need(keys.extra, n, "keys.extra")

    def eval_keys(self, keys, group, party, seeds, xs, ocws=None, out=None):
        """ys[k] = Eval(party, seeds[k], key k, xs[k]) on a LevelMajorKeys batch.
        Grotto keys: the path-walk evaluator for domains with no table.  It reconstructs 1[alpha <= x], but its
        per-party share is NOT the share of eval_all / grotto_eval_table / the reference's GrottoDcf::Eval (they
        differ by a party-independent mask): both parties must use eval_keys for the same (key, x)."""
        seeds, xs = self._dev(seeds, torch.int32), self._dev(xs)
        n, nb = keys.nkeys, keys.in_bits
        self._check_keys(keys)
        xb = self._xbytes(xs, nb)
        self._need(seeds, 4 * n, "seeds"); self._need(xs, n * (2 if xb == 16 else 1), "xs")
        if out is not None:
            if keys.scheme == GROTTO:
                self._out(out, n, "out", (torch.uint8, torch.bool))
            else:
                self._out(out, 16 * n, "out")
        L, st = self._L, self._stream()
        if keys.scheme == GROTTO:
            ys = out if out is not None else torch.empty((n,), dtype=torch.uint8, device=self.device)
            rc = L.fss_amd_grotto_eval_keys(self._h, nb, party, _ptr(seeds), _ptr(keys.cw_s), _ptr(keys.tr_bits),
                                            _ptr(xs), xb, n, _ptr(ys), st)
            _lib.check(rc, "grotto_eval_keys")
            return ys
        ys = out if out is not None else torch.empty((n, 4), dtype=torch.int32, device=self.device)
        if keys.scheme == DPF:
            rc = L.fss_amd_dpf_eval_keys(self._h, nb, group, party, _ptr(seeds), _ptr(keys.cw_s), _ptr(keys.tr_bits),
                                         _ptr(keys.out_cw), _ptr(xs), xb, n, _ptr(ys), st)
        elif keys.scheme == DCF:
            rc = L.fss_amd_dcf_eval_keys(self._h, nb, group, party, _ptr(seeds), _ptr(keys.cw_s), _ptr(keys.cw_v),
                                         _ptr(keys.out_cw), _ptr(xs), xb, n, _ptr(ys), st)
        else:
            ocws = self._dev(self._need(ocws, 4 * n, "ocws"), torch.int32)
            rc = L.fss_amd_ht_eval_keys(self._h, nb, group, party, _ptr(seeds), _ptr(keys.cw_s), _ptr(keys.extra),
                                        _ptr(ocws), _ptr(xs), xb, n, _ptr(ys), st)
        _lib.check(rc, "eval_keys")
        return ys

    # ------------------------------------------------------------ full domain
    def eval_all(self, scheme, in_bits, group, party, seed, cws, ocw=None, first_leaf=0, nleaves=None, out=None):
        seed, cws = self._dev(seed, torch.int32), self._dev(cws, torch.int32)
        if nleaves is None:
            nleaves = 1 << in_bits
        L, st = self._L, self._stream()
        ws_bytes = L.fss_amd_eval_all_workspace(scheme, in_bits, nleaves)
        if ws_bytes == 0:
            raise ValueError(f"eval_all: unsupported block (in_bits={in_bits}, nleaves={nleaves})")
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=self.device)
        self._need(seed, 4, "seed"); self._need(cws, 8 * self._slots(DPF if scheme == GROTTO else scheme, in_bits), "cws")
        if out is not None:
            if scheme == GROTTO:
                self._out(out, nleaves, "out", (torch.uint8, torch.bool))
            else:
                self._out(out, 16 * nleaves, "out")
        if scheme == GROTTO:
            ys = out if out is not None else torch.empty((nleaves,), dtype=torch.uint8, device=self.device)
            rc = L.fss_amd_grotto_eval_all(self._h, in_bits, party, _ptr(seed), _ptr(cws), first_leaf, nleaves,
                                           _ptr(ys), _ptr(ws), ws_bytes, st)
        else:
            ys = out if out is not None else torch.empty((nleaves, 4), dtype=torch.int32, device=self.device)
            if scheme == DPF:
                rc = L.fss_amd_dpf_eval_all(self._h, in_bits, group, party, _ptr(seed), _ptr(cws), first_leaf, nleaves,
                                            _ptr(ys), _ptr(ws), ws_bytes, st)
            elif scheme == DCF:
                rc = L.fss_amd_dcf_eval_all(self._h, in_bits, group, party, _ptr(seed), _ptr(cws), first_leaf, nleaves,
                                            _ptr(ys), _ptr(ws), ws_bytes, st)
            else:
                ocw = self._dev(self._need(ocw, 4, "ocw"), torch.int32)
                rc = L.fss_amd_ht_eval_all(self._h, in_bits, group, party, _ptr(seed), _ptr(cws), _ptr(ocw),
                                           first_leaf, nleaves, _ptr(ys), _ptr(ws), ws_bytes, st)
        _lib.check(rc, "eval_all")
        # the workspace must outlive the enqueued kernels
        ws.record_stream(torch.cuda.current_stream(self.device))
        return ys

    def eval_all_batch(self, scheme, in_bits, group, party, seeds, cws, ocws=None, out=None):
        """Full domain of every key of a key-major batch: seeds int32[n,4], cws int32[n,slots,8] ->
        int32[n, 2^in_bits, 4]   (fss::gpu::*EvalAllGpuBatch)."""
        seeds, cws = self._dev(seeds, torch.int32), self._dev(cws, torch.int32)
        if cws.dim() != 3 or cws.shape[1] != self._slots(scheme, in_bits) or cws.shape[2] != 8:
            raise ValueError(f"eval_all_batch: cws must be int32[nkeys, {self._slots(scheme, in_bits)}, 8], got {tuple(cws.shape)}")
        n = cws.shape[0]
        self._need(seeds, 4 * n, "seeds")
        L, st = self._L, self._stream()
        ws_bytes = L.fss_amd_eval_all_batch_workspace(scheme, in_bits, n)
        if ws_bytes == 0:
            raise ValueError(f"eval_all_batch: unsupported (scheme={scheme}, in_bits={in_bits}, nkeys={n})")
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=self.device)
        ys = (self._out(out, 16 * n << in_bits, "out") if out is not None
              else torch.empty((n, 1 << in_bits, 4), dtype=torch.int32, device=self.device))
        if scheme == DPF:
            rc = L.fss_amd_dpf_eval_all_batch(self._h, in_bits, group, party, _ptr(seeds), _ptr(cws), n, _ptr(ys), _ptr(ws), ws_bytes, st)
        elif scheme == DCF:
            rc = L.fss_amd_dcf_eval_all_batch(self._h, in_bits, group, party, _ptr(seeds), _ptr(cws), n, _ptr(ys), _ptr(ws), ws_bytes, st)
        elif scheme == HT:
            ocws = self._dev(self._need(ocws, 4 * n, "ocws"), torch.int32)
            rc = L.fss_amd_ht_eval_all_batch(self._h, in_bits, group, party, _ptr(seeds), _ptr(cws), _ptr(ocws), n, _ptr(ys),
                                             _ptr(ws), ws_bytes, st)
        else:
            raise ValueError("eval_all_batch: DPF, DCF or Half-Tree")
        _lib.check(rc, "eval_all_batch")
        ws.record_stream(torch.cuda.current_stream(self.device))
        return ys

    # ------------------------------------------------------------ Grotto parity table
    def grotto_preprocess(self, in_bits, party, seed, cws):
        """GrottoDcf::Preprocess: uint8[2N-1] parity tree (device)."""
        seed, cws = self._dev(seed, torch.int32), self._dev(cws, torch.int32)
        N = 1 << in_bits
        L, st = self._L, self._stream()
        ws_bytes = L.fss_amd_eval_all_workspace(GROTTO, in_bits, N) + N
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=self.device)
        pt = torch.empty((2 * N - 1,), dtype=torch.uint8, device=self.device)
        _lib.check(L.fss_amd_grotto_preprocess(self._h, in_bits, party, _ptr(seed), _ptr(cws), _ptr(pt), _ptr(ws),
                                               ws_bytes, st), "grotto_preprocess")
        ws.record_stream(torch.cuda.current_stream(self.device))
        return pt

    def grotto_eval_table(self, in_bits, pt, xs, out=None):
        """GrottoDcf::Eval over a batch: uint8[n]."""
        if in_bits > 40:
            raise ValueError("grotto_eval_table: the parity table has 2^(in_bits+1) - 1 bytes; in_bits <= 40")
        pt, xs = self._dev(pt, torch.uint8), self._dev(xs)
        xb = self._xbytes(xs, in_bits)
        if xb == 16:
            raise ValueError("grotto_eval_table covers in_bits <= 64: pass int32 / int64 xs")
        n = xs.numel()
        self._need(pt, (2 << in_bits) - 1, "pt")
        ys = self._out(out, n, "out", (torch.uint8, torch.bool)) if out is not None else torch.empty((n,), dtype=torch.uint8, device=self.device)
        _lib.check(self._L.fss_amd_grotto_eval_table(self._h, in_bits, _ptr(pt), _ptr(xs), xb, n,
                                                     _ptr(ys), self._stream()), "grotto_eval_table")
        return ys

    # ------------------------------------------------------------ key generation
    def gen(self, scheme, in_bits, group, s0s, alphas, betas, pred=0):
        """s0s int32[n,2,4], alphas int32/int64[n], betas int32[n,4] -> (cws int32[n,slots,8], ocws|None)."""
        s0s, alphas, betas = self._dev(s0s, torch.int32), self._dev(alphas), self._dev(betas, torch.int32)
        xb = self._xbytes(alphas, in_bits)
        n = alphas.shape[0] if xb == 16 else alphas.numel()
        self._need(s0s, 8 * n, "s0s"); self._need(betas, 4 * n, "betas")
        slots = in_bits if scheme == HT else in_bits + 1
        cws = torch.empty((n, slots, 8), dtype=torch.int32, device=self.device)
        L, st = self._L, self._stream()
        ocws = None
        if scheme in (DPF, GROTTO):
            rc = L.fss_amd_dpf_gen(self._h, in_bits, group, _ptr(s0s), _ptr(alphas), xb, _ptr(betas), n, _ptr(cws), st)
        elif scheme == DCF:
            rc = L.fss_amd_dcf_gen(self._h, in_bits, group, pred, _ptr(s0s), _ptr(alphas), xb, _ptr(betas), n,
                                   _ptr(cws), st)
        else:
            ocws = torch.empty((n, 4), dtype=torch.int32, device=self.device)
            rc = L.fss_amd_ht_gen(self._h, in_bits, group, _ptr(s0s), _ptr(alphas), xb, _ptr(betas), n, _ptr(cws),
                                  _ptr(ocws), st)
        _lib.check(rc, "gen")
        return cws, ocws

    def gen_level_major(self, scheme, in_bits, group, s0s, alphas, betas, pred=0):
        """Same keys as gen() written level-major directly -> (LevelMajorKeys, ocws|None); equals
        relayout(gen(...)) without the key-major intermediate."""
        s0s, alphas, betas = self._dev(s0s, torch.int32), self._dev(alphas), self._dev(betas, torch.int32)
        xb = self._xbytes(alphas, in_bits)
        n = alphas.shape[0] if xb == 16 else alphas.numel()
        self._need(s0s, 8 * n, "s0s"); self._need(betas, 4 * n, "betas")
        dev, L, st = self.device, self._L, self._stream()
        cw_s = torch.empty((in_bits, n, 4), dtype=torch.int32, device=dev)
        if scheme in (DPF, GROTTO):
            tr = torch.empty(((in_bits + 63) // 64 * n,), dtype=torch.int64, device=dev)
            oc = torch.empty((n, 4), dtype=torch.int32, device=dev)
            rc = L.fss_amd_dpf_gen_level_major(self._h, in_bits, group, _ptr(s0s), _ptr(alphas), xb, _ptr(betas), n,
                                               _ptr(cw_s), _ptr(tr), _ptr(oc), st)
            _lib.check(rc, "dpf_gen_level_major")
            return LevelMajorKeys(scheme, in_bits, n, cw_s, tr_bits=tr, out_cw=oc), None
        if scheme == DCF:
            cw_v = torch.empty((in_bits, n, 4), dtype=torch.int32, device=dev)
            oc = torch.empty((n, 4), dtype=torch.int32, device=dev)
            rc = L.fss_amd_dcf_gen_level_major(self._h, in_bits, group, pred, _ptr(s0s), _ptr(alphas), xb, _ptr(betas),
                                               n, _ptr(cw_s), _ptr(cw_v), _ptr(oc), st)
            _lib.check(rc, "dcf_gen_level_major")
            return LevelMajorKeys(scheme, in_bits, n, cw_s, cw_v=cw_v, out_cw=oc), None
        ex = torch.empty((n,), dtype=torch.int32, device=dev)
        ocws = torch.empty((n, 4), dtype=torch.int32, device=dev)
        rc = L.fss_amd_ht_gen_level_major(self._h, in_bits, group, _ptr(s0s), _ptr(alphas), xb, _ptr(betas), n,
                                          _ptr(cw_s), _ptr(ex), _ptr(ocws), st)
        _lib.check(rc, "ht_gen_level_major")
        return LevelMajorKeys(scheme, in_bits, n, cw_s, extra=ex), ocws


    def set_uint_modulus(self, mod):
        """fss::group::Uint<T, mod> for the u8..u64 groups (0 = the wrapping group, the default)."""
        _lib.check(self._L.fss_amd_ctx_set_uint_modulus(self._h, int(mod)), "fss_amd_ctx_set_uint_modulus")

    # ---------------------------------------------------------------- verifiable DPF
    def set_hash_iv(self, iv):
        """Key of the keyed-BLAKE3 hashes H and H' of the VDPF (fss::hash::Blake3's iv, 8 words)."""
        v = (C.c_int32 * 8)(*[int(np.uint32(int(x) & 0xFFFFFFFF).astype(np.int32)) for x in iv])
        _lib.check(self._L.fss_amd_ctx_set_hash_iv(self._h, v), "fss_amd_ctx_set_hash_iv")

    def set_hash_sha256(self, key):
        """Use fss::hash::Sha256(key) (4 words) for the VDPF's H and H' instead of keyed BLAKE3."""
        v = (C.c_int32 * 4)(*[int(np.uint32(int(x) & 0xFFFFFFFF).astype(np.int32)) for x in key])
        _lib.check(self._L.fss_amd_ctx_set_hash_sha256(self._h, v), "fss_amd_ctx_set_hash_sha256")

    def vdpf_gen(self, in_bits, group, s0s, alphas, betas, level_major=False):
        """-> (cws int32[n, in_bits, 8] | LevelMajorKeys, cs int32[n,4,4], ocws int32[n,4], retry int32[n])."""
        s0s, alphas, betas = self._dev(s0s, torch.int32), self._dev(alphas), self._dev(betas, torch.int32)
        xb = self._xbytes(alphas, in_bits)
        n = alphas.shape[0] if xb == 16 else alphas.numel()
        dev, L, st = self.device, self._L, self._stream()
        cs = torch.empty((n, 4, 4), dtype=torch.int32, device=dev)
        ocws = torch.empty((n, 4), dtype=torch.int32, device=dev)
        retry = torch.empty((n,), dtype=torch.int32, device=dev)
        if level_major:
            cw_s = torch.empty((in_bits, n, 4), dtype=torch.int32, device=dev)
            tr = torch.empty(((in_bits + 63) // 64 * n,), dtype=torch.int64, device=dev)
            rc = L.fss_amd_vdpf_gen_level_major(self._h, in_bits, group, _ptr(s0s), _ptr(alphas), xb, _ptr(betas), n,
                                                _ptr(cw_s), _ptr(tr), _ptr(cs), _ptr(ocws), _ptr(retry), st)
            _lib.check(rc, "vdpf_gen_level_major")
            return LevelMajorKeys(VDPF, in_bits, n, cw_s, tr_bits=tr), cs, ocws, retry
        cws = torch.empty((n, in_bits, 8), dtype=torch.int32, device=dev)
        rc = L.fss_amd_vdpf_gen(self._h, in_bits, group, _ptr(s0s), _ptr(alphas), xb, _ptr(betas), n, _ptr(cws), _ptr(cs),
                                _ptr(ocws), _ptr(retry), st)
        _lib.check(rc, "vdpf_gen")
        return cws, cs, ocws, retry

    def vdpf_relayout(self, in_bits, cws):
        cws = self._dev(cws, torch.int32)
        if cws.dim() != 3 or cws.shape[1] != in_bits or cws.shape[2] != 8:
            raise ValueError(f"vdpf_relayout: cws must be int32[nkeys, {in_bits}, 8], got {tuple(cws.shape)}")
        n = cws.shape[0]
        cw_s = torch.empty((in_bits, n, 4), dtype=torch.int32, device=self.device)
        tr = torch.empty(((in_bits + 63) // 64 * n,), dtype=torch.int64, device=self.device)
        _lib.check(self._L.fss_amd_vdpf_relayout(self._h, in_bits, _ptr(cws), n, _ptr(cw_s), _ptr(tr), self._stream()),
                   "vdpf_relayout")
        return LevelMajorKeys(VDPF, in_bits, n, cw_s, tr_bits=tr)

    def vdpf_eval_points(self, in_bits, group, party, s0, cws, cs, ocw, xs):
        """One key at many points -> (ys int32[n,4], pi_tildes int32[n,4,4])."""
        s0, cws, cs, ocw, xs = (self._dev(s0, torch.int32), self._dev(cws, torch.int32), self._dev(cs, torch.int32),
                                self._dev(ocw, torch.int32), self._dev(xs))
        xb = self._xbytes(xs, in_bits)
        n = xs.shape[0] if xb == 16 else xs.numel()
        self._need(s0, 4, "s0"); self._need(cws, 8 * in_bits, "cws"); self._need(cs, 16, "cs"); self._need(ocw, 4, "ocw")
        ys = torch.empty((n, 4), dtype=torch.int32, device=self.device)
        pis = torch.empty((n, 4, 4), dtype=torch.int32, device=self.device)
        rc = self._L.fss_amd_vdpf_eval_points(self._h, in_bits, group, party, _ptr(s0), _ptr(cws), _ptr(cs), _ptr(ocw),
                                              _ptr(xs), xb, n, _ptr(ys), _ptr(pis), self._stream())
        _lib.check(rc, "vdpf_eval_points")
        return ys, pis

    def vdpf_eval_keys(self, keys, group, party, seeds, cs, ocws, xs, out=None):
        """ys[k], pi_tildes[k] = Vdpf::Eval(party, seeds[k], key k, xs[k]) on a LevelMajorKeys batch."""
        seeds, cs, ocws, xs = (self._dev(seeds, torch.int32), self._dev(cs, torch.int32), self._dev(ocws, torch.int32),
                               self._dev(xs))
        n, nb = keys.nkeys, keys.in_bits
        xb = self._xbytes(xs, nb)
        self._check_keys(keys)
        self._need(seeds, 4 * n, "seeds"); self._need(cs, 16 * n, "cs"); self._need(ocws, 4 * n, "ocws")
        self._need(xs, n * (2 if xb == 16 else 1), "xs")
        if out is not None:
            ys, pis = self._out(out[0], 16 * n, "out[0]"), self._out(out[1], 64 * n, "out[1]")
        else:
            ys, pis = (torch.empty((n, 4), dtype=torch.int32, device=self.device),
                       torch.empty((n, 4, 4), dtype=torch.int32, device=self.device))
        rc = self._L.fss_amd_vdpf_eval_keys(self._h, nb, group, party, _ptr(seeds), _ptr(keys.cw_s), _ptr(keys.tr_bits),
                                            _ptr(cs), _ptr(ocws), _ptr(xs), xb, n, _ptr(ys), _ptr(pis), self._stream())
        _lib.check(rc, "vdpf_eval_keys")
        return ys, pis

    def vdpf_prove(self, pi_tildes, cs):
        """Vdpf::Prove over the hashes of ONE key (sequential chain) -> int32[4,4]."""
        pis, cs = self._dev(pi_tildes, torch.int32), self._dev(cs, torch.int32)
        self._need(cs, 16, "cs")
        out = torch.empty((4, 4), dtype=torch.int32, device=self.device)
        _lib.check(self._L.fss_amd_vdpf_prove(self._h, _ptr(pis), pis.numel() // 16, _ptr(cs), _ptr(out), self._stream()),
                   "vdpf_prove")
        return out

    def vdpf_prove_keys(self, pi_tildes, cs):
        """Per key, one evaluation each -> int32[n,4,4]."""
        pis, cs = self._dev(pi_tildes, torch.int32), self._dev(cs, torch.int32)
        n = pis.numel() // 16
        self._need(cs, 16 * n, "cs")
        out = torch.empty((n, 4, 4), dtype=torch.int32, device=self.device)
        _lib.check(self._L.fss_amd_vdpf_prove_keys(self._h, _ptr(pis), _ptr(cs), n, _ptr(out), self._stream()),
                   "vdpf_prove_keys")
        return out


_default = {}


def default_engine(device=None, prg="aes128_mmo"):
    """Process-wide engine per (device, prg): the reference's bench AES keys / test ChaCha nonce."""
    if not torch.cuda.is_available():
        raise RuntimeError("fss_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
    idx = torch.cuda.current_device() if device is None else (device if isinstance(device, int) else torch.device(device).index or 0)
    if (idx, prg) not in _default:
        _default[(idx, prg)] = Engine(idx, prg=prg)
    return _default[(idx, prg)]
