"""ctypes/numpy front end of the CPU oracle (oracle/fss_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (fss_amd) never imports this.

Shapes follow the reference's Python binding (fss_crypto/_csrc/dpf_binding_impl.cuh:27-131):
seeds are int32[4], correction words int32[n_slots, 8] (raw 32-byte Cw slots),
outputs int32[..., 4].
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfss_oracle.so")

BYTES, U8, U16, U32, U64, U128M127 = 0, 1, 2, 3, 4, 5
LT, GT = 0, 1
DPF, DCF, HT, GROTTO = 0, 1, 2, 3

# AES keys and Half-Tree hash key of the reference's benches/samples
# (src/bench_cpu.cu:37-40, :509; samples/dpf_dcf_cpu.cu:96-103).
KEYS = [
    bytes(range(1, 17)),
    bytes(range(16, 0, -1)),
    bytes([1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8]),
    bytes([8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1]),
]
HASH_KEY = np.array([0x12345678, 0x9ABCDEF0, 0x13572468, 0x2468ACE0], dtype=np.uint32).view(np.int32)


def build(force=False):
    """Compile oracle/libfss_oracle.so with gcc (no-op when up to date)."""
    src = os.path.join(_HERE, "fss_oracle.c")
    hdr = os.path.join(_HERE, "fss_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libfss_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class _I4(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("z", C.c_int32), ("w", C.c_int32)]


class _Prg(C.Structure):
    _fields_ = [("mul", C.c_int), ("use_evp", C.c_int), ("key", (C.c_uint8 * 16) * 4),
                ("rk", (C.c_uint32 * 44) * 4), ("evp", C.c_void_p * 4), ("kind", C.c_int), ("rounds", C.c_int),
                ("nonce", C.c_int32 * 2)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i, u64, sz = C.c_void_p, C.c_int, C.c_uint64, C.c_size_t
        P = C.POINTER(_Prg)
        L.fo_prg_init.argtypes = [P, vp, i, i]
        L.fo_prg_init.restype = i
        L.fo_prg_free.argtypes = [P]
        L.fo_prg_gen.argtypes = [P, _I4, vp]
        L.fo_aes128_encrypt.argtypes = [vp, vp, vp]
        L.fo_prg_init_chacha.argtypes = [P, i, i, vp]
        L.fo_prg_init_chacha.restype = i
        L.fo_chacha_permute.argtypes = [vp, i]
        L.fo_dpf_gen.argtypes = [P, i, i, vp, vp, u64, u64, _I4]
        L.fo_dpf_eval.argtypes = [P, i, i, i, _I4, vp, u64, u64]
        L.fo_dpf_eval.restype = _I4
        L.fo_dpf_eval_all.argtypes = [P, i, i, i, _I4, vp, vp]
        L.fo_dpf_eval_all_tasks.argtypes = [P, i, i, i, _I4, vp, vp, i]
        L.fo_dpf_eval_all_repeat.argtypes = [P, i, i, i, _I4, vp, vp, i, i]
        L.fo_dcf_gen.argtypes = [P, i, i, i, vp, vp, u64, u64, _I4]
        L.fo_dcf_eval.argtypes = [P, i, i, i, _I4, vp, u64, u64]
        L.fo_dcf_eval.restype = _I4
        L.fo_dcf_eval_all.argtypes = [P, i, i, i, _I4, vp, vp]
        L.fo_ht_gen.argtypes = [P, _I4, i, i, vp, vp, vp, u64, u64, _I4]
        L.fo_ht_eval.argtypes = [P, _I4, i, i, i, _I4, vp, _I4, u64, u64]
        L.fo_ht_eval.restype = _I4
        L.fo_ht_eval_all.argtypes = [P, _I4, i, i, i, _I4, vp, _I4, vp]
        L.fo_grotto_gen.argtypes = [P, i, vp, vp, u64, u64]
        L.fo_grotto_preprocess.argtypes = [P, i, i, _I4, vp, vp]
        L.fo_grotto_eval.argtypes = [i, vp, u64]
        L.fo_grotto_eval.restype = i
        L.fo_grotto_eval_all.argtypes = [P, i, i, _I4, vp, vp]
        L.fo_grotto_eval_walk.argtypes = [P, i, i, _I4, vp, u64]
        L.fo_grotto_eval_walk.restype = i
        u32 = C.c_uint32
        L.fo_blake3_compress.argtypes = [vp, vp, u64, u32, u32, vp]
        L.fo_blake3_hash.argtypes = [vp, vp, vp]
        L.fo_blake3_xorhash.argtypes = [vp, _I4, _I4, vp]
        L.fo_vdpf_gen.argtypes = [P, vp, i, i, vp, vp, vp, vp, u64, u64, _I4]
        L.fo_vdpf_gen.restype = i
        L.fo_vdpf_eval.argtypes = [P, vp, i, i, i, _I4, vp, vp, _I4, u64, u64, vp]
        L.fo_vdpf_eval.restype = _I4
        L.fo_vdpf_prove.argtypes = [vp, vp, sz, vp, vp]
        L.fo_set_uint_modulus.argtypes = [u64]
        L.fo_sha256.argtypes = [vp, sz, vp]
        L.fo_set_hash_kind.argtypes = [i]
        L.fo_eval_points.argtypes = [P, _I4, i, i, i, i, _I4, vp, _I4, vp, vp, sz, vp, i]
        L.fo_eval_keys.argtypes = [P, _I4, i, i, i, i, vp, vp, sz, vp, vp, vp, sz, vp, i]
        L.fo_gen_keys.argtypes = [P, _I4, i, i, i, i, vp, vp, vp, vp, sz, vp, sz, vp, i]
        _lib = L
    return _lib


def _i4(a):
    a = np.asarray(a).astype(np.int64)
    a = (a & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    return _I4(int(a[0]), int(a[1]), int(a[2]), int(a[3]))


def _arr(v):
    return np.array([v.x, v.y, v.z, v.w], dtype=np.int32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _split(v):
    v = int(v)
    return v & ((1 << 64) - 1), (v >> 64) & ((1 << 64) - 1)


def set_uint_modulus(mod: int):
    """Uint<T, mod> for the U8..U64 groups (0 = wrapping); process-wide."""
    lib().fo_set_uint_modulus(int(mod))


def have_aesni() -> bool:
    return bool(lib().fo_have_aesni())


def aes128_encrypt(key: bytes, block: bytes) -> bytes:
    out = (C.c_uint8 * 16)()
    lib().fo_aes128_encrypt(C.c_char_p(key), C.c_char_p(block), out)
    return bytes(out)


CHACHA_NONCE = (0x12345678, 0x9ABCDEF0)  # the nonce of the reference's C++ tests (src/dcf_test.cu:35)


def chacha_permute(words, rounds=20):
    a = np.array(words, dtype=np.uint32)
    lib().fo_chacha_permute(_ptr(a), rounds)
    return a


class Prg:
    """fss::prg::Aes128Mmo<mul> (include/fss/prg/aes128_mmo.cuh:27-95), or with chacha=True
    fss::prg::ChaCha<mul, rounds> (include/fss/prg/chacha.cuh:24-128)."""

    def __init__(self, mul, keys=None, use_evp=False, chacha=False, nonce=CHACHA_NONCE, rounds=20, aesni=False):
        self.mul = mul
        self._p = _Prg()
        if chacha:
            nn = (C.c_int32 * 2)(*[int(np.uint32(v).astype(np.int32)) for v in nonce])
            if lib().fo_prg_init_chacha(C.byref(self._p), mul, rounds, nn) != 0:
                raise RuntimeError("fo_prg_init_chacha failed")
            return
        keys = KEYS[:mul] if keys is None else keys
        assert len(keys) == mul
        buf = b"".join(keys)
        rc = lib().fo_prg_init(C.byref(self._p), C.c_char_p(buf), mul, 2 if aesni else (1 if use_evp else 0))
        if rc != 0:
            raise RuntimeError("fo_prg_init failed (OpenSSL missing, or no AES-NI on this CPU?)")

    def __del__(self):
        try:
            lib().fo_prg_free(C.byref(self._p))
        except Exception:
            pass

    @property
    def ref(self):
        return C.byref(self._p)

    def gen(self, seed):
        out = np.zeros((self.mul, 4), dtype=np.int32)
        lib().fo_prg_gen(self.ref, _i4(seed), _ptr(out))
        return out


def _cws_buf(n):
    return np.zeros((n, 8), dtype=np.int32)


def _seeds2(s0s):
    return np.ascontiguousarray(np.asarray(s0s, dtype=np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32).reshape(2, 4)


# ---------------------------------------------------------------- DPF
def dpf_gen(prg, in_bits, group, s0s, alpha, beta):
    cws = _cws_buf(in_bits + 1)
    s = _seeds2(s0s)
    lo, hi = _split(alpha)
    lib().fo_dpf_gen(prg.ref, in_bits, group, _ptr(cws), _ptr(s), lo, hi, _i4(beta))
    return cws


def dpf_eval(prg, in_bits, group, b, s0, cws, x):
    lo, hi = _split(x)
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    return _arr(lib().fo_dpf_eval(prg.ref, in_bits, group, int(b), _i4(s0), _ptr(cws), lo, hi))


def dpf_eval_all(prg, in_bits, group, b, s0, cws, task_threads=0, out=None):
    """task_threads = 0: the flattened breadth-first form; >= 1: the reference's depth-first recursion with OpenMP
    tasks above ceil(log2(task_threads)) levels (dpf.cuh:232-303) -- same leaves."""
    ys = np.zeros((1 << in_bits, 4), dtype=np.int32) if out is None else out
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    if task_threads:
        lib().fo_dpf_eval_all_tasks(prg.ref, in_bits, group, int(b), _i4(s0), _ptr(cws), _ptr(ys), int(task_threads))
    else:
        lib().fo_dpf_eval_all(prg.ref, in_bits, group, int(b), _i4(s0), _ptr(cws), _ptr(ys))
    return ys


def dpf_eval_all_repeat(prg, in_bits, group, b, s0, cws, task_threads, reps):
    """`reps` EvalAll calls inside one foreign call (timing of small domains); returns the last outputs."""
    ys = np.zeros((1 << in_bits, 4), dtype=np.int32)
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    lib().fo_dpf_eval_all_repeat(prg.ref, in_bits, group, int(b), _i4(s0), _ptr(cws), _ptr(ys), int(task_threads), int(reps))
    return ys


# ---------------------------------------------------------------- DCF
def dcf_gen(prg, in_bits, group, pred, s0s, alpha, beta):
    cws = _cws_buf(in_bits + 1)
    s = _seeds2(s0s)
    lo, hi = _split(alpha)
    lib().fo_dcf_gen(prg.ref, in_bits, group, pred, _ptr(cws), _ptr(s), lo, hi, _i4(beta))
    return cws


def dcf_eval(prg, in_bits, group, b, s0, cws, x):
    lo, hi = _split(x)
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    return _arr(lib().fo_dcf_eval(prg.ref, in_bits, group, int(b), _i4(s0), _ptr(cws), lo, hi))


def dcf_eval_all(prg, in_bits, group, b, s0, cws):
    ys = np.zeros((1 << in_bits, 4), dtype=np.int32)
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    lib().fo_dcf_eval_all(prg.ref, in_bits, group, int(b), _i4(s0), _ptr(cws), _ptr(ys))
    return ys


# ---------------------------------------------------------------- Half-Tree
def ht_gen(prg, in_bits, group, s0s, alpha, beta, hash_key=HASH_KEY):
    cws = _cws_buf(in_bits)
    ocw = np.zeros(4, dtype=np.int32)
    s = _seeds2(s0s)
    lo, hi = _split(alpha)
    lib().fo_ht_gen(prg.ref, _i4(hash_key), in_bits, group, _ptr(cws), _ptr(ocw), _ptr(s), lo, hi, _i4(beta))
    return cws, ocw


def ht_eval(prg, in_bits, group, b, s0, cws, ocw, x, hash_key=HASH_KEY):
    lo, hi = _split(x)
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    return _arr(lib().fo_ht_eval(prg.ref, _i4(hash_key), in_bits, group, int(b), _i4(s0), _ptr(cws), _i4(ocw), lo, hi))


def ht_eval_all(prg, in_bits, group, b, s0, cws, ocw, hash_key=HASH_KEY):
    ys = np.zeros((1 << in_bits, 4), dtype=np.int32)
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    lib().fo_ht_eval_all(prg.ref, _i4(hash_key), in_bits, group, int(b), _i4(s0), _ptr(cws), _i4(ocw), _ptr(ys))
    return ys


# ---------------------------------------------------------------- Grotto
def grotto_gen(prg, in_bits, s0s, alpha):
    cws = _cws_buf(in_bits + 1)
    s = _seeds2(s0s)
    lo, hi = _split(alpha)
    lib().fo_grotto_gen(prg.ref, in_bits, _ptr(cws), _ptr(s), lo, hi)
    return cws


def grotto_preprocess(prg, in_bits, b, s0, cws):
    pt = np.zeros(2 * (1 << in_bits) - 1, dtype=np.uint8)
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    lib().fo_grotto_preprocess(prg.ref, in_bits, int(b), _i4(s0), _ptr(cws), _ptr(pt))
    return pt


def grotto_eval(in_bits, pt, x):
    return lib().fo_grotto_eval(in_bits, _ptr(pt), int(x))


def grotto_eval_all(prg, in_bits, b, s0, cws):
    ys = np.zeros(1 << in_bits, dtype=np.uint8)
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    lib().fo_grotto_eval_all(prg.ref, in_bits, int(b), _i4(s0), _ptr(cws), _ptr(ys))
    return ys


def grotto_eval_walk(prg, in_bits, b, s0, cws, x):
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    return lib().fo_grotto_eval_walk(prg.ref, in_bits, int(b), _i4(s0), _ptr(cws), int(x))


# ---------------------------------------------------------------- batched
# ---------------------------------------------------------------- BLAKE3 + VDPF
BLAKE3_IV = np.array([0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19],
                     dtype=np.uint32)
# hash IV of the reference's VDPF tests (src/vdpf_test.cu:35-36)
HASH_IV = np.array([0x11111111, 0x22222222, 0x33333333, 0x44444444, 0x55555555, 0x66666666, 0x77777777, 0x88888888],
                   dtype=np.uint32).view(np.int32)


def sha256(data: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().fo_sha256(C.c_char_p(data), len(data), out)
    return bytes(out)


def set_hash_kind(kind: int):
    """0 = keyed BLAKE3 (fss::hash::Blake3), 1 = SHA-256 (fss::hash::Sha256, key = first 4 IV words); process-wide."""
    lib().fo_set_hash_kind(int(kind))


def blake3_compress(h, msg, counter=0, block_len=64, flags=0):
    h = np.ascontiguousarray(h, dtype=np.uint32)
    msg = np.ascontiguousarray(msg, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    lib().fo_blake3_compress(_ptr(h), _ptr(msg), counter, block_len, flags, _ptr(out))
    return out


def _iv(iv):
    return np.ascontiguousarray(np.asarray(iv, dtype=np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def blake3_hash(iv, msg4):
    """fss::hash::Blake3::Hash(span<int4,4>) -> int32[2,4]."""
    m = np.ascontiguousarray(msg4, dtype=np.int32).reshape(4, 4)
    out = np.zeros((2, 4), dtype=np.int32)
    lib().fo_blake3_hash(_ptr(_iv(iv)), _ptr(m), _ptr(out))
    return out


def blake3_xorhash(iv, a, b):
    """fss::hash::Blake3::Hash(tuple<int4, int4>) -> int32[4,4]."""
    out = np.zeros((4, 4), dtype=np.int32)
    lib().fo_blake3_xorhash(_ptr(_iv(iv)), _i4(a), _i4(b), _ptr(out))
    return out


def vdpf_gen(prg, in_bits, group, s0s, alpha, beta, iv=HASH_IV):
    """-> (cws int32[in_bits, 8], cs int32[4, 4], ocw int32[4], retry flag)."""
    cws = _cws_buf(in_bits)
    cs = np.zeros((4, 4), dtype=np.int32)
    ocw = np.zeros(4, dtype=np.int32)
    s = _seeds2(s0s)
    lo, hi = _split(alpha)
    rc = lib().fo_vdpf_gen(prg.ref, _ptr(_iv(iv)), in_bits, group, _ptr(cws), _ptr(cs), _ptr(ocw), _ptr(s), lo, hi,
                           _i4(beta))
    return cws, cs, ocw, rc


def vdpf_eval(prg, in_bits, group, b, s0, cws, cs, ocw, x, iv=HASH_IV):
    """-> (y int32[4], pi_tilde int32[4, 4])."""
    lo, hi = _split(x)
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    cs = np.ascontiguousarray(cs, dtype=np.int32)
    pt = np.zeros((4, 4), dtype=np.int32)
    y = _arr(lib().fo_vdpf_eval(prg.ref, _ptr(_iv(iv)), in_bits, group, int(b), _i4(s0), _ptr(cws), _ptr(cs), _i4(ocw),
                                lo, hi, _ptr(pt)))
    return y, pt


def vdpf_prove(pi_tildes, cs, iv=HASH_IV):
    pts = np.ascontiguousarray(pi_tildes, dtype=np.int32).reshape(-1, 4, 4)
    cs = np.ascontiguousarray(cs, dtype=np.int32)
    pi = np.zeros((4, 4), dtype=np.int32)
    lib().fo_vdpf_prove(_ptr(_iv(iv)), _ptr(pts), pts.shape[0], _ptr(cs), _ptr(pi))
    return pi


def _lohi(vals):
    """uint64 array, or a list of Python ints (any width <= 128) -> (lo uint64[n], hi uint64[n] or None)."""
    if isinstance(vals, np.ndarray) and vals.dtype != object:
        return np.ascontiguousarray(vals, dtype=np.uint64), None
    vals = [int(v) for v in vals]
    lo = np.array([v & 0xFFFFFFFFFFFFFFFF for v in vals], dtype=np.uint64)
    if all(v < (1 << 64) for v in vals):
        return lo, None
    return lo, np.array([v >> 64 for v in vals], dtype=np.uint64)


def eval_points(prg, scheme, in_bits, group, b, s0, cws, xs, ocw=None, hash_key=HASH_KEY, nthreads=1):
    """One key, many points.  xs: uint64[n] or a list of ints (<= 128 bits).  Returns int32[n,4]."""
    xs, xh = _lohi(xs)
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    ys = np.zeros((xs.size, 4), dtype=np.int32)
    oc = _i4(ocw if ocw is not None else [0, 0, 0, 0])
    lib().fo_eval_points(prg.ref, _i4(hash_key), scheme, in_bits, group, int(b), _i4(s0), _ptr(cws), oc, _ptr(xs),
                         None if xh is None else _ptr(xh), xs.size, _ptr(ys), nthreads)
    return ys


def eval_keys(prg, scheme, in_bits, group, b, s0s, cws, xs, ocws=None, hash_key=HASH_KEY, nthreads=1):
    """Many keys x one point each.  s0s int32[n,4]; cws int32[n,stride,8]; xs uint64[n] or list of ints."""
    xs, xh = _lohi(xs)
    cws = np.ascontiguousarray(cws, dtype=np.int32)
    s0s = np.ascontiguousarray(s0s, dtype=np.int32)
    n, stride = cws.shape[0], cws.shape[1]
    ys = np.zeros((n, 4), dtype=np.int32)
    op = None
    if ocws is not None:
        ocws = np.ascontiguousarray(ocws, dtype=np.int32)
        op = _ptr(ocws)
    lib().fo_eval_keys(prg.ref, _i4(hash_key), scheme, in_bits, group, int(b), _ptr(s0s), _ptr(cws), stride, op,
                       _ptr(xs), None if xh is None else _ptr(xh), n, _ptr(ys), nthreads)
    return ys


def gen_keys(prg, scheme, in_bits, group, s0s, alphas, betas, pred=LT, hash_key=HASH_KEY, nthreads=1):
    """Many keys.  s0s int32[n,2,4]; alphas uint64[n]; betas int32[n,4].
    Returns (cws int32[n,stride,8], ocws int32[n,4] or None)."""
    s0s = np.ascontiguousarray(s0s, dtype=np.int32)
    alphas, ah = _lohi(alphas)
    betas = np.ascontiguousarray(betas, dtype=np.int32)
    n = alphas.size
    stride = in_bits if scheme == HT else in_bits + 1
    cws = np.zeros((n, stride, 8), dtype=np.int32)
    ocws = np.zeros((n, 4), dtype=np.int32)
    lib().fo_gen_keys(prg.ref, _i4(hash_key), scheme, in_bits, group, pred, _ptr(s0s), _ptr(alphas),
                      None if ah is None else _ptr(ah), _ptr(betas), n, _ptr(cws), stride, _ptr(ocws), nthreads)
    return cws, (ocws if scheme == HT else None)
