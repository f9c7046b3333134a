"""Pins the CPU oracle (oracle/fss_oracle.c): FIPS-197, OpenSSL EVP (the reference's
PRG dependency), and the reference's own outputs recorded in SURVEY.md 8(c)."""
import hashlib

import numpy as np
import pytest

from kat import load, words

K = load()
S0, S1, BETA = words(K["s0"]), words(K["s1"]), words(K["beta"])


def test_te0_known_answer_is_the_fips197_sbox_shape():
    """SURVEY 8c recorded te0[0] = c66363a5, te0[1] = f87c7c84: (2S, S, S, 3S) of S(0) = 0x63, S(1) = 0x7c (FIPS-197 fig. 7)."""
    K = load()
    for x, s in ((0, 0x63), (1, 0x7C)):
        s2 = ((s << 1) ^ (0x1B if s & 0x80 else 0)) & 0xFF
        assert int(K["te0_be"][x], 16) == (s2 << 24) | (s << 16) | (s << 8) | (s2 ^ s)


def test_fips197_appendix_c1(O):
    key = bytes(range(16))
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    assert O.aes128_encrypt(key, pt).hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"


def test_fips197_appendix_b(O):
    key = bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c")
    pt = bytes.fromhex("3243f6a8885a308d313198a2e0370734")
    assert O.aes128_encrypt(key, pt).hex() == "3925841d02dc09fbdc118597196a0b32"


def test_prg_kat(O):
    out = O.Prg(4).gen(S0)
    for i in range(4):
        assert (out[i] == words(K["prg4"][i])).all()


@pytest.mark.parametrize("mul", [1, 2, 4])
def test_prg_matches_openssl_evp(O, mul, rng):
    own, evp = O.Prg(mul), O.Prg(mul, use_evp=True)
    for _ in range(200):
        s = rng.i4(clear_lsb=False)
        assert (own.gen(s) == evp.gen(s)).all()


def test_dpf8_bytes(O):
    k, p = K["dpf8_bytes"], O.Prg(2)
    cws = O.dpf_gen(p, 8, O.BYTES, [S0, S1], k["alpha"], BETA)
    assert (cws[0, :4] == words(k["cw0_s"])).all() and cws[0, 4] == k["cw0_tr"]
    assert (cws[8, :4] == words(k["cw8_s"])).all()
    assert (O.dpf_eval(p, 8, O.BYTES, 0, S0, cws, 42) == words(k["y0_42"])).all()
    assert (O.dpf_eval(p, 8, O.BYTES, 1, S1, cws, 42) == words(k["y1_42"])).all()
    assert (O.dpf_eval(p, 8, O.BYTES, 0, S0, cws, 100) == words(k["y0_100"])).all()
    assert hashlib.sha256(O.dpf_eval_all(p, 8, O.BYTES, 0, S0, cws).tobytes()).hexdigest() == k["evalall_sha256_p0"]
    assert hashlib.sha256(O.dpf_eval_all(p, 8, O.BYTES, 1, S1, cws).tobytes()).hexdigest() == k["evalall_sha256_p1"]


def test_dcf8_bytes(O):
    k, p = K["dcf8_bytes_lt"], O.Prg(4)
    cws = O.dcf_gen(p, 8, O.BYTES, O.LT, [S0, S1], k["alpha"], BETA)
    assert (cws[0, :4] == words(k["cw0_s"])).all() and (cws[0, 4:] == words(k["cw0_v"])).all()
    assert hashlib.sha256(cws.tobytes()).hexdigest() == k["cws_sha256"]
    assert (O.dcf_eval(p, 8, O.BYTES, 0, S0, cws, 10) == words(k["y0_10"])).all()
    assert (O.dcf_eval(p, 8, O.BYTES, 1, S1, cws, 10) == words(k["y1_10"])).all()
    assert hashlib.sha256(O.dcf_eval_all(p, 8, O.BYTES, 0, S0, cws).tobytes()).hexdigest() == k["evalall_sha256_p0"]


def test_dpf32_u64(O):
    k, p = K["dpf32_u64"], O.Prg(2)
    cws = O.dpf_gen(p, 32, O.U64, [S0, S1], k["alpha"], BETA)
    assert (cws[32, :4] == words(k["cw32_s"])).all()
    assert (O.dpf_eval(p, 32, O.U64, 0, S0, cws, k["alpha"]) == words(k["y0_alpha"])).all()
    assert (O.dpf_eval(p, 32, O.U64, 1, S1, cws, k["alpha"]) == words(k["y1_alpha"])).all()
    assert (O.dpf_eval(p, 32, O.U64, 0, S0, cws, k["x_other"]) == words(k["y0_other"])).all()


def test_ht64_u64(O):
    k, p = K["ht64_u64"], O.Prg(1)
    cws, ocw = O.ht_gen(p, 64, O.U64, [S0, S1], k["alpha"], BETA)
    assert (cws[0, :4] == words(k["cw0_s"])).all() and cws[63, 4] == k["cw63_extra"]
    assert (ocw == words(k["ocw"])).all()
    assert (O.ht_eval(p, 64, O.U64, 0, S0, cws, ocw, k["alpha"]) == words(k["y0_alpha"])).all()
    assert (O.ht_eval(p, 64, O.U64, 1, S1, cws, ocw, k["alpha"]) == words(k["y1_alpha"])).all()
    assert (O.ht_eval(p, 64, O.U64, 0, S0, cws, ocw, k["x_other"]) == words(k["y0_other"])).all()


# ---------------------------------------------------------------- ChaCha PRG (include/fss/prg/chacha.cuh)
def test_chacha_permutation_rfc7539(O):
    """The permutation core against RFC 7539 section 2.3.2 (block = permute(state) + state)."""
    key = bytes(range(32))
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(np.frombuffer(key, dtype="<u4")) + \
         [1, 0x09000000, 0x4A000000, 0]
    out = O.chacha_permute(st)
    block = (out.astype(np.uint64) + np.array(st, dtype=np.uint64)) & 0xFFFFFFFF
    want = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
            0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]
    assert [int(v) for v in block] == want


def test_chacha_gen_structure(O, rng):
    """Gen as chacha.cuh:95-127 builds it: rows of one permutation of [const | seed | seed | 0,0,nonce]
    with the XOR feed-forward; mul = 1 is row 1 of the 16-byte-constant state, mul = 2 rows 0-1."""
    nonce = (0x12345678, 0x9ABCDEF0)
    for _ in range(20):
        seed = rng.i4(clear_lsb=False)
        sd = [int(v) & 0xFFFFFFFF for v in seed]
        for mul, const in ((1, [0x61707865, 0x3120646E, 0x79622D36, 0x6B206574]),
                           (2, [0x61707865, 0x3120646E, 0x79622D36, 0x6B206574]),
                           (4, [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574])):
            st = const + sd + sd + [0, 0, nonce[0], nonce[1]]
            x = [int(v) for v in O.chacha_permute(st)]
            rows = [[x[j] ^ const[j] for j in range(4)], [x[4 + j] ^ sd[j] for j in range(4)],
                    [x[8 + j] ^ sd[j] for j in range(4)], [x[12], x[13], x[14] ^ nonce[0], x[15] ^ nonce[1]]]
            want = {1: [rows[1]], 2: rows[:2], 4: rows}[mul]
            got = O.Prg(mul, chacha=True, nonce=nonce).gen(seed).view(np.uint32)
            assert [[int(v) for v in r] for r in got] == want


# ---------------------------------------------------------------- BLAKE3 (VDPF hashes)
def test_blake3_compress_official_vectors(O):
    """The compression function against the official BLAKE3 test vectors (BLAKE3 test_vectors.json: `hash`
    of the empty input and of one zero byte; `keyed_hash` of the empty input with the key
    "whats the Elvish word for friend") -- single-block inputs are one compression with
    CHUNK_START | CHUNK_END | ROOT (| KEYED_HASH)."""
    iv = O.BLAKE3_IV
    z = np.zeros(16, dtype=np.uint32)
    assert O.blake3_compress(iv, z, 0, 0, 1 | 2 | 8)[:8].tobytes().hex() == \
        "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert O.blake3_compress(iv, z, 0, 1, 1 | 2 | 8)[:8].tobytes().hex() == \
        "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213"
    key = np.frombuffer(b"whats the Elvish word for friend", dtype=np.uint32)
    assert O.blake3_compress(key, z, 0, 0, 1 | 2 | 8 | 16)[:8].tobytes().hex() == \
        "92b2b75604ed3c761f9d6f62392c8a9227ad0ea3f09573e783f1498a4ed60d26"


def test_blake3_hash_forms(O):
    """fss::hash::Blake3's two Hash overloads are keyed single-block compressions (blake3.cuh:151-169)."""
    iv = O.HASH_IV
    a = np.array([0x01020305, 7, 8, 9], dtype=np.int32)
    b = np.array([11, 12, 13, 14], dtype=np.int32)
    h4 = O.blake3_xorhash(iv, a, b)
    for half, lsb in ((0, 0), (1, 1)):
        m = np.zeros(16, dtype=np.uint32)
        m[:4] = a.view(np.uint32)
        m[3] = (m[3] & ~np.uint32(1)) | np.uint32(lsb)
        m[4:8] = b.view(np.uint32)
        want = O.blake3_compress(iv.view(np.uint32), m, 0, 32, 1 | 2 | 8 | 16)[:8]
        assert (h4[2 * half:2 * half + 2].reshape(-1).view(np.uint32) == want).all()
    msg = np.arange(16, dtype=np.int32).reshape(4, 4)
    h2 = O.blake3_hash(iv, msg)
    want = O.blake3_compress(iv.view(np.uint32), msg.reshape(-1).view(np.uint32), 0, 64, 1 | 2 | 8 | 16)[:8]
    assert (h2.reshape(-1).view(np.uint32) == want).all()


def test_aesni_prg_equals_evp_and_table_aes(O, rng):
    """The AES-NI PRG (the reference's Aes128MmoRaw, prg/aes128_mmo_raw.cuh:62-87; the baseline's "best CPU"
    line) is the same function as the OpenSSL and table paths."""
    if not O.have_aesni():
        pytest.skip("no AES-NI on this CPU")
    for mul in (1, 2, 4):
        a, b, c = O.Prg(mul), O.Prg(mul, use_evp=True), O.Prg(mul, aesni=True)
        for _ in range(50):
            s = rng.i4(clear_lsb=False)
            assert (a.gen(s) == c.gen(s)).all() and (b.gen(s) == c.gen(s)).all()


def test_sha256_against_openssl_and_nist(O):
    """The oracle's SHA-256 (the reference's fss::hash::Sha256 = OpenSSL EVP_Digest over key | message,
    hash/sha256.cuh:15-62) against hashlib (OpenSSL) at every padding edge, the FIPS 180-4 "abc" vector, and the two
    Hash overloads as digests of the concatenations the reference builds."""
    import hashlib
    assert O.sha256(b"abc").hex() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"
    for n in (0, 1, 47, 48, 55, 56, 63, 64, 79, 80, 119, 120, 200):
        m = bytes((7 * i + n) & 0xFF for i in range(n))
        assert O.sha256(m) == hashlib.sha256(m).digest(), n
    key = O.HASH_IV[:4].copy()
    a = np.array([0x01020305, 7, 8, 9], dtype=np.int32)
    b = np.array([11, 12, 13, 14], dtype=np.int32)
    O.set_hash_kind(1)
    try:
        a0, a1 = a.copy(), a.copy()
        a0[3] &= ~1
        a1[3] |= 1
        want = (hashlib.sha256(key.tobytes() + a0.tobytes() + b.tobytes()).digest()
                + hashlib.sha256(key.tobytes() + a1.tobytes() + b.tobytes()).digest())
        assert O.blake3_xorhash(O.HASH_IV, a, b).tobytes() == want
        msg = np.arange(16, dtype=np.int32).reshape(4, 4)
        assert O.blake3_hash(O.HASH_IV, msg).tobytes() == hashlib.sha256(key.tobytes() + msg.tobytes()).digest()
    finally:
        O.set_hash_kind(0)
