"""Reference group arithmetic on int32[...,4] buffers, for share reconstruction in tests
(include/fss/group/bytes.cuh:19-43, include/fss/group/uint.cuh:27-88)."""
import numpy as np

BYTES, U8, U16, U32, U64, U128 = 0, 1, 2, 3, 4, 5
_BITS = {U8: 8, U16: 16, U32: 32, U64: 64}


def to_int(g, buf):
    w = [int(v) & 0xFFFFFFFF for v in buf]
    if g == BYTES:
        return w[0] | w[1] << 32 | w[2] << 64 | w[3] << 96
    if g == U128:
        return (w[0] | w[1] << 32 | w[2] << 64 | (w[3] >> 1) << 96) % (1 << 127)
    return (w[0] | w[1] << 32) & ((1 << _BITS[g]) - 1)


def from_int(g, v):
    if g == U128:
        w = [v & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF, (v >> 64) & 0xFFFFFFFF, ((v >> 96) << 1) & 0xFFFFFFFF]
    elif g == BYTES:
        w = [(v >> (32 * i)) & 0xFFFFFFFF for i in range(4)]
    else:
        w = [v & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF, 0, 0]
    return np.array(w, dtype=np.uint32).view(np.int32)


def add(g, a, b):
    """a + b of two int4 buffers in group g, as an int4 buffer."""
    x, y = to_int(g, a), to_int(g, b)
    if g == BYTES:
        return from_int(g, x ^ y)
    if g == U128:
        return from_int(g, (x + y) % (1 << 127))
    return from_int(g, (x + y) & ((1 << _BITS[g]) - 1))


def add_rows(g, A, B):
    """Vectorised a + b for int32[n,4] arrays (Bytes, U8..U64 only)."""
    A, B = np.ascontiguousarray(A), np.ascontiguousarray(B)
    if g == BYTES:
        return A ^ B
    a = A.view(np.uint32).astype(np.uint64)
    b = B.view(np.uint32).astype(np.uint64)
    s = (a[:, 0] | (a[:, 1] << np.uint64(32))) + (b[:, 0] | (b[:, 1] << np.uint64(32)))
    if g != U64:
        s &= np.uint64((1 << _BITS[g]) - 1)
    out = np.zeros_like(A)
    out[:, 0] = (s & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
    out[:, 1] = (s >> np.uint64(32)).astype(np.uint32).view(np.int32)
    return out


def canon(g, buf):
    """Group element of a clamped 16-byte buffer, as the canonical int4 (From then Into)."""
    return from_int(g, to_int(g, buf))
