#!/usr/bin/env python3
"""Generator for tools/ubench/aes_bitslice.hip: table-free (bit-sliced) AES-128 rounds as straight-line
v_bitop3_b32 / v_xor / v_and code, to price the formulation that moves the S-box off the LDS pipe.

State layout A ("wide", 32 blocks per lane): plane[b][i] = one VGPR holding bit i of state byte b of the lane's 32
blocks (bit j of the register = block j); 128 state registers.
State layout B ("packed", 16 blocks per lane): register [row][q][i] holds bit i of the bytes in columns q (low
16 bits) and q + 2 (high 16 bits) of `row`, 16 blocks each; 64 state registers, ShiftRows costs 32 half-word
rotates per round (v_alignbit), everything else is the same bitwise netlist on half as many registers.

The round is built as a netlist of 2-input gates (S-box: Boyar-Peralta's depth-16 circuit, checked below against the
S-box table; MixColumns and AddRoundKey as XORs), fused greedily into 3-input look-up tables (one v_bitop3_b32 each),
checked gate-for-gate in Python against a byte-wise AES, and emitted as one device function per round shape."""
import sys

# ---- Boyar-Peralta S-box, U0 = msb .. U7 = lsb, S0 = msb ('+' xor, 'x' and, '#' xnor)
BP = """T1=U0+U3 T2=U0+U5 T3=U0+U6 T4=U3+U5 T5=U4+U6 T6=T1+T5 T7=U1+U2 T8=U7+T6 T9=U7+T7 T10=T6+T7 T11=U1+U5 T12=U2+U5
T13=T3+T4 T14=T6+T11 T15=T5+T11 T16=T5+T12 T17=T9+T16 T18=U3+U7 T19=T7+T18 T20=T1+T19 T21=U6+U7 T22=T7+T21 T23=T2+T22
T24=T2+T10 T25=T20+T17 T26=T3+T16 T27=T1+T12
M1=T13xT6 M2=T23xT8 M3=T14+M1 M4=T19xU7 M5=M4+M1 M6=T3xT16 M7=T22xT9 M8=T26+M6 M9=T20xT17 M10=M9+M6 M11=T1xT15
M12=T4xT27 M13=M12+M11 M14=T2xT10 M15=M14+M11 M16=M3+M2 M17=M5+T24 M18=M8+M7 M19=M10+M15 M20=M16+M13 M21=M17+M15
M22=M18+M13 M23=M19+T25 M24=M22+M23 M25=M22xM20 M26=M21+M25 M27=M20+M21 M28=M23+M25 M29=M28xM27 M30=M26xM24
M31=M20xM23 M32=M27xM31 M33=M27+M25 M34=M21xM22 M35=M24xM34 M36=M24+M25 M37=M21+M29 M38=M32+M33 M39=M23+M30
M40=M35+M36 M41=M38+M40 M42=M37+M39 M43=M37+M38 M44=M39+M40 M45=M42+M41 M46=M44xT6 M47=M40xT8 M48=M39xU7
M49=M43xT16 M50=M38xT9 M51=M37xT17 M52=M42xT15 M53=M45xT27 M54=M41xT10 M55=M44xT13 M56=M40xT23 M57=M39xT19
M58=M43xT3 M59=M38xT22 M60=M37xT20 M61=M42xT1 M62=M45xT4 M63=M41xT2
L0=M61+M62 L1=M50+M56 L2=M46+M48 L3=M47+M55 L4=M54+M58 L5=M49+M61 L6=M62+L5 L7=M46+L3 L8=M51+M59 L9=M52+M53
L10=M53+L4 L11=M60+L2 L12=M48+M51 L13=M50+L0 L14=M52+M61 L15=M55+L1 L16=M56+L0 L17=M57+L1 L18=M58+L8 L19=M63+L4
L20=L0+L1 L21=L1+L7 L22=L3+L12 L23=L18+L2 L24=L15+L9 L25=L6+L10 L26=L7+L9 L27=L8+L10 L28=L11+L14 L29=L11+L17
S0=L6+L24 S1=L16#L26 S2=L19#L28 S3=L6+L21 S4=L20+L22 S5=L25+L29 S6=L13#L27 S7=L6#L23"""


def bp_gates():
    out = []
    for tok in BP.split():
        dst, rhs = tok.split("=")
        for op in "+x#":
            if op in rhs:
                a, b = rhs.split(op)
                out.append((dst, op, a, b))
                break
    return out


def sbox_table():
    p = q = 1
    s = [0] * 256
    while True:
        p = p ^ ((p << 1) & 0xFF) ^ (0x1B if p & 0x80 else 0)
        q ^= (q << 1) & 0xFF
        q ^= (q << 2) & 0xFF
        q ^= (q << 4) & 0xFF
        if q & 0x80:
            q ^= 0x09
        x = q ^ ((q << 1 | q >> 7) & 0xFF) ^ ((q << 2 | q >> 6) & 0xFF) ^ ((q << 3 | q >> 5) & 0xFF) ^ ((q << 4 | q >> 4) & 0xFF)
        s[p] = x ^ 0x63
        if p == 1:
            break
    s[0] = 0x63
    return s


# ---- netlist: nodes are ("in", name) | ("const", 0/1) | (op, a, b) with op in {"xor", "and", "xnor"}
class Net:
    def __init__(self):
        self.nodes = []      # (op, a, b) by id
        self.cse = {}

    def inp(self, name):
        self.nodes.append(("in", name, None))
        return len(self.nodes) - 1

    def gate(self, op, a, b):
        if op != "and" or True:
            if a > b:
                a, b = b, a
        key = (op, a, b)
        if key in self.cse:
            return self.cse[key]
        self.nodes.append(key)
        self.cse[key] = len(self.nodes) - 1
        return len(self.nodes) - 1

    def xor(self, a, b):
        return self.gate("xor", a, b)

    def rot(self, a):
        """Half-word rotate of a packed register (its own instruction: bitwise tables cannot move bits)."""
        key = ("rot", a, None)
        if key not in self.cse:
            self.nodes.append(key)
            self.cse[key] = len(self.nodes) - 1
        return self.cse[key]


def sbox_net(net, bits):
    """bits[i] = node of bit i (0 = lsb) -> list of 8 output nodes."""
    v = {f"U{k}": bits[7 - k] for k in range(8)}
    for dst, op, a, b in bp_gates():
        v[dst] = net.gate({"+": "xor", "x": "and", "#": "xnor"}[op], v[a], v[b])
    return [v[f"S{7 - i}"] for i in range(8)]


def xtime_bits(net, v):
    """bits of 2*v in GF(2^8): b0 = v7, b1 = v0^v7, b2 = v1, b3 = v2^v7, b4 = v3^v7, b5 = v4, b6 = v5, b7 = v6."""
    return [v[7], net.xor(v[0], v[7]), v[1], net.xor(v[2], v[7]), net.xor(v[3], v[7]), v[4], v[5], v[6]]


def round_net(last, with_key, packed=False):
    """One AES round on bit planes.  Wide: 16 byte positions b = 4 * col + row, a register per (b, bit).  Packed: 8
    register-bytes b = 4 * q + row holding columns q (low half-word) and q + 2 (high half-word); ShiftRows then needs
    a half-word rotate ("rot" node) wherever the two columns swap places.  Inputs s[b][i], key planes k[b][i] when
    with_key.  Output: nodes after SubBytes, ShiftRows, [MixColumns], AddRoundKey."""
    net = Net()
    nb = 8 if packed else 16
    ncol = nb // 4
    s = [[net.inp(f"s{b}_{i}") for i in range(8)] for b in range(nb)]
    k = [[net.inp(f"k{b}_{i}") for i in range(8)] for b in range(nb)] if with_key else None
    sb = [sbox_net(net, s[b]) for b in range(nb)]
    sr = [None] * nb
    for c in range(ncol):
        for r in range(4):
            if not packed:
                sr[4 * c + r] = sb[4 * ((c + r) % 4) + r]
            else:
                # new {lo: old col m, hi: old col m + 2} with m = (c + r) % 4: register-byte m % 2, halves swapped for m >= 2
                m = (c + r) % 4
                src = sb[4 * (m % 2) + r]
                sr[4 * c + r] = [net.rot(x) for x in src] if m >= 2 else src
    out = [None] * nb
    for c in range(ncol):
        a = [sr[4 * c + r] for r in range(4)]
        if last:
            mc = a
        else:
            mc = []
            for r in range(4):
                t = [net.xor(a[r][i], a[(r + 1) % 4][i]) for i in range(8)]      # a_r ^ a_{r+1}
                x2 = xtime_bits(net, t)                                            # 2 (a_r ^ a_{r+1})
                rest = [net.xor(a[(r + 2) % 4][i], a[(r + 3) % 4][i]) for i in range(8)]
                # out_r = 2 a_r ^ 3 a_{r+1} ^ a_{r+2} ^ a_{r+3} = 2 (a_r ^ a_{r+1}) ^ a_{r+1} ^ a_{r+2} ^ a_{r+3}
                mc.append([net.xor(net.xor(x2[i], a[(r + 1) % 4][i]), rest[i]) for i in range(8)])
        for r in range(4):
            out[4 * c + r] = [net.xor(mc[r][i], k[4 * c + r][i]) if with_key else mc[r][i] for i in range(8)]
    return net, s, k, out


# ---- greedy fusion into 3-input LUTs
def truth(op):
    return {"xor": lambda a, b: a ^ b, "and": lambda a, b: a & b, "xnor": lambda a, b: 1 ^ a ^ b}[op]


def fuse(net, outputs):
    """Returns a list of LUT nodes [(id, inputs (<=3 node ids), imm8 or op)] in topological order.  A gate is merged
    into its consumer when it has exactly one use, is not an output, and the merged support stays <= 3 inputs."""
    uses = {}
    outs = set(o for byte in outputs for o in byte)
    for i, (op, a, b) in enumerate(net.nodes):
        if op == "rot":
            uses[a] = uses.get(a, 0) + 2   # never merged into the rotate
        elif op not in ("in",):
            uses[a] = uses.get(a, 0) + 1
            uses[b] = uses.get(b, 0) + 1
    # each live gate: support (tuple of node ids) and function as dict over support assignments
    fn = {}
    rots = {}
    for i, (op, a, b) in enumerate(net.nodes):
        if op == "in":
            continue
        if op == "rot":
            rots[i] = a
            continue
        sup = [a, b]
        f = lambda env, op=op, a=a, b=b: truth(op)(env[a], env[b])
        # try to inline single-use gate operands (prefer the one that keeps support small), repeatedly
        changed = True
        while changed:
            changed = False
            for x in list(sup):
                if x in fn and uses.get(x, 0) == 1 and x not in outs:
                    xs, xf = fn[x]
                    new_sup = [y for y in sup if y != x] + [y for y in xs if y not in sup]
                    new_sup = list(dict.fromkeys(new_sup))
                    if len(new_sup) <= 3:
                        f = (lambda env, f=f, x=x, xf=xf: f({**env, x: xf(env)}))
                        sup = new_sup
                        fn.pop(x)
                        changed = True
                        break
        fn[i] = (sup, f)
    luts = []
    for i in sorted(list(fn) + list(rots)):
        if i in rots:
            luts.append((i, [rots[i]], "rot"))
            continue
        sup, f = fn[i]
        sup = list(sup)
        imm = 0
        n = len(sup)
        padded = sup + [sup[-1]] * (3 - n)
        for a in (0, 1):
            for b in (0, 1):
                for c in (0, 1):
                    env = {padded[0]: a, padded[1]: b, padded[2]: c}
                    if n < 3 and ((n == 2 and c != b) or (n == 1 and (b != a or c != a))):
                        # inconsistent assignment of a duplicated input: copy the consistent one (value irrelevant)
                        env = {padded[0]: a, padded[1]: b if n >= 2 else a, padded[2]: (b if n == 2 else a)}
                    if f(env):
                        imm |= 1 << ((a << 2) | (b << 1) | c)   # index = a*4 + b*2 + c with a = 0xF0, b = 0xCC, c = 0xAA
        luts.append((i, sup, imm))
    return luts


def eval_luts(net, luts, values):
    """values: dict input node id -> int bit-vector; returns dict node -> value (python ints as 32-bit planes)."""
    v = dict(values)
    M = 0xFFFFFFFF
    for i, sup, imm in luts:
        if imm == "rot":
            x = v[sup[0]]
            v[i] = ((x << 16) | (x >> 16)) & M
            continue
        s3 = sup + [sup[-1]] * (3 - len(sup))
        a, b, c = (v[x] for x in s3)
        r = 0
        for idx in range(8):
            if imm >> idx & 1:
                ta = a if idx & 4 else ~a & M
                tb = b if idx & 2 else ~b & M
                tc = c if idx & 1 else ~c & M
                r |= ta & tb & tc
        v[i] = r & M
    return v


# ---- byte-wise reference for the self-check
def ref_round(state, key, last):
    S = sbox_table()
    sb = [S[x] for x in state]
    sr = [0] * 16
    for c in range(4):
        for r in range(4):
            sr[4 * c + r] = sb[4 * ((c + r) % 4) + r]
    xt = lambda x: ((x << 1) & 0xFF) ^ (0x1B if x & 0x80 else 0)
    out = [0] * 16
    for c in range(4):
        a = sr[4 * c:4 * c + 4]
        for r in range(4):
            if last:
                m = a[r]
            else:
                m = xt(a[r]) ^ (xt(a[(r + 1) % 4]) ^ a[(r + 1) % 4]) ^ a[(r + 2) % 4] ^ a[(r + 3) % 4]
            out[4 * c + r] = m ^ key[4 * c + r]
    return out


def self_check():
    import random
    rnd = random.Random(1)
    S = sbox_table()
    # S-box circuit alone
    net = Net()
    ins = [net.inp(f"u{i}") for i in range(8)]
    outs = sbox_net(net, ins)
    for x in range(256):
        v = {ins[i]: (x >> i) & 1 for i in range(8)}
        for i, (op, a, b) in enumerate(net.nodes):
            if op != "in":
                v[i] = truth(op)(v[a], v[b])
        assert sum(v[outs[i]] << i for i in range(8)) == S[x], x
    for last in (False, True):
        net, s, k, out = round_net(last, True)
        luts = fuse(net, out)
        blocks = [[rnd.randrange(256) for _ in range(16)] for _ in range(32)]
        key = [rnd.randrange(256) for _ in range(16)]
        vals = {}
        for b in range(16):
            for i in range(8):
                vals[s[b][i]] = sum(((blocks[j][b] >> i) & 1) << j for j in range(32))
                vals[k[b][i]] = 0xFFFFFFFF if (key[b] >> i) & 1 else 0
        v = eval_luts(net, luts, vals)
        for j in range(32):
            want = ref_round(blocks[j], key, last)
            got = [sum(((v[out[b][i]] >> j) & 1) << i for i in range(8)) for b in range(16)]
            assert got == want, (last, j)
        # packed: register-byte 4 q + r = columns q (bits 0..15) and q + 2 (bits 16..31), 16 blocks
        net, s, k, out = round_net(last, True, packed=True)
        luts = fuse(net, out)
        vals = {}
        pk = lambda f: {(q, r, i): sum(f(j, 4 * q + r, i) << j for j in range(16)) | sum(f(j, 4 * (q + 2) + r, i) << (16 + j) for j in range(16))
                        for q in range(2) for r in range(4) for i in range(8)}
        sv = pk(lambda j, b, i: (blocks[j][b] >> i) & 1)
        kv = pk(lambda j, b, i: (key[b] >> i) & 1)
        for q in range(2):
            for r in range(4):
                for i in range(8):
                    vals[s[4 * q + r][i]] = sv[(q, r, i)]
                    vals[k[4 * q + r][i]] = kv[(q, r, i)]
        v = eval_luts(net, luts, vals)
        for j in range(16):
            want = ref_round(blocks[j], key, last)
            got = [0] * 16
            for q in range(2):
                for r in range(4):
                    for i in range(8):
                        x = v[out[4 * q + r][i]]
                        got[4 * q + r] |= ((x >> j) & 1) << i
                        got[4 * (q + 2) + r] |= ((x >> (16 + j)) & 1) << i
            assert got == want, ("packed", last, j)
    return True


def emit_round(name, last, with_key, packed=False):
    net, s, k, out = round_net(last, with_key, packed)
    luts = fuse(net, out)
    nb = 8 if packed else 16
    lines = [f"// {len(luts)} instructions ({sum(1 for l in luts if len(l[1]) == 3)} v_bitop3, "
             f"{sum(1 for l in luts if l[2] == 'rot')} half-word rotates, rest 2-input) for {nb} register-bytes x 8 planes"]
    args = f"uint32_t (&p)[{nb}][8]" + (f", const uint32_t (&k)[{nb}][8]" if with_key else "")
    lines.append(f"__device__ __forceinline__ void {name}({args}) {{")
    names = {}
    for b in range(nb):
        for i in range(8):
            names[s[b][i]] = f"p[{b}][{i}]"
            if with_key:
                names[k[b][i]] = f"k[{b}][{i}]"
    for i, sup, imm in luts:
        names[i] = f"t{i}"
        a = [names[x] for x in sup]
        if imm == "rot":
            lines.append(f"  const uint32_t t{i} = __builtin_amdgcn_alignbit({a[0]}, {a[0]}, 16);")
        elif len(sup) == 3:
            lines.append(f"  const uint32_t t{i} = __builtin_amdgcn_bitop3_b32({a[0]}, {a[1]}, {a[2]}, 0x{imm:02x});")
        elif len(sup) == 2:
            # truth table over (a, b) with c = b: pick plain operators where they exist
            tt = tuple((imm >> ((x << 2) | (y << 1) | y)) & 1 for x in (0, 1) for y in (0, 1))
            expr = {(0, 1, 1, 0): f"{a[0]} ^ {a[1]}", (0, 0, 0, 1): f"{a[0]} & {a[1]}", (1, 0, 0, 1): f"~({a[0]} ^ {a[1]})",
                    (0, 1, 1, 1): f"{a[0]} | {a[1]}"}.get(tt)
            if expr is None:
                expr = f"__builtin_amdgcn_bitop3_b32({a[0]}, {a[1]}, {a[1]}, 0x{imm:02x})"
            lines.append(f"  const uint32_t t{i} = {expr};")
        else:
            lines.append(f"  const uint32_t t{i} = ~{a[0]};")
    for b in range(nb):
        for i in range(8):
            lines.append(f"  p[{b}][{i}] = {names[out[b][i]]};")
    lines.append("}")
    return "\n".join(lines), len(luts)


if __name__ == "__main__":
    assert self_check()
    body = []
    counts = {}
    for nm, last, wk, pk in (("bs_round", False, True, False), ("bs_round_last", True, True, False),
                             ("bs_round_nokey", False, False, False), ("bsp_round", False, True, True),
                             ("bsp_round_last", True, True, True), ("bsp_round_nokey", False, False, True)):
        code, n = emit_round(nm, last, wk, pk)
        body.append(code)
        counts[nm] = n
    sys.stdout.write("// generated by tools/ubench/aes_bitslice_gen.py -- do not edit\n")
    sys.stdout.write(f"// instructions per round: {counts}\n")
    sys.stdout.write("\n\n".join(body) + "\n")
