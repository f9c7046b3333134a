"""Error behaviour of the C ABI on a GPU box: bad arguments come back as FSS_AMD_EINVAL (-1) before any
kernel is launched; the reference has asserts only (dpf.cuh:209,237), so these are the library's own contract
(include/fss_amd.h)."""
import ctypes as C

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

EINVAL = -1


@pytest.fixture(scope="module")
def raw():
    import fss_amd
    from fss_amd import _lib
    eng = fss_amd.default_engine(0)
    bufs = {k: torch.zeros(4096, dtype=torch.int32, device="cuda") for k in ("seed", "cws", "xs", "ys", "ws", "ocw")}
    return _lib.lib(), eng._h, {k: C.c_void_p(v.data_ptr()) for k, v in bufs.items()}, bufs


def test_point_eval_argument_errors(raw):
    L, h, p, _ = raw
    ok = lambda **kw: L.fss_amd_dpf_eval_points(kw.get("ctx", h), kw.get("n", 32), kw.get("g", 4), kw.get("party", 0),
                                                kw.get("seed", p["seed"]), kw.get("cws", p["cws"]), kw.get("xs", p["xs"]),
                                                kw.get("xb", 4), kw.get("cnt", 8), kw.get("ys", p["ys"]), None)
    assert ok() == 0
    assert ok(cnt=0) == 0  # empty batch is fine
    for bad in (dict(ctx=None), dict(n=0), dict(n=33), dict(n=129, xb=16), dict(g=6), dict(g=-1), dict(party=2),
                dict(xb=3), dict(xb=2), dict(cnt=-1), dict(seed=None), dict(cws=None), dict(xs=None), dict(ys=None)):
        assert ok(**bad) == EINVAL, bad
    assert ok(n=64, xb=8) == 0 and ok(n=65, xb=8) == EINVAL and ok(n=128, xb=16) == 0
    # Half-Tree needs its output correction word
    assert L.fss_amd_ht_eval_points(h, 32, 4, 0, p["seed"], p["cws"], None, p["xs"], 4, 8, p["ys"], None) == EINVAL


def test_eval_all_argument_errors(raw):
    L, h, p, bufs = raw
    L.fss_amd_eval_all_workspace.restype = C.c_size_t
    need = L.fss_amd_eval_all_workspace(0, 12, 256)
    assert 0 < need <= bufs["ws"].numel() * 4
    call = lambda first, n, ws=need, in_bits=12, ys=p["ys"]: L.fss_amd_dpf_eval_all(
        h, in_bits, 4, 0, p["seed"], p["cws"], first, n, ys, p["ws"], ws, None)
    assert call(0, 256) == 0 and call(256, 256) == 0 and call(4096 - 256, 256) == 0
    assert call(128, 256) == EINVAL       # block not aligned to its size
    assert call(0, 255) == EINVAL         # not a power of two
    assert call(4096, 256) == EINVAL      # outside the domain
    assert call(0, 256, ws=need - 16) == EINVAL
    assert call(0, 256, in_bits=63) == EINVAL
    assert call(0, 256, ys=C.c_void_p(bufs["ys"].data_ptr() + 4)) == EINVAL  # misaligned output


def test_misaligned_pointers_are_einval(raw):
    """int4 arrays are 16-byte accessed, x arrays sizeof(In)-accessed (include/fss_amd.h, Conventions): a
    misaligned pointer is refused on the host instead of faulting on the device."""
    L, h, p, bufs = raw
    off = lambda k, b: C.c_void_p(bufs[k].data_ptr() + b)
    pts = lambda **kw: L.fss_amd_dpf_eval_points(h, 32, 4, 0, kw.get("seed", p["seed"]), kw.get("cws", p["cws"]),
                                                 kw.get("xs", p["xs"]), kw.get("xb", 4), 8, kw.get("ys", p["ys"]), None)
    assert pts(seed=off("seed", 16), cws=off("cws", 32), xs=off("xs", 4), ys=off("ys", 16)) == 0
    for bad in (dict(seed=off("seed", 4)), dict(cws=off("cws", 8)), dict(ys=off("ys", 8)), dict(xs=off("xs", 2)),
                dict(xs=off("xs", 4), xb=8)):
        assert pts(**bad) == EINVAL, bad
    keys = lambda **kw: L.fss_amd_dpf_eval_keys(h, 32, 4, 0, kw.get("seeds", p["seed"]), kw.get("cw_s", p["cws"]),
                                                kw.get("tr", p["xs"]), kw.get("ocw", p["ocw"]), p["xs"], 4, 8,
                                                kw.get("ys", p["ys"]), None)
    assert keys() == 0
    for bad in (dict(seeds=off("seed", 8)), dict(cw_s=off("cws", 4)), dict(tr=off("xs", 4)), dict(ocw=off("ocw", 4)),
                dict(ys=off("ys", 12))):
        assert keys(**bad) == EINVAL, bad
    assert L.fss_amd_dpf_gen(h, 32, 4, p["seed"], p["xs"], 4, p["ocw"], 4, off("cws", 8), None) == EINVAL
    assert L.fss_amd_prg_gen(h, 2, off("seed", 4), 4, p["ys"], None) == EINVAL
    assert L.fss_amd_dpf_relayout(h, 32, p["cws"], 4, p["ws"], off("xs", 4), p["ocw"], None) == EINVAL


def test_keys_gen_and_misc_argument_errors(raw):
    L, h, p, _ = raw
    assert L.fss_amd_dpf_eval_keys(h, 32, 4, 0, p["seed"], p["cws"], p["xs"], p["ocw"], p["xs"], 4, 8, p["ys"], None) == 0
    assert L.fss_amd_dpf_eval_keys(h, 32, 4, 0, p["seed"], p["cws"], None, p["ocw"], p["xs"], 4, 8, p["ys"], None) == EINVAL
    assert L.fss_amd_dcf_gen(h, 32, 4, 2, p["seed"], p["xs"], 4, p["ocw"], 4, p["cws"], None) == EINVAL  # pred
    assert L.fss_amd_dcf_gen(h, 32, 4, 1, p["seed"], p["xs"], 4, None, 4, p["cws"], None) == 0           # beta = 0
    assert L.fss_amd_grotto_eval_keys(h, 65, 0, p["seed"], p["cws"], p["xs"], p["xs"], 16, 4, p["ys"], None) == EINVAL
    assert L.fss_amd_prg_gen(h, 3, p["seed"], 4, p["ys"], None) == EINVAL
    assert L.fss_amd_ctx_set_prg(h, 7, None, 20) == EINVAL
    assert L.fss_amd_ctx_set_prg(h, 1, None, 21) == EINVAL  # odd round count
    assert L.fss_amd_ctx_set_prg(h, 0, None, 0) == 0
    assert L.fss_amd_ctx_set_hash_iv(h, None) == EINVAL
    L.fss_amd_strerror.restype = C.c_char_p
    assert L.fss_amd_strerror(EINVAL) == b"invalid argument"
    bad = C.c_void_p()
    assert L.fss_amd_ctx_create(C.byref(bad), 99, C.c_char_p(b"\0" * 64), None) == EINVAL and not bad.value


def test_python_layer_raises(raw):
    import fss_amd
    d = fss_amd.Dpf(in_bits=16)
    s0 = torch.zeros(4, dtype=torch.int32)
    cws = torch.zeros((17, 8), dtype=torch.int32)
    with pytest.raises(fss_amd._lib.FssAmdError, match="invalid argument"):
        d.engine.eval_all(fss_amd.DPF, 16, fss_amd.U64, 0, s0, cws, first_leaf=3, nleaves=8)
    with pytest.raises(ValueError):
        d.eval(0, s0, cws, 1 << 16)
    with pytest.raises(ValueError):
        d.eval(2, s0, cws, 1)


@pytest.mark.parametrize("env", [{"FSS_AMD_DYN": "1"}, {"FSS_AMD_DYN": "0", "FSS_AMD_ROTPRIO": "0", "FSS_AMD_QUAD_POINTS_MAX": "0"},
                                 {"FSS_AMD_ILP": "1", "FSS_AMD_BS": "256", "FSS_AMD_FRONTIER_TAIL": "0"}, {"FSS_AMD_LEAF_E": "1", "FSS_AMD_TOPMAX": "3"},
                                 {"FSS_AMD_MID_E": "2", "FSS_AMD_GROTTO_DEFER": "0", "FSS_AMD_RELAYOUT": "0", "FSS_AMD_QUAD_FRONTIER": "0"},
                                 # round 2: the chain of launches instead of the one-launch EvalAll; its stages on the chunk
                                 # counter; rotating instead of ranked priorities, ranked ones under the dynamic queue
                                 {"FSS_AMD_FUSED": "0", "FSS_AMD_STATIC_PRIO": "1", "FSS_AMD_DYN_MIN": "1"},
                                 {"FSS_AMD_FUSED_RANK": "0", "FSS_AMD_FUSED_CHACHA": "0", "FSS_AMD_ROTPRIO": "2", "FSS_AMD_PRIO_EVERY": "0",
                                  "FSS_AMD_LEAF_E": "2", "FSS_AMD_FUSED_GRID": "7"},
                                 # round 3: "two, then one" forced onto the small oracle-checked batches -- the dynamic split
                                 # with a large one-per-lane share, the fixed-share split, and two per lane throughout
                                 {"FSS_AMD_ILP": "2", "FSS_AMD_DYN": "1", "FSS_AMD_MIX_MID_PCT": "30", "FSS_AMD_QUAD_POINTS_MAX": "0"},
                                 {"FSS_AMD_ILP": "2", "FSS_AMD_DYN": "0", "FSS_AMD_MIX_TAIL_UNITS": "1", "FSS_AMD_QUAD_POINTS_MAX": "0",
                                  "FSS_AMD_TAIL_MINWORK": "1"},
                                 {"FSS_AMD_ILP": "2", "FSS_AMD_MIX_MID_PCT": "0", "FSS_AMD_MIX_TAIL_UNITS": "0", "FSS_AMD_DYN": "1",
                                  "FSS_AMD_QUAD_POINTS_MAX": "0"},
                                 # mid + tail share above 100 % of the batch: the one-per-lane region is the whole batch
                                 # (the split must clamp, never start the body at a negative index)
                                 {"FSS_AMD_ILP": "2", "FSS_AMD_DYN": "1", "FSS_AMD_TAIL_PCT": "100", "FSS_AMD_QUAD_POINTS_MAX": "0",
                                  "FSS_AMD_TAIL_MINWORK": "1"},
                                 {"FSS_AMD_ILP": "2", "FSS_AMD_DYN": "1", "FSS_AMD_TAIL_PCT": "100", "FSS_AMD_MIX_MID_PCT": "90",
                                  "FSS_AMD_QUAD_POINTS_MAX": "0", "FSS_AMD_TAIL_MINWORK": "1"},
                                 # round 5: FSS_AMD_POINTS_AUTO from one point on -- every one-key point evaluation of the
                                 # oracle-checked cases takes the shared-prefix plan (node table, then the walk that requests
                                 # each chunk's start nodes half a chunk ahead) instead of the quad-lane / plain walk
                                 {"FSS_AMD_AUTO_PREFIX_MIN": "1", "FSS_AMD_QUAD_POINTS_MAX": "0"},
                                 {"FSS_AMD_AUTO_PREFIX_MIN": "1", "FSS_AMD_QUAD_POINTS_MAX": "0", "FSS_AMD_ILP": "1", "FSS_AMD_DYN": "1",
                                  "FSS_AMD_AUTO_PREFIX_BACK": "0"},
                                 # ... the node table as a chain of launches, batch EvalAll with per-lane key addressing
                                 {"FSS_AMD_PREFIX_FUSED": "0", "FSS_AMD_BATCH_UNI": "0"}])
def test_parity_under_forced_tunables(env):
    """The launch heuristics (dynamic vs static distribution, lanes per workgroup, evaluations per lane, stage
    depths) pick by batch size, so the small oracle-checked cases would only ever see one of each; run the
    oracle comparisons again with the other choices forced (the tunables are read once per process, and only by a
    process that sets FSS_AMD_TUNE=1)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_parity.py"), os.path.join(root, "tests", "test_gpu_vdpf.py"),
                        os.path.join(root, "tests", "test_gpu_chacha.py"),
                        "-k", "vs_oracle or kats or subblocks or deep_prefix or chacha"],
                       env={**os.environ, "FSS_AMD_TUNE": "1", **env}, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_round5_entry_point_errors(raw):
    """fss_amd_ctx_set_points_plan, fss_amd_points_prefix_bits and the slice functions refuse what they cannot do."""
    L, h, _, _ = raw
    assert L.fss_amd_ctx_set_points_plan(h, 2) == -1 and L.fss_amd_ctx_set_points_plan(None, 0) == -1
    assert L.fss_amd_ctx_set_points_plan(h, 1) == 0 and L.fss_amd_ctx_set_points_plan(h, 0) == 0
    assert L.fss_amd_points_prefix_bits(3, 32, 1 << 24) == 0      # Grotto keys have no one-key point walk
    assert L.fss_amd_points_prefix_bits(0, 65, 1 << 24) == 0      # the node table is indexed by a 64-bit x
    assert L.fss_amd_points_prefix_bits(0, 32, 1 << 24) == 23 and L.fss_amd_points_prefix_bits(1, 32, 1 << 20) == 19
    lo, hi = C.c_int64(), C.c_int64()
    assert L.fss_amd_batch_slice(10, 3, 3, C.byref(lo), C.byref(hi)) == -1 and L.fss_amd_batch_slice(-1, 0, 1, C.byref(lo), C.byref(hi)) == -1
    assert L.fss_amd_batch_slice(10, 2, 3, C.byref(lo), C.byref(hi)) == 0 and (lo.value, hi.value) == (7, 10)
    assert L.fss_amd_domain_block(28, 0, 3, C.byref(lo), C.byref(hi)) == -1 and L.fss_amd_domain_block(2, 0, 8, C.byref(lo), C.byref(hi)) == -1
    assert L.fss_amd_domain_block(63, 0, 1, C.byref(lo), C.byref(hi)) == -1
    assert L.fss_amd_grotto_carries(None, 2, None) == -1
    y = torch.zeros(1000, dtype=torch.uint8, device="cuda")
    assert L.fss_amd_grotto_apply_carry(h, C.c_void_p(y.data_ptr()), 1000, 2, None) == -1
    assert L.fss_amd_grotto_apply_carry(h, C.c_void_p(y.data_ptr() + 3), 997, 1, None) == 0   # unaligned start: byte-exact
    torch.cuda.synchronize()
    assert int(y.sum()) == 997 and int(y[:3].sum()) == 0
    assert L.fss_amd_grotto_apply_carry(h, C.c_void_p(y.data_ptr()), 1000, 0, None) == 0 and int(y.sum()) == 997


def test_engine_refuses_undersized_tensors():
    """The Python layer checks sizes the raw-pointer ABI cannot."""
    import fss_amd
    eng = fss_amd.default_engine(0)
    z = lambda *shape: torch.zeros(shape, dtype=torch.int32, device="cuda")
    xs = z(100)
    with pytest.raises(ValueError, match="cws"):
        eng.eval_points(fss_amd.DPF, 32, fss_amd.U64, 0, z(4), z(32, 8), xs)          # one slot short
    with pytest.raises(ValueError, match="seed"):
        eng.eval_points(fss_amd.DPF, 32, fss_amd.U64, 0, z(3), z(33, 8), xs)
    with pytest.raises(ValueError, match="ocw"):
        eng.eval_points(fss_amd.HT, 32, fss_amd.U64, 0, z(4), z(32, 8), xs)           # Half-Tree without ocw
    with pytest.raises(ValueError, match="out"):
        eng.eval_points(fss_amd.DPF, 32, fss_amd.U64, 0, z(4), z(33, 8), xs, out=z(99, 4))
    with pytest.raises(ValueError, match="relayout"):
        eng.relayout(fss_amd.DCF, 32, z(5, 32, 8))
    keys = eng.relayout(fss_amd.DPF, 32, z(5, 33, 8))
    with pytest.raises(ValueError, match="seeds"):
        eng.eval_keys(keys, fss_amd.U64, 0, z(4, 4), z(5))
    with pytest.raises(ValueError, match="xs"):
        eng.eval_keys(keys, fss_amd.U64, 0, z(5, 4), z(4))
    with pytest.raises(ValueError, match="betas"):
        eng.gen(fss_amd.DPF, 32, fss_amd.U64, z(5, 2, 4), z(5), z(4, 4))
    with pytest.raises(ValueError, match="cws"):
        eng.eval_all(fss_amd.DCF, 10, fss_amd.U64, 0, z(4), z(10, 8))


def test_engine_refuses_unusable_out_tensors():
    """A caller-supplied `out` reaches the kernels as a raw pointer: wrong element size (numel alone would pass),
    wrong device, a strided view or a wrong dtype must be ValueError, never a device heap overflow."""
    import fss_amd
    eng = fss_amd.default_engine(0)
    z = lambda *shape, dt=torch.int32: torch.zeros(shape, dtype=dt, device="cuda")
    xs, seed, cws = z(100), z(4), z(33, 8)
    ev = lambda out: eng.eval_points(fss_amd.DPF, 32, fss_amd.U64, 0, seed, cws, xs, out=out)
    with pytest.raises(ValueError, match="dtype"):
        ev(torch.zeros(4 * 100, dtype=torch.uint8, device="cuda"))       # 400 elements but 400 bytes, not 1600
    with pytest.raises(ValueError, match="must be on"):
        ev(torch.zeros((100, 4), dtype=torch.int32))                     # host memory
    with pytest.raises(ValueError, match="contiguous"):
        ev(z(100, 8)[:, ::2])
    with pytest.raises(ValueError, match="bytes"):
        ev(z(99, 4))
    assert ev(z(100, 4)).shape == (100, 4)
    with pytest.raises(ValueError, match="dtype"):
        eng.eval_points_shared(fss_amd.DPF, 32, fss_amd.U64, 0, seed, cws, xs, out=z(100, 4, dt=torch.int16))
    with pytest.raises(ValueError, match="dtype"):
        eng.eval_all(fss_amd.DPF, 10, fss_amd.U64, 0, seed, z(11, 8), out=z(1024 * 4, dt=torch.uint8))
    with pytest.raises(ValueError, match="dtype"):
        eng.eval_all(fss_amd.GROTTO, 10, fss_amd.BYTES, 0, seed, z(11, 8), out=z(1024))   # Grotto writes bytes
    with pytest.raises(ValueError, match="bytes"):
        eng.eval_all_batch(fss_amd.DPF, 6, fss_amd.U64, 0, z(3, 4), z(3, 7, 8), out=z(3, 63, 4))
    keys = eng.relayout(fss_amd.DPF, 32, z(5, 33, 8))
    with pytest.raises(ValueError, match="dtype"):
        eng.eval_keys(keys, fss_amd.U64, 0, z(5, 4), z(5), out=z(20, dt=torch.int8))
    # (n, 2) int64 pairs are __uint128_t points: the shared-prefix path (in_bits <= 64) refuses them instead of
    # reading 2n 16-byte values from an n-element buffer
    with pytest.raises(ValueError, match="pairs"):
        eng.eval_points_shared(fss_amd.DPF, 32, fss_amd.U64, 0, seed, cws, torch.zeros((50, 2), dtype=torch.int64, device="cuda"))
    # short parity table / short VDPF key material
    with pytest.raises(ValueError, match="pt"):
        eng.grotto_eval_table(10, torch.zeros(2047 - 1, dtype=torch.uint8, device="cuda"), z(5))
    vk = fss_amd.engine.LevelMajorKeys(fss_amd.engine.VDPF, 16, 8, z(16, 8, 4), tr_bits=torch.zeros(8, dtype=torch.int64, device="cuda"))
    with pytest.raises(ValueError, match="cs"):
        eng.vdpf_eval_keys(vk, fss_amd.U64, 0, z(8, 4), z(7, 4, 4), z(8, 4), z(8))
    with pytest.raises(ValueError, match="ocws"):
        eng.vdpf_eval_keys(vk, fss_amd.U64, 0, z(8, 4), z(8, 4, 4), z(7, 4), z(8))
    short = fss_amd.engine.LevelMajorKeys(fss_amd.DPF, 32, 5, z(31, 5, 4), tr_bits=keys.tr_bits, out_cw=keys.out_cw)
    with pytest.raises(ValueError, match="cw_s"):
        eng.key_major(short)


def test_grotto_pathwalk_reconstructs_only_with_itself():
    """The n = 64-capable path walk and the table / EvalAll evaluators give different per-party shares of the same
    predicate (include/fss_amd.h, schemes.GrottoDcf.eval_keys_pathwalk): each reconstructs with itself; mixing
    them across the two parties does not."""
    import numpy as np
    import fss_amd
    n, nkeys = 10, 256
    g = fss_amd.GrottoDcf(n)
    gen = torch.Generator().manual_seed(5)
    s0s = torch.randint(-2**31, 2**31, (nkeys, 2, 4), dtype=torch.int32, generator=gen)
    alphas = torch.randint(0, 1 << n, (nkeys,), dtype=torch.int32, generator=gen)
    xs = torch.randint(0, 1 << n, (nkeys,), dtype=torch.int32, generator=gen)
    cws = g.gen_batch(s0s, alphas)
    keys = g.relayout(cws)
    walk = [g.eval_keys_pathwalk(b, s0s[:, b].contiguous(), keys, xs).cpu().numpy() for b in (0, 1)]
    want = (alphas <= xs).numpy().astype(np.uint8)
    assert ((walk[0] ^ walk[1]) == want).all()
    table = []
    for b in (0, 1):
        row = [int(g.eval(g.preprocess(b, s0s[k, b], cws[k].cpu()), xs[k:k + 1])[0]) for k in range(nkeys)]
        table.append(np.array(row, dtype=np.uint8))
    assert ((table[0] ^ table[1]) == want).all()
    mixed = walk[0] ^ table[1]
    assert (mixed != want).any(), "walk and table shares are documented as incompatible"


def test_a_fresh_context_walks_plainly_and_auto_is_opt_in():
    """The C ABI's own default is FSS_AMD_POINTS_PLAIN -- an eval call allocates nothing and waits for nothing, as the
    reference's launchers (include/fss/point_eval_gpu.cuh:448-460) -- and FSS_AMD_POINTS_AUTO is an explicit choice of
    the context (fss_amd_ctx_set_points_plan); the Python Engine makes it at creation.  Same outputs either way."""
    import fss_amd
    from fss_amd import _lib
    L = _lib.lib()
    L.fss_amd_last_kernel.restype = C.c_char_p
    h = C.c_void_p()
    keys = b"".join(bytes(k) for k in fss_amd.engine.DEFAULT_AES_KEYS)
    assert L.fss_amd_ctx_create(C.byref(h), 0, C.c_char_p(keys), None) == 0
    try:
        n = 1 << 19
        gen = torch.Generator(device="cuda").manual_seed(60)
        xs = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda", generator=gen)
        seed = torch.randint(-2**31, 2**31, (4,), dtype=torch.int32, device="cuda", generator=gen)
        cws = torch.randint(-2**31, 2**31, (33, 8), dtype=torch.int32, device="cuda", generator=gen)
        ys = [torch.empty((n, 4), dtype=torch.int32, device="cuda") for _ in range(2)]
        P = lambda t: C.c_void_p(t.data_ptr())
        free0 = torch.cuda.mem_get_info()[0]
        g, b = C.c_int(0), C.c_int(0)
        assert L.fss_amd_dpf_eval_points(h, 32, 4, 0, P(seed), P(cws), P(xs), 4, n, P(ys[0]), None) == 0
        assert L.fss_amd_last_kernel(h, C.byref(g), C.byref(b)) == b"dpf_eval_points"
        torch.cuda.synchronize()
        assert free0 - torch.cuda.mem_get_info()[0] < (1 << 20)  # nothing allocated by the call
        assert L.fss_amd_ctx_set_points_plan(h, 0) == 0  # FSS_AMD_POINTS_AUTO
        assert L.fss_amd_dpf_eval_points(h, 32, 4, 0, P(seed), P(cws), P(xs), 4, n, P(ys[1]), None) == 0
        assert L.fss_amd_last_kernel(h, C.byref(g), C.byref(b)) == b"dpf_eval_points_shared"
        torch.cuda.synchronize()
        assert torch.equal(ys[0], ys[1])
        assert fss_amd.Engine(0).points_plan == "auto"
    finally:
        L.fss_amd_ctx_destroy(h)
