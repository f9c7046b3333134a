"""Build recipe for libfss_amd.so (hipcc, gfx950 only, in-tree)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO = os.path.join(_HERE, "libfss_amd.so")
UNITS = ["fss_amd.hip", "fss_abi_evalall.hip", "fss_abi_gen.hip", "fss_abi_fused.hip", "fss_abi_fused_batch.hip",
         "fss_abi_fused_chacha.hip", "fss_abi_fused_nodes.hip"]  # compiled in parallel, linked into one .so
HEADERS = ["fss_dev.hpp", "fss_point.hpp", "fss_quad.hpp", "fss_evalall.hpp", "fss_gen.hpp", "fss_vdpf.hpp", "fss_host.hpp",
           "fss_fused_launch.hpp"]
OBJDIR = os.path.join(_HERE, "csrc", "build")


STAMP = os.path.join(_HERE, "libfss_amd.so.sources.sha256")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _sources_digest():
    """SHA-256 over the library's sources and compile flags: what the built .so is stamped with."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in [os.path.join(CSRC, s) for s in UNITS + HEADERS] + [os.path.join(_HERE, "..", "include", "fss_amd.h")]:
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


HEADLINE_SOURCES = ("fss_dev.hpp", "fss_point.hpp", "fss_host.hpp", "fss_amd.hip")


# Sources the kernels of bench.py's other workloads (and their launch plans) are compiled from: the full-domain
# evaluation is the one-launch form of fss_abi_fused.hip over fss_evalall.hpp; the independent-key walks are the point
# walk's kernel; the Grotto path walk's launch lives in fss_abi_evalall.hip.
_EVALALL_SOURCES = ("fss_dev.hpp", "fss_point.hpp", "fss_quad.hpp", "fss_evalall.hpp", "fss_host.hpp",
                    "fss_fused_launch.hpp", "fss_abi_fused.hip", "fss_abi_evalall.hip")
WORKLOAD_SOURCES = {
    "headline": HEADLINE_SOURCES, "dpf_2e20": HEADLINE_SOURCES, "dcf_2e20": HEADLINE_SOURCES,
    "evalall_c4": _EVALALL_SOURCES,
    "ht_c5": HEADLINE_SOURCES,
    "grotto_c5": HEADLINE_SOURCES + ("fss_abi_evalall.hip",),
}


def _digest_of(files):
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


def headline_digest():
    """SHA-256 over the sources (and flags) the point-walk kernels and their launch plan are compiled from: the
    rocprofv3 PMC summaries under profiles/ carry it, and bench.py says whether the counters it replays were
    captured for these sources."""
    return _digest_of(HEADLINE_SOURCES)


def workload_digest(name):
    """The same for one of bench.py's workloads (WORKLOAD_SOURCES); profiles/r<N>/pmc_workloads.json carries it."""
    return _digest_of(WORKLOAD_SOURCES[name])


def _stale():
    """The .so is rebuilt unless its stamp equals the digest of the sources it would be built from (file times are
    not trusted: a checkout or a copied snapshot resets them)."""
    if not os.path.exists(SO) or not os.path.exists(STAMP):
        return True
    return open(STAMP).read().strip() != _sources_digest()


def build_variant(tag, defines, verbose=False):
    """A tuning / diagnostic build of the same sources with extra -D flags -> fss_amd/libfss_amd_<tag>.so
    (used through FSS_AMD_LIB=...; never the product library)."""
    return build(force=True, verbose=verbose, so=os.path.join(_HERE, f"libfss_amd_{tag}.so"), defines=defines, tag=tag)


DIAG_SO = os.path.join(_HERE, "libfss_amd_diag.so")


def build_diag(force=False, verbose=False):
    """The diagnostic build bench.py measures the in-kernel clock with (fss_amd._lib.diag_lib): the same sources with
    -DFSS_DIAG_STAMPS=1 (every wave of the walks records s_memtime / s_memrealtime at entry and exit).  Stamped like the
    product library, rebuilt only when the sources changed.  Never loaded by the product path."""
    stamp = DIAG_SO + ".sources.sha256"
    want = _sources_digest() + " FSS_DIAG_STAMPS=1"
    if not force and os.path.exists(DIAG_SO) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return DIAG_SO
    build_variant("diag", ["FSS_DIAG_STAMPS=1"], verbose=verbose)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return DIAG_SO


def remove_stale_variants():
    """Delete tuning builds (fss_amd/libfss_amd_<tag>.so, tag != diag) left by an earlier session: nothing loads them,
    and every .so in the tree is pushed to the GPU box with each gpurun call."""
    import glob
    gone = []
    if os.environ.get("FSS_AMD_KEEP_VARIANTS"):  # an A/B session that builds variants and the library in turn
        return gone
    for f in glob.glob(os.path.join(_HERE, "libfss_amd_*.so")) + glob.glob(os.path.join(_HERE, "libfss_amd_*.so.sources.sha256")):
        if os.path.basename(f).startswith("libfss_amd_diag.so"):
            continue
        os.remove(f)
        gone.append(f)
    # ... and their objects (csrc/build/<unit>.<tag>.o; the product's are <unit>.o, the diagnostic build's <unit>.diag.o)
    for f in glob.glob(os.path.join(CSRC, "build", "*.*.o")):
        if not f.endswith(".diag.o"):
            os.remove(f)
            gone.append(f)
    return gone


def build(force=False, verbose=False, so=None, defines=(), tag=""):
    """Compile fss_amd/csrc/*.hip for gfx950 (one hipcc per unit, in parallel) and link fss_amd/libfss_amd.so."""
    so = so or SO
    if so == SO:
        for f in remove_stale_variants():
            if verbose:
                print("removed stale variant", f)
    if so == SO and not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJDIR, exist_ok=True)
    flags = FLAGS + [f"-D{d}" for d in defines]
    procs = []
    for u in UNITS:
        obj = os.path.join(OBJDIR, u.replace(".hip", (f".{tag}" if tag else "") + ".o"))
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, u), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, obj, subprocess.Popen(cmd)))
    objs = []
    for cmd, obj, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
        objs.append(obj)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    if so == SO:
        with open(STAMP, "w") as f:
            f.write(_sources_digest() + "\n")
    return so


SAMPLE_SRC = os.path.join(_HERE, "..", "samples", "dpf_dcf_gpu.cpp")
SAMPLE_BIN = os.path.join(_HERE, "..", "samples", "dpf_dcf_gpu")


def build_sample(force=False):
    """Compile the C++ sample (include/fss_amd.hpp over the C ABI) against libfss_amd.so."""
    build()
    hdrs = [os.path.join(_HERE, "..", "include", h) for h in PUBLIC_HEADERS]
    if (not force and os.path.exists(SAMPLE_BIN)
            and os.path.getmtime(SAMPLE_BIN) >= max(os.path.getmtime(f) for f in [SAMPLE_SRC, SO] + hdrs)):
        return SAMPLE_BIN
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++20", "-I" + os.path.join(_HERE, "..", "include"),
                           SAMPLE_SRC, "-L" + _HERE, "-lfss_amd", "-Wl,-rpath,$ORIGIN/../fss_amd", "-o", SAMPLE_BIN])
    return SAMPLE_BIN


CXX_TESTS = os.path.join(_HERE, "..", "tests", "cxx")
DROPIN_SRC = os.path.join(CXX_TESTS, "dropin_test.cpp")
DROPIN_BIN = os.path.join(CXX_TESTS, "dropin_test")
PUBLIC_HEADERS = ("fss_amd.h", "fss_amd.hpp", "fss_amd_thread.hpp", "fss_amd_multi.hpp")


def _build_cxx_test(name, force=False):
    build()
    src, exe = os.path.join(CXX_TESTS, name + ".cpp"), os.path.join(CXX_TESTS, name)
    hdrs = [os.path.join(_HERE, "..", "include", h) for h in PUBLIC_HEADERS]
    if not force and os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(f) for f in [src, SO] + hdrs):
        return exe
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++20", "-I" + os.path.join(_HERE, "..", "include"),
                           src, "-L" + _HERE, "-lfss_amd", "-Wl,-rpath,$ORIGIN/../../fss_amd", "-o", exe])
    return exe


def build_dropin_test(force=False):
    """Compile tests/cxx/dropin_test.cpp: caller code in the reference's own call syntax against include/fss_amd.hpp."""
    return _build_cxx_test("dropin_test", force)


def build_host_kat_test(force=False):
    """Compile tests/cxx/host_kat_test.cpp: the software AES of include/fss_amd.hpp called on the host (runs without a GPU)."""
    return _build_cxx_test("host_kat_test", force)


def build_dropin_device_test(force=False):
    """Compile tests/cxx/dropin_device_test.cpp: Gen / Eval called per thread inside the caller's own kernels."""
    return _build_cxx_test("dropin_device_test", force)


def build_reference_cases_test(force=False):
    """Compile tests/cxx/reference_cases_test.cpp: the reference's own gtest case list (group axioms, DPF / DCF / Half-Tree /
    Grotto / VDPF properties) under its suite and case names, through the <fss/...cuh> include lines."""
    return _build_cxx_test("reference_cases_test", force)


def build_multi_gpu_test(force=False):
    """Compile tests/cxx/multi_gpu_test.cpp: fss::gpu::DeviceSet, eight members on one GPU against one launch."""
    return _build_cxx_test("multi_gpu_test", force)


MULTI_SAMPLE_SRC = os.path.join(_HERE, "..", "samples", "multi_gpu.cpp")
MULTI_SAMPLE_BIN = os.path.join(_HERE, "..", "samples", "multi_gpu")


def build_multi_gpu_sample(force=False):
    """Compile samples/multi_gpu.cpp (fss::gpu::DeviceSet over every visible device)."""
    build()
    hdrs = [os.path.join(_HERE, "..", "include", h) for h in PUBLIC_HEADERS]
    if (not force and os.path.exists(MULTI_SAMPLE_BIN)
            and os.path.getmtime(MULTI_SAMPLE_BIN) >= max(os.path.getmtime(f) for f in [MULTI_SAMPLE_SRC, SO] + hdrs)):
        return MULTI_SAMPLE_BIN
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++20", "-I" + os.path.join(_HERE, "..", "include"),
                           MULTI_SAMPLE_SRC, "-L" + _HERE, "-lfss_amd", "-Wl,-rpath,$ORIGIN/../fss_amd", "-o", MULTI_SAMPLE_BIN])
    return MULTI_SAMPLE_BIN
