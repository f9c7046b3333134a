"""The C-ABI library loads without a GPU and exports every symbol include/fss_amd.h declares;
the Python prototypes cover exactly that set.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fss_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fss_amd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    syms = declared_symbols()
    for must in ["fss_amd_ctx_create", "fss_amd_dpf_eval_points", "fss_amd_dcf_eval_points", "fss_amd_ht_eval_keys",
                 "fss_amd_dpf_eval_all", "fss_amd_dpf_relayout", "fss_amd_grotto_eval_all", "fss_amd_dpf_gen"]:
        assert must in syms


def test_library_exports_every_declared_symbol():
    from fss_amd import _build, _lib
    _build.build()
    lib = ctypes.CDLL(_lib.SO)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_python_prototypes_match_header():
    from fss_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_symbols()


def test_no_gpu_means_loud_failure():
    """Without a HIP device the product refuses to run (there is no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import fss_amd
    with pytest.raises(RuntimeError, match="no CPU path"):
        fss_amd.default_engine()
    d = fss_amd.Dpf(16, "bytes")
    with pytest.raises(RuntimeError, match="no CPU path"):
        d.gen(torch.zeros((2, 4), dtype=torch.int32), 1, torch.zeros(4, dtype=torch.int32))


def test_error_strings_without_gpu():
    from fss_amd import _lib
    L = _lib.lib()
    assert L.fss_amd_strerror(0) == b"ok"
    assert L.fss_amd_strerror(-1) == b"invalid argument"
    assert L.fss_amd_eval_all_workspace(0, 20, 1 << 20) > 0
    assert L.fss_amd_eval_all_workspace(0, 20, 3) == 0       # not a power of two
    assert L.fss_amd_eval_all_workspace(2, 20, 1) == 0       # Half-Tree blocks hold >= 2 leaves


def test_product_does_not_import_the_oracle():
    """The shipped package must never route through oracle/."""
    pkg = os.path.join(ROOT, "fss_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), (dirpath, f)


def test_cxx_layer_compiles():
    """include/fss_amd.hpp (the fss::Dpf / fss::gpu::* mirror) and the sample build with hipcc."""
    from fss_amd import _build
    assert os.path.exists(_build.build_sample())


def test_cxx_dropin_test_compiles():
    """Caller code in the reference's call syntax (tests/cxx/dropin_test.cpp) compiles against include/fss_amd.hpp
    unchanged; it runs under -m gpu (tests/test_gpu_parity.py::test_cxx_dropin_boundary)."""
    from fss_amd import _build
    assert os.path.exists(_build.build_dropin_test())


def test_cxx_dropin_device_test_compiles():
    """Gen / Eval called per thread inside caller kernels, with the built-in PRGs and with caller-defined Prgable /
    Groupable types (tests/cxx/dropin_device_test.cpp), compiles; it runs under -m gpu
    (tests/test_gpu_parity.py::test_cxx_dropin_device_side)."""
    from fss_amd import _build
    assert os.path.exists(_build.build_dropin_device_test())


def test_cxx_reference_test_list_compiles_and_its_group_axioms_hold():
    """tests/cxx/reference_cases_test.cpp (the reference's own gtest list through its <fss/...cuh> include lines) compiles; the 33
    group-axiom cases of src/group_test.cu -- eleven group types, Uint<__uint128_t, 2^126> among them -- are host arithmetic
    and run here; the scheme cases run under -m gpu (tests/test_gpu_parity.py::test_cxx_reference_test_list)."""
    import subprocess
    from fss_amd import _build
    exe = _build.build_reference_cases_test()
    r = subprocess.run([exe, "--filter", "GroupAxiomTest"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "33 cases, 0 failed" in r.stdout and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_public_cxx_headers_are_warning_free_under_wall_wextra():
    """include/fss_amd.hpp, fss_amd_thread.hpp and fss_amd_multi.hpp (through the <fss/...cuh> include lines) with the templates the
    drop-in tests instantiate -- host signatures, launchers, per-thread walks in kernels, DeviceSet, the reference's test list -- compile for host and gfx950 under
    -Wall -Wextra -Werror: a caller who builds with warnings as errors can include them."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for src in ("reference_cases_test.cpp", "dropin_test.cpp", "dropin_device_test.cpp", "multi_gpu_test.cpp"):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++20", "-Wall", "-Wextra", "-Werror", "-Wno-unused-command-line-argument",
                            "-I" + os.path.join(root, "include"), "-fsyntax-only", os.path.join(root, "tests", "cxx", src)],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]


def test_cxx_host_side_aes_known_answers():
    """The software AES of include/fss_amd.hpp on the HOST (tests/cxx/host_kat_test.cpp: aes_detail helpers, FIPS-197 C.1,
    the reference's PRG known answers through Aes128Soft / Aes128SoftBanked Gen and GenSide) -- the cipher code the per-thread
    device path instantiates with its LDS tables, run here without a GPU."""
    import subprocess
    from fss_amd import _build
    exe = _build.build_host_kat_test()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_cxx_multi_gpu_test_and_sample_compile():
    """fss::gpu::DeviceSet (include/fss_amd_multi.hpp): tests/cxx/multi_gpu_test.cpp and samples/multi_gpu.cpp compile; they
    run under -m gpu (tests/test_gpu_parity.py::test_cxx_multi_gpu_device_set)."""
    from fss_amd import _build
    assert os.path.exists(_build.build_multi_gpu_test()) and os.path.exists(_build.build_multi_gpu_sample())


def test_header_is_c99_and_links_from_c(tmp_path):
    """include/fss_amd.h is the drop-in boundary of a C FFI (cgo / JNI / ctypes): tests/c/abi_c99.c compiles as pedantic
    C99 against it, links libfss_amd.so and runs the entry points that need no GPU (slice arithmetic, plan query)."""
    import subprocess
    from fss_amd import _build
    _build.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_c99")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "c", "abi_c99.c"), "-L" + os.path.join(root, "fss_amd"), "-lfss_amd",
                           "-Wl,-rpath," + os.path.join(root, "fss_amd"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.returncode, r.stdout, r.stderr)


def test_makefile_stamp_equals_python_digest():
    """`make -C fss_amd/csrc` stamps the library with sources_digest.sh; fss_amd/_build.py must accept that stamp
    (same digest), or Python would rebuild a library make has just built."""
    import subprocess
    from fss_amd import _build
    sh = os.path.join(os.path.dirname(_build.CSRC), "csrc", "sources_digest.sh")
    out = subprocess.run(["sh", sh], capture_output=True, text=True, check=True).stdout.strip()
    assert out == _build._sources_digest()
