"""Build recipe for libfss_amd.so (hipcc, gfx950 only, in-tree)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO = os.path.join(_HERE, "libfss_amd.so")
SOURCES = ["fss_amd.hip", "fss_dev.hpp", "fss_point.hpp", "fss_evalall.hpp", "fss_gen.hpp", "fss_vdpf.hpp", "fss_abi_rest.inc",
           "fss_abi_vdpf.inc"]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(_HERE, "..", "include", "fss_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile fss_amd/csrc/fss_amd.hip -> fss_amd/libfss_amd.so for gfx950."""
    if not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", SO,
           os.path.join(CSRC, "fss_amd.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


SAMPLE_SRC = os.path.join(_HERE, "..", "samples", "dpf_dcf_gpu.cpp")
SAMPLE_BIN = os.path.join(_HERE, "..", "samples", "dpf_dcf_gpu")


def build_sample(force=False):
    """Compile the C++ sample (include/fss_amd.hpp over the C ABI) against libfss_amd.so."""
    build()
    hdrs = [os.path.join(_HERE, "..", "include", h) for h in ("fss_amd.h", "fss_amd.hpp")]
    if (not force and os.path.exists(SAMPLE_BIN)
            and os.path.getmtime(SAMPLE_BIN) >= max(os.path.getmtime(f) for f in [SAMPLE_SRC, SO] + hdrs)):
        return SAMPLE_BIN
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(_HERE, "..", "include"),
                           SAMPLE_SRC, "-L" + _HERE, "-lfss_amd", "-Wl,-rpath,$ORIGIN/../fss_amd", "-o", SAMPLE_BIN])
    return SAMPLE_BIN
