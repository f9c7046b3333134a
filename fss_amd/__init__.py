"""fss_amd: MI355X-native batched DPF/DCF evaluation (AES-128-MMO PRG) behind the
reference's scheme interfaces.  The compute path is libfss_amd.so (hand-written HIP
for gfx950); importing this package without that library raises on first use."""
from .engine import (BYTES, DCF, DEFAULT_AES_KEYS, DEFAULT_HASH_KEY, DPF, GROTTO, HT, U8, U16, U32, U64, U128, Engine,
                     LevelMajorKeys, default_engine)
from .schemes import Dcf, Dpf, GrottoDcf, HalfTreeDpf, Vdpf

__all__ = ["Dcf", "Dpf", "GrottoDcf", "HalfTreeDpf", "Vdpf", "Engine", "LevelMajorKeys", "default_engine"]
