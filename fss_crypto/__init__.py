"""Drop-in for the reference's Python binding: `import fss_crypto` keeps working with the same two
classes, constructor defaults and method signatures (fss_crypto/__init__.py, fss_crypto/dpf.py:23-109,
fss_crypto/dcf.py:24-113 in myl7/fss), computed by the MI355X engine in `fss_amd` -- and with the
reference's module layout: `fss_crypto.dpf.Dpf`, `fss_crypto.dcf.Dcf`, `fss_crypto._validate`, `fss_crypto._int`
(there is no `_jit`: nothing is compiled at import time, the library is built once by `fss_amd._build`).

Only the reference's surface is exported here; the batched methods (eval_points, eval_keys, relayout,
eval_all sub-blocks) and the other schemes live in `fss_amd`."""
from fss_crypto.dcf import Dcf
from fss_crypto.dpf import Dpf

__all__ = ["Dcf", "Dpf"]
