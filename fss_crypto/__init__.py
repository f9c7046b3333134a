"""Drop-in for the reference's Python binding: `import fss_crypto` keeps working with the same two
classes, constructor defaults and method signatures (fss_crypto/__init__.py, fss_crypto/dpf.py:23-109,
fss_crypto/dcf.py:24-113 in myl7/fss), computed by the MI355X engine in `fss_amd`.

Only the reference's surface is exported here; the batched methods (eval_points, eval_keys, relayout,
eval_all sub-blocks) and the other schemes live in `fss_amd`."""
import fss_amd


class Dpf(fss_amd.Dpf):
    def __init__(self, in_bits, group="bytes", prg="chacha"):  # the reference's defaults (dpf.py:32-33)
        super().__init__(in_bits, group=group, prg=prg)


class Dcf(fss_amd.Dcf):
    def __init__(self, in_bits, group="bytes", prg="chacha", pred="lt"):  # dcf.py:34-35
        super().__init__(in_bits, group=group, prg=prg, pred=pred)


__all__ = ["Dcf", "Dpf"]
