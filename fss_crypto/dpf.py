"""`fss_crypto.dpf.Dpf` -- where the reference keeps the class (fss_crypto/dpf.py:23-109 in myl7/fss): the same constructor
defaults and methods, computed by the MI355X engine (`fss_amd.Dpf`)."""
import fss_amd


class Dpf(fss_amd.Dpf):
    def __init__(self, in_bits, group="bytes", prg="chacha"):  # the reference's defaults (dpf.py:32-33)
        super().__init__(in_bits, group=group, prg=prg)
