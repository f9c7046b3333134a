"""`fss_crypto.dcf.Dcf` -- where the reference keeps the class (fss_crypto/dcf.py:24-113 in myl7/fss): the same constructor
defaults and methods, computed by the MI355X engine (`fss_amd.Dcf`)."""
import fss_amd


class Dcf(fss_amd.Dcf):
    def __init__(self, in_bits, group="bytes", prg="chacha", pred="lt"):  # dcf.py:34-35
        super().__init__(in_bits, group=group, prg=prg, pred=pred)
