"""`fss_crypto._int` -- the reference's module name for its integer helper (fss_crypto/_int.py:4-8)."""
from fss_amd._int import split_uint128  # noqa: F401
