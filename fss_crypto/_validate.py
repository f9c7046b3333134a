"""`fss_crypto._validate` -- the module name under which the reference keeps its argument validators and under which its
own tests import them (test/test_validation.py:3-17).  The validators are `fss_amd._validate`'s (the messages the
reference's tests match by regex; pinned by tests/golden/validation_contract.json)."""
from fss_amd._validate import (  # noqa: F401
    validate_alpha,
    validate_beta,
    validate_cpu_only,
    validate_cws,
    validate_device_match,
    validate_domain_value,
    validate_group,
    validate_in_bits,
    validate_party,
    validate_pred,
    validate_prg,
    validate_s0,
    validate_s0s,
)
