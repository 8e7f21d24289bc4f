"""sdnq.kernel_wrappers of the import-name drop-in: the names of sdnq_amd.kernel_wrappers (see sdnq/__init__.py)."""
from sdnq_amd.kernel_wrappers import *  # noqa: F401,F403
from sdnq_amd import kernel_wrappers as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
