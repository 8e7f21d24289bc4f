"""sdnq.forward of the import-name drop-in: the names of sdnq_amd.forward (see sdnq/__init__.py)."""
from sdnq_amd.forward import *  # noqa: F401,F403
from sdnq_amd import forward as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
