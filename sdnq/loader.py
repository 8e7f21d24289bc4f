"""sdnq.loader of the import-name drop-in: the names of sdnq_amd.loader (see sdnq/__init__.py)."""
from sdnq_amd.loader import *  # noqa: F401,F403
from sdnq_amd import loader as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
