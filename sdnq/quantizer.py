"""``sdnq.quantizer`` of the import-name drop-in: ``sdnq_amd.quantizer`` plus the plugin classes of ``sdnq_amd.hf_quantizer``."""
from sdnq_amd import quantizer as _m
from sdnq_amd.hf_quantizer import SDNQConfig, SDNQQuantizer, register  # noqa: F401

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__") and k != "SDNQConfig"})
