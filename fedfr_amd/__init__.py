"""fedfr_amd — MI355X-native implementation of FedFR's per-client training hot path.

Host side mirrors the reference's Python surface (backbones.iresnet*, losses, partial_fc, client,
server); all math runs in libfedfr_hip.so (hand-written HIP for gfx950) through the C ABI declared in
include/fedfr_hip.h.  There is no CPU fallback.
"""
__version__ = "0.1.0"
