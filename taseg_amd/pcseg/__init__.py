"""pcseg-compatible surface of the hot path: `pcseg.model.build_network`, the MinkUNet family
segmentors and the loss they call (reference: /root/reference/pcseg/{model,loss})."""
from . import loss, model  # noqa: F401
