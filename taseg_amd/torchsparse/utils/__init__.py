from .misc import make_ntuple  # noqa: F401
from .quantize import sparse_quantize  # noqa: F401
from .collate import sparse_collate, sparse_collate_fn  # noqa: F401
