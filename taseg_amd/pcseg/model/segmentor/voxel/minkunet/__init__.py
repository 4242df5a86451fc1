from .minkunet import MinkUNet  # noqa: F401
from .minkunet_ms import MinkUNetMs  # noqa: F401
