"""TEST INFRASTRUCTURE: stand-in (`from PIL import Image` only has to resolve)."""
from . import Image  # noqa: F401
