from .model import Model  # noqa: F401
from .archival import Archive, load_archive  # noqa: F401
