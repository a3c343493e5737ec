from .checks import ConfigurationError  # noqa: F401
from .from_params import FromParams, Lazy, Registrable  # noqa: F401
from .params import Params  # noqa: F401
from .tqdm import Tqdm  # noqa: F401
from . import util  # noqa: F401
