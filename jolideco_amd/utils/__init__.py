from .numpy import *  # noqa: F401,F403
from .torch import *  # noqa: F401,F403
