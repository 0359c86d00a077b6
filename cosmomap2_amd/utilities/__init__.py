"""Flat namespace like the reference's ``from utilities import *``
(utilities/__init__.py:6-10 of the reference)."""
from .utilities_functions import *                           # noqa: F401,F403
from .linear_algebra_funcs import *                          # noqa: F401,F403
from .process_ces import *                                   # noqa: F401,F403
from .healpy_functions import *                              # noqa: F401,F403
from .IOfiles import *                                       # noqa: F401,F403
