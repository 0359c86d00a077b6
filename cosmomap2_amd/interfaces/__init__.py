"""Flat namespace like the reference's ``from interfaces import *``
(interfaces/__init__.py:4-6 of the reference)."""
from .. import linop as lp                                   # noqa: F401
from .linearoperators import *                               # noqa: F401,F403
from .blkop import BlockDiagonalLinearOperator               # noqa: F401
from .deflationlib import *                                  # noqa: F401,F403
