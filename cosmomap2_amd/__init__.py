"""
cosmomap2_amd -- the PCG map-making hot path of giuspugl/COSMOMAP2 on AMD MI355X.

    from cosmomap2_amd.interfaces import *     # SparseLO, BlockLO, ... , lp
    from cosmomap2_amd.utilities import *      # ProcessTimeSamples, dgemm, system_setup, ...
    from cosmomap2_amd import cg               # device PCG (scipy.sparse.linalg.cg convention)

The arithmetic lives in libcosmomap2_hip.so (include/cosmomap2.h), built in-tree by
``python -m cosmomap2_amd.build``.  There is no CPU fallback.
"""
from . import linop                                           # noqa: F401
from .solvers import cg                                       # noqa: F401

__version__ = "0.1.0"
