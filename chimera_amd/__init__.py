# -*- coding: utf-8 -*-
"""chimera_amd -- MI355X-native implementation of the CHIMERA hyper-likelihood path.

Same public names as the reference package (CHIMERA/__init__.py:27-39): ``hyperlikelihood``, ``selection_function``,
``population``, ``compute_z_grids``, the ``cosmo`` / ``mass`` / ``rate`` / ``completeness`` modules (also importable as
``chimera_amd.cosmo`` etc.), ``data``.  Compute goes through libchimera_hip.so (HIP, gfx950) via ctypes.
"""
__version__ = "0.1.0"
__description__ = "MI355X-native hot path of CHIMERA (gravitational-wave cosmology with galaxy catalogues)"

import sys

from . import utils
from . import data
from .population import *          # cosmo, mass, rate, population, compute_z_grids, ...
from .likelihood import hyperlikelihood
from .selection_function import selection_function
from .catalog import completeness
from . import catalog
from . import parallel

sys.modules[__name__ + ".cosmo"] = cosmo
sys.modules[__name__ + ".mass"] = mass
sys.modules[__name__ + ".rate"] = rate
sys.modules[__name__ + ".completeness"] = completeness
