from .pop_wrapper import *
from . import cosmo
from . import mass
from . import rate
