from .catalog import *
from .completeness import *
from . import completeness
