"""Runtime configuration (reference: CHIMERA/utils/config.py:1-24): same logger name and format; the compute
backend is always the HIP library, so the reference's ``CHIMERA_ENABLE_GPU`` switch is accepted and ignored."""
import logging
import os

USE_GPU = True
_ = os.getenv('CHIMERA_ENABLE_GPU', 'True')

logger = logging.getLogger('CHIMERA')
if not logger.handlers:
  logger.setLevel(logging.INFO if os.getenv('CHIMERA_LOG', '').lower() in ('1', 'info', 'true') else logging.WARNING)
  _h = logging.StreamHandler()
  _h.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
  logger.addHandler(_h)
