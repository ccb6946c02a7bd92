"""Event / injection sharding across GPUs: one process per GPU, one RCCL all-reduce per evaluation.

Replaces the design of the reference's (un-importable) MPI layer, CHIMERA/parallel.py: contiguous chunks of events
(parallel.py:94-99) and of injections (:68-73) per rank, and a SUM all-reduce of the per-rank log-numerators and
selection sums (:366-376, 406-407).  Here the payload is ``3 * nbatch`` doubles -- [sum_i log L_i, nansum dN, sum dN^2]
per draw -- reduced by ``ncclAllReduce`` inside ``chm_eval`` on the evaluation stream.

The RCCL unique id is exchanged out of band: through ``torch.distributed`` if a process group is up (plumbing only),
else through a file (``CHIMERA_COMM_FILE``).
"""
import ctypes as C
import os
import time
import numpy as np
from . import _lib


def chunk_bounds(n, nranks, rank):
  """Contiguous chunk [lo, hi) of ``n`` items for ``rank``: n//R each, the first n%R ranks get one extra
  (CHIMERA/parallel.py:68-73, 94-99)."""
  base, extra = divmod(int(n), int(nranks))
  lo = rank * base + min(rank, extra)
  return lo, lo + base + (1 if rank < extra else 0)


class Comm(object):
  """A rank of an RCCL communicator bound to one GPU."""

  def __init__(self, nranks, rank, device=None, unique_id=None):
    self.nranks, self.rank = int(nranks), int(rank)
    self.device = _lib.default_device() if device is None else int(device)
    self._h = C.c_void_p()
    if unique_id is None:
      unique_id = exchange_unique_id(self.nranks, self.rank)
    buf = C.create_string_buffer(bytes(unique_id), 128)
    _lib.check(_lib.lib().chm_comm_init_rank(buf, self.nranks, self.rank, self.device, C.byref(self._h)))

  @classmethod
  def from_env(cls):
    """Build from torchrun-style environment variables (RANK, WORLD_SIZE, LOCAL_RANK)."""
    n, r = int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('RANK', 0))
    return cls(n, r, int(os.environ.get('LOCAL_RANK', 0)))

  @property
  def handle(self):
    return self._h

  def allreduce_sum(self, x):
    x = _lib.as_f64(x).copy()
    _lib.check(_lib.lib().chm_comm_allreduce_sum(self._h, _lib.dptr(x), x.size))
    return x

  def close(self):
    if self._h:
      _lib.lib().chm_comm_destroy(self._h)
      self._h = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


class HostComm(object):
  """Fallback communicator for hosts where RCCL cannot bring up a communicator: the ``3 * nbatch`` partial sums travel through
  ``torch.distributed`` (any initialised backend, e.g. gloo) on the host and the combination (likelihood.py:298-316,
  selection_function.py:38-47) is formed by :func:`combine_partials` instead of ``k_combine``.  Same sharding, same sums; only the
  transport differs (an extra D2H/H2D of 3 doubles per draw per call)."""
  handle = None

  def __init__(self, nranks, rank, device=None):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
      raise RuntimeError("HostComm needs an initialised torch.distributed process group")
    self.nranks, self.rank = int(nranks), int(rank)
    self.device = _lib.default_device() if device is None else int(device)

  def allreduce_sum(self, x):
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(_lib.as_f64(x).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()

  def close(self):
    pass


def combine_partials(partials, E_total, N_inj, N_eff, scale_free, R0, Tobs, has_like=True, has_sel=True):
  """Host form of the device's ``combine_one``: [sum_i log L_i, nansum dN, sum dN^2] (summed over ranks) ->
  (log_hyper, log_num, N_exp)   (selection_function.py:38-47, likelihood.py:298-300, 313-316)."""
  log_num, s1, s2 = (float(v) for v in partials)
  with np.errstate(all='ignore'):
    Nexp = np.nan
    if has_sel:
      xi = s1 / N_inj
      Nexp = Tobs * xi
      if N_eff is not None:
        variance2 = s2 / N_inj**2 - xi**2 / N_inj
        if xi**2 / variance2 < N_eff:
          Nexp = 0.0
    log_hyper = np.nan
    if has_like:
      if not scale_free:
        log_num = log_num + E_total * np.log(R0 * Tobs)
      if has_sel:
        log_hyper = log_num - E_total * np.log(Nexp) if scale_free else log_num - Nexp
    else:
      log_num = np.nan
  return log_hyper, log_num, Nexp


def new_unique_id():
  buf = C.create_string_buffer(128)
  _lib.check(_lib.lib().chm_comm_unique_id(buf))
  return buf.raw


def exchange_unique_id(nranks, rank):
  """Rank 0 creates the RCCL unique id; the 128 bytes reach the other ranks through torch.distributed (if
  initialised) or a rendezvous file."""
  if nranks == 1:
    return new_unique_id()
  try:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
      obj = [new_unique_id() if rank == 0 else None]
      dist.broadcast_object_list(obj, src=0)
      return obj[0]
  except ImportError:
    pass
  path = os.environ.get('CHIMERA_COMM_FILE')
  if not path:
    raise RuntimeError("multi-rank Comm needs an initialised torch.distributed group or CHIMERA_COMM_FILE")
  if rank == 0:
    uid = new_unique_id()
    with open(path + '.tmp', 'wb') as f:
      f.write(uid)
    os.replace(path + '.tmp', path)
    return uid
  t0 = time.time()
  while not os.path.exists(path):
    if time.time() - t0 > 120:
      raise RuntimeError(f"timed out waiting for {path}")
    time.sleep(0.05)
  with open(path, 'rb') as f:
    return f.read()
