"""Event / injection sharding across GPUs: one process per GPU, one RCCL all-reduce per evaluation.

Replaces the design of the reference's (un-importable) MPI layer, CHIMERA/parallel.py: contiguous chunks of events
(parallel.py:94-99) and of injections (:68-73) per rank, and a SUM all-reduce of the per-rank log-numerators and
selection sums (:366-376, 406-407).  Here the payload is ``3 * nbatch`` doubles -- [sum_i log L_i, nansum dN, sum dN^2]
per draw -- reduced by ``ncclAllReduce`` inside ``chm_eval`` on the evaluation stream.

Control plane (no PyTorch, no MPI): :class:`Rendezvous` is a small socket star over the ranks of one job -- rank 0 listens
on a Unix-domain socket (one node; the path is derived from the launcher's MASTER_PORT and parent pid) or on TCP
(``CHIMERA_COMM_ADDR=host:port``, several nodes) -- and carries only the 128-byte RCCL unique id, barriers and a few
doubles (the max-over-ranks of a wall time).  :class:`HostComm` reduces the ``3 * nbatch`` partial sums over the same
star for hosts on which RCCL cannot bring a communicator up; the data path of a healthy job never touches it.
"""
import ctypes as C
import os
import socket
import struct
import time
import numpy as np
from . import _lib


def chunk_bounds(n, nranks, rank):
  """Contiguous chunk [lo, hi) of ``n`` items for ``rank``: n//R each, the first n%R ranks get one extra
  (CHIMERA/parallel.py:68-73, 94-99)."""
  base, extra = divmod(int(n), int(nranks))
  lo = rank * base + min(rank, extra)
  return lo, lo + base + (1 if rank < extra else 0)


# ----------------------------------------------------------------------------------------------------------
# control plane
# ----------------------------------------------------------------------------------------------------------
_OP_SUM, _OP_MAX, _OP_BCAST = 1, 2, 3


def _recv_exact(sock, n):
  buf = bytearray()
  while len(buf) < n:
    chunk = sock.recv(n - len(buf))
    if not chunk:
      raise ConnectionError("chimera_amd.parallel: peer closed the rendezvous connection")
    buf.extend(chunk)
  return bytes(buf)


def _send_msg(sock, op, payload):
  sock.sendall(struct.pack('<BQ', op, len(payload)) + payload)


def _recv_msg(sock):
  op, n = struct.unpack('<BQ', _recv_exact(sock, 9))
  return op, _recv_exact(sock, n)


def default_address(world=None):
  """Where the ranks of this job meet.  ``CHIMERA_COMM_ADDR=host:port`` -> TCP; otherwise a Unix-domain socket under
  /tmp named after the launcher: MASTER_PORT (set by ``torch.distributed.run`` / any torchrun-style launcher) and the
  parent pid shared by the ranks it started, so two jobs on one node -- or a stale path of an earlier job -- never meet."""
  addr = os.environ.get('CHIMERA_COMM_ADDR')
  if addr:
    host, port = addr.rsplit(':', 1)
    return (host, int(port))
  tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
  return os.path.join(os.environ.get('CHIMERA_COMM_DIR', '/tmp'), f"chimera_rdzv_{tag}.sock")


class Rendezvous(object):
  """Socket star over the ranks of one job: rank 0 is the hub.  Every method is a collective (all ranks call it, in the
  same order).  Reductions run in rank order on the hub, so every rank receives the same bits."""

  def __init__(self, nranks, rank, address=None, timeout=300.):
    self.nranks, self.rank = int(nranks), int(rank)
    self.address = default_address() if address is None else address
    self._peers, self._sock, self._listener = [], None, None
    if self.nranks == 1:
      return
    unix = isinstance(self.address, str)
    fam = socket.AF_UNIX if unix else socket.AF_INET
    if self.rank == 0:
      ls = socket.socket(fam, socket.SOCK_STREAM)
      if unix:
        try:
          os.unlink(self.address)                               # a stale path of a dead job with the same launcher pid
        except FileNotFoundError:
          pass
      else:
        ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
      ls.bind(self.address)
      ls.listen(self.nranks)
      ls.settimeout(timeout)
      self._listener = ls
      peers = {}
      while len(peers) < self.nranks - 1:
        conn, _ = ls.accept()
        conn.settimeout(timeout)
        r, n = struct.unpack('<II', _recv_exact(conn, 8))
        if n != self.nranks or r in peers or not (0 < r < self.nranks):
          conn.close()
          raise RuntimeError(f"chimera_amd.parallel: unexpected peer (rank {r} of {n}) at the rendezvous of a {self.nranks}-rank job")
        peers[r] = conn
      self._peers = [peers[r] for r in range(1, self.nranks)]
    else:
      t0 = time.time()
      while True:
        s = socket.socket(fam, socket.SOCK_STREAM)
        try:
          s.connect(self.address)
          break
        except (FileNotFoundError, ConnectionRefusedError, OSError):
          s.close()
          if time.time() - t0 > timeout:
            raise RuntimeError(f"chimera_amd.parallel: rank {self.rank} could not reach rank 0 at {self.address}")
          time.sleep(0.02)
      s.settimeout(timeout)
      s.sendall(struct.pack('<II', self.rank, self.nranks))
      self._sock = s

  # -- collectives ---------------------------------------------------------------------------------------
  def _reduce(self, x, op):
    x = np.ascontiguousarray(x, dtype=np.float64)
    if self.nranks == 1:
      return x.copy()
    if self.rank == 0:
      acc = x.copy()
      for p in self._peers:                                     # rank order: the same sum on every run
        o, payload = _recv_msg(p)
        if o != op or len(payload) != acc.nbytes:
          raise RuntimeError("chimera_amd.parallel: ranks disagree on the collective being run")
        y = np.frombuffer(payload, dtype=np.float64).reshape(acc.shape)
        acc = acc + y if op == _OP_SUM else np.maximum(acc, y)
      out = acc.tobytes()
      for p in self._peers:
        _send_msg(p, op, out)
      return acc
    _send_msg(self._sock, op, x.tobytes())
    o, payload = _recv_msg(self._sock)
    return np.frombuffer(payload, dtype=np.float64).reshape(x.shape).copy()

  def allreduce_sum(self, x):
    return self._reduce(x, _OP_SUM)

  def allreduce_max(self, x):
    return self._reduce(x, _OP_MAX)

  def barrier(self):
    self._reduce(np.zeros(1), _OP_SUM)

  def broadcast_bytes(self, data=None):
    """Rank 0's ``data`` on every rank."""
    if self.nranks == 1:
      return bytes(data)
    if self.rank == 0:
      for p in self._peers:
        _send_msg(p, _OP_BCAST, bytes(data))
      return bytes(data)
    o, payload = _recv_msg(self._sock)
    if o != _OP_BCAST:
      raise RuntimeError("chimera_amd.parallel: ranks disagree on the collective being run")
    return payload

  def close(self):
    for s in self._peers + [self._sock, self._listener]:
      if s is not None:
        try:
          s.close()
        except OSError:
          pass
    if self._listener is not None and isinstance(self.address, str):
      try:
        os.unlink(self.address)
      except OSError:
        pass
    self._peers, self._sock, self._listener = [], None, None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


def env_ranks():
  """(rank, world, local_rank) from the torchrun-style environment (RANK, WORLD_SIZE, LOCAL_RANK)."""
  return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


# ----------------------------------------------------------------------------------------------------------
# data plane
# ----------------------------------------------------------------------------------------------------------
class Comm(object):
  """A rank of an RCCL communicator bound to one GPU.  The 128-byte unique id comes from ``unique_id`` (every rank passes
  the same bytes) or travels over ``rendezvous`` (rank 0 creates it)."""

  def __init__(self, nranks, rank, device=None, unique_id=None, rendezvous=None):
    self.nranks, self.rank = int(nranks), int(rank)
    self.device = _lib.default_device() if device is None else int(device)
    self._h = C.c_void_p()
    if unique_id is None:
      unique_id = exchange_unique_id(self.nranks, self.rank, rendezvous)
    buf = C.create_string_buffer(bytes(unique_id), 128)
    _lib.check(_lib.lib().chm_comm_init_rank(buf, self.nranks, self.rank, self.device, C.byref(self._h)))
    got = _lib.lib().chm_comm_nranks(self._h)
    if got != self.nranks:
      self.close()
      raise RuntimeError(f"RCCL communicator reports {got} ranks, expected {self.nranks}")

  @classmethod
  def from_env(cls, rendezvous=None):
    """Build from torchrun-style environment variables (RANK, WORLD_SIZE, LOCAL_RANK)."""
    r, n, lr = env_ranks()
    ndev = _lib.lib().chm_device_count()
    return cls(n, r, lr % ndev if ndev > 0 else lr, rendezvous=rendezvous)

  @property
  def handle(self):
    return self._h

  def allreduce_sum(self, x):
    x = _lib.as_f64(x).copy()
    _lib.check(_lib.lib().chm_comm_allreduce_sum(self._h, _lib.dptr(x), x.size))
    return x

  def set_ticket(self, ticket):
    """The next evaluation on this communicator enqueues its all-reduce in ticket order (``chm_comm_set_ticket``): several lanes per rank,
    each with its own communicator and host thread, number their steps identically on every rank so that the collectives reach the devices
    in one order everywhere."""
    _lib.check(_lib.lib().chm_comm_set_ticket(self._h, int(ticket)))

  @staticmethod
  def reset_tickets(next_ticket=0):
    _lib.check(_lib.lib().chm_comm_ticket_reset(int(next_ticket)))

  @staticmethod
  def skip_ticket(ticket):
    """Forfeit a ticket (``chm_comm_ticket_skip``): the step that carried it will not reach its collective -- call it from the ``except`` /
    ``finally`` of a failed step so that the other lanes' higher tickets are served instead of waiting for the timeout."""
    _lib.check(_lib.lib().chm_comm_ticket_skip(int(ticket)))

  @staticmethod
  def ticket_timeout(seconds):
    """How long a ticketed evaluation waits for the lower tickets before it fails (``chm_comm_ticket_timeout``; default 120 s).  A step that
    times out raises once and forfeits its ticket."""
    _lib.check(_lib.lib().chm_comm_ticket_timeout(int(round(1e3 * float(seconds)))))

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
  """Fallback communicator for hosts where RCCL cannot bring up a communicator: the ``3 * nbatch`` partial sums travel over the
  :class:`Rendezvous` sockets on the host and the combination (likelihood.py:298-316, selection_function.py:38-47) is formed by
  :func:`combine_partials` instead of ``k_combine``.  Same sharding, same sums; only the transport differs (an extra D2H of 3
  doubles per draw per call).  ``handle`` is None: ``chm_eval`` then runs without a communicator and returns the shard's partials."""
  handle = None

  def __init__(self, nranks, rank, device=None, rendezvous=None):
    self.nranks, self.rank = int(nranks), int(rank)
    self.device = _lib.default_device() if device is None else int(device)
    self._own = rendezvous is None
    self.rendezvous = Rendezvous(self.nranks, self.rank) if rendezvous is None else rendezvous

  def allreduce_sum(self, x):
    return self.rendezvous.allreduce_sum(_lib.as_f64(x))

  def close(self):
    if self._own and self.rendezvous is not None:
      self.rendezvous.close()
    self.rendezvous = None


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


# ----------------------------------------------------------------------------------------------------------
# the 'both' scheme: groups of ranks, each group a 'data'-sharded replica that takes its own slice of the draws of a batch
# ----------------------------------------------------------------------------------------------------------
def group_layout(nranks, rank, ngroups):
  """(group id, rank inside the group, size of the group) of ``rank`` when ``nranks`` ranks form ``ngroups`` groups of consecutive ranks, the first
  ``nranks % ngroups`` groups one rank larger -- the reference's split of MPI processes into parameter batches (CHIMERA/parallel.py:136-156)."""
  nranks, rank, ngroups = int(nranks), int(rank), int(ngroups)
  if not (1 <= ngroups <= nranks):
    raise ValueError("the number of parameter batches must be between 1 and the number of ranks")      # parallel.py:133-134
  base, rem = divmod(nranks, ngroups)
  first = 0
  for g in range(ngroups):
    size = base + 1 if g < rem else base
    if first <= rank < first + size:
      return g, rank - first, size
    first += size
  raise ValueError(f"rank {rank} outside a job of {nranks} ranks")


def draws_of_group(n, ngroups, group):
  """[i0, i1): the draws of a batch of ``n`` that parameter batch ``group`` evaluates -- n // ngroups each, the first n % ngroups batches one more
  (CHIMERA/parallel.py:311-321, 384-394)."""
  per, rem = divmod(int(n), int(ngroups))
  i0 = group * per + min(group, rem)
  return i0, i0 + per + (1 if group < rem else 0)


def _group_address(address, group):
  """where the ranks of one group meet: the job's address with the group number appended (Unix socket) / added to the port (TCP)"""
  if isinstance(address, str):
    return f"{address}.g{group}"
  return (address[0], int(address[1]) + 1 + int(group))


def _split_address(address, ngroups):
  """where the ranks of an RCCL world meet to split it (distinct from the job's own address and from every group address)"""
  if isinstance(address, str):
    return f"{address}.split"
  return (address[0], int(address[1]) + 1 + int(ngroups))


def split(world, ngroups, rendezvous=None):
  """A COLLECTIVE over ``world``: the communicator of this rank's group under the 'both' scheme (the reference's ``comm.Split(color=batch_id,
  key=rank)``, CHIMERA/parallel.py:149-156).  The result is what the selection function and the likelihood of the group are built on
  (``selection_function(..., comm=group)``, ``hyperlikelihood(..., comm=group, scheme='both')``): inside a group events and injections are
  sharded as in the 'data' scheme; the groups share the draws of a batch.  It remembers ``world`` (the values of all draws are assembled over
  it), ``group_id`` and ``ngroups``.

  RCCL worlds: rank 0 creates one unique id per group; the ids travel over ``rendezvous`` (a temporary one at the job's default address when
  none is passed).  Host-socket worlds (:class:`HostComm`): every group meets at its own address derived from the world's."""
  g, r, n = group_layout(world.nranks, world.rank, ngroups)
  if getattr(world, 'handle', None) is None and hasattr(world, 'rendezvous'):       # HostComm
    sub = HostComm(n, r, device=world.device, rendezvous=Rendezvous(n, r, address=_group_address(world.rendezvous.address, g)))
    sub._own = True
    world.rendezvous.barrier()                                 # every group is connected before anybody goes on
  else:
    own = rendezvous is None
    #  the temporary rendezvous meets at an address of its own -- the job's default address may still be held by the job's
    # rendezvous: a second listener would unlink a live Unix socket, and fail with EADDRINUSE on TCP while the other ranks wait on the old one
    rdzv = Rendezvous(world.nranks, world.rank, address=_split_address(default_address(), ngroups)) if own else rendezvous
    try:
      ids = rdzv.broadcast_bytes(b''.join(new_unique_id() for _ in range(int(ngroups))) if world.rank == 0 else None)
    finally:
      if own:
        rdzv.barrier()
        rdzv.close()
    sub = Comm(n, r, device=world.device, unique_id=ids[128 * g:128 * (g + 1)])
  sub.world, sub.group_id, sub.ngroups = world, g, int(ngroups)
  return sub


def new_unique_id():
  buf = C.create_string_buffer(128)
  _lib.check(_lib.lib().chm_comm_unique_id(buf))
  return buf.raw


def exchange_unique_id(nranks, rank, rendezvous=None):
  """Rank 0 creates the RCCL unique id; the 128 bytes reach the other ranks over the rendezvous sockets (a temporary
  :class:`Rendezvous` at the default address when none is passed)."""
  if nranks == 1:
    return new_unique_id()
  own = rendezvous is None
  rdzv = Rendezvous(nranks, rank) if own else rendezvous
  try:
    return rdzv.broadcast_bytes(new_unique_id() if rank == 0 else None)
  finally:
    if own:
      rdzv.barrier()                                          # every rank holds the id before the hub goes away
      rdzv.close()
