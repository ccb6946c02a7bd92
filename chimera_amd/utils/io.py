"""I/O helpers (reference: CHIMERA/utils/io.py:7-66).  Same functions; files are ``.npz`` (always available) or HDF5 when
``h5py`` can be imported (the reference's Zenodo products are HDF5).  Group members are stored in ``.npz`` as ``group/key``."""
import numpy as np

try:                                    # optional: not present in the build image
  import h5py
except Exception:                       # pragma: no cover
  h5py = None


def _is_h5(fname):
  return str(fname).endswith(('.h5', '.hdf5'))


def _need_h5py(fname):
  if h5py is None:
    raise ImportError(f"{fname}: reading/writing HDF5 needs h5py, which is not installed; use .npz")


def save_set(obj, dir_file, attrs=[], datasets=[], groups=[]):
  """io.py:7-18."""
  if _is_h5(dir_file):
    _need_h5py(dir_file)
    with h5py.File(dir_file, 'w') as f:
      for a in attrs:
        f.attrs[a] = getattr(obj, a)
      for d in datasets:
        f.create_dataset(d, data=np.asarray(getattr(obj, d)))
      for g in groups:
        dg = f.create_group(g)
        for k, v in getattr(obj, g).items():
          dg.create_dataset(k, data=v)
    return
  out = {}
  for a in attrs:
    out['attr/' + a] = np.asarray(getattr(obj, a))
  for d in datasets:
    out[d] = np.asarray(getattr(obj, d))
  for g in groups:
    for k, v in (getattr(obj, g) or {}).items():
      out[f'{g}/{k}'] = np.asarray(v)
  np.savez(dir_file, **out)


def load_set(obj, dir_file, attrs=[], datasets=[], groups=[]):
  """io.py:20-41: returns a new object for the immutable theta_* containers, updates mutable objects in place."""
  new_fields = {}
  if _is_h5(dir_file):
    _need_h5py(dir_file)
    with h5py.File(dir_file, 'r') as f:
      for a in attrs:
        new_fields[a] = f.attrs[a]
      for d in datasets:
        new_fields[d] = np.array(f[d][:])
      for g in groups:
        new_fields[g] = {k: np.array(f[g][k][:]) for k in f[g].keys()}
  else:
    with np.load(dir_file, allow_pickle=False) as f:
      for a in attrs:
        v = f['attr/' + a]
        new_fields[a] = v[()] if v.ndim == 0 else v
      for d in datasets:
        new_fields[d] = f[d]
      for g in groups:
        new_fields[g] = {k[len(g) + 1:]: f[k] for k in f.files if k.startswith(g + '/')}
  if hasattr(obj, '_fields') and hasattr(obj, 'update'):
    return obj.update(**new_fields)
  for k, v in new_fields.items():
    setattr(obj, k, v)
  return obj


def load_data_h5(fname, group_h5=None, backend='numpy', require_keys=None):
  """io.py:44-66 (also reads ``.npz`` with optional ``group/`` prefixes)."""
  data = {}
  if _is_h5(fname):
    _need_h5py(fname)
    with h5py.File(fname, 'r') as f:
      target = f if group_h5 is None else f[group_h5]
      keys = list(target.keys())
      if require_keys:
        missing = [k for k in require_keys if k not in keys]
        if missing:
          raise ValueError(f"Missing required keys in {fname}: {missing}")
      for key in keys:
        data[key] = np.array(target[key][:])
    return data
  with np.load(fname, allow_pickle=False) as f:
    prefix = '' if group_h5 is None else group_h5 + '/'
    keys = [k[len(prefix):] for k in f.files if k.startswith(prefix)] if prefix else list(f.files)
    if prefix and not keys:            # flat file without group prefixes
      prefix, keys = '', list(f.files)
    if require_keys:
      missing = [k for k in require_keys if k not in keys]
      if missing:
        raise ValueError(f"Missing required keys in {fname}: {missing}")
    for key in keys:
      data[key] = f[prefix + key]
  return data
