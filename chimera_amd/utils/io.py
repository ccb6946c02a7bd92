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


def _flatten(obj, attrs, datasets, groups):
  """(attributes, arrays keyed by their path in the file) of a set: ``name`` for a dataset, ``group/key`` for a member of a dict-valued field."""
  arrays = {d: np.asarray(getattr(obj, d)) for d in datasets}
  for g in groups:
    arrays.update({f'{g}/{k}': np.asarray(v) for k, v in (getattr(obj, g) or {}).items()})
  return {a: getattr(obj, a) for a in attrs}, arrays


def save_set(obj, dir_file, attrs=[], datasets=[], groups=[]):
  """io.py:7-18: the listed attributes, datasets and dict-valued fields (one HDF5 group each) of ``obj`` in one file.  Both back ends write the
  same flat (path -> array) map: HDF5 creates the groups from the paths, ``.npz`` keeps the paths as names and the attributes under ``attr/``."""
  meta, arrays = _flatten(obj, attrs, datasets, groups)
  if not _is_h5(dir_file):
    np.savez(dir_file, **{'attr/' + a: np.asarray(v) for a, v in meta.items()}, **arrays)
    return
  _need_h5py(dir_file)
  with h5py.File(dir_file, 'w') as f:
    f.attrs.update(meta)
    for g in groups:                                          # (also when the field is empty: readers look the group up)
      f.require_group(g)
    for path, value in arrays.items():
      f[path] = value


def load_set(obj, dir_file, attrs=[], datasets=[], groups=[]):
  """io.py:20-41: returns a new object for the immutable theta_* containers, updates mutable objects in place."""
  if _is_h5(dir_file):
    _need_h5py(dir_file)
    with h5py.File(dir_file, 'r') as f:
      new_fields = {a: f.attrs[a] for a in attrs}
      new_fields.update({d: f[d][()] for d in datasets})
      new_fields.update({g: ({k: member[()] for k, member in f[g].items()} if g in f else {}) for g in groups})
  else:
    with np.load(dir_file, allow_pickle=False) as f:
      new_fields = {a: (f['attr/' + a][()] if f['attr/' + a].ndim == 0 else f['attr/' + a]) for a in attrs}
      new_fields.update({d: f[d] for d in datasets})
      new_fields.update({g: {k[len(g) + 1:]: f[k] for k in f.files if k.startswith(g + '/')} for g in groups})
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
