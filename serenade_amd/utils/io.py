"""HDF5 helpers with the semantics of serenade/utils/utils.py:46-116 (read_hdf5 / write_hdf5).

h5py is an optional dependency (it is not in the build image); the functions raise a clear
error when it is missing instead of silently doing something else."""
import logging
import os

import numpy as np


def _h5py():
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - depends on the image
        raise RuntimeError("h5py is required to read/write .h5 feature / stats files") from e
    return h5py


def read_hdf5(hdf5_name, hdf5_path):
    """Return the dataset as ndarray, or None (after logging an error) when file/key is missing."""
    if not os.path.exists(hdf5_name):
        logging.error(f"There is no such a hdf5 file ({hdf5_name}).")
        return None
    h5py = _h5py()
    with h5py.File(hdf5_name, "r") as f:
        if hdf5_path not in f:
            logging.error(f"There is no such a data in hdf5 file. ({hdf5_path})")
            return None
        return f[hdf5_path][()]


def write_hdf5(hdf5_name, hdf5_path, write_data, is_overwrite=True):
    write_data = np.array(write_data)
    folder = os.path.dirname(hdf5_name)
    if folder and not os.path.exists(folder):
        os.makedirs(folder)
    h5py = _h5py()
    mode = "r+" if os.path.exists(hdf5_name) else "w"
    with h5py.File(hdf5_name, mode) as f:
        if hdf5_path in f:
            if not is_overwrite:
                logging.error("Dataset in hdf5 file already exists. if you want to overwrite, set is_overwrite = True.")
                return
            logging.warning("Dataset in hdf5 file already exists. recreate dataset in hdf5.")
            del f[hdf5_path]
        f.create_dataset(hdf5_path, data=write_data)
