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


# ---------------------------------------------------------------------------------------------
# Feature files.  The reference stores one HDF5 per utterance with datasets
# wave, hubert, logmel, loud, est_lf0_score / gt_lf0_score, f0, midi (preprocess.py:567-611).  The same keys in
# a NumPy ".npz" archive are accepted too (handy where h5py is not installed).
def read_feats(path, key):
    if path.endswith(".npz"):
        if not os.path.exists(path):
            logging.error(f"There is no such a feature file ({path}).")
            return None
        with np.load(path) as f:
            if key not in f:
                logging.error(f"There is no such a data in feature file. ({key})")
                return None
            return f[key]
    return read_hdf5(path, key)


def write_feats(path, key, data):
    if path.endswith(".npz"):
        old = {}
        if os.path.exists(path):
            with np.load(path) as f:
                old = {k: f[k] for k in f.files}
        old[key] = np.asarray(data)
        np.savez(path, **old)
        return
    write_hdf5(path, key, data)


def find_files(root_dir, query="*.h5", include_root_dir=True):
    """serenade/utils/utils.py:27-43."""
    import fnmatch
    files = []
    for root, _, filenames in os.walk(root_dir, followlinks=True):
        for filename in fnmatch.filter(filenames, query):
            files.append(os.path.join(root, filename))
    if not include_root_dir:
        files = [f.replace(root_dir + "/", "") for f in files]
    return files


def write_wav_pcm16(path, wave, sr):
    """`soundfile.write(path, wave, sr, "PCM_16")` (ssc_decode.py:361-366,449-455).  Without the soundfile package
    the stdlib `wave` module writes the same samples: libsndfile converts normalised floats to PCM_16 as
    `lrint(x * 0x7FFF)` (round-half-even, scale 32767); values outside [-1, 1] are clipped here instead of wrapping
    (the generator ends in tanh, so converted audio never gets there)."""
    x = np.asarray(wave)
    try:
        import soundfile as sf
        sf.write(path, x, sr, "PCM_16")
        return
    except ImportError:
        pass
    import wave as _wave
    if x.dtype.kind == "f":
        x = np.clip(np.rint(x.astype(np.float64) * 32767.0), -32768, 32767).astype("<i2")
    else:
        x = x.astype("<i2")
    with _wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(sr))
        f.writeframes(x.tobytes())


def read_wav(path):
    """`soundfile.read(path)` for the files this pipeline writes: (float64 samples in [-1, 1), sampling rate).
    PCM_16 is normalised by 32768 as libsndfile does (ssc_postprocessing.py:143)."""
    try:
        import soundfile as sf
        return sf.read(path)
    except ImportError:
        pass
    import wave as _wave
    with _wave.open(path, "rb") as f:
        if f.getsampwidth() != 2:
            raise RuntimeError(f"{path}: only PCM_16 wav files are readable without the soundfile package")
        sr, ch = f.getframerate(), f.getnchannels()
        x = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2").astype(np.float64) / 32768.0
    return (x.reshape(-1, ch) if ch > 1 else x), sr
