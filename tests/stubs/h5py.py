"""Minimal stand-in for h5py used ONLY by the CPU tests of qpnet_amd.loaders.read_hdf5 / write_hdf5 (h5py is not installed
in the build image).  One file = one pickled {dataset path: ndarray} dict; only what those wrappers call is provided."""
import os
import pickle

import numpy as np


class _Dataset:
    def __init__(self, arr):
        self._a = arr
        self.shape = arr.shape

    def __getitem__(self, key):
        return self._a if key == () else self._a[key]


class File:
    def __init__(self, name, mode="r"):
        self.name, self.mode = name, mode
        self.data = {}
        if mode in ("r", "r+", "a") and os.path.exists(name):
            with open(name, "rb") as f:
                self.data = pickle.load(f)
        elif mode in ("r", "r+"):
            raise OSError("Unable to open file (%s)" % name)

    @staticmethod
    def _k(path):
        return "/" + path.strip("/")

    def __contains__(self, path):
        return self._k(path) in self.data

    def __getitem__(self, path):
        return _Dataset(self.data[self._k(path)])

    def __delitem__(self, path):
        del self.data[self._k(path)]

    def create_dataset(self, path, data=None):
        self.data[self._k(path)] = np.array(data)

    def flush(self):
        if self.mode != "r":
            with open(self.name, "wb") as f:
                pickle.dump(self.data, f)

    def close(self):
        self.flush()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
