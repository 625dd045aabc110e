"""ctypes binding of libgridhip_io.so (include/gridhip_io.h): the native HDF5 I/O that replaces the
reference's hdf5/hdf5.cc behind the unchanged src/Hdf5.hs interface.  Function names follow
src/Hdf5.hs:69-205 (createh5File, readDataset*, createDataset*, listGroupMembers)."""
import ctypes as C
import os

import numpy as np

from ._lib import ROOT

LIB_PATH = os.path.join(ROOT, "lib", "libgridhip_io.so")
_lib = None

SYMBOLS = ["createh5File", "getRankDataset", "getDimsDataset", "readDatasetInt", "readDatasetLLong",
           "readDatasetDouble", "readDatasetComplex", "readDatasetsDouble", "readDatasetsComplex",
           "createDatasetInt", "createDatasetLLong", "createDatasetDouble", "createDatasetComplex",
           "listGroupMembers", "h5io_last_error", "h5io_free_list"]


class H5Error(RuntimeError):
    pass


def load(path=None):
    global _lib
    if path is None and _lib is not None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError(f"{p} not found: build it with __graft_entry__.build()")
    lib = C.CDLL(p)
    lib.getRankDataset.restype = C.c_int
    lib.listGroupMembers.restype = C.POINTER(C.c_char_p)
    if hasattr(lib, "h5io_last_error"):
        lib.h5io_last_error.restype = C.c_char_p
    if path is None:
        _lib = lib
    return lib


def _b(s):
    # the reference's shim appends ".h5" IN PLACE: always hand over a buffer with room for it
    return C.create_string_buffer(s.encode(), len(s.encode()) + 8)


def _check(lib):
    if hasattr(lib, "h5io_last_error"):
        e = lib.h5io_last_error()
        if e:
            raise H5Error(e.decode())


def createh5File(name, lib=None):
    lib = lib or load()
    lib.createh5File(_b(name))
    _check(lib)


def shape(name, dataset, lib=None):
    lib = lib or load()
    rank = lib.getRankDataset(_b(name), _b(dataset))
    _check(lib)
    if rank < 0:
        raise H5Error(f"no dataset {dataset}")
    dims = (C.c_int * max(rank, 1))()
    lib.getDimsDataset(_b(name), _b(dataset), rank, dims)
    _check(lib)
    return tuple(int(dims[i]) for i in range(rank))


_READ = {np.dtype(np.float64): "readDatasetDouble", np.dtype(np.complex128): "readDatasetComplex",
         np.dtype(np.int64): "readDatasetLLong", np.dtype(np.int32): "readDatasetInt"}
_WRITE = {np.dtype(np.float64): "createDatasetDouble", np.dtype(np.complex128): "createDatasetComplex",
          np.dtype(np.int64): "createDatasetLLong", np.dtype(np.int32): "createDatasetInt"}


def readDataset(name, dataset, dtype, lib=None):
    """src/Hdf5.hs:113-137 — shape from the file, element type from the caller."""
    lib = lib or load()
    out = np.empty(shape(name, dataset, lib), dtype=dtype)
    getattr(lib, _READ[np.dtype(dtype)])(_b(name), _b(dataset), out.ctypes.data_as(C.c_void_p))
    _check(lib)
    return out


def readDatasets(name, datasets, dtype, lib=None):
    """src/Hdf5.hs:139-167 — equally shaped datasets stacked along a new leading axis."""
    lib = lib or load()
    sh = shape(name, datasets[0], lib)
    out = np.empty((len(datasets),) + sh, dtype=dtype)
    bufs = [_b(d) for d in datasets]
    arr = (C.c_char_p * (len(datasets) + 1))(*[C.cast(b, C.c_char_p) for b in bufs], None)
    fn = "readDatasetsComplex" if np.dtype(dtype) == np.complex128 else "readDatasetsDouble"
    getattr(lib, fn)(_b(name), arr, out.ctypes.data_as(C.c_void_p))
    _check(lib)
    return out


def createDataset(name, dataset, array, lib=None):
    """src/Hdf5.hs:169-205"""
    lib = lib or load()
    a = np.ascontiguousarray(array)
    dims = (C.c_int * max(a.ndim, 1))(*a.shape)
    getattr(lib, _WRITE[a.dtype])(_b(name), _b(dataset), a.ndim, dims, a.ctypes.data_as(C.c_void_p))
    _check(lib)


def listGroupMembers(name, group, lib=None):
    """src/Hdf5.hs:105-111"""
    lib = lib or load()
    p = lib.listGroupMembers(_b(name), _b(group))
    _check(lib)
    out = []
    i = 0
    while p and p[i]:
        out.append(p[i].decode())
        i += 1
    if hasattr(lib, "h5io_free_list"):
        lib.h5io_free_list.argtypes = [C.POINTER(C.c_char_p)]
        lib.h5io_free_list(p)
    return out
