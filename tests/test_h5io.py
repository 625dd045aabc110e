"""Native HDF5 I/O (libgridhip_io.so) — the §8(f) "I/O" row.  CPU tests: the round trip of the
reference's test/Hdf5.hs:36-58 with the arrays of :20-28, the exported symbols, error reporting,
and interchange with the reference's own shim built from /root/reference/hdf5/hdf5.cc into
oracle/_ref/ (skipped where that build is absent, e.g. never required on the GPU box).
GPU test: synthetic HDF5 dataset -> aw_gridding driver (src/ImageDataset.hs:29-86) -> /img,
checked against the numpy oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from gridhip import h5io

REF = os.path.join(ROOT, "oracle", "_ref", "libhdf5ref.so")


def ref_arrays():
    d = np.arange(2000, dtype=np.float64).reshape(1000, 2)                      # testData
    c = (np.arange(1000) * (1 + 1j)).astype(np.complex128)                      # testDataC
    t3 = (np.arange(12000) * (1 + 1j)).astype(np.complex128).reshape(2, 3, 1000, 2)  # test3
    return d, c, t3


def test_exports_the_symbols_hdf5_hs_binds():
    hdr = open(os.path.join(ROOT, "include", "gridhip_io.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(\w+)\s*\(", hdr)) - {"defined"}
    lib = C.CDLL(h5io.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s
    assert declared == set(h5io.SYMBOLS)


def test_round_trip_like_reference_testIO(tmp_path):
    name = str(tmp_path / "test")  # no extension: ".h5" is appended, as the reference does
    d, c, t3 = ref_arrays()
    h5io.createh5File(name)
    h5io.createDataset(name, "/testD", d)
    h5io.createDataset(name, "/testC", c)
    h5io.createDataset(name, "/test3", t3)
    assert os.path.exists(name + ".h5")
    assert h5io.shape(name, "/test3") == (2, 3, 1000, 2)
    assert np.array_equal(h5io.readDataset(name, "/testD", np.float64), d)
    assert np.array_equal(h5io.readDataset(name + ".h5", "/testC", np.complex128), c)
    assert np.array_equal(h5io.readDataset(name, "/test3", np.complex128), t3)
    ints = np.arange(-5, 5, dtype=np.int64)
    h5io.createDataset(name, "/grp/sub/ints", ints)  # intermediate groups are created
    assert np.array_equal(h5io.readDataset(name, "/grp/sub/ints", np.int64), ints)
    assert sorted(h5io.listGroupMembers(name, "/")) == ["grp", "test3", "testC", "testD"]
    stack = h5io.readDatasets(name, ["/testC", "/testC"], np.complex128)
    assert stack.shape == (2, 1000) and np.array_equal(stack[1], c)


def test_errors_are_reported_not_swallowed(tmp_path):
    name = str(tmp_path / "missing")
    with pytest.raises(h5io.H5Error):
        h5io.readDataset(name, "/x", np.float64)
    h5io.createh5File(name)
    with pytest.raises(h5io.H5Error):
        h5io.shape(name, "/nope")
    with pytest.raises(h5io.H5Error):
        h5io.listGroupMembers(name, "/nogroup")


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref (reference hdf5.cc build) not present")
def test_interchange_with_reference_shim(tmp_path):
    """Files written by either shim are read identically by the other."""
    ref = h5io.load(REF)
    d, c, t3 = ref_arrays()
    a, b = str(tmp_path / "mine"), str(tmp_path / "theirs")
    h5io.createh5File(a)
    h5io.createDataset(a, "/testD", d)
    h5io.createDataset(a, "/testC", c)
    h5io.createDataset(a, "/test3", t3)
    h5io.createh5File(b, lib=ref)
    h5io.createDataset(b, "/testD", d, lib=ref)
    h5io.createDataset(b, "/testC", c, lib=ref)
    h5io.createDataset(b, "/test3", t3, lib=ref)
    for name, lib in ((a, ref), (b, None)):
        assert h5io.shape(name, "/test3", lib) == (2, 3, 1000, 2)
        assert np.array_equal(h5io.readDataset(name, "/testD", np.float64, lib), d)
        assert np.array_equal(h5io.readDataset(name, "/testC", np.complex128, lib), c)
        assert np.array_equal(h5io.readDataset(name, "/test3", np.complex128, lib), t3)
        assert sorted(h5io.listGroupMembers(name, "/", lib)) == ["test3", "testC", "testD"]
        st = h5io.readDatasets(name, ["/testC", "/testC"], np.complex128, lib)
        assert np.array_equal(st[0], c) and np.array_equal(st[1], c)


def test_dataset_schema_round_trip(tmp_path):
    from gridhip import dataset
    visf, wf, af = dataset.write_synthetic_dataset(str(tmp_path / "syn"), n=50, nant=3, nw=4, Q=2, S=7)
    wk, wb = dataset.getWKernels(wf, 0.008)
    assert wk.shape == (4, 2, 2, 7, 7) and np.all(np.diff(wb) > 0)
    ak = dataset.getAKernels(af, 0.008, 58000.25, 1.0e8)
    assert ak.shape == (3, 7, 7)
    assert dataset.readVis(visf).shape == (50,) and dataset.readBaselines(visf).shape == (50, 3)
    a1, a2, t, f = dataset.readSource(visf)
    assert a1.dtype == np.int64 and f == 1.0e8
    assert dataset.findClosestList([1.0, 2.0, 4.0, 8.0], 3.1) == (4.0, 2)


@pytest.mark.gpu
def test_aw_gridding_driver_on_synthetic_hdf5(tmp_path, ctx):
    """Replacement for BASELINE config 1 (the LFS data files are absent): HDF5 in, /img out."""
    from gridhip import dataset
    from oracle import gridref_np as P
    theta, lam = 0.008, 8000  # N = 64
    visf, wf, af = dataset.write_synthetic_dataset(str(tmp_path / "syn"), n=300, nant=4, nw=5, Q=2, S=15,
                                                   theta=theta, lam=lam, seed=3)
    out = str(tmp_path / "out")
    img, mx = dataset.aw_gridding(ctx, wf, af, visf, outfile=out, theta=theta, lam=lam)
    assert np.array_equal(h5io.readDataset(out, "/img", np.float64), img)
    # oracle for the same chain
    vis, uvw = dataset.readVis(visf), dataset.readBaselines(visf)
    a1, a2, ts, f = dataset.readSource(visf)
    wk, wb = dataset.getWKernels(wf, theta)
    ak = dataset.getAKernels(af, theta, ts[0], f)
    p = dataset.uvw_lambda(f, uvw)
    N = P.haskell_round(theta * lam)
    wt = P.doweight(N, p[:, 0] / lam, p[:, 1] / lam, np.ones(len(vis), dtype=np.complex128))
    u1, v1, w1, vis1 = P.mirror_uvw(p[:, 0], p[:, 1], p[:, 2], vis)
    wbin = np.array([P.find_closest(wb, x) for x in w1])
    G = P.awgrid(wk, ak, np.zeros((N, N), dtype=np.complex128), u1 / lam, v1 / lam, wbin, a1, a2, vis1 * wt)
    ref = np.real(P.ifft_c(P.make_grid_hermitian(G)))
    assert np.abs(img - ref).max() / np.abs(ref).max() < 1e-10
    assert abs(mx - ref.max()) <= 1e-10 * abs(ref.max())
