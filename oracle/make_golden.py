#!/usr/bin/env python3
"""Generate tests/golden/*.npz (small fixtures; data only) — CPU ORACLE tooling, not product code.

Provenance of each fixture:

  brokennumbers.npz   RECORDED by the reference: inputs are the literals of old/BrokenNumbers.hs:47-48
                      (`vcomplex`), expected grid is the interpreter output printed at :86-91.
  brokennumbers_real.npz  RECORDED by the reference: the real-valued twin of the above (`vdouble`, old/BrokenNumbers.hs:50-51),
                      expected grid is the output printed at :101-106 (`test2`, interpreter and CPU backend agree).
  fixbounds.npz       DERIVED: inputs are the literals of test/GridTesting.hs:365-387 (`testFixbounds`: the ten points
                      scattered four times, with offsets (0,0), (1,1), (1,0), (0,1), through fixoutofboundsOLD :428-439);
                      expected grid evaluated here from that definition with plain Python loops.  It is convgrid with a
                      2 x 2 kernel of ones.
  fixbounds2.npz      DERIVED: inputs are the literals of test/GridTesting.hs:389-426 (`testFixbounds2`),
                      expected grid evaluated here from that test's own definition with plain Python loops.
  smalltest_aw.npz    DERIVED: inputs are the literals of test/SmallTest.hs:51-76; expected grid from the
                      numpy restatement (FFT path of convolve2d, as the reference computes it).
  convgrid2_small.npz DERIVED: seeded random case; expected grid from the numpy restatement.
  wkernel_*.npz       DERIVED: parameter sets of test/GridTesting.hs:85-93,139-150 (theta=0.1, Q=2,
                      npixFF=256, S=31, w in {100,1000}); expected kernels from the numpy restatement.

Nothing is read from /root/reference at run time; the literals above are restated here.
Run:  python oracle/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import gridref_np as P  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def brokennumbers():
    xs = np.array([(2 * x) % 5 for x in range(10)], dtype=np.int64)
    ys = np.array([(3 * x + 1) % 5 for x in range(10)], dtype=np.int64)
    val = np.array([complex(x + 5, 1.0) for x in range(10)])
    expected = np.zeros((5, 5), dtype=np.complex128)
    # old/BrokenNumbers.hs:86-91 (interpreter output `test1`), rows top to bottom
    expected[0, 1] = 42 + 4j
    expected[1, 0] = 30 + 4j
    expected[2, 4] = 38 + 4j
    expected[3, 3] = 46 + 4j
    expected[4, 2] = 34 + 4j
    np.savez(os.path.join(OUT, "brokennumbers.npz"), x=xs, y=ys, val=val, passes=np.int64(2), expected=expected)


def brokennumbers_real():
    xs = np.array([(2 * x) % 5 for x in range(10)], dtype=np.int64)
    ys = np.array([(3 * x + 1) % 5 for x in range(10)], dtype=np.int64)
    val = np.array([float(x + 5) for x in range(10)])
    expected = np.zeros((5, 5), dtype=np.float64)
    # old/BrokenNumbers.hs:101-106 (`test2`), rows top to bottom
    expected[0, 1] = 42.0
    expected[1, 0] = 30.0
    expected[2, 4] = 38.0
    expected[3, 3] = 46.0
    expected[4, 2] = 34.0
    np.savez(os.path.join(OUT, "brokennumbers_real.npz"), x=xs, y=ys, val=val, passes=np.int64(2), expected=expected)


def fixbounds():
    x = np.array([(2 * k) % 5 for k in range(10)], dtype=np.int64)
    y = np.array([(3 * k + 1) % 5 for k in range(10)], dtype=np.int64)
    vis = np.array([complex(k + 5, 1.0) for k in range(10)])
    G = np.zeros((5, 5), dtype=np.complex128)
    for offy, offx in ((0, 0), (1, 1), (1, 0), (0, 1)):      # fullrepli i j: offsety = i, offsetx = j (:373-386)
        for k in range(10):
            idx, idy = x[k] + offx, y[k] + offy
            if idx < 0 or idy < 0 or idx >= 5 or idy >= 5:
                continue  # fixoutofboundsOLD: (-offx, -offy, 0) -> adds 0 to G[0, 0]
            G[idy, idx] += vis[k]
    # the same through convgrid: a [1,1,2,2] kernel of ones, coordinates whose footprint origin (cell - gw/2) is (x, y)
    pu = (x + 1 - 2) / 5.0
    pv = (y + 1 - 2) / 5.0
    np.savez(os.path.join(OUT, "fixbounds.npz"), x=x, y=y, vis=vis, gcf=np.ones((1, 1, 2, 2), dtype=np.complex128),
             pu=pu, pv=pv, expected=G)


def fixbounds2():
    x = np.array([(2 * k) % 5 for k in range(10)], dtype=np.int64)
    xf = np.array([k % 2 for k in range(10)], dtype=np.int64)
    y = np.array([(3 * k + 2) % 5 for k in range(10)], dtype=np.int64)
    yf = np.array([k % 2 for k in range(10)], dtype=np.int64)
    vis = np.array([complex(k + 5, 1.0) for k in range(10)])
    gcf = np.array([complex(k, k) for k in range(16)]).reshape(2, 2, 2, 2)
    G = np.zeros((5, 5), dtype=np.complex128)
    for k in range(10):
        for i in range(2):
            for j in range(2):
                xx, yy = x[k] + j, y[k] + i
                if xx < 0 or yy < 0 or xx >= 5 or yy >= 5:
                    continue  # fixoutofbounds: value 0 added to G[0,0]
                G[yy, xx] += vis[k] * gcf[yf[k], xf[k], i, j]
    # coordinates that make convgrid's frac_coords (Q=2, N=5, minus gw/2=1) produce the integers above
    pu = (x + 1 + xf / 2.0 - 2) / 5.0
    pv = (y + 1 + yf / 2.0 - 2) / 5.0
    np.savez(os.path.join(OUT, "fixbounds2.npz"), x=x, xf=xf, y=y, yf=yf, vis=vis, gcf=gcf, pu=pu, pv=pv,
             expected=G)


def smalltest_aw():
    S = 15
    xx = np.arange(S, dtype=np.float64)
    wk = np.broadcast_to(xx[None, :] * (0.01 + 0.005j) + 0.1, (S, S)).astype(np.complex128)
    wkerns = wk.reshape(1, 1, 1, S, S).copy()
    akerns = np.full((3, S, S), 0.1 + 0j)
    u = np.array([0.1, -0.1])
    v = np.array([0.2, 0.4])
    w = np.array([0.3, 0.1])
    wbin = np.array([0, 0], dtype=np.int64)
    a1 = np.array([0, 0], dtype=np.int64)
    a2 = np.array([1, 2], dtype=np.int64)
    vis = np.array([0.3 + 0.5j, 0.4 + 0.2j])
    G = np.zeros((10, 10), dtype=np.complex128)
    P.awgrid(wkerns, akerns, G, u, v, wbin, a1, a2, vis)
    np.savez(os.path.join(OUT, "smalltest_aw.npz"), wkerns=wkerns, akerns=akerns, u=u, v=v, w=w, wbin=wbin,
             a1=a1, a2=a2, vis=vis, expected=G)


def convgrid2_small():
    rng = np.random.default_rng(20261004)
    N, W, Q, S, n = 32, 2, 2, 5, 60
    gcf = rng.normal(size=(W, Q, Q, S, S)) + 1j * rng.normal(size=(W, Q, Q, S, S))
    u = rng.uniform(-0.56, 0.56, n)
    v = rng.uniform(-0.56, 0.56, n)
    wbin = rng.integers(0, W, n)
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    G = np.zeros((N, N), dtype=np.complex128)
    P.convgrid2(gcf, G, u, v, wbin, vis)
    d = P.degrid2(gcf, G, u, v, wbin)
    np.savez(os.path.join(OUT, "convgrid2_small.npz"), gcf=gcf, u=u, v=v, wbin=wbin, vis=vis, expected=G,
             degrid=d)


def wkernels():
    for w in (100.0, 1000.0):
        k = P.w_kernel(0.1, w, 256, 31, 2)
        np.savez(os.path.join(OUT, f"wkernel_w{int(w)}.npz"), theta=0.1, w=w, npixFF=256, npixKern=31, qpx=2,
                 expected=k)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    brokennumbers()
    brokennumbers_real()
    fixbounds()
    fixbounds2()
    smalltest_aw()
    convgrid2_small()
    wkernels()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
