"""Seeded fuzz: random grid shapes, kernel shapes, oversampling, plane counts, distributions and
tuning options; convgrid2 and degrid2 through the C ABI against the CPU oracle (1e-10 relative).
Catches shape-dependent indexing mistakes (LDS pitch, tile offsets, halo clipping, chunk/batch
boundaries) that the fixed-shape parity tests could miss."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def make_case(seed):
    rng = np.random.default_rng(1000 + seed)
    H = int(rng.integers(17, 400))
    Wd = H if rng.random() < 0.5 else int(rng.integers(17, 400))
    square = rng.random() < 0.7
    gh = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 11, 13, 15, 17, 21]))
    gw = gh if square else int(rng.choice([1, 3, 4, 5, 7, 9, 12, 15, 19]))
    Q = int(rng.choice([1, 2, 3, 4, 8]))
    W = int(rng.choice([1, 2, 5, 8, 13, 32]))
    n = int(rng.choice([1, 7, 300, 5000, 40000, 120000]))
    spread = float(rng.choice([0.3, 0.5, 0.56]))
    if rng.random() < 0.3:
        u = np.clip(rng.normal(0, 0.05, n), -0.7, 0.7)
        v = np.clip(rng.normal(0, 0.05, n), -0.7, 0.7)
    else:
        u, v = rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n)
    gcf = rng.normal(size=(W, Q, Q, gh, gw)) + 1j * rng.normal(size=(W, Q, Q, gh, gw))
    wb = rng.integers(0, W, n)
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    opts = {}
    if rng.random() < 0.6:
        opts["tile"] = int(rng.choice([8, 16, 32, 64]))
    if rng.random() < 0.6:
        opts["block"] = int(rng.choice([64, 128, 256, 512, 1024]))
    if rng.random() < 0.5:
        opts["wgroups"] = int(rng.choice([1, 2, 3, 8]))
    if rng.random() < 0.5:
        opts["chunk"] = int(rng.choice([64, 100, 1000, 5000]))
    opts["sort"] = int(rng.choice([0, 1, 2]))
    opts["prepass"] = int(rng.choice([0, 1, 2]))  # one- or two-level scatter in the binning pre-pass
    return (H, Wd, gcf, u, v, wb, vis, opts)


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_convgrid2_and_degrid2(ctx, oracle, seed):
    H, Wd, gcf, u, v, wb, vis, opts = make_case(seed)
    keys = ("tile", "block", "wgroups", "chunk", "sort", "prepass")
    G0 = np.zeros((H, Wd), dtype=np.complex128)
    ref = oracle.convgrid2(gcf, G0.copy(), u, v, wb, vis)
    rng = np.random.default_rng(seed)
    Gd = rng.normal(size=(H, Wd)) + 1j * rng.normal(size=(H, Wd))
    dref = oracle.degrid2(gcf, Gd, u, v, wb)
    try:
        try:
            for k, val in opts.items():
                ctx.set_option(k, val)
            got = ctx.convgrid2(gcf, G0.copy(), (u, v, None), wb, vis)
            dgot = ctx.degrid2(gcf, Gd, (u, v, None), wb)
        except Exception as e:  # a forced tile that cannot fit its halo in LDS is a legal refusal
            if "LDS" in str(e) or "unsupported" in str(e).lower():
                pytest.skip(f"shape refused with these options: {e}")
            raise
    finally:
        for k in keys:
            ctx.set_option(k, 0)
    assert ctx.get_option("errors") == 0
    assert rel(got, ref) < TOL, (H, Wd, gcf.shape, len(u), opts)
    if np.abs(dref).max() > 0:
        assert rel(dgot, dref) < TOL, (H, Wd, gcf.shape, len(u), opts)
