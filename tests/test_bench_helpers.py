"""bench.py's bookkeeping (no GPU): the algorithmic-bytes yardstick of SURVEY.md §8(d) and the LDS-atomic floor that
DESIGN.md §4 reports next to it."""
import pytest

import bench


def test_algorithmic_bytes_per_visibility():
    # 40 B stream + per tap 16 B kernel read + 16 B grid read + 16 B grid write
    assert bench.alg_bytes_per_vis(15) == 10840
    assert bench.alg_bytes_per_vis(7) == 2392
    assert bench.alg_bytes_per_vis(1) == 88


@pytest.mark.parametrize("S,cycles", [
    (8, 16),        # 64 taps: one full step, two 8-cycle instructions
    (16, 64),       # 256 taps: four full steps
    (15, 56.25),    # 225 taps: three full steps, a 32-tap step shared by two visibilities, 1 tap per block of 64
    (7, 16),        # 49 taps: four 16-lane groups -> 8 cycles per instruction
    (5, 12),        # 25 taps: two groups -> 6 cycles
    (11, 32),       # 121 taps: 64 + 57
    (13, 46),       # 169 taps: 2 x 64 + 41 (three groups: 7 cycles)
    (17, 72.25),    # 289 taps: four full steps, a shared 32-tap step, 1 tap per block (parts of 3 + 2 steps)
    (21, 112),      # 441 taps: 6 x 64 + 57
    (25, 160),      # 625 taps: 9 x 64 + 49
    (31, 240.25),   # 961 taps: 15 x 64, the last tap once per block of 64 records
    (32, 256),
])
def test_lds_atomic_cycles_per_visibility(S, cycles):
    assert bench.lds_atomic_cycles_per_vis(S) == cycles


def test_workloads_are_the_baseline_configs():
    assert bench.WORKLOADS["cfg3"] == (100_000_000, 4096, 128, 8, 15)
    assert bench.WORKLOADS["cfg2"] == (1_000_000, 2048, 16, 8, 7)
    assert bench.WORKLOADS["cfg5"] == (125_000_000, 8192, 128, 8, 15)  # 10^9 visibilities over 8 GPUs
    assert bench.WORKLOADS["cfg4"][1:] == (4096, 128, 8, 15)


def test_compulsory_bytes_per_visibility():
    # SURVEY.md §8(d) B_min: 40 + (32 N^2 + 16 W Q^2 S^2) / n  -> 45.7 B/vis in the headline configuration
    assert bench.compulsory_bytes_per_vis(10**8, 4096, 128, 8, 15) == pytest.approx(45.66, abs=0.01)


def test_lds_atomic_peak_matches_the_cycle_floor():
    # the byte-rate form of the bound (roofline.achieved / peak) and the cycle form (lds_floor_ms) agree where
    # the tap count packs perfectly into 64-lane instructions
    S, n, cus, clk = 16, 10**8, 256, 2.1
    floor_ms = n * bench.lds_atomic_cycles_per_vis(S) / cus / (clk * 1e9) * 1e3
    peak = cus * bench.LDS_ATOMIC_B_PER_CLK * clk  # GB/s
    assert 2 * S * S * 8 * n / (peak * 1e9) * 1e3 == pytest.approx(floor_ms)


def test_traffic_is_only_quoted_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    import json
    import os
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "ska-sdp-accelerate-gridding_amd" / "csrc")
    src = tmp_path / "ska-sdp-accelerate-gridding_amd" / "csrc" / "tile_k.hip"
    src.write_text("// v1\nint f(int x) { return x + 1; }\n")
    fp = bench.csrc_fingerprint()
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps({"cfg3": {"hbm_bytes_per_launch": 1e9, "csrc_sha16": fp}}))
    assert bench.committed_traffic("cfg3")[0] == 1e9
    src.write_text("// another comment /* and */\nint f(int x)   {\n    return x + 1;  /* same code */\n}\n")
    assert bench.committed_traffic("cfg3")[0] == 1e9      # comments and white space do not count
    src.write_text("// v1\nint f(int x) { return x + 2; }\n")
    assert bench.committed_traffic("cfg3")[0] is None
    assert bench.committed_traffic("cfg2")[0] is None


def test_counter_based_stream_is_the_same_everywhere():
    """a counter-based generator, identical on CPU and GPU, so that any rank can draw any range of the
    one global stream.  The numpy uint64 twin reproduces torch's int64 arithmetic bit for bit, a range drawn on its
    own equals that range of the whole stream, and the stream has the properties the bench relies on (every tap in
    range, mirrored to v >= 0, w-bins inside the table)."""
    import numpy as np
    import torch
    seed = 0x5EEDC0DE
    k = np.arange(12345, 12345 + 4096)
    for j in range(8):
        a = bench.counter_uniform_numpy(seed, k, j)
        b = bench.counter_uniform(seed, torch.from_numpy(k), j).numpy()
        assert np.array_equal(a, b) and a.min() >= 0.0 and a.max() < 1.0
    N, W, S = 4096, 128, 15
    dev = torch.device("cpu")
    whole = bench.synth_vis(5000, N, W, S, seed, dev)
    for lo, hi in ((0, 1250), (1250, 3750), (3750, 5000)):
        part = bench.synth_vis(hi - lo, N, W, S, seed, dev, lo=lo)
        for x, y in zip(whole, part):
            assert torch.equal(x[lo:hi], y)
    u, v, wb, vis = whole
    m = (S / 2 + 1) / N
    assert u.abs().max() <= 0.5 - m and v.min() >= 0.0 and v.max() <= 0.5 - m
    assert wb.min() >= 0 and wb.max() <= W - 1 and len(torch.unique(wb)) > 100
    # u from the twin: (U - 0.5) * (1 - 2 m), mirrored where v < 0 - single IEEE operations, so bit-exact on any device
    pu = (bench.counter_uniform_numpy(seed, np.arange(5000), 0) - 0.5) * (1 - 2 * m)
    pv = (bench.counter_uniform_numpy(seed, np.arange(5000), 1) - 0.5) * (1 - 2 * m)
    assert np.array_equal(np.where(pv < 0, -pu, pu), u.numpy()) and np.array_equal(np.abs(pv), v.numpy())
    other = bench.synth_vis(100, N, W, S, seed + 1, dev)
    assert not torch.equal(other[0], u[:100])
    core = bench.synth_vis(2000, N, W, S, seed, dev, dist="core")
    assert core[0].std() < 0.12 and core[1].min() >= 0.0


def test_expected_checksum_is_what_the_oracle_grids(oracle):
    """The bench line's self-check: sum(G) of a correct convgrid2 equals sum_k vis_k * sum_ij K[slice_k] when every tap
    is in range.  Here the C oracle is the gridder; the mirrored stream leaves the rows below mirrored_first_row zero."""
    import numpy as np
    import torch
    from gridhip.distributed import mirrored_first_row
    N, W, Q, S, n = 256, 8, 4, 7, 20000
    dev = torch.device("cpu")
    u, v, wb, vis = bench.synth_vis(n, N, W, S, 99, dev)
    gcf = bench.synth_kernels(W, Q, S, dev)
    G = np.zeros((N, N), dtype=np.complex128)
    oracle.convgrid2(gcf.numpy(), G, u.numpy(), v.numpy(), wb.numpy(), vis.numpy())
    expect, scale = bench.expected_checksum(u, v, wb, vis, gcf, N)
    assert abs(G.sum() - complex(expect.item())) / scale.item() < 1e-13
    y0 = mirrored_first_row(N, S)
    assert y0 == N // 2 - S // 2 - 1 and not G[:y0].any() and G[y0 + 1].any()
    # a visibility short, or a wrong slice, is seen
    e2, _ = bench.expected_checksum(u[:-1], v[:-1], wb[:-1], vis[:-1], gcf, N)
    assert abs(G.sum() - complex(e2.item())) / scale.item() > 1e-8


def test_plain_command_spawns_the_ranks():
    """`python bench.py --gpus 2` as a plain command starts the two rank processes itself (before any GPU call) and
    relays rank 0's line; --dry-run keeps the rendezvous on gloo so this runs on a CPU box."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, bench.__file__, "--gpus", "2", "--dry-run", "--workload", "cfg5"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rank_sum"] == 3.0 and rec["vis_per_gpu"] == 125_000_000


def test_a_failing_rank_ends_the_job():
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    # without --dry-run the ranks need GPUs; on a CPU box every rank exits non-zero and so must the parent
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    out = subprocess.run([sys.executable, bench.__file__, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
