"""Parity of the HIP kernels (through the C-ABI, via caspr_amd.ops / caspr_amd.models) against the CPU
oracle on identical seeded inputs.  Integer outputs (FPS / ball-query / three-NN indices) must be
bit-exact; floating-point outputs within the tolerance written next to each check (north_star: 1e-5 abs
for T-NOCS and CNF-sampled xyz).  Every measured error is also appended to gpurun_out/parity_report.json.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import model as O
from oracle import point_ops as P
from caspr_amd.utils.synthetic import car_sequences, dense_sequences

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REPORT = {}


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).normal(0, 1, shape) * scale).astype(np.float32))


def record(name, got, want, tol):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    want = want.detach().cpu().double().numpy() if torch.is_tensor(want) else np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    err = float(np.abs(got - want).max()) if got.size else 0.0
    REPORT[name] = {"max_abs_err": err, "tol": tol, "ref_absmax": float(np.abs(want).max()) if want.size else 0.0}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)
    assert np.isfinite(got).all(), "%s: non-finite output" % name
    assert err <= tol, "%s: max abs err %.3e > %.1e (|ref|max %.3e)" % (name, err, tol, REPORT[name]["ref_absmax"])


def record_f64(name, got, want32, want64, tol=1e-5, fixture_dist=None):
    """The pipeline-level criterion: |hip - f64| <= tol, FLAT, against the f64 evaluation of the same graph (north_star: 1e-5 abs on
    T-NOCS / sampled xyz; the f32 oracle's own distance from f64 is recorded next to it -- on sparse inputs the reference's f32
    arithmetic itself is 4e-5 ... 1e-3 away, duplicate-padded neighbourhoods amplify f32 rounding inside GroupNorm by up to
    1/sqrt(eps)).  No check carries a slack any more (round 5: neighbourhoods of 2..4 distinct samples are evaluated in f64).
    fixture_dist: `want32` is a golden fixture of the REAL reference (tests/golden/reference_golden.npz) and fixture_dist its distance
    from the f64 oracle as recorded when the fixture was pinned (profiles/r04_parity_report.json).  Then the chain
    hip ~ f64-oracle ~ fixture is asserted link by link ON THIS BOX: |f64-oracle - fixture| <= 2 x fixture_dist (the f64 oracle
    still is the evaluation of the graph the reference ran, to the reference's own f32 error) and |hip - fixture| <= that + tol."""
    def a(t):
        return t.detach().cpu().double().numpy() if torch.is_tensor(t) else np.asarray(t, dtype=np.float64)
    got, want32, want64 = a(got), a(want32), a(want64)
    e_gpu, e_ref, e_direct = float(np.abs(got - want64).max()), float(np.abs(want32 - want64).max()), float(np.abs(got - want32).max())
    REPORT[name] = {"max_abs_err_vs_f64": e_gpu, "oracle32_vs_f64": e_ref, "max_abs_err_vs_oracle32": e_direct, "bound": tol}
    if fixture_dist is not None:
        REPORT[name].update({"fixture_vs_f64_recorded": fixture_dist, "fixture_vs_f64_bound": 2 * fixture_dist, "hip_vs_fixture_bound": 2 * fixture_dist + tol})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)
    assert np.isfinite(got).all(), "%s: non-finite output" % name
    assert e_gpu <= tol, "%s: |hip-f64| %.3e > %.2e (|oracle32-f64| %.3e)" % (name, e_gpu, tol, e_ref)
    if fixture_dist is not None:
        assert e_ref <= 2 * fixture_dist, "%s: |f64 oracle - reference fixture| %.3e > 2 x the pinned %.3e" % (name, e_ref, fixture_dist)
        assert e_direct <= 2 * fixture_dist + tol, "%s: |hip - reference fixture| %.3e > %.3e" % (name, e_direct, 2 * fixture_dist + tol)


def exact(name, got, want):
    got, want = got.cpu().numpy(), want.cpu().numpy()
    bad = int((got != want).sum())
    REPORT[name] = {"mismatches": bad, "count": int(want.size)}
    assert bad == 0, "%s: %d / %d entries differ" % (name, bad, want.size)


@pytest.fixture(autouse=True)
def _inference_mode():
    """This file checks the inference kernels: like the reference's callers (test.py:124,141) everything runs under
    torch.no_grad() -- with grad mode on, eval() models take the taped, differentiable path (tests/test_hip_train.py)."""
    with torch.no_grad():
        yield


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the -m gpu tests need a ROCm GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from caspr_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def model(dev, seeded_sd):
    from caspr_amd.models import CaSPR
    m = CaSPR(cnf_rk4_steps=8, latent_rk4_steps=4)   # the step counts the golden fixtures were generated with
    m.load_state_dict(seeded_sd)
    return m.to(dev).eval()


def clouds(B, n, seed=0, dup=False):
    x, _ = car_sequences(B, 1, n, seed=100 + seed)
    c = x[:, 0, :, :3].contiguous()
    if dup:  # dataset padding duplicates leading points (caspr_dataset.py:188-195) -> exact ties
        c[:, n // 2:] = c[:, : n - n // 2]
    return c


# ---------------------------------------------------------------------------------------------
# index operators: bit-exact
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,M,dup", [(2048, 1024, False), (1024, 512, False), (512, 256, True), (256, 64, False),
                                    (64, 16, False), (512, 1024, False), (4096, 1024, False), (100, 37, True)])
def test_fps_bit_exact(ops, dev, n, M, dup):
    c = clouds(3, n, seed=n + M, dup=dup)
    want = P.furthest_point_sampling(c, M)
    got, new_xyz = ops.furthest_point_sampling(c.to(dev), M, return_xyz=True)
    exact("fps_idx_n%d_M%d" % (n, M), got, want)
    exact("fps_newxyz_n%d_M%d" % (n, M), new_xyz, torch.gather(c, 1, want.long().unsqueeze(-1).expand(-1, -1, 3)))


@pytest.mark.parametrize("n,M,dup", [(5000, 300, False), (8192, 512, True), (20000, 64, False), (36864, 16, False)])
def test_fps_bit_exact_beyond_4096_points(ops, dev, n, M, dup):
    """Clouds larger than any configuration of the reference (the running minimum moves from registers to LDS)."""
    c = clouds(2, n, seed=n + M, dup=dup)
    want = P.furthest_point_sampling(c, M)
    got, new_xyz = ops.furthest_point_sampling(c.to(dev), M, return_xyz=True)
    exact("fps_big_idx_n%d_M%d" % (n, M), got, want)
    exact("fps_big_newxyz_n%d_M%d" % (n, M), new_xyz, torch.gather(c, 1, want.long().unsqueeze(-1).expand(-1, -1, 3)))
    with pytest.raises(Exception):
        ops.furthest_point_sampling(torch.zeros(1, 36865, 3, device=dev), 4)


def test_fps_guard_and_origin(ops, dev):
    c = clouds(2, 256, seed=5)
    c[:, 10:40] = 0.0           # padded origin points are skipped by the guard
    c[0, 0] = 0.0               # even the start index may be guarded
    for guard in (True, False):
        exact("fps_guard%d" % guard, ops.furthest_point_sampling(c.to(dev), 64, guard=guard), P.furthest_point_sampling(c, 64, guard=guard))


def test_fps_tie_order_tree_reduction(ops, dev, golden):
    """Exact ties: duplicates at an offset that is not a multiple of the 512-thread block (dataset padding,
    caspr_dataset.py:188-195) and grid clouds where every round ties.  The winner follows the upstream shared-memory
    tree (bit-reversed k mod blockDim, then k) -- fixture produced by oracle/point_ops.c, itself checked against a
    literal restatement of the block algorithm in tests/test_oracle_golden.py."""
    x, _ = car_sequences(1, 2, 1024, seed=1234)
    dup = x.reshape(2, 1024, 4)[:, :, :3].contiguous()
    dup[:, 700:1024] = dup[:, 37:361].clone()
    got = ops.furthest_point_sampling(dup.to(dev), 1024)
    exact("fps_dup_offset700_vs_golden", got, torch.from_numpy(golden["ops_fps_idx_dup"]))
    for n, M in [(8, 8), (23, 23), (100, 100), (700, 300), (2048, 600), (4096, 512)]:
        rng = np.random.default_rng(n)
        pts = torch.from_numpy((rng.integers(0, 6, (2, n, 3)) / 4.0 + np.array([1.0, 0.5, 2.0])).astype(np.float32))
        exact("fps_grid_ties_n%d" % n, ops.furthest_point_sampling(pts.to(dev), M), P.furthest_point_sampling(pts, M))


@pytest.mark.parametrize("n,M,r,ns", [(2048, 1024, 0.02, 16), (2048, 1024, 0.05, 32), (1024, 512, 0.1, 32), (256, 64, 0.4, 32),
                                     (64, 16, 0.8, 32), (64, 16, 0.4, 16), (300, 50, 0.2, 16),
                                     # the LDS kernel's edges (csrc/point_ops.hip: ball_query_lds_kernel): a ragged last workgroup / wave, a
                                     # cloud above 64 KB (LDS opt-in), the largest cloud it takes, and the first one it leaves to the plain kernel
                                     (2048, 100, 0.05, 32), (8192, 1024, 0.03, 32), (12288, 128, 0.02, 16), (12289, 128, 0.02, 16)])
def test_ball_query_bit_exact(ops, dev, n, M, r, ns):
    c = clouds(2, n, seed=n, dup=(n == 300))
    ctr = torch.gather(c, 1, P.furthest_point_sampling(c, M).long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    exact("ball_n%d_r%g_ns%d" % (n, r, ns), ops.ball_query(r, ns, c.to(dev), ctr.to(dev)), P.ball_query(r, ns, c, ctr))


@pytest.mark.parametrize("n,M,ra,nsa,rb,nsb", [(2048, 1024, 0.02, 16, 0.05, 32), (1024, 512, 0.05, 16, 0.1, 32), (512, 256, 0.2, 32, 0.1, 16), (256, 64, 0.2, 16, 0.4, 32),
                                               (64, 16, 0.4, 16, 0.8, 32),          # below the LDS kernel's shapes: two plain launches behind the same entry
                                               (2048, 100, 0.05, 32, 0.05, 32), (8192, 1024, 0.01, 16, 0.03, 32), (300, 64, 0.1, 16, 0.2, 16)])
def test_ball_query_pair_is_the_two_queries(ops, dev, n, M, ra, nsa, rb, nsb):
    """caspr_ball_query2_f32 (round 6: both scales of a set-abstraction level in one pass over the cloud, pointnet2.py:338-342,391): each
    scale's rows bit-identical to its own caspr_ball_query_f32 call AND to the oracle -- car clouds, duplicate padding (exact ties, full
    balls: a scale that has its ns hits stops recording while the other goes on), radii in either order, equal radii, a ragged last wave."""
    c = clouds(2, n, seed=n + 1, dup=(n == 300))
    ctr = torch.gather(c, 1, P.furthest_point_sampling(c, M).long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    ga, gb = ops.ball_query_pair(ra, nsa, rb, nsb, c.to(dev), ctr.to(dev))
    exact("ball_pair_a_n%d_r%g_ns%d" % (n, ra, nsa), ga, ops.ball_query(ra, nsa, c.to(dev), ctr.to(dev)))
    exact("ball_pair_b_n%d_r%g_ns%d" % (n, rb, nsb), gb, ops.ball_query(rb, nsb, c.to(dev), ctr.to(dev)))
    exact("ball_pair_a_oracle_n%d" % n, ga, P.ball_query(ra, nsa, c, ctr))
    exact("ball_pair_b_oracle_n%d" % n, gb, P.ball_query(rb, nsb, c, ctr))


def test_ball_query_empty_ball(ops, dev):
    c = clouds(1, 128)
    ctr = torch.tensor([[[10.0, 10.0, 10.0], [c[0, 5, 0], c[0, 5, 1], c[0, 5, 2]]]])
    exact("ball_empty", ops.ball_query(0.05, 16, c.to(dev), ctr.to(dev)), P.ball_query(0.05, 16, c, ctr))


def test_group_and_gather(ops, dev):
    c = clouds(2, 512, seed=3)
    feat = rnd(1, 2, 512, 8)
    idx = P.furthest_point_sampling(c, 128)
    ctr = torch.gather(c, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    bidx = P.ball_query(0.2, 16, c, ctr)
    want = P.group(c, ctr, feat.transpose(1, 2).contiguous(), bidx)
    got = ops.group_points(c.to(dev), ctr.to(dev), feat.to(dev), bidx.to(dev))
    exact("group_points", got, want)
    exact("gather_points", ops.gather_points(feat.to(dev), idx.to(dev)), torch.gather(feat, 1, idx.long().unsqueeze(-1).expand(-1, -1, 8)))


@pytest.mark.parametrize("n,m", [(2048, 1024), (1024, 512), (256, 64), (64, 16), (77, 33)])
def test_three_nn(ops, dev, n, m):
    c = clouds(2, n, seed=n, dup=(n == 77))
    kn = c[:, :m].contiguous() if n != 77 else clouds(2, m, seed=9)
    d, i = P.three_nn(c, kn)
    gd, gi, gw = ops.three_nn(c.to(dev), kn.to(dev), with_weights=True)
    exact("three_nn_idx_%d_%d" % (n, m), gi, i)
    record("three_nn_dist_%d_%d" % (n, m), gd, d, 1e-7)
    inv = 1.0 / (d + 1e-8)
    record("three_nn_weight_%d_%d" % (n, m), gw, inv / inv.sum(dim=2, keepdim=True), 1e-6)


def test_three_interpolate(ops, dev):
    c = clouds(2, 256, seed=1)
    kn = c[:, :64].contiguous()
    d, i = P.three_nn(c, kn)
    inv = 1.0 / (d + 1e-8)
    w = (inv / inv.sum(dim=2, keepdim=True)).contiguous()
    feat = rnd(2, 2, 64, 512)
    skip = rnd(3, 2, 256, 8)
    want = P.three_interpolate(feat.transpose(1, 2).contiguous(), i, w).transpose(1, 2)
    got = ops.three_interpolate(feat.to(dev), i.to(dev), w.to(dev), skip=skip.to(dev), skip_channels=6)
    assert got.shape == (2, 256, 520)
    record("three_interp", got[:, :, :512], want, 1e-6)
    exact("three_interp_skip", got[:, :, 512:518], skip[:, :, :6])
    assert float(got[:, :, 518:].abs().max()) == 0.0
    # with the producer's GroupNorm+ReLU folded into the load
    sc, sh = rnd(4, 2, 512).abs() + 0.5, rnd(5, 2, 512)
    fa = torch.relu(feat * sc.unsqueeze(1) + sh.unsqueeze(1))
    want = P.three_interpolate(fa.transpose(1, 2).contiguous(), i, w).transpose(1, 2)
    got = ops.three_interpolate(feat.to(dev), i.to(dev), w.to(dev), in_scale=sc.to(dev), in_shift=sh.to(dev), in_relu=True)
    record("three_interp_lazy", got, want, 2e-6)


def test_chamfer(ops, dev):
    p, q = clouds(3, 2048, seed=1), clouds(3, 1500, seed=2)
    d1, d2 = P.chamfer(p, q)
    g1, g2 = ops.chamfer_distance(p.to(dev), q.to(dev))
    exact("chamfer_d1", g1, d1)
    exact("chamfer_d2", g2, d2)
    z1, _ = ops.chamfer_distance(p.to(dev), p.to(dev))
    assert float(z1.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------
# MFMA pointwise conv + GroupNorm statistics
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,P_,Cin,Cout", [(2, 300, 4, 64), (2, 256, 64, 128), (1, 130, 128, 1024), (3, 64, 1536, 512),
                                          (2, 200, 518, 512), (1, 256, 576, 1600), (1, 384, 1600, 1600), (2, 100, 1600, 4),
                                          (1, 20, 1600, 3078), (1, 1, 1024, 1600),
                                          # streaming kernel (P >= 128, Cin >= 192): narrow outputs, ragged rows and K, one wave's worth of rows + 1
                                          (2, 300, 1600, 4), (1, 129, 200, 130), (2, 260, 1539, 3), (3, 128, 192, 16),
                                          # narrow kernel (P >= 128, Cin < 192): every row-tile count, ragged K / rows / outputs, two slabs
                                          (2, 1000, 9, 32), (1, 300, 131, 64), (2, 257, 99, 96), (1, 129, 191, 130), (2, 200, 16, 16),
                                          (1, 640, 32, 48), (2, 128, 3, 5)])
def test_conv1x1(ops, dev, B, P_, Cin, Cout):
    ldx = (Cin + 3) // 4 * 4
    x = torch.zeros(B, P_, ldx)
    x[:, :, :Cin] = rnd(Cin + Cout, B, P_, Cin)
    x[:, :, Cin:] = 7.0  # padding columns must be ignored
    w = rnd(1, Cout, Cin, scale=1.0 / np.sqrt(Cin))
    b = rnd(2, Cout, scale=0.1)
    want = (x[:, :, :Cin].double() @ w.double().t() + b.double())
    pw = ops.PackedWeight(w.to(dev))
    got = ops.conv1x1(pw, b.to(dev), x.to(dev))
    assert got.shape == (B, P_, (Cout + 3) // 4 * 4)
    record("conv1x1_%dx%d" % (Cin, Cout), got[:, :, :Cout], want, 2e-6 * max(1.0, float(want.abs().max())))


def test_conv1x1_fused_input_and_epilogues(ops, dev):
    B, P_, Cin, Cout = 2, 333, 576, 200
    x = rnd(1, B, P_, Cin)
    w, b = rnd(2, Cout, Cin, scale=0.05), rnd(3, Cout, scale=0.1)
    sc, sh, bb = rnd(4, B, Cin).abs() + 0.5, rnd(5, B, Cin), rnd(6, B, Cout)
    xin = x * sc.unsqueeze(1) + sh.unsqueeze(1)
    xin[:, :, 512:] = torch.relu(xin[:, :, 512:])
    want = torch.sigmoid(xin.double() @ w.double().t() + b.double() + bb.double().unsqueeze(1))
    pw = ops.PackedWeight(w.to(dev))
    buf = torch.zeros(B, P_, 256, device=dev)
    got = ops.conv1x1(pw, b.to(dev), x.to(dev), bbias=bb.to(dev), in_scale=sc.to(dev), in_shift=sh.to(dev), in_relu=True,
                      in_relu_from=512, act=1, out=buf[:, :, 32:232])
    record("conv1x1_fused", got, want, 2e-6)
    assert float(buf[:, :, :32].abs().max()) == 0.0 and float(buf[:, :, 232:].abs().max()) == 0.0
    # column-sliced packing == packing the slice
    pw2 = ops.PackedWeight(torch.cat([torch.zeros(Cout, 1), w], 1).to(dev), col0=1)
    assert torch.equal(pw2.data, pw.data)


def test_conv1x1_narrow_epilogues_and_batch_invariance(ops, dev):
    """The narrow kernel (the training encoder's set-abstraction convs): per-entry bias + sigmoid epilogue, a strided output, and an
    entry's rows giving the same bits whatever the batch around them."""
    B, P_, Cin, Cout = 3, 777, 99, 80
    x, w, b, bb = rnd(1, B, P_, 100), rnd(2, Cout, Cin, scale=0.1), rnd(3, Cout, scale=0.1), rnd(4, B, Cout)
    want = torch.sigmoid(x[:, :, :Cin].double() @ w.double().t() + b.double() + bb.double().unsqueeze(1))
    pw = ops.PackedWeight(w.to(dev))
    buf = torch.zeros(B, P_, 128, device=dev)
    got = ops.conv1x1(pw, b.to(dev), x.to(dev), bbias=bb.to(dev), act=1, out=buf[:, :, 16:96])
    record("conv1x1_narrow_epilogues", got, want, 2e-6)
    assert float(buf[:, :, :16].abs().max()) == 0.0 and float(buf[:, :, 96:].abs().max()) == 0.0
    one = ops.conv1x1(pw, b.to(dev), x[1:2].to(dev), bbias=bb[1:2].to(dev), act=1)
    exact("conv1x1_narrow_batch_invariance", one[:, :, :Cout], got[1:2])


def test_conv1x1_repeatable_under_load(ops, dev):
    """Many co-resident blocks, fused input transform staged through LDS: two launches must agree bit for bit
    (caught a missing barrier after the per-block scale/shift staging once)."""
    B, P_, Cin, Cout = 4, 4096, 512, 512
    x, w = rnd(1, B, P_, Cin).to(dev), rnd(2, Cout, Cin, scale=0.05).to(dev)
    sc, sh = (rnd(3, B, Cin).abs() + 0.5).to(dev), rnd(4, B, Cin).to(dev)
    pw = ops.PackedWeight(w)
    y1 = ops.conv1x1(pw, None, x, in_scale=sc, in_shift=sh, in_relu=True)
    want = torch.relu(x * sc.unsqueeze(1) + sh.unsqueeze(1)).double() @ w.double().t()
    record("conv1x1_under_load", y1, want, 2e-5)
    for _ in range(5):
        exact("conv1x1_repeat", ops.conv1x1(pw, None, x, in_scale=sc, in_shift=sh, in_relu=True), y1)


@pytest.mark.parametrize("P_,Cin,Cout", [(8, 64, 512), (8, 512, 512), (16, 512, 64), (1, 7, 3), (5, 515, 130)])
def test_conv1x1_few_rows(ops, dev, P_, Cin, Cout):
    """conv1x1 over at most 16 rows of one batch entry (the latent ODE's layers in training) takes conv1x1_skinny_kernel:
    against float64, with bias, the sigmoid epilogue, ragged widths and garbage in the input's pad columns."""
    w, b = rnd(1, Cout, Cin, scale=1.0 / np.sqrt(Cin)), rnd(2, Cout, scale=0.1)
    ldx = (Cin + 3) // 4 * 4
    xw = rnd(3, 1, P_, ldx)                       # pad columns hold noise
    x = xw[:, :, :Cin]
    pw = ops.PackedWeight(w.to(dev))
    want = x.double() @ w.double().t() + b.double()
    got = ops.conv1x1(pw, b.to(dev), xw.to(dev)[:, :, :Cin] if ldx == Cin else xw.to(dev))
    record("conv1x1_few_rows_%dx%dx%d" % (P_, Cin, Cout), got[:, :, :Cout], want, 2e-6 * max(1.0, float(want.abs().max())))
    got = ops.conv1x1(pw, None, xw.to(dev), act=1)
    record("conv1x1_few_rows_sigmoid", got[:, :, :Cout], torch.sigmoid(x.double() @ w.double().t()), 2e-6)
    exact("conv1x1_few_rows_repeat", ops.conv1x1(pw, None, xw.to(dev), act=1)[:, :, :Cout], got[:, :, :Cout])
    # a batch entry's rows give the same bits whatever the batch around them (the kernel is chosen by P, not by B)
    xb = torch.cat([rnd(9, 2, P_, ldx), xw], dim=0).to(dev)
    exact("conv1x1_few_rows_batch_invariance", ops.conv1x1(pw, None, xb, act=1)[2:, :, :Cout], got[:, :, :Cout])


@pytest.mark.parametrize("B,P_,Cin,Cout", [(2, 256, 512, 512), (1, 128, 1600, 1600), (3, 384, 608, 512), (1, 1024, 1536, 132)])
def test_conv1x1_bf16x6(ops, dev, monkeypatch, B, P_, Cin, Cout):
    """Default kernel of the large pointwise convs (csrc/gemm_bf16x6.hip): exact three-way bf16 split of both operands, six
    MFMA products, f32 accumulation.  Held to the SAME tolerance as the f32 MFMA kernel, and compared with it directly."""
    w = rnd(1, Cout, Cin + 4, scale=1.0 / np.sqrt(Cin))
    b, bb = rnd(2, Cout, scale=0.1), rnd(3, B, Cout, scale=0.1)
    xw = rnd(Cin + Cout, B, P_, Cin + 8)            # the conv reads a column slice of a wider buffer
    x = xw[:, :, 8:]
    sc, sh = rnd(4, B, Cin).abs() + 0.5, rnd(5, B, Cin)
    pw = ops.PackedWeight(w.to(dev), col0=4)
    pw32 = pw                                        # same object: the mode flag picks the kernel at call time
    assert pw.x6_ok == (Cin % 32 == 0 and Cout % 4 == 0)
    xd = xw.to(dev)[:, :, 8:]

    def conv(pw_, *a, f32=False, **k):
        prev = ops.set_matmul_mode(conv=not f32)
        try:
            return ops.conv1x1(pw_, *a, **k)
        finally:
            ops.set_matmul_mode(conv=prev[0])
    # plain
    want = x.double() @ w[:, 4:].double().t() + b.double()
    got = conv(pw, b.to(dev), xd)
    ref = conv(pw32, b.to(dev), xd, f32=True)
    assert not torch.equal(got, ref), "the two modes must run different kernels"
    tol = 2e-6 * max(1.0, float(want.abs().max()))
    record("conv1x1_bf16x6_%dx%d" % (Cin, Cout), got[:, :, :Cout], want, tol)
    e6, e32 = float((got.cpu().double()[:, :, :Cout] - want).abs().max()), float((ref.cpu().double()[:, :, :Cout] - want).abs().max())
    assert e6 <= 1.5 * e32 + 1e-7, (e6, e32)         # as close to f64 as the f32 MFMA kernel
    # fused producer GroupNorm + ReLU (from channel 64 on), per-batch bias, into a slice of a wider output
    xin = x * sc.unsqueeze(1) + sh.unsqueeze(1)
    xin[:, :, 64:] = torch.relu(xin[:, :, 64:])
    want = xin.double() @ w[:, 4:].double().t() + b.double() + bb.double().unsqueeze(1)
    buf = torch.zeros(B, P_, Cout + 8, device=dev)
    got = conv(pw, b.to(dev), xd, bbias=bb.to(dev), in_scale=sc.to(dev), in_shift=sh.to(dev), in_relu=True, in_relu_from=64,
               out=buf[:, :, 4:4 + Cout])
    record("conv1x1_bf16x6_fused_%dx%d" % (Cin, Cout), got, want, 2e-6 * max(1.0, float(want.abs().max())))
    assert float(buf[:, :, :4].abs().max()) == 0.0 and float(buf[:, :, 4 + Cout:].abs().max()) == 0.0
    exact("conv1x1_bf16x6_repeat", conv(pw, b.to(dev), xd), conv(pw, b.to(dev), xd))
    # unsupported row counts fall back to the f32 kernel
    xs = xd[:, :100].contiguous()
    exact("conv1x1_bf16x6_fallback", conv(pw, b.to(dev), xs), conv(pw32, b.to(dev), xs, f32=True))


@pytest.mark.parametrize("B,P_,C", [(2, 2500, 64), (1, 1024, 1600), (3, 64, 512), (2, 1100, 1024), (2, 333, 128)])
def test_gn_stats(ops, dev, B, P_, C):
    y = rnd(C, B, P_, C) * 2.0 + 0.7
    gamma, beta = rnd(1, C) * 0.2 + 1.0, rnd(2, C) * 0.1
    gamma[::7] *= -1.0  # negative scales exercise the min branch of the fused max
    want = F.group_norm(y.transpose(1, 2).double(), 16, gamma.double(), beta.double(), 1e-5).transpose(1, 2)
    sc, sh, pm = ops.gn_stats(y.to(dev), C, gamma.to(dev), beta.to(dev), want_max=True)
    got = y.to(dev) * sc.unsqueeze(1) + sh.unsqueeze(1)
    record("gn_apply_C%d" % C, got, want, 5e-6)
    record("gn_max_C%d" % C, pm, want.max(dim=1)[0], 5e-6)


@pytest.mark.parametrize("B,P_,Cin,Cout", [(2, 256, 512, 512), (1, 2560, 1600, 1600), (3, 384, 128, 1024), (2, 128, 544, 512), (1, 200, 512, 512), (2, 256, 64, 128)])
def test_conv1x1_gn_fused(ops, dev, B, P_, Cin, Cout):
    """conv -> GroupNorm statistics in one pass (caspr_conv1x1_gn_bf16x6_f32): the statistics taken from the conv's
    accumulators (f32 per 128-point tile, f64 across tiles) against an f64 GroupNorm of the f64 conv, at gn_stats' tolerance;
    against the two-pass form (conv1x1 + gn_stats) directly; with and without storing the output; moments for training."""
    from caspr_amd import train_ops as T
    w = rnd(1, Cout, Cin, scale=1.0 / np.sqrt(Cin))
    b, bb = rnd(2, Cout, scale=0.3), rnd(3, B, Cout, scale=0.1)
    x = rnd(Cin + Cout, B, P_, Cin)
    sc_in, sh_in = rnd(4, B, Cin).abs() + 0.5, rnd(5, B, Cin)
    gamma, beta = rnd(6, Cout) * 0.2 + 1.0, rnd(7, Cout) * 0.1
    gamma[::7] *= -1.0
    pw = ops.PackedWeight(w.to(dev))
    xin = torch.relu(x * sc_in.unsqueeze(1) + sh_in.unsqueeze(1))
    y64 = xin.double() @ w.double().t() + b.double() + bb.double().unsqueeze(1)
    want = F.group_norm(y64.transpose(1, 2), 16, gamma.double(), beta.double(), 1e-5).transpose(1, 2)
    kw = dict(bbias=bb.to(dev), in_scale=sc_in.to(dev), in_shift=sh_in.to(dev), in_relu=True)
    y, sc, sh, mean, rstd, pm = ops.conv1x1_gn(pw, b.to(dev), x.to(dev), gamma.to(dev), beta.to(dev), want_max=True, want_moments=True, **kw)
    tol_y = 2e-6 * max(1.0, float(y64.abs().max()))
    record("conv_gn_y_%dx%d" % (Cin, Cout), y[:, :, :Cout], y64, tol_y)
    got = y64.to(dev) * sc.double().unsqueeze(1) + sh.double().unsqueeze(1)        # the statistics alone, applied to the exact output
    record("conv_gn_apply_%dx%d" % (Cin, Cout), got, want, 5e-6)
    record("conv_gn_max_%dx%d" % (Cin, Cout), pm, want.max(dim=1)[0], 5e-6 + 2 * tol_y)
    m64 = y64.view(B, P_, 16, Cout // 16).transpose(1, 2).reshape(B, 16, -1)
    record("conv_gn_mean", mean, m64.mean(dim=2), 2e-6)
    record("conv_gn_rstd", rstd, 1.0 / torch.sqrt(m64.var(dim=2, unbiased=False) + 1e-5), 5e-6 * float((1.0 / torch.sqrt(m64.var(dim=2, unbiased=False) + 1e-5)).max()))
    # two-pass form on the same conv output
    y2 = ops.conv1x1(pw, b.to(dev), x.to(dev), **kw)
    if pw.x6_ok and P_ % 128 == 0:
        exact("conv_gn_y_same_kernel", y, y2)       # (narrower inputs: conv1x1 alone prefers the f32 kernel)
    sc2, sh2, mean2, rstd2, pm2 = T.gn_stats_train(y2, Cout, gamma.to(dev), beta.to(dev), want_max=True)
    record("conv_gn_scale_vs_two_pass", sc, sc2, 2e-6 * float(sc2.abs().max()))
    record("conv_gn_shift_vs_two_pass", sh, sh2, 2e-6 * max(1.0, float(sh2.abs().max())))
    # statistics only: no output
    res = ops.conv1x1_gn(pw, b.to(dev), x.to(dev), gamma.to(dev), beta.to(dev), want_max=True, write=False, **kw)
    if P_ % 128 == 0:
        assert res[0] is None
        exact("conv_gn_nowrite_scale", res[1], sc)
        exact("conv_gn_nowrite_shift", res[2], sh)
        exact("conv_gn_nowrite_max", res[3], pm)
        # a batch entry's statistics do not depend on the batch around it
        r1 = ops.conv1x1_gn(pw, b.to(dev), x[B - 1:].to(dev).contiguous(), gamma.to(dev), beta.to(dev), want_max=True, bbias=bb[B - 1:].to(dev).contiguous(),
                            in_scale=sc_in[B - 1:].to(dev).contiguous(), in_shift=sh_in[B - 1:].to(dev).contiguous(), in_relu=True)
        exact("conv_gn_batch_invariance_scale", r1[1], sc[B - 1:])
        exact("conv_gn_batch_invariance_max", r1[3], pm[B - 1:])
    else:
        assert res[0] is not None       # unsupported row count: conv1x1 + gn_stats


@pytest.mark.parametrize("B,P_,Cin,Cout", [(1, 1280, 1600, 1600), (2, 256, 512, 512), (1, 384, 1536, 512), (1, 128, 256, 1088), (3, 128, 512, 560)])
def test_conv1x1_x6w_kernel(ops, dev, B, P_, Cin, Cout):
    """The 128-point x 512-channel conv (csrc/gemm_bf16x6w.hip; a remainder of <= 64 channels on conv1x1_x6tail_kernel -- 64 of 1600 and of
    1088, 48 of 560 with an odd number of 128-point tiles -- a larger one on the 256-channel kernel): plain and
    with the producer's GroupNorm + ReLU fused, bias + per-batch bias, against the f64 contraction at the conv kernels' common
    tolerance; the GroupNorm statistics of its epilogue against the other kernel's; and it IS the kernel that ran."""
    w = rnd(1, Cout, Cin, scale=1.0 / np.sqrt(Cin))
    b, bb = rnd(2, Cout, scale=0.3), rnd(3, B, Cout, scale=0.1)
    x = rnd(Cin + Cout, B, P_, Cin)
    sc_in, sh_in = rnd(4, B, Cin).abs() + 0.5, rnd(5, B, Cin)
    gamma, beta = rnd(6, Cout) * 0.2 + 1.0, rnd(7, Cout) * 0.1
    min_cin, ops._X6W_MIN_CIN = ops._X6W_MIN_CIN, 256       # by default the host sends layers with >= 512 input channels and >= 1024 rows there
    min_rows, ops._X6W_MIN_ROWS = ops._X6W_MIN_ROWS, 128
    try:
        pw = ops.PackedWeight(w.to(dev))
    finally:
        ops._X6W_MIN_CIN = min_cin
    assert pw.x6w_ok
    kw = dict(bbias=bb.to(dev), in_scale=sc_in.to(dev), in_shift=sh_in.to(dev), in_relu=True, in_relu_from=8)
    xin = x * sc_in.unsqueeze(1) + sh_in.unsqueeze(1)
    xin[:, :, 8:] = torch.relu(xin[:, :, 8:])
    y64 = xin.double() @ w.double().t() + b.double() + bb.double().unsqueeze(1)
    prev = ops.CONV_X6W
    try:
        ops.CONV_X6W = True
        y = ops.conv1x1(pw, b.to(dev), x.to(dev), **kw)
        yp = ops.conv1x1(pw, b.to(dev), x.to(dev))
        res = ops.conv1x1_gn(pw, b.to(dev), x.to(dev), gamma.to(dev), beta.to(dev), want_max=True, want_moments=True, **kw)
        ops.CONV_X6W = False
        y_old = ops.conv1x1(pw, b.to(dev), x.to(dev), **kw)
        res_old = ops.conv1x1_gn(pw, b.to(dev), x.to(dev), gamma.to(dev), beta.to(dev), want_max=True, want_moments=True, **kw)
    finally:
        ops.CONV_X6W = prev
    tol = 2e-6 * max(1.0, float(y64.abs().max()))
    try:
        _x6w_rest(ops, dev, B, P_, Cin, Cout, pw, w, b, bb, x, sc_in, sh_in, y, yp, y_old, res, res_old, y64, tol)
    finally:
        ops._X6W_MIN_ROWS = min_rows


def _x6w_rest(ops, dev, B, P_, Cin, Cout, pw, w, b, bb, x, sc_in, sh_in, y, yp, y_old, res, res_old, y64, tol):
    record("conv_x6w_fused_%dx%d" % (Cin, Cout), y[:, :, :Cout], y64, tol)
    record("conv_x6w_plain_%dx%d" % (Cin, Cout), yp[:, :, :Cout], x.double() @ w.double().t() + b.double(), tol)
    assert not torch.equal(y, y_old), "both settings ran the same kernel"
    exact("conv_x6w_gn_output", res[0], y)
    for i, nm in ((1, "scale"), (2, "shift"), (3, "mean"), (4, "rstd"), (5, "max")):
        record("conv_x6w_gn_%s_%dx%d" % (nm, Cin, Cout), res[i], res_old[i], 3e-6 * max(1.0, float(res_old[i].abs().max())))
    # a batch entry does not depend on the batch around it
    if B > 1:
        y1 = ops.conv1x1(pw, b.to(dev), x[B - 1:].to(dev).contiguous(), bbias=bb[B - 1:].to(dev).contiguous(), in_scale=sc_in[B - 1:].to(dev).contiguous(),
                         in_shift=sh_in[B - 1:].to(dev).contiguous(), in_relu=True, in_relu_from=8)
        exact("conv_x6w_batch_invariance", y1, y[B - 1:])


def test_conv1x1_gn_fused_large_mean(ops, dev):
    """The fused statistics when a group's mean dwarfs its spread (|mean| / sigma ~ 1e3: a large bias in front of the GroupNorm,
    what a trained checkpoint may hold): the epilogue takes per-tile mean and squared deviations in two passes over the
    accumulators and the finalize kernel combines tiles pairwise in f64, so the variance keeps its digits (E[x^2] - mean^2 in f32
    would lose (mean / sigma)^2 2^-24 = 6 % of it here)."""
    B, P_, Cin, Cout = 2, 512, 512, 512
    w = rnd(1, Cout, Cin, scale=0.02 / np.sqrt(Cin))
    b = rnd(2, Cout, scale=0.01) + 20.0
    x = rnd(3, B, P_, Cin)
    gamma, beta = rnd(6, Cout) * 0.2 + 1.0, rnd(7, Cout) * 0.1
    pw = ops.PackedWeight(w.to(dev))
    y64 = x.double() @ w.double().t() + b.double()
    want = F.group_norm(y64.transpose(1, 2), 16, gamma.double(), beta.double(), 1e-5).transpose(1, 2)
    y, sc, sh = ops.conv1x1_gn(pw, b.to(dev), x.to(dev), gamma.to(dev), beta.to(dev))
    got = y64.to(dev) * sc.double().unsqueeze(1) + sh.double().unsqueeze(1)        # the statistics alone, applied to the exact output
    m64 = y64.view(B, P_, 16, Cout // 16).transpose(1, 2).reshape(B, 16, -1)
    REPORT["conv_gn_large_mean_ratio"] = {"mean_over_sigma": float((m64.mean(dim=2).abs() / m64.std(dim=2)).max())}
    assert REPORT["conv_gn_large_mean_ratio"]["mean_over_sigma"] > 500
    # f32 scale / shift (shift ~ mean * rstd ~ 1e3, one rounding = 6e-5) bound what any f32 (scale, shift) pair can deliver; a
    # variance off by 6 % would show as 3e-2 of the normalised output
    record("conv_gn_apply_large_mean", got, want, 5e-4)


# ---------------------------------------------------------------------------------------------
# the Kaolin-named operator surface
# ---------------------------------------------------------------------------------------------
def test_kaolin_compat_forward(dev):
    """caspr_amd.compat.kaolin_amd: the six symbols of pointnet2.py:7 at Kaolin's channels-first layouts against the oracle's
    restatement of the same operators -- exactly the calls PointNet2SetAbstraction.forward / PointNet2FeaturePropagator.forward
    make (pointnet2.py:384-398, 514-519)."""
    from caspr_amd.compat import kaolin_amd as K
    B, n, M, C, ns = 2, 512, 128, 7, 16                      # C not a multiple of 4: the row padding of the binding is exercised
    pts = torch.cat([clouds(B, n, seed=3), rnd(5, B, n, C)], dim=2)
    xyz, feat = K.separate_xyz_and_features(pts.to(dev))
    wxyz, wfeat = P.separate_xyz_and_features(pts)
    exact("kaolin_separate_xyz", xyz, wxyz)
    exact("kaolin_separate_feat", feat, wfeat)
    idx = K.furthest_point_sampling(xyz, M)
    widx = P.furthest_point_sampling(wxyz, M)
    exact("kaolin_fps", idx, widx)
    new_xyz = K.fps_gather_by_index(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    wnew = P.fps_gather_by_index(wxyz.transpose(1, 2).contiguous(), widx).transpose(1, 2).contiguous()
    exact("kaolin_fps_gather", new_xyz, wnew)
    grouper = K.PointNet2GroupingLayer(0.1, ns, use_xyz_feature=True, use_random_ball_query=False)
    g = grouper(xyz, new_xyz, feat)
    wg = P.group(wxyz, wnew, wfeat, P.ball_query(0.1, ns, wxyz, wnew))
    assert tuple(g.shape) == (B, M, 3 + C, ns)
    exact("kaolin_grouping_layer", g, wg)
    g0 = grouper(xyz, new_xyz, None)                          # xyz-only level (tpointnet2.py:79: PointNet++ sees xyz + 6 features, here none)
    exact("kaolin_grouping_layer_xyz_only", g0, P.group(wxyz, wnew, None, P.ball_query(0.1, ns, wxyz, wnew)))
    dist, i3 = K.three_nn(xyz, new_xyz)
    wdist, wi3 = P.three_nn(wxyz, wnew)
    exact("kaolin_three_nn_idx", i3, wi3)
    exact("kaolin_three_nn_dist", dist, wdist)
    inv = 1.0 / (dist + 1e-8)
    w = inv / inv.sum(dim=2, keepdim=True)                    # pointnet2.py:516-518
    fprev = rnd(6, B, C, M)
    out = K.three_interpolate(fprev.to(dev), i3, w)
    want = P.three_interpolate(fprev, wi3, w.cpu())
    assert tuple(out.shape) == (B, C, n)
    record("kaolin_three_interpolate", out, want, 1e-6)


def test_kaolin_compat_gradients(dev):
    """The three Kaolin operators that carry a gradient, against torch.autograd through the oracle's differentiable (plain
    indexing) route in f64."""
    from caspr_amd.compat import kaolin_amd as K
    B, n, M, C, ns = 2, 256, 64, 6, 16
    xyz = clouds(B, n, seed=4)
    idx = P.furthest_point_sampling(xyz, M)
    new_xyz = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    feat = rnd(8, B, C, n)
    with torch.enable_grad():
        # grouping layer
        f = feat.clone().to(dev).requires_grad_(True)
        R = rnd(9, B, M, 3 + C, ns)
        (K.PointNet2GroupingLayer(0.15, ns)(xyz.to(dev), new_xyz.to(dev), f) * R.to(dev)).sum().backward()
        f6 = feat.double().requires_grad_(True)
        bidx = P.ball_query(0.15, ns, xyz, new_xyz)
        (P.group(xyz.double(), new_xyz.double(), f6, bidx) * R.double()).sum().backward()
        record("kaolin_grouping_grad", f.grad, f6.grad, 1e-5 * float(f6.grad.abs().max()))
        # three_interpolate
        dist, i3 = P.three_nn(xyz, new_xyz)
        inv = 1.0 / (dist + 1e-8)
        w = inv / inv.sum(dim=2, keepdim=True)
        fp = rnd(10, B, C, M)
        a = fp.clone().to(dev).requires_grad_(True)
        R2 = rnd(11, B, C, n)
        (K.three_interpolate(a, i3.to(dev), w.to(dev)) * R2.to(dev)).sum().backward()
        a6 = fp.double().requires_grad_(True)
        (P.three_interpolate(a6, i3, w.double()) * R2.double()).sum().backward()
        record("kaolin_three_interpolate_grad", a.grad, a6.grad, 1e-5 * float(a6.grad.abs().max()))
        # fps_gather_by_index (repeated indices accumulate)
        gi = torch.cat([idx[:, :M - 4], idx[:, :4]], dim=1).contiguous()
        b = feat.clone().to(dev).requires_grad_(True)
        R3 = rnd(12, B, C, M)
        (K.fps_gather_by_index(b, gi.to(dev)) * R3.to(dev)).sum().backward()
        b6 = feat.double().requires_grad_(True)
        (P.fps_gather_by_index(b6, gi) * R3.double()).sum().backward()
        record("kaolin_gather_grad", b.grad, b6.grad, 1e-6 * float(b6.grad.abs().max()))


# ---------------------------------------------------------------------------------------------
# fused set-abstraction scale
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("level,scale", [(0, 0), (0, 1), (1, 0), (2, 1), (3, 0), (4, 1)])
def test_sa_mlp_max(dev, seeded_sd, model, level, scale):
    from caspr_amd import ops
    sa = model.encoder.local_extract.set_abstractions[level]
    n_in = [2048, 1024, 512, 256, 64][level]
    C = [6, 96, 128, 256, 512][level]
    M = sa.num_points_out
    c = clouds(2, n_in, seed=level)
    if level > 0:
        c = c * [1, 1.5, 2.0, 3.0, 4.0][level]  # coarser levels see sparser clouds
    feat = rnd(level + 7, 2, n_in, C, scale=0.7)
    idx = P.furthest_point_sampling(c, M)
    ctr = torch.gather(c, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    g = sa.grouper_modules[scale]
    bidx = P.ball_query(g.radius, g.num_samples, c, ctr)
    grouped = P.group(c, ctr, feat.transpose(1, 2).contiguous(), bidx)
    pre = "encoder.local_extract.set_abstractions.%d.pointnet_modules.%d" % (level, scale)
    want = O.feature_extractor(seeded_sd, pre, grouped.view(-1, C + 3, g.num_samples)).view(2, M, -1)
    sd64 = {k: v.double() for k, v in seeded_sd.items() if k.startswith(pre)}
    want64 = O.feature_extractor(sd64, pre, grouped.view(-1, C + 3, g.num_samples).double()).view(2, M, -1)
    ldf = (C + 3) // 4 * 4
    fpad = torch.zeros(2, n_in, ldf)
    fpad[:, :, :C] = feat
    out = torch.zeros(2, M, want.shape[2] + 8, device=dev)
    ops.sa_mlp_max(c.to(dev), ctr.to(dev), fpad.to(dev), bidx.to(dev), C, sa.pointnet_modules[scale].kernel_layers(), out, 8)
    # FLAT on every level and scale since round 5.  Rounds 3-4 kept a capped slack on the two 16-sample scales: level 1 scale 0 (2.0e-5)
    # and level 0 scale 0 (4.1e-4, the f32 reference's own error: its worst neighbourhood holds TWO distinct points, every deviation
    # column a multiple of one vector d, and a one-channel GroupNorm group whose W d cancels to ~1e-3 puts the f32 accumulation error of
    # that product through 1 / sqrt(var + eps)).  Now neighbourhoods of 2..8 distinct samples are re-evaluated in f64
    # (csrc/sa_mlp.hip: sa_repair_f64_kernel): 1.7e-6 / 4.9e-6.
    record_f64("sa_mlp_l%d_s%d" % (level, scale), out[:, :, 8:], want, want64, 1e-5)
    assert float(out[:, :, :8].abs().max()) == 0.0


@pytest.mark.parametrize("level,scale", [(2, 0), (2, 1), (3, 0), (3, 1)])
def test_sa_mlp_max_pre_aggregated(dev, seeded_sd, model, level, scale):
    """The wide levels with the first layer's feature part computed once per SOURCE point (ops.sa_mlp_max_pre, csrc/sa_mlp.hip): against
    the f64 oracle of the reference's formulation at the flat 1e-5, and against the fused call's own result."""
    from caspr_amd import ops
    sa = model.encoder.local_extract.set_abstractions[level]
    n_in = [2048, 1024, 512, 256, 64][level]
    C = [6, 96, 128, 256, 512][level]
    M = sa.num_points_out
    c = clouds(2, n_in, seed=level) * [1, 1.5, 2.0, 3.0, 4.0][level]
    feat = rnd(level + 7, 2, n_in, C, scale=0.7)
    idx = P.furthest_point_sampling(c, M)
    ctr = torch.gather(c, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    g = sa.grouper_modules[scale]
    bidx = P.ball_query(g.radius, g.num_samples, c, ctr)
    grouped = P.group(c, ctr, feat.transpose(1, 2).contiguous(), bidx)
    pre_ = "encoder.local_extract.set_abstractions.%d.pointnet_modules.%d" % (level, scale)
    want = O.feature_extractor(seeded_sd, pre_, grouped.view(-1, C + 3, g.num_samples)).view(2, M, -1)
    sd64 = {k: v.double() for k, v in seeded_sd.items() if k.startswith(pre_)}
    want64 = O.feature_extractor(sd64, pre_, grouped.view(-1, C + 3, g.num_samples).double()).view(2, M, -1)
    pn = sa.pointnet_modules[scale]
    pw_f, wx = pn.pre_layers()
    fdev = feat.to(dev).contiguous()
    pre = ops.conv1x1(pw_f, None, fdev)
    out = torch.zeros(2, M, want.shape[2] + 8, device=dev)
    ops.sa_mlp_max_pre(c.to(dev), ctr.to(dev), pre, bidx.to(dev), wx, pn.kernel_layers(), out, 8)
    record_f64("sa_mlp_pre_l%d_s%d" % (level, scale), out[:, :, 8:], want, want64, 1e-5)
    assert float(out[:, :, :8].abs().max()) == 0.0
    fused = torch.zeros(2, M, want.shape[2], device=dev)
    ldf = (C + 3) // 4 * 4
    fpad = torch.zeros(2, n_in, ldf)
    fpad[:, :, :C] = feat
    ops.sa_mlp_max(c.to(dev), ctr.to(dev), fpad.to(dev), bidx.to(dev), C, pn.kernel_layers(), fused, 0)
    record("sa_mlp_pre_vs_fused_l%d_s%d" % (level, scale), out[:, :, 8:], fused, 1e-5)


@pytest.mark.parametrize("level,scale", [(0, 0), (0, 1), (1, 0)])
def test_sa_small_balls_with_foreign_index_layouts(dev, seeded_sd, model, level, scale):
    """The f64 re-evaluation of small balls (sa_repair_f64_kernel) and the register kernel's early exit for waves made of such balls both
    recognise them by ball query's row layout (entry 0 the first hit, entries 1..K-1 the other hits, then copies of entry 0).  An index
    row with ANOTHER layout -- here: the same samples shuffled behind entry 0, which is a legal input of the grouping layer
    (pointnet2.py:340-342 takes any index tensor) -- must fall to the register kernel, not between the two: every output finite and
    equal to the oracle's on the same rows (to the register kernel's f32 accuracy on such balls; the point is that no row is skipped
    by both), and rows that keep the layout still get the f64 treatment (1e-5)."""
    from caspr_amd import ops
    sa = model.encoder.local_extract.set_abstractions[level]
    n_in, C = [2048, 1024][level], [6, 96][level]
    M = sa.num_points_out
    c = clouds(2, n_in, seed=level) * [1, 1.5][level]
    feat = rnd(level + 7, 2, n_in, C, scale=0.7)
    idx = P.furthest_point_sampling(c, M)
    ctr = torch.gather(c, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    g = sa.grouper_modules[scale]
    bidx = P.ball_query(g.radius, g.num_samples, c, ctr)
    ns = g.num_samples
    gen = torch.Generator().manual_seed(11)
    shuffled = torch.zeros(2, M, dtype=torch.bool)
    for b_ in range(2):
        for m_ in range(0, M, 3):                       # every third row: entries 1.. shuffled (entry 0 stays the reference sample)
            perm = 1 + torch.randperm(ns - 1, generator=gen)
            bidx[b_, m_, 1:] = bidx[b_, m_, perm]
            shuffled[b_, m_] = True
    grouped = P.group(c, ctr, feat.transpose(1, 2).contiguous(), bidx)
    pre = "encoder.local_extract.set_abstractions.%d.pointnet_modules.%d" % (level, scale)
    sd64 = {k: v.double() for k, v in seeded_sd.items() if k.startswith(pre)}
    want64 = O.feature_extractor(sd64, pre, grouped.view(-1, C + 3, ns).double()).view(2, M, -1)
    ldf = (C + 3) // 4 * 4
    fpad = torch.zeros(2, n_in, ldf)
    fpad[:, :, :C] = feat
    out = torch.full((2, M, want64.shape[2]), float("nan"), device=dev)
    ops.sa_mlp_max(c.to(dev), ctr.to(dev), fpad.to(dev), bidx.to(dev), C, sa.pointnet_modules[scale].kernel_layers(), out, 0)
    got = out.cpu().double()
    assert torch.isfinite(got).all(), "a neighbourhood was left to neither kernel"
    err = (got - want64).abs().amax(dim=2)
    REPORT["sa_foreign_layout_l%d_s%d" % (level, scale)] = {"shuffled_rows_max_err": float(err[shuffled].max()), "ball_query_rows_max_err": float(err[~shuffled].max())}
    assert float(err[shuffled].max()) <= 2e-3, float(err[shuffled].max())
    assert float(err[~shuffled].max()) <= 1e-5, float(err[~shuffled].max())


@pytest.mark.parametrize("level,scale", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_sa_register_kernel_is_slot_invariant(dev, seeded_sd, model, level, scale):
    """The register kernel lists, per workgroup, the neighbourhoods the f64 re-evaluation will NOT overwrite and walks that list
    (csrc/sa_mlp.hip: sa_small_entry), so the wave slot a neighbourhood is computed in depends on how many small balls precede it.
    Its arithmetic must not: with the centres (and their index rows) in another order every output row is the SAME BITS, moved."""
    from caspr_amd import ops
    sa = model.encoder.local_extract.set_abstractions[level]
    n_in, C = [2048, 1024][level], [6, 96][level]
    M = sa.num_points_out
    c = clouds(2, n_in, seed=level + 20) * [1, 1.5][level]
    feat = rnd(level + 9, 2, n_in, C, scale=0.7)
    idx = P.furthest_point_sampling(c, M)
    ctr = torch.gather(c, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    g = sa.grouper_modules[scale]
    bidx = P.ball_query(g.radius, g.num_samples, c, ctr)
    distinct = torch.tensor([[len(set(r.tolist())) for r in bb] for bb in bidx])
    ldf = (C + 3) // 4 * 4
    fpad = torch.zeros(2, n_in, ldf)
    fpad[:, :, :C] = feat
    layers = sa.pointnet_modules[scale].kernel_layers()
    cout = sa.pointnet_layer_dims_list[scale][-1]
    out = torch.full((2, M, cout), float("nan"), device=dev)
    ops.sa_mlp_max(c.to(dev), ctr.to(dev), fpad.to(dev), bidx.to(dev), C, layers, out, 0)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(5))
    out_p = torch.full((2, M, cout), float("nan"), device=dev)
    ops.sa_mlp_max(c.to(dev), ctr[:, perm].contiguous().to(dev), fpad.to(dev), bidx[:, perm].contiguous().to(dev), C, layers, out_p, 0)
    assert torch.isfinite(out).all() and torch.isfinite(out_p).all()
    assert torch.equal(out.cpu()[:, perm], out_p.cpu())
    REPORT["sa_slot_invariance_l%d_s%d" % (level, scale)] = {"neighbourhoods": int(distinct.numel()), "with_at_most_8_distinct_samples": int((distinct <= 8).sum()),
                                                             "with_at_most_4": int((distinct <= 4).sum())}


@pytest.mark.parametrize("level,scale", [(0, 1), (1, 0), (2, 0)])
def test_sa_call_in_two_halves_writes_the_same_bits(dev, seeded_sd, model, level, scale):
    """include/caspr_hip.h: CASPR_SA_ONLY_MFMA + CASPR_SA_ONLY_F64 (ops.sa_mlp_max(part=...)) together write exactly what the plain call
    writes, in either order; on the LDS kernel's shapes (level 2) the f64 half is a no-op and the MFMA half is the call."""
    from caspr_amd import ops
    sa = model.encoder.local_extract.set_abstractions[level]
    n_in, C = [2048, 1024, 512][level], [6, 96, 128][level]
    M = sa.num_points_out
    c = clouds(2, n_in, seed=level + 30) * [1, 1.5, 2.0][level]
    feat = rnd(level + 11, 2, n_in, C, scale=0.7)
    idx = P.furthest_point_sampling(c, M)
    ctr = torch.gather(c, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    g = sa.grouper_modules[scale]
    bidx = P.ball_query(g.radius, g.num_samples, c, ctr)
    ldf = (C + 3) // 4 * 4
    fpad = torch.zeros(2, n_in, ldf)
    fpad[:, :, :C] = feat
    layers = sa.pointnet_modules[scale].kernel_layers()
    cout = sa.pointnet_layer_dims_list[scale][-1]
    a = (c.to(dev), ctr.to(dev), fpad.to(dev), bidx.to(dev), C, layers)
    whole = torch.full((2, M, cout), float("nan"), device=dev)
    ops.sa_mlp_max(*a, whole, 0)
    for order in (("mfma", "f64"), ("f64", "mfma")):
        out = torch.full((2, M, cout), float("nan"), device=dev)
        for part in order:
            ops.sa_mlp_max(*a, out, 0, part=part)
        assert torch.isfinite(out).all()
        assert torch.equal(out, whole), order


# ---------------------------------------------------------------------------------------------
# latent ODE and CNF
# ---------------------------------------------------------------------------------------------
def test_latent_rk4(dev, seeded_sd, model):
    z0 = rnd(1, 5, 1600)
    times = torch.tensor([0.0, 0.1, 0.35, 0.5, 1.0])
    want = O.latent_solve(seeded_sd, z0[:, :64], times, "rk4", 4)
    got = model.latent_ode(z0.to(dev)[:, :64], times.to(dev))
    record("latent_rk4", got, want, 1e-5)
    assert model.latent_ode.num_evals() == 4 * 4 * 4


@pytest.mark.parametrize("n,steps", [(256, 8), (100, 3)])
def test_cnf_sample(dev, seeded_sd, model, n, steps):
    BT = 3
    c, y = rnd(31, BT, 1600), rnd(32, BT, n, 3)
    want = O.point_cnf(seeded_sd, y, c, None, True, "rk4", steps)
    model.point_cnf.chain[1].rk4_steps = steps
    got = model.point_cnf(y.to(dev), c.to(dev), reverse=True)
    model.point_cnf.chain[1].rk4_steps = 8
    record("cnf_sample_n%d_s%d" % (n, steps), got, want, 1e-5)


@pytest.mark.parametrize("n,steps", [(256, 8), (100, 3), (2048, 2), (333, 4)])
def test_cnf_sample_bf16x6(dev, seeded_sd, model, n, steps):
    """Default sampling kernel (csrc/ode_bf16x6w.hip, 128 points per workgroup; n = 100 / 333: ragged last workgroup): hidden
    layers as six bf16 MFMA products of exactly split operands, activations kept in registers between the layers.  Same 1e-5
    criterion against the oracle as the f32 kernel, and within 5e-6 of the f32 kernel itself."""
    from caspr_amd import ops
    BT = 3
    c, y = rnd(31, BT, 1600), rnd(32, BT, n, 3)
    want = O.point_cnf(seeded_sd, y, c, None, True, "rk4", steps)
    cnf = model.point_cnf.chain[1]
    cnf.rk4_steps = steps
    prev = ops.set_matmul_mode(cnf=False)
    try:
        ref = model.point_cnf(y.to(dev), c.to(dev), reverse=True)
        ops.set_matmul_mode(cnf=True)
        got = model.point_cnf(y.to(dev), c.to(dev), reverse=True)
        again = model.point_cnf(y.to(dev), c.to(dev), reverse=True)
    finally:
        ops.set_matmul_mode(cnf=prev[1])
        cnf.rk4_steps = 8
    assert not torch.equal(got, ref), "the two modes must run different kernels"
    record("cnf_sample_f32_n%d_s%d" % (n, steps), ref, want, 1e-5)
    record("cnf_sample_bf16x6_n%d_s%d" % (n, steps), got, want, 1e-5)
    record("cnf_sample_bf16x6_vs_f32_n%d_s%d" % (n, steps), got, ref, 5e-6)
    exact("cnf_sample_bf16x6_repeat", again, got)


@pytest.mark.parametrize("mode,n", [("bf16x6", 96), ("bf16x6", 100), ("bf16x6", 1024), ("f32", 96), ("f32", 37)])
def test_cnf_forward_with_divergence(dev, seeded_sd, model, mode, n):
    """forward()/NLL direction with the Hutchinson divergence (odefunc.py:119-142, injected noise) on both kernels: the
    bf16x6 kernel's divergence variant (tangents as the upper 8 columns of every wave) and the f32-MFMA kernel."""
    from caspr_amd import ops
    BT = 2
    c, x, e = rnd(41, BT, 1600), rnd(42, BT, n, 3, scale=0.5), rnd(43, BT, n, 3)
    lp0 = rnd(44, BT, n, 1)
    wy, wlp = O.point_cnf(seeded_sd, x, c, lp0, False, "rk4", 8, e)
    prev = ops.set_matmul_mode(cnf=(mode == "bf16x6"))
    try:
        gy, glp = model.point_cnf(x.to(dev), c.to(dev), lp0.to(dev), e=e.to(dev))
        record("cnf_fwd_y_%s_n%d" % (mode, n), gy, wy, 1e-5)
        record("cnf_fwd_logp_%s_n%d" % (mode, n), glp, wlp, 1e-4)
        # sampling ignores the divergence: xyz of the with-div kernel == xyz of the plain kernel (same direction)
        gy2 = model.point_cnf(x.to(dev), c.to(dev), reverse=False)
        record("cnf_fwd_y_nodiv_vs_div_%s_n%d" % (mode, n), gy2, gy, 1e-6)
        # flow then inverse flow returns to the start (fixed-step RK4 is reversible to O(h^5)); with the log-density too
        back, lpb = model.point_cnf(gy, c.to(dev), glp, reverse=True, e=e.to(dev))
        record("cnf_roundtrip_%s_n%d" % (mode, n), back, x, 2e-4)
        record("cnf_roundtrip_logp_%s_n%d" % (mode, n), lpb, lp0, 2e-4)
        again = model.point_cnf(x.to(dev), c.to(dev), lp0.to(dev), e=e.to(dev))
        exact("cnf_fwd_repeat_%s_n%d" % (mode, n), again[1], glp)
    finally:
        ops.set_matmul_mode(cnf=prev[1])


# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sd64(seeded_sd):
    return {k: v.double() for k, v in seeded_sd.items()}


def test_encode_parity_dense(dev, seeded_sd, sd64, model):
    """Well-conditioned input: T-NOCS against the f32 oracle at the north_star tolerance.  f32 itself sits at
    ~8e-6 from the f64 evaluation of this graph (|oracle32 - f64| is recorded next to |hip - f64|), so the
    direct hip-vs-oracle32 difference is bounded by 2e-5 and each side by 1e-5 + the other's own f64 error."""
    x, _ = dense_sequences(2, 2, 1024)
    inter = []
    z0, tnocs = O.encode(seeded_sd, x, intermediates=inter)
    z64, t64 = O.encode(sd64, x.double())
    model.encoder.record = []
    gz0, gt = model.encode(x.to(dev))
    rec, model.encoder.record = model.encoder.record, None
    for l in range(5):
        exact("dense_fps_l%d" % l, rec[l]["fps_idx"], inter[l]["fps_idx"])
        for s in range(2):
            exact("dense_ball_l%d_s%d" % (l, s), rec[l]["ball_idx"][s], inter[l]["ball_idx"][s])
    record("dense_tnocs", gt, tnocs, 1e-5 + max(0.0, float((tnocs.double() - t64).abs().max()) - 8.5e-6))
    record("dense_tnocs_hip_vs_f64", gt, t64, 3e-6)
    record("dense_z0_hip_vs_f64", gz0, z64, 1e-5)
    record("dense_z0", gz0, z0, 1e-4)   # direct: the f32 oracle's own error on this 1600-wide max feature is 7.5e-5 (|z0| ~ 4)
    record_f64("dense_z0_vs_f64", gz0, z0, z64, 1e-5)


def test_encode_parity_cars(dev, seeded_sd, sd64, model):
    """Sparse car-like clouds: indices bit-exact; floats within 1e-5 of the f64 evaluation, flat."""
    x, _ = car_sequences(2, 2, 1024, seed=1234)
    inter = []
    z0, tnocs = O.encode(seeded_sd, x, intermediates=inter)
    z64, t64 = O.encode(sd64, x.double())
    model.encoder.record = []
    gz0, gt = model.encode(x.to(dev))
    rec, model.encoder.record = model.encoder.record, None
    for l in range(5):
        exact("enc_fps_l%d" % l, rec[l]["fps_idx"], inter[l]["fps_idx"])
        for s in range(2):
            exact("enc_ball_l%d_s%d" % (l, s), rec[l]["ball_idx"][s], inter[l]["ball_idx"][s])
    record_f64("enc_tnocs", gt, tnocs, t64, 1e-5)
    record_f64("enc_z0", gz0, z0, z64, 1e-5)    # FLAT since the f64 reference column (round 4): 4.1e-6 (round 3: 2.1e-5; f32 oracle 5.9e-4): the max over 20,480 pre-ReLU values


@pytest.mark.parametrize("B,m,n,C,C2", [(3, 128, 1000, 512, 6), (2, 64, 256, 128, 0), (1, 256, 512, 64, 8)])
def test_three_interp_add_gn_op(ops, dev, B, m, n, C, C2):
    """caspr_three_interp_add_gn_f32 alone: the three-neighbour combination of u + skip part + bias against f64 torch, and the GroupNorm
    scale / shift it returns against the statistics of its own output in f64 (ragged row blocks, no skip, the maximal skip width)."""
    u = rnd(1, B, m, C)
    unk, known = clouds(B, n, seed=5), clouds(B, m, seed=6)
    _, idx, w = ops.three_nn(unk.to(dev), known.to(dev), with_weights=True)
    skip = rnd(2, B, n, 8) if C2 else None
    wsk = rnd(3, C, C2, scale=0.3) if C2 else None
    bias, gamma, beta = rnd(4, C, scale=0.2), rnd(5, C, scale=0.5) + 1.0, rnd(6, C, scale=0.1)
    y, s, t_ = ops.three_interp_add_gn(u.to(dev), idx, w, None if skip is None else skip.to(dev), C2, None if wsk is None else wsk.to(dev), bias.to(dev),
                                       gamma.to(dev), beta.to(dev))
    i64, w64 = idx.cpu().long(), w.cpu().double()
    ref = torch.zeros(B, n, C, dtype=torch.float64)
    for k in range(3):
        ref += w64[:, :, k:k + 1] * torch.gather(u.double(), 1, i64[:, :, k:k + 1].expand(-1, -1, C))
    if C2:
        ref += skip[:, :, :C2].double() @ wsk.double().t()
    ref += bias.double()
    record("three_interp_add_gn_y_%d_%d" % (n, C), y, ref, 2e-6 * float(ref.abs().max()))
    G = 16
    yg = y.cpu().double().view(B, n, G, C // G)
    mean = yg.mean(dim=(1, 3), keepdim=True)
    var = ((yg - mean) ** 2).mean(dim=(1, 3), keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    sc = (gamma.double().view(1, 1, G, C // G) * rstd).reshape(B, C)
    sh = (beta.double().view(1, 1, G, C // G) - mean * gamma.double().view(1, 1, G, C // G) * rstd).reshape(B, C)
    record("three_interp_add_gn_scale_%d_%d" % (n, C), s, sc, 1e-5 * float(sc.abs().max()))
    record("three_interp_add_gn_shift_%d_%d" % (n, C), t_, sh, 1e-5 * max(1.0, float(sh.abs().max())))


def test_feature_propagation_conv_on_the_coarse_level(dev, seeded_sd, sd64, model):
    """The finest feature-propagation level's first conv runs over the COARSE rows (interpolation and a pointwise conv commute:
    models/pointnet2.py FP_COMMUTE, csrc/gemm.hip three_interp_add_gn_kernel) when the fine level has twice the points (N = 2048).
    Against the reference's order (interpolate, concatenate, conv) on the same kernels, and both against the f64 oracle: flat 1e-5."""
    from caspr_amd import ops
    import caspr_amd.models.pointnet2 as P2
    x, _ = car_sequences(1, 2, 2048, seed=5)
    z64, t64 = O.encode(sd64, x.double())
    res = {}
    prev = P2.FP_COMMUTE
    try:
        for on in (True, False):
            P2.FP_COMMUTE = on
            ops.TIMERS.clear()
            ops.TIMING = 2
            z, tn = model.encode(x.to(dev))
            torch.cuda.synchronize()
            ops.TIMING = False
            took = any(k.startswith("k:three_interp_add_gn") for k in ops.TIMERS)
            assert took == on, "the coarse-level path %s" % ("was not taken" if on else "ran although switched off")
            res[on] = (z.cpu(), tn.cpu())
    finally:
        P2.FP_COMMUTE = prev
        ops.TIMING = False
    record("fp_commute_tnocs_vs_f64", res[True][1], t64, 1e-5)
    record("fp_commute_z0_vs_f64", res[True][0], z64, 1e-5)
    record("fp_reference_order_tnocs_vs_f64", res[False][1], t64, 1e-5)
    record("fp_commute_vs_reference_order_tnocs", res[True][1], res[False][1], 5e-6)
    record("fp_commute_vs_reference_order_z0", res[True][0], res[False][0], 1e-5)


@pytest.mark.parametrize("mode", ["bf16x6", "f32"])
@pytest.mark.parametrize("B,T,N", [(2, 3, 1024), (1, 2, 1000)])
def test_head_fold_matches_the_separate_last_layer(dev, seeded_sd, sd64, model, mode, B, T, N):
    """The encoder with PointNet++'s last (purely linear) layer folded into the head's first conv (conv over frames, statistics
    pooled per sequence: the default path) against the same encoder with the two layers apart (the recording path), and both
    against the f64 evaluation: z0 and T-NOCS.  N = 1000 takes the conv + separate-statistics fallback of the pooled call."""
    from caspr_amd import ops
    prev = ops.set_matmul_mode(mode)
    try:
        x, _ = car_sequences(B, T, N, seed=77)
        z64, t64 = O.encode(sd64, x.double())
        gz0, gt = model.encode(x.to(dev))
        model.encoder.record = []
        sz0, st = model.encode(x.to(dev))
        model.encoder.record = None
    finally:
        ops.set_matmul_mode(conv=prev[0], cnf=prev[1])
    scale = float(z64.abs().max())
    record("head_fold_vs_separate_z0_%s_%d" % (mode, N), gz0, sz0, 2e-5 * max(1.0, scale))
    record("head_fold_vs_separate_tnocs_%s_%d" % (mode, N), gt, st, 5e-6)
    # ... and the fold is no further from the f64 evaluation than the separate layers are (the input's own conditioning is the
    # business of test_encode_parity_*: the default path there IS the folded one)
    for nm, g, sref, w64 in (("tnocs", gt, st, t64), ("z0", gz0, sz0, z64)):
        e_fold = float((g.cpu().double() - w64).abs().max())
        e_sep = float((sref.cpu().double() - w64).abs().max())
        REPORT["head_fold_vs_f64_%s_%s_%d" % (nm, mode, N)] = {"fold": e_fold, "separate": e_sep}
        assert e_fold <= e_sep + 3e-6 * max(1.0, float(w64.abs().max())), (nm, e_fold, e_sep)


@pytest.mark.parametrize("mode", ["bf16x6", "f32"])
def test_reconstruct_dense_vs_oracle(dev, seeded_sd, model, mode):
    """encode -> advect -> sample on the well-conditioned input: T-NOCS and sampled xyz within 1e-5 of the oracle, with the
    matrix products on the default bf16x6 kernels and on the f32 MFMA kernels (ops.set_matmul_mode)."""
    from caspr_amd import ops
    prev = ops.set_matmul_mode(mode)
    try:
        _reconstruct_dense_vs_oracle(dev, seeded_sd, model, mode)
    finally:
        ops.set_matmul_mode(conv=prev[0], cnf=prev[1])


def _reconstruct_dense_vs_oracle(dev, seeded_sd, model, mode):
    x, sp = dense_sequences(1, 3, 1024)
    torch.manual_seed(0)
    ybase = torch.randn(1, 3, 512, 3)
    _, wlp, wx, wt = O.reconstruct(seeded_sd, x, ybase, timestamps=sp[0, :, 0, 3])
    sd64 = {k: v.double() for k, v in seeded_sd.items()}
    _, _, x64, t64 = O.reconstruct(sd64, x.double(), ybase.double(), timestamps=sp[0, :, 0, 3].double())
    _, glp, gx, gt = model.reconstruct(x.to(dev), num_points=512, timestamps=sp[0, :, 0, 3].to(dev), y=ybase.to(dev))
    tag = "" if mode == "bf16x6" else "_f32mfma"
    # north_star as written: within 1e-5 of the (f32) reference restatement.  What is left of that difference is the f32
    # oracle's own distance from the f64 evaluation (7-8e-6 on this input); the HIP path itself sits within 4e-6 of f64.
    # (the slack term only opens if the f32 oracle itself lands further from f64 than it does on the build / driver hosts
    # -- its rounding depends on the CPU's GEMM blocking and thread count, which is not this build's to fix)
    record("dense_recon_tnocs" + tag, gt, wt, 1e-5 + max(0.0, float((wt.double() - t64).abs().max()) - 8.5e-6))
    record("dense_recon_x" + tag, gx, wx, 1e-5 + max(0.0, float((wx.double() - x64).abs().max()) - 8.5e-6))
    record("dense_recon_tnocs_hip_vs_f64" + tag, gt, t64, 3e-6)
    record("dense_recon_x_hip_vs_f64" + tag, gx, x64, 4e-6)
    record("dense_recon_logp_y" + tag, glp, wlp, 1e-5)
    from caspr_amd import ops
    gt_pts = sp[0, :, :512, :3].contiguous()
    d1, d2 = ops.chamfer_distance(gx.view(3, 512, 3).contiguous(), gt_pts.to(dev))
    cd = d1.mean(dim=1) + d2.mean(dim=1)
    record("dense_recon_chamfer_l2_hip_vs_f64" + tag, cd, O.chamfer_l2(x64.view(3, 512, 3).float(), gt_pts), 1e-5)
    record("dense_recon_chamfer_l2" + tag, cd, O.chamfer_l2(wx.view(3, 512, 3), gt_pts), 2e-5)   # the f32 oracle's samples are 7e-6 off themselves


def test_reconstruct_vs_reference_golden(dev, seeded_sd, sd64, model, golden):
    """Fixture produced by the REAL reference's reconstruct() (shimmed third-party ops) on a sparse car cloud."""
    x, sp = car_sequences(1, 2, 1024, seed=1234)
    ybase = torch.from_numpy(golden["pipe_ybase"])
    _, _, x64, t64 = O.reconstruct(sd64, x.double(), ybase.double(), timestamps=sp[0, :, 0, 3].double())
    gy, glp, gx, gt = model.reconstruct(x.to(dev), num_points=256, timestamps=sp[0, :, 0, 3].to(dev), y=ybase.to(dev))
    record_f64("recon_x_vs_reference_golden", gx, golden["pipe_recon_x"], x64, 1e-5, fixture_dist=3.82e-5)
    record_f64("recon_tnocs_vs_reference_golden", gt, golden["pipe_tnocs"], t64, 1e-5, fixture_dist=7.43e-5)
    record("recon_logp_y", glp, golden["pipe_logp_y"], 1e-5)
    assert [int(v) for v in model.get_nfe()] == [int(v) for v in golden["pipe_nfe"]]


def test_reconstruct_base_samples_from_cpu_generator(dev, model):
    """Base samples come from the CPU generator exactly as models/utils.py:25 (reproducible from manual_seed)."""
    x, sp = car_sequences(1, 2, 1024, seed=7)
    torch.manual_seed(0)
    y1, _, x1, _ = model.reconstruct(x.to(dev), num_points=128, timestamps=sp[0, :, 0, 3].to(dev))
    torch.manual_seed(0)
    want = torch.randn(2, 128, 3).view(1, 2, 128, 3)
    exact("decode_base_samples", y1, want)
    torch.manual_seed(0)
    y2, _, x2, _ = model.reconstruct(x.to(dev), num_points=128, timestamps=sp[0, :, 0, 3].to(dev))
    exact("reconstruct_deterministic", x2, x1)


def test_forward_nll(dev, seeded_sd, sd64, model, golden):
    """CaSPR.forward loss values: dense input strict; the reference's golden (sparse cars) within 1e-5 (T-NOCS) / 2e-6 relative (NLL) of the f64 evaluation."""
    x, sp = dense_sequences(1, 2, 1024)
    e = rnd(23, 2, 1024, 3)
    wr, wt = O.forward_nll(seeded_sd, x, sp, e)
    with torch.no_grad():            # evaluation as the reference's callers run it (test.py:124,141)
        recon, tl = model(x.to(dev), sp.to(dev), e=e.to(dev))
    record("dense_fwd_tnocs_loss", tl, wt, 2e-5)
    record("dense_fwd_recon_loss", recon, wr, 2e-4)      # NLL ~ 1e1, accumulates ~1e3 f32 ops per point
    x, sp = car_sequences(1, 2, 1024, seed=1234)
    r64, t64 = O.forward_nll(sd64, x.double(), sp.double(), e.double())
    with torch.no_grad():
        recon, tl = model(x.to(dev), sp.to(dev), e=e.to(dev))
    record_f64("fwd_tnocs_loss_vs_reference_golden", tl, golden["fwd_tnocs_loss"], t64, 1e-5, fixture_dist=7.43e-5)
    record_f64("fwd_recon_loss_vs_reference_golden", recon, golden["fwd_recon_loss"], r64, 2e-5, fixture_dist=1.15e-4)      # NLL ~ 1e1: 2e-6 relative


def test_demo_config_shape(dev, seeded_sd, sd64, model):
    """configs[0] of BASELINE.json: seq-len 5, 512 points (N < 1024 = SA1 centres: FPS repeats indices)."""
    x, sp = car_sequences(1, 5, 512, seed=3)
    z0, tnocs = O.encode(seeded_sd, x)
    z64, t64 = O.encode(sd64, x.double())
    gz0, gt = model.encode(x.to(dev))
    record_f64("demo_tnocs", gt, tnocs, t64, 1e-5)                     # N = 512 < 1024: FPS repeats indices, every level degenerate; FLAT since round 4: 6.4e-6 (round 3: 4.3e-5; f32 oracle 4.7e-4)
    record_f64("demo_z0", gz0, z0, z64, 1e-5)                          # FLAT since round 4: 5.5e-6 (round 3: 2.3e-5; f32 oracle 2.4e-3)


def test_real_demo_sequence_vs_reference_golden(dev, seeded_sd, sd64, model, golden):
    """REAL data (data/demo, 5 steps x 512 points) through the real reference (fixture) vs the HIP path."""
    x = torch.from_numpy(golden["demo_x"]).unsqueeze(0)
    sp = torch.from_numpy(golden["demo_nocs"]).unsqueeze(0)
    yb = torch.from_numpy(golden["demo_ybase"])
    z64, t64 = O.encode(sd64, x.double())
    _, _, x64, _ = O.reconstruct(sd64, x.double(), yb.double(), timestamps=sp[0, :, 0, 3].double())
    _, _, gx, gt = model.reconstruct(x.to(dev), num_points=128, timestamps=sp[0, :, 0, 3].to(dev), y=yb.to(dev))
    record_f64("realdemo_tnocs", gt, golden["demo_tnocs"], t64, 1e-5, fixture_dist=3.61e-4)
    record_f64("realdemo_recon_x", gx, golden["demo_recon_x"], x64, 1e-5, fixture_dist=1.87e-5)


def test_warping_config_no_tnocs(dev, seeded_sd):
    """BASELINE.json configs[3] surface: regress_tnocs=False (no conv3, tnocs None) and max_timestamp=1.0 (warping_cars.cfg)."""
    from caspr_amd.models import CaSPR
    m = CaSPR(regress_tnocs=False, cnf_rk4_steps=8, latent_rk4_steps=4)
    m.load_state_dict({k: v for k, v in seeded_sd.items() if not k.startswith("encoder.conv3")})
    m = m.to(dev).eval()
    x, sp = dense_sequences(1, 3, 1024, max_timestamp=1.0)
    torch.manual_seed(0)
    yb = torch.randn(1, 3, 256, 3)
    _, _, wx, wt = O.reconstruct(seeded_sd, x, yb, max_timestamp=1.0, regress_tnocs=False)
    _, _, gx, gt = m.reconstruct(x.to(dev), num_points=256, max_timestamp=1.0, y=yb.to(dev))
    assert gt is None and wt is None
    record("warping_recon_x", gx, wx, 2e-5)


@pytest.mark.parametrize("B,T,N,npts", [(1, 1, 200, 50), (3, 2, 333, 97)])
def test_ragged_shapes(dev, seeded_sd, sd64, model, B, T, N, npts):
    """Single-step sequences, point counts that are not multiples of any tile (N < every FPS target: repeated
    indices; GEMM / CNF tail tiles) and a sample count that is not a multiple of the 32-column CNF tile."""
    x, sp = dense_sequences(B, T, N, seed=B * 100 + N, side=0.2)
    torch.manual_seed(B)
    yb = torch.randn(B, T, npts, 3)
    ts = sp[0, :, 0, 3]
    _, wlp, wx, wt = O.reconstruct(seeded_sd, x, yb, timestamps=ts)
    _, _, x64, t64 = O.reconstruct(sd64, x.double(), yb.double(), timestamps=ts.double())
    _, glp, gx, gt = model.reconstruct(x.to(dev), num_points=npts, timestamps=ts.to(dev), y=yb.to(dev))
    assert gx.shape == (B, T, npts, 3) and gt.shape == (B, T, N, 4)
    # N < 1024: FPS repeats indices, neighbourhoods degenerate; still within the flat 1e-5 of the f64 evaluation
    record_f64("ragged_%dx%dx%d_tnocs" % (B, T, N), gt, wt, t64, 1e-5)
    record_f64("ragged_%dx%dx%d_x" % (B, T, N), gx, wx, x64, 1e-5)
    assert [int(v) for v in model.get_nfe()] == [4 * 4 * (T - 1), 32]


def test_bad_arguments_raise(dev, model):
    """Error behaviour: Python raises (the reference prints and exit()s), the C ABI reports through its error string."""
    from caspr_amd import ops, lib
    with pytest.raises(ValueError):
        ops.conv1x1(ops.PackedWeight(torch.zeros(8, 8, device=dev)), None, torch.zeros(1, 4, 6, device=dev))   # row stride not a multiple of 4
    with pytest.raises(lib.CasprHipError, match="ns="):
        ops.sa_mlp_max(torch.zeros(1, 8, 3, device=dev), torch.zeros(1, 2, 3, device=dev), None,
                       torch.zeros(1, 2, 8, dtype=torch.int32, device=dev), 0,
                       model.encoder.local_extract.set_abstractions[0].pointnet_modules[0].kernel_layers(),
                       torch.zeros(1, 2, 32, device=dev), 0)
    with pytest.raises(lib.CasprHipError, match="n=40000"):
        ops.furthest_point_sampling(torch.zeros(1, 40000, 3, device=dev), 16)
    with pytest.raises(ValueError):
        model.point_cnf(torch.zeros(2, 8, 3, device=dev), torch.zeros(2, 1600, device=dev), integration_times=torch.tensor([0.0, 1.0]))


def test_eval_protocols_vs_oracle(dev, seeded_sd, model):
    """evaluations.py protocols (10 steps x 2048 points; 3 observed / 7 unobserved) on the HIP path vs the oracle."""
    from caspr_amd.utils import evaluations as E
    x, sp = dense_sequences(1, 10, 2048, seed=11)
    torch.manual_seed(5)
    yb = torch.randn(1, 10, 2048, 3)
    res = E.test_shape_recon(model, [(x, sp)], dev, E.SPLIT_OBSERVED_STEPS, E.SPLIT_UNOBSERVED_STEPS, base_samples=[yb])
    # oracle: encode only the observed steps, reconstruct at all 10 timestamps (evaluations.py:105-114)
    _, _, wx, _ = O.reconstruct(seeded_sd, x[:, E.SPLIT_OBSERVED_STEPS].contiguous(), yb, timestamps=sp[0, :, 0, 3])
    want_obs = O.chamfer_l2(wx[0, E.SPLIT_OBSERVED_STEPS], sp[0, E.SPLIT_OBSERVED_STEPS, :, :3].contiguous())
    want_un = O.chamfer_l2(wx[0, E.SPLIT_UNOBSERVED_STEPS], sp[0, E.SPLIT_UNOBSERVED_STEPS, :, :3].contiguous())
    # north_star: Chamfer-L2 to 1e-5.  With random weights the reconstruction is far from the ground truth
    # (Chamfer ~ 1.6, a trained model gives ~1e-3), so the bound is taken relative to the value.
    record("eval_chamfer_observed", torch.tensor(res["observed_chamfer"]), want_obs, 1e-5 * max(1.0, float(want_obs.max())))
    record("eval_chamfer_unobserved", torch.tensor(res["unobserved_chamfer"]), want_un, 1e-5 * max(1.0, float(want_un.max())))
    assert res["nfe_mean"] == [4 * 4 * 9, 32] and res["infer_time_mean"] > 0
    want_emd = O.approx_emd(wx[0, E.SPLIT_UNOBSERVED_STEPS].double(), sp[0, E.SPLIT_UNOBSERVED_STEPS, :, :3].double()) / 2048   # evaluations.py:45-46
    record("eval_emd_unobserved", torch.tensor(res["unobserved_emd"]), want_emd, 1e-4 * float(want_emd.max()))
    tn = E.test_tnocs_regression(model, [(x, sp)], dev)
    _, wt = O.encode(seeded_sd, x)
    want_space = torch.mean(torch.norm(wt[..., :3] - sp[..., :3], dim=3), dim=2).mean()
    record("eval_tnocs_space_mean", torch.tensor(tn["space"]["mean"]), want_space, 1e-5)
    with pytest.raises(ValueError):
        E.test_shape_recon(model, [(x[:, :5], sp[:, :5])], dev)


def test_decode_options_and_variants(dev, seeded_sd, model, tmp_path):
    """The remaining branches of CaSPR.decode / ctor variants (caspr.py:204-267, 23-70) on the HIP path."""
    from caspr_amd.models import CaSPR
    from caspr_amd.utils.torch_utils import load_weights
    x, sp = dense_sequences(2, 3, 1024, seed=21)
    xd, ts = x.to(dev), sp[0, :, 0, 3].to(dev)
    # constant_in_time: one base cloud per sequence, repeated over the steps (caspr.py:254-256)
    torch.manual_seed(2)
    y, logp, xr, _ = model.reconstruct(xd, num_points=96, timestamps=ts, constant_in_time=True)
    assert torch.equal(y[:, 0], y[:, 2]) and not torch.equal(xr[:, 0], xr[:, 2])
    torch.manual_seed(2)
    assert torch.equal(y[:, 0].cpu(), torch.randn(2, 96, 3))
    record("decode_logp_y", logp, (-0.5 * np.log(2 * np.pi) - y.cpu() ** 2 / 2).sum(-1), 1e-6)
    # truncated base samples (models/utils.py:15-22) and Gaussian contours (caspr.py:232-250)
    y, _, xr, _ = model.reconstruct(xd, num_points=64, timestamps=ts, truncate_std=1.5)
    # 4 candidates per element: (1 - 0.866)^4 = 3e-4 of the elements keep an out-of-range first candidate, as in the reference
    assert float((y.abs() < 1.5).float().mean()) > 0.995 and torch.isfinite(xr).all()
    y, _, xr, _ = model.reconstruct(xd, num_points=90, timestamps=ts, sample_contours=[0.5, 1.0, 2.0])
    r = y.norm(dim=-1)[0, 0].cpu()
    assert torch.allclose(r[:30], torch.full((30,), 0.5), atol=1e-5) and torch.allclose(r[60:], torch.full((30,), 2.0), atol=1e-5)
    # timestamps=None: times come from the input cloud divided by max_timestamp (caspr.py:299-300)
    torch.manual_seed(17)
    yb = torch.randn(2, 3, 64, 3)
    _, _, xa, _ = model.reconstruct(xd, num_points=64, y=yb.to(dev))
    _, _, wa, _ = O.reconstruct(seeded_sd, x, yb)
    record("recon_times_from_input", xa, wa, 2e-5)
    # pretrain_tnocs surface: encoder only, forward returns (tnocs_loss,) (caspr.py:55-57,97-99)
    pm = CaSPR(pretrain_tnocs=True)
    load_weights(pm, {k: v for k, v in seeded_sd.items() if k.startswith("encoder.")})
    pm = pm.to(dev).eval()
    with torch.no_grad():
        (tl,) = pm(xd, sp.to(dev))
    _, wt = O.encode(seeded_sd, x)
    record("pretrain_tnocs_loss", tl, (wt - sp).abs(), 2e-5)
    # two stacked CNF blocks (flow.py:68-72): chain [MBN, CNF, CNF, MBN]; checkpoint round trip through torch.save
    m2 = CaSPR(cnf_blocks=2, cnf_rk4_steps=4, latent_rk4_steps=4)
    from caspr_amd.utils.synthetic import seeded_state_dict
    sd2 = seeded_state_dict(m2.state_dict(), 3)
    m2.load_state_dict(sd2)
    torch.save(m2.state_dict(), tmp_path / "ck.pth")
    m3 = CaSPR(cnf_blocks=2, cnf_rk4_steps=4, latent_rk4_steps=4)
    load_weights(m3, torch.load(tmp_path / "ck.pth"))
    m3 = m3.to(dev).eval()
    c, yy = rnd(51, 4, 1600), rnd(52, 4, 80, 3)
    want = O.point_cnf(sd2, yy, c, None, True, "rk4", 4, blocks=2)
    record("cnf_two_blocks", m3.point_cnf(yy.to(dev), c.to(dev), reverse=True), want, 1e-5)
    assert m3.get_nfe()[1] == 2 * 16


def test_full_size_properties(dev, model):
    """cars.cfg recon shape (T=10, N=2048) through size-independent properties (the oracle is too slow here)."""
    from caspr_amd import ops
    B, T, N = 4, 10, 2048
    x, sp = car_sequences(B, T, N, seed=99)
    xd = x.to(dev)
    xyz = xd.view(B * T, N, 4)[:, :, :3].contiguous()
    # FPS: indices distinct, and the M=512 run is a prefix of the M=1024 run
    i1024 = ops.furthest_point_sampling(xyz, 1024)
    i512 = ops.furthest_point_sampling(xyz, 512)
    assert torch.equal(i1024[:, :512], i512)
    assert all(len(set(r.tolist())) == 1024 for r in i1024[:3].cpu())
    # ball query: every neighbour lies inside the ball, valid prefix ascending
    ctr = torch.gather(xyz, 1, i1024.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    bi = ops.ball_query(0.05, 32, xyz, ctr)
    nb = torch.gather(xyz.unsqueeze(1).expand(-1, 1024, -1, -1)[:2], 2, bi[:2].long().unsqueeze(-1).expand(-1, -1, -1, 3))
    d2 = ((nb - ctr[:2].unsqueeze(2)) ** 2).sum(-1)
    assert float(d2.max()) < 0.05 * 0.05 * (1 + 1e-5)
    # sequences are independent: reconstructing a batch == reconstructing its halves (sharding invariance)
    torch.manual_seed(1)
    ybase = torch.randn(B, T, N, 3)
    ts = sp[0, :, 0, 3].to(dev)
    _, _, xa, ta = model.reconstruct(xd, num_points=N, timestamps=ts, y=ybase.to(dev))
    _, _, xb, tb = model.reconstruct(xd[2:], num_points=N, timestamps=ts, y=ybase[2:].to(dev))
    exact("shard_invariance_x", xa[2:], xb)
    exact("shard_invariance_tnocs", ta[2:], tb)
    assert torch.isfinite(xa).all() and float(ta.min()) > 0.0 and float(ta.max()) < 1.0
    assert [int(v) for v in model.get_nfe()] == [4 * 4 * 9, 32]


def test_cfg5_random_clouds(dev, seeded_sd, sd64, model):
    """BASELINE.json configs[4]: synthetic random clouds, T=20, N=4096 (64 sequences per GPU in the 8-GPU run).
    (a) oracle comparison on ONE sequence at the full (20, 4096) shape -- U(0,1)^3 clouds at this density make every
        r=0.02 ball a singleton padded with 15 / 31 copies of its centre, the worst case for GroupNorm conditioning: flat 1e-5
        against f64 since round 5 (7e-7 measured: f64 small balls + low parts between the first two levels), indices bit-exact;
    (b) size-independent properties at (2, 20, 4096): FPS prefix property at n = 4096, finiteness / range, bitwise
        sharding invariance, NFE."""
    from caspr_amd import ops
    from caspr_amd.utils.synthetic import random_clouds
    T, N = 20, 4096
    x = random_clouds(2, T, N, seed=7)
    ts = x[0, :, 0, 3] / 5.0
    torch.manual_seed(5)
    yb = torch.randn(2, T, 256, 3)
    # (a)
    inter = []
    _, _, wx, wt = O.reconstruct(seeded_sd, x[:1], yb[:1], timestamps=ts, cnf_steps=8, latent_steps=4)
    _, _, x64, t64 = O.reconstruct(sd64, x[:1].double(), yb[:1].double(), timestamps=ts.double(), cnf_steps=8, latent_steps=4)
    O.encode(seeded_sd, x[:1], intermediates=inter)
    model.encoder.record = []
    _, _, gx, gt = model.reconstruct(x[:1].to(dev), num_points=256, timestamps=ts.to(dev), y=yb[:1].to(dev))
    rec, model.encoder.record = model.encoder.record, None
    for l in range(5):
        exact("cfg5_fps_l%d" % l, rec[l]["fps_idx"], inter[l]["fps_idx"])
        for s_ in range(2):
            exact("cfg5_ball_l%d_s%d" % (l, s_), rec[l]["ball_idx"][s_], inter[l]["ball_idx"][s_])
    record_f64("cfg5_tnocs", gt, wt, t64, 1e-5)                       # i.i.d. uniform clouds: most neighbourhoods hold one to four points (round 3: 1.5e-4, round 4: 3.5e-5; f32 oracle 9.9e-4)
    record_f64("cfg5_recon_x", gx, wx, x64, 1e-5)                     # FLAT since round 4: 3.4e-6 (round 3: 1.7e-5; f32 oracle 1.1e-4)
    # (b)
    xd = x.to(dev)
    xyz = xd.view(2 * T, N, 4)[:, :, :3].contiguous()
    i1024 = ops.furthest_point_sampling(xyz, 1024)
    assert torch.equal(i1024[:, :300], ops.furthest_point_sampling(xyz, 300))
    assert all(len(set(r.tolist())) == 1024 for r in i1024[:3].cpu())
    torch.manual_seed(6)
    yfull = torch.randn(2, T, N, 3)
    _, _, xa, ta = model.reconstruct(xd, num_points=N, timestamps=ts.to(dev), y=yfull.to(dev))
    _, _, xb, tb = model.reconstruct(xd[1:], num_points=N, timestamps=ts.to(dev), y=yfull[1:].to(dev))
    exact("cfg5_shard_invariance_x", xa[1:], xb)
    exact("cfg5_shard_invariance_tnocs", ta[1:], tb)
    assert xa.shape == (2, T, N, 3) and torch.isfinite(xa).all() and float(ta.min()) > 0.0 and float(ta.max()) < 1.0
    assert [int(v) for v in model.get_nfe()] == [4 * 4 * (T - 1), 32]


def test_cfg4_warping_full_size(dev, seeded_sd, sd64):
    """BASELINE.json configs[3] (warping_cars.cfg): regress_tnocs=False, max_timestamp=1.0, input = the NOCS-space cloud
    itself (caspr_dataset.py:173-175), at the config's own T=10, N=2048.  Oracle comparison on one sequence (dense NOCS
    cloud: strict bound; car-like NOCS cloud: 1e-5 against the f64 evaluation), properties + sharding invariance at B=2, and the RK4
    step count an adaptive solver would choose on these weights (calibrate_rk4_steps), recorded in the parity report."""
    from caspr_amd.models import CaSPR
    m = CaSPR(regress_tnocs=False, cnf_rk4_steps=8, latent_rk4_steps=4)
    m.load_state_dict({k: v for k, v in seeded_sd.items() if not k.startswith("encoder.conv3")})
    m = m.to(dev).eval()
    T, N = 10, 2048
    _, sp = car_sequences(2, T, N, seed=77)
    x = sp.clone()                                   # the deformable datasets feed the NOCS cloud, time stamps in [0, 1]
    torch.manual_seed(4)
    yb = torch.randn(2, T, 256, 3)
    _, _, wx, wt = O.reconstruct(seeded_sd, x[:1], yb[:1], max_timestamp=1.0, regress_tnocs=False)
    _, _, x64, _ = O.reconstruct(sd64, x[:1].double(), yb[:1].double(), max_timestamp=1.0, regress_tnocs=False)
    _, _, gx, gt = m.reconstruct(x[:1].to(dev), num_points=256, max_timestamp=1.0, y=yb[:1].to(dev))
    assert gt is None and wt is None
    record_f64("cfg4_recon_x", gx, wx, x64, 1e-5)
    xd, _ = dense_sequences(1, T, N, seed=78, max_timestamp=1.0)
    _, _, wxd, _ = O.reconstruct(seeded_sd, xd, yb[:1], max_timestamp=1.0, regress_tnocs=False)
    _, _, gxd, _ = m.reconstruct(xd.to(dev), num_points=256, max_timestamp=1.0, y=yb[:1].to(dev))
    record("cfg4_dense_recon_x", gxd, wxd, 2e-5)
    torch.manual_seed(8)
    yfull = torch.randn(2, T, N, 3)
    _, _, xa, _ = m.reconstruct(x.to(dev), num_points=N, max_timestamp=1.0, y=yfull.to(dev))
    _, _, xb, _ = m.reconstruct(x[1:].to(dev), num_points=N, max_timestamp=1.0, y=yfull[1:].to(dev))
    exact("cfg4_shard_invariance_x", xa[1:], xb)
    assert torch.isfinite(xa).all() and [int(v) for v in m.get_nfe()] == [4 * 4 * (T - 1), 32]
    chosen, diffs = m.calibrate_rk4_steps(x.to(dev), tol=1e-5, max_timestamp=1.0)
    REPORT["cfg4_calibrated_cnf_steps"] = {"chosen": chosen, "step_doubling_diffs": {str(k): v for k, v in diffs.items()}}
    assert 1 <= chosen <= 16 and all(np.isfinite(v) for v in diffs.values())


def test_latent_team_kernel_under_concurrent_load(dev, seeded_sd, model):
    """The 32-workgroup latent team kernel synchronises with a hand-rolled barrier: its workgroups must become co-resident
    while OTHER streams keep the GPU busy (the encoder's side-stream index chain, a co-scheduled CNF launch).  Under such
    load the solve must return the single-workgroup kernel's values (never NaN), and the deferred status check stays quiet."""
    from caspr_amd import ops
    B, T = 16, 10
    z0 = rnd(61, B, 1600).to(dev)
    times = torch.linspace(0, 1, T).view(1, T).repeat(B, 1).to(dev)
    lat = model.latent_ode
    quiet = lat.solve_at(z0[:, :64], times)
    saved, ops.LATENT_TEAM = ops.LATENT_TEAM, False
    try:
        single = lat.solve_at(z0[:, :64], times)
    finally:
        ops.LATENT_TEAM = saved
    record("latent_team_vs_single", quiet, single, 2e-6)
    side1, side2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    xyz = rnd(62, 160, 2048, 3).to(dev)
    yc, cc = rnd(63, 40, 2048, 3).to(dev), rnd(64, 40, 1600).to(dev)
    torch.cuda.synchronize()
    for rep in range(4):
        with torch.cuda.stream(side1):
            for _ in range(3):
                ops.furthest_point_sampling(xyz, 1024)            # 160 long-running workgroups
        with torch.cuda.stream(side2):
            model.point_cnf(yc, cc, reverse=True)                  # fills every CU with MFMA workgroups
        loaded = lat.solve_at(z0[:, :64], times)
        torch.cuda.synchronize()
        assert torch.isfinite(loaded).all(), "team barrier gave up under load (rep %d)" % rep
        exact("latent_team_under_load_rep%d" % rep, loaded, quiet)
    ops.check_deferred_errors()


@pytest.mark.parametrize("B,n,m", [(3, 512, 512), (2, 300, 512), (2, 1024, 256), (1, 2048, 2048)])
def test_emd_matches_oracle(ops, dev, B, n, m):
    """Approximate EMD (utils/emd.py, evaluations.py:45-46) vs the oracle's restatement of approxmatch + matchcost, f64.
    Also the metric's basic properties: zero-ish for identical clouds, symmetric under swapping equal-size clouds."""
    g = np.random.default_rng(n + m)
    p = torch.from_numpy(g.uniform(0, 1, (B, n, 3)).astype(np.float32))
    q = torch.from_numpy(g.uniform(0, 1, (B, m, 3)).astype(np.float32))
    want = O.approx_emd(p.double(), q.double())
    got = ops.earth_mover_distance(p.to(dev), q.to(dev), transpose=False)
    err = float(((got.cpu().double() - want) / want).abs().max())
    REPORT["emd[%d,%d]" % (n, m)] = {"max_rel_err": err, "tol": 1e-4, "cost_per_point": float((want / n).mean())}
    assert err <= 1e-4, "EMD rel err %.3e" % err
    # reference call shape: (b,3,n) with transpose=True (emd.py:24)
    got_t = ops.earth_mover_distance(p.transpose(1, 2).to(dev), q.transpose(1, 2).to(dev))
    assert torch.equal(got_t, got)
    same = ops.earth_mover_distance(p.to(dev), p.to(dev), transpose=False) / n
    assert float(same.max()) < 0.05 * float((want / n).min()) + 1e-3


def test_calibrate_rk4_steps(dev, seeded_sd):
    """Step-doubling calibration of the CNF step count: the chosen count is the smallest candidate meeting the tolerance,
    it is installed on the CNF blocks, and the result at that count agrees with a 32-step solve.  (On the seeded weights
    the dynamics are so mild that every candidate differs from its doubled solve only by f32 rounding, 5e-7 .. 2e-6
    growing with the step count, so the tolerance here is set above that floor.)"""
    from caspr_amd.models import CaSPR
    m = CaSPR()
    m.load_state_dict(seeded_sd)
    m = m.to(dev).eval()
    x, sp = dense_sequences(1, 3, 1024, seed=41)
    torch.manual_seed(123)
    chosen, diffs = m.calibrate_rk4_steps(x.to(dev), tol=5e-6)
    REPORT["calibrate_rk4"] = {"chosen": chosen, "diffs": {str(k): v for k, v in diffs.items()}}
    assert diffs[chosen] <= 5e-6 * 15 / 16 and all(diffs[s] > 5e-6 * 15 / 16 for s in diffs if s < chosen)
    assert m.point_cnf.chain[1].rk4_steps == chosen and m.cnf_args.rk4_steps == chosen
    torch.manual_seed(7)
    yb = torch.randn(1, 3, 256, 3)
    a = m.reconstruct(x.to(dev), num_points=256, y=yb.to(dev))[2]
    m.point_cnf.chain[1].rk4_steps = 32
    b = m.reconstruct(x.to(dev), num_points=256, y=yb.to(dev))[2]
    record("calibrated_vs_32_steps", a, b, 1e-5)


@pytest.mark.parametrize("B,Tu", [(1, 3), (16, 10), (17, 4), (64, 5)])
def test_latent_team_kernel_matches_single_workgroup_kernel(ops, dev, seeded_sd, B, Tu):
    """caspr_latent_rk4_team_f32 (32 workgroups per 16 sequences, LDS-resident weights, team barriers) against
    caspr_latent_rk4_f32 and the oracle: same RK4, sums re-associated (K split over waves / workgroups): <= 5e-6."""
    from caspr_amd.models import CaSPR
    m = CaSPR()
    m.load_state_dict(seeded_sd)
    m = m.to(dev).eval()
    z0 = rnd(B, B, 64, scale=0.5)
    t = torch.linspace(0.0, 1.0, Tu)
    wts = m.latent_ode._weights()
    prev = ops.LATENT_TEAM
    try:
        ops.LATENT_TEAM = False
        a = ops.latent_rk4(z0.to(dev), t.to(dev), 2, wts)
        ops.LATENT_TEAM = True
        b = ops.latent_rk4(z0.to(dev), t.to(dev), 2, wts)
        c = ops.latent_rk4(z0.to(dev), t.to(dev), 2, wts)
    finally:
        ops.LATENT_TEAM = prev
    assert torch.equal(b, c), "team kernel is not repeatable"
    record("latent_team_vs_single[%d,%d]" % (B, Tu), b, a, 5e-6)
    want = O.latent_solve(seeded_sd, z0, t, steps_per_interval=2)
    record("latent_team_vs_oracle[%d,%d]" % (B, Tu), b, want, 1e-5)


# ---------------------------------------------------------------------------------------------
# STRESS dynamics (round-3 review, missing #1): the same kernels on weights whose flow is HARD to integrate
# (synthetic.stress_state_dict: gates that switch in time, both softplus tails, T = 1, a latent field that moves; on the f64
# oracle the reference's dopri5(1e-5) spends 68 evaluations there and RK4 at S = 8 is 6e-3 off -- tests/test_oracle_golden.py pins
# that).  Same criterion as everywhere: |hip - f64 oracle AT THE SAME STEP COUNT| <= 1e-5, flat, both product modes.
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def stress_model(dev, stress_sd):
    from caspr_amd.models import CaSPR
    m = CaSPR(cnf_rk4_steps=32, latent_rk4_steps=8)      # the step counts of the golden capture (gen_golden.py section 7)
    m.load_state_dict(stress_sd)
    return m.to(dev).eval()


@pytest.fixture(scope="module")
def stress_sd64(stress_sd):
    return {k: v.double() for k, v in stress_sd.items()}


@pytest.mark.parametrize("mode", ["bf16x6", "f32"])
@pytest.mark.parametrize("n,steps", [(256, 8), (100, 32), (2048, 64)])
def test_stress_cnf_sample(dev, stress_sd, stress_sd64, stress_model, mode, n, steps):
    """Sampling direction (cnf.py:70-128 with reverse=True, logpx None) on the stress weights: 128-point bf16x6 kernel and the
    f32-MFMA kernel against the f64 and f32 oracle at the same S -- S = 8 (where the INTEGRATION error is 6e-3: both sides must make
    the same one), 32 and 64 (where it converges).  n = 100: ragged last workgroup; n = 2048: the headline's frame size."""
    from caspr_amd import ops
    BT = 3 if n < 2048 else 1
    c, y = rnd(31, BT, 1600), rnd(32, BT, n, 3)
    w32 = O.point_cnf(stress_sd, y, c, None, True, "rk4", steps)
    w64 = O.point_cnf(stress_sd64, y.double(), c.double(), None, True, "rk4", steps)
    cnf = stress_model.point_cnf.chain[1]
    prev_steps, cnf.rk4_steps = cnf.rk4_steps, steps
    prev = ops.set_matmul_mode(cnf=(mode == "bf16x6"))
    try:
        got = stress_model.point_cnf(y.to(dev), c.to(dev), reverse=True)
        again = stress_model.point_cnf(y.to(dev), c.to(dev), reverse=True)
    finally:
        ops.set_matmul_mode(cnf=prev[1])
        cnf.rk4_steps = prev_steps
    record_f64("stress_cnf_sample_%s_n%d_s%d" % (mode, n, steps), got, w32, w64, 1e-5)
    exact("stress_cnf_sample_repeat_%s_n%d_s%d" % (mode, n, steps), again, got)


@pytest.mark.parametrize("mode", ["bf16x6", "f32"])
def test_stress_flow_vs_reference_golden(dev, stress_sd64, stress_model, golden, mode):
    """The REAL reference's SequentialFlow / CNF.forward / ODEfunc (autograd Hutchinson divergence) on the stress weights, both
    directions, RK4 shim at 32 steps (gen_golden.py section 7) against the HIP kernels at 32 steps: sampled x / forward y within 1e-5
    of the f64 evaluation, log-density 1e-4; the fixture itself (the reference's f32 arithmetic) is recorded next to it."""
    from caspr_amd import ops
    S = int(golden["stress_cnf_steps"])
    y, c, e = rnd(51, 2, 48, 3), rnd(52, 2, 1600), rnd(53, 2, 48, 3)
    xs, lp0 = rnd(55, 2, 48, 3, scale=0.5), rnd(56, 2, 48, 1)
    w64 = O.point_cnf(stress_sd64, y.double(), c.double(), None, True, "rk4", S)
    fy64, flp64 = O.point_cnf(stress_sd64, xs.double(), c.double(), lp0.double(), False, "rk4", S, e.double())
    cnf = stress_model.point_cnf.chain[1]
    assert cnf.rk4_steps == S
    prev = ops.set_matmul_mode(cnf=(mode == "bf16x6"))
    try:
        gx = stress_model.point_cnf(y.to(dev), c.to(dev), reverse=True)
        gy, glp = stress_model.point_cnf(xs.to(dev), c.to(dev), lp0.to(dev), e=e.to(dev))
    finally:
        ops.set_matmul_mode(cnf=prev[1])
    record_f64("stress_golden_rev_x_" + mode, gx, golden["stress_flow_rev_x"], w64, 1e-5)
    record_f64("stress_golden_fwd_y_" + mode, gy, golden["stress_flow_fwd_y"], fy64, 1e-5)
    record_f64("stress_golden_fwd_logp_" + mode, glp, golden["stress_flow_fwd_logp"], flp64, 1e-4)


@pytest.mark.parametrize("mode,n,steps", [("bf16x6", 96, 8), ("bf16x6", 100, 32), ("bf16x6", 1024, 64), ("f32", 96, 8), ("f32", 37, 32)])
def test_stress_cnf_forward_with_divergence(dev, stress_sd, stress_sd64, stress_model, mode, n, steps):
    """forward()/NLL direction with the Hutchinson divergence (odefunc.py:119-142, injected noise) on the stress weights: y 1e-5,
    log-density 1e-4 against f64 at the same S; sampling must not depend on the divergence; flow -> inverse flow round trip of
    (x, logp) at the step count where RK4 has converged."""
    from caspr_amd import ops
    BT = 2
    c, x, e = rnd(41, BT, 1600), rnd(42, BT, n, 3, scale=0.5), rnd(43, BT, n, 3)
    lp0 = rnd(44, BT, n, 1)
    wy, wlp = O.point_cnf(stress_sd, x, c, lp0, False, "rk4", steps, e)
    wy64, wlp64 = O.point_cnf(stress_sd64, x.double(), c.double(), lp0.double(), False, "rk4", steps, e.double())
    cnf = stress_model.point_cnf.chain[1]
    prev_steps, cnf.rk4_steps = cnf.rk4_steps, steps
    prev = ops.set_matmul_mode(cnf=(mode == "bf16x6"))
    try:
        gy, glp = stress_model.point_cnf(x.to(dev), c.to(dev), lp0.to(dev), e=e.to(dev))
        tag = "%s_n%d_s%d" % (mode, n, steps)
        record_f64("stress_cnf_fwd_y_" + tag, gy, wy, wy64, 1e-5)
        record_f64("stress_cnf_fwd_logp_" + tag, glp, wlp, wlp64, 1e-4)
        gy2 = stress_model.point_cnf(x.to(dev), c.to(dev), reverse=False)
        record("stress_cnf_fwd_y_nodiv_vs_div_" + tag, gy2, gy, 2e-6)
        if steps >= 64:
            back, lpb = stress_model.point_cnf(gy, c.to(dev), glp, reverse=True, e=e.to(dev))
            record("stress_cnf_roundtrip_" + tag, back, x, 5e-5)
            record("stress_cnf_roundtrip_logp_" + tag, lpb, lp0, 5e-4)
    finally:
        ops.set_matmul_mode(cnf=prev[1])
        cnf.rk4_steps = prev_steps


@pytest.mark.parametrize("team", [True, False])
@pytest.mark.parametrize("steps", [2, 8])
def test_stress_latent_rk4(ops, dev, stress_sd, stress_sd64, stress_model, golden, team, steps):
    """The latent solve (latent_ode_model.py:45-70,139-147) on a field that MOVES the state ~6 units over [0, 1] (at 2 steps per
    interval RK4 is 1e-2 from its converged solution: both sides must make the same error): team kernel and single-workgroup kernel
    against the f64 oracle at the same step count, flat 1e-5 (measured 1.4-2.3e-6)."""
    z0 = rnd(1, 5, 64)
    times = torch.tensor([0.0, 0.1, 0.35, 0.5, 1.0])
    w32 = O.latent_solve(stress_sd, z0, times, "rk4", steps)
    w64 = O.latent_solve(stress_sd64, z0.double(), times.double(), "rk4", steps)
    prev = ops.LATENT_TEAM
    try:
        ops.LATENT_TEAM = team
        got = ops.latent_rk4(z0.to(dev), times.to(dev), steps, stress_model.latent_ode._weights())
    finally:
        ops.LATENT_TEAM = prev
    REPORT["stress_latent_scale"] = float(w64.abs().max())
    record_f64("stress_latent_rk4_%s_s%d" % ("team" if team else "single", steps), got, w32, w64, 1e-5)      # flat, on |z| ~ 7
    # DynamicsNet itself against the real reference's module on these weights (one evaluation = one RK4 step of length 0 + f)
    if steps == 2 and team:
        zz = rnd(54, 4, 64)
        h = 1e-3
        one = ops.latent_rk4(zz.to(dev), torch.tensor([0.0, h]).to(dev), 1, stress_model.latent_ode._weights())[:, 1]
        f0 = torch.from_numpy(golden["stress_dynamics"])
        # z(h) = z + h f(z) + O(h^2 |f'| |f|): a consistency check of the field's magnitude and direction, not a tight bound
        record("stress_dynamics_first_order", (one.cpu() - zz) / h, f0, 0.05 * float(f0.abs().max()))


@pytest.mark.parametrize("mode", ["bf16x6", "f32"])
def test_stress_reconstruct(dev, stress_sd, stress_sd64, stress_model, golden, mode):
    """encode -> advect -> sample end to end on the stress weights (dense input, three distinct stamps, CNF 32 / latent 8 steps):
    against the f64 oracle at the same step counts AND against the real reference's reconstruct() on these weights
    (gen_golden.py section 7).  The f32 evaluations (oracle and reference fixture) sit ~8e-5 from f64 on xyz here -- the stressed
    hyper-networks amplify the encoder's f32 rounding of z0 -- which is recorded; the HIP path is held to the flat 1e-5."""
    from caspr_amd import ops
    x, sp = dense_sequences(1, 3, 1024, seed=41)
    yb = torch.from_numpy(golden["stress_pipe_ybase"])
    ts = sp[0, :, 0, 3]
    S, L = int(golden["stress_cnf_steps"]), int(golden["stress_latent_steps"])
    _, _, x64, t64 = O.reconstruct(stress_sd64, x.double(), yb.double(), timestamps=ts.double(), cnf_steps=S, latent_steps=L)
    prev = ops.set_matmul_mode(mode)
    try:
        _, _, gx, gt = stress_model.reconstruct(x.to(dev), num_points=96, timestamps=ts.to(dev), y=yb.to(dev))
        z0, _ = stress_model.encode(x.to(dev))
        gz = stress_model.aggregate_and_solve_latent(z0, sp[:, :, 0, 3].to(dev))[:, :, :64]
    finally:
        ops.set_matmul_mode(conv=prev[0], cnf=prev[1])
    z064, _ = O.encode(stress_sd64, x.double())
    z64 = O.aggregate_and_solve_latent(stress_sd64, z064, sp[:, :, 0, 3].double(), method="rk4", steps_per_interval=L)[:, :, :64]
    record_f64("stress_recon_tnocs_" + mode, gt, golden["stress_pipe_tnocs"], t64, 1e-5)
    record_f64("stress_recon_latent_" + mode, gz, golden["stress_pipe_latent"], z64, 1e-5 * max(1.0, float(z64.abs().max())))
    record_f64("stress_recon_x_" + mode, gx, golden["stress_pipe_recon_x"], x64, 1e-5)


def test_stress_calibrated_steps_agree_with_dopri5(dev, stress_sd, stress_sd64):
    """calibrate_rk4_steps on the stress weights: step doubling at tol = 1e-5 must pick S > 8 (the seeded weights: 1), the latent
    calibration more than the default 2 steps per interval; the HIP result at the chosen S must lie within 1e-5 of the CONVERGED f64
    solution (RK4, 256 steps) and agree with the oracle's restatement of the reference's integrator -- dopri5 at atol = rtol = 1e-5,
    flow.py:96-99 -- as closely as dopri5 itself agrees with the converged solution (its global error at that tolerance, ~1e-3 here).
    The CNF is compared on the SAME context (the HIP path's latent codes), so this isolates the integrator."""
    from caspr_amd.models import CaSPR
    m = CaSPR()
    m.load_state_dict(stress_sd)
    m = m.to(dev).eval()
    x, sp = dense_sequences(1, 3, 1024, seed=41)
    torch.manual_seed(123)
    chosen, diffs, lchosen, ldiffs = m.calibrate_rk4_steps(x.to(dev), tol=1e-5, latent_tol=1e-4, max_timestamp=5.0)
    assert chosen > 8 and diffs[chosen] <= 1e-5 * 15 / 16 and all(v > 1e-5 * 15 / 16 for s, v in diffs.items() if s < chosen), (chosen, diffs)
    # refined between the powers of two (round 6): every count below the chosen one that was tried failed, the chosen one was verified
    REPORT["stress_calibration_refined"] = {"chosen": chosen, "power_of_two_would_be": min(k for k in diffs if k >= chosen and (k & (k - 1)) == 0)}
    assert diffs[8] >= 1e-4, diffs
    assert lchosen > 2 and m.latent_ode.rk4_steps == lchosen, (lchosen, ldiffs)
    assert m.point_cnf.chain[1].rk4_steps == chosen and m.cnf_args.rk4_steps == chosen
    z0, _ = m.encode(x.to(dev))
    z = m.aggregate_and_solve_latent(z0, (x[:, :, 0, 3] / 5.0).to(dev))
    torch.manual_seed(7)
    yb = torch.randn(1, 3, 64, 3)
    got = m.decode(z, 64, y=yb.to(dev))[2].cpu().double().view(3, 64, 3)
    ctx = z.cpu().double().view(3, -1)
    cnt = [0]
    dop = O.point_cnf(stress_sd64, yb.double().view(3, 64, 3), ctx, None, True, "dopri5", counter=cnt)
    conv = O.point_cnf(stress_sd64, yb.double().view(3, 64, 3), ctx, None, True, "rk4", 256)
    e_dop = float((dop - conv).abs().max())
    REPORT["stress_calibration"] = {"chosen": chosen, "step_doubling_diffs": {str(k): v for k, v in diffs.items()}, "latent_chosen": lchosen,
                                    "latent_diffs": {str(k): v for k, v in ldiffs.items()}, "dopri5_nfe": cnt[0], "rk4_nfe": 4 * chosen,
                                    "dopri5_vs_converged": e_dop, "hip_vs_converged": float((got - conv).abs().max()),
                                    "hip_vs_dopri5": float((got - dop).abs().max())}
    assert cnt[0] >= 60, cnt
    # the calibration bounds the INTEGRATION error of the chosen count by 1e-5 (Richardson estimate (16/15) diff <= tol; the count need
    # not be a power of two any more, so there is no factor-of-16 margin to hide in); the arithmetic error of the f32 kernels at that
    # count -- HIP against the f64 evaluation of the SAME discrete map -- comes on top and is bounded on its own
    same = O.point_cnf(stress_sd64, yb.double().view(3, 64, 3), ctx, None, True, "rk4", chosen)
    record("stress_calibrated_vs_f64_same_steps", got, same, 1e-5)
    assert 16.0 / 15.0 * diffs[chosen] <= 1e-5
    record("stress_calibrated_vs_converged_f64", got, conv, 1e-5 + float((got - same).abs().max()))
    record("stress_calibrated_vs_dopri5", got, dop, e_dop + 1e-5)


def test_latent_solve_beside_the_last_head_layer(dev, seeded_sd, sd64):
    """reconstruct() starts the latent solve from INSIDE the encoder's last layer (ops.conv1x1_gn_early: the ODE's initial state is
    final after that layer's first channel tile; the team kernel then runs on 32 reserved compute units beside the remaining tiles).
    Against the serial order of rounds 1-3 (CASPR_EARLY_LATENT=0): the encoder's outputs are bit-identical (the pieces of the layer
    are the layer), the latent codes and the samples identical too (same kernel, same inputs); and the path is the one that ran."""
    from caspr_amd.models import CaSPR
    import caspr_amd.models.caspr as C
    m = CaSPR()
    m.load_state_dict(seeded_sd)
    m = m.to(dev).eval()
    x, sp = car_sequences(3, 4, 1024, seed=77)
    torch.manual_seed(5)
    yb = torch.randn(3, 4, 256, 3)
    outs, used = {}, {}
    prev = C.EARLY_LATENT
    try:
        for flag in (False, True):
            C.EARLY_LATENT = flag
            outs[flag] = m.reconstruct(x.to(dev), num_points=256, y=yb.to(dev))
            torch.cuda.synchronize()
            used[flag] = m._early_latent_used
            assert [int(v) for v in m.get_nfe()] == [4 * m.latent_ode.rk4_steps * 3, 4 * m.cnf_args.rk4_steps]
    finally:
        C.EARLY_LATENT = prev
    assert used[True] and not used[False]
    exact("early_latent_tnocs", outs[True][3], outs[False][3])
    exact("early_latent_x", outs[True][2], outs[False][2])
    _, _, x64, t64 = O.reconstruct(sd64, x.double(), yb.double(), cnf_steps=m.cnf_args.rk4_steps, latent_steps=m.latent_ode.rk4_steps)
    _, _, x32, t32 = O.reconstruct(seeded_sd, x, yb, cnf_steps=m.cnf_args.rk4_steps, latent_steps=m.latent_ode.rk4_steps)
    record_f64("early_latent_x_vs_f64", outs[True][2], x32, x64, 1e-5)


def test_latent_team_switch_covers_the_early_solve(dev, seeded_sd):
    """Round-4 advice: with the team kernel switched off (ops.LATENT_TEAM = False, what its own failure message recommends) the solve
    that reconstruct() starts beside the head's last layer must leave the team kernel too -- otherwise B <= 16 took the team kernel
    and B > 16 the single-workgroup one (5e-6 apart: results depended on the batch size).  Early and serial order: identical bits."""
    from caspr_amd import ops
    from caspr_amd.models import CaSPR
    import caspr_amd.models.caspr as C
    m = CaSPR()
    m.load_state_dict(seeded_sd)
    m = m.to(dev).eval()
    x, sp = car_sequences(2, 3, 1024, seed=78)
    torch.manual_seed(6)
    yb = torch.randn(2, 3, 128, 3)
    outs = {}
    prev_team, prev_early = ops.LATENT_TEAM, C.EARLY_LATENT
    try:
        ops.LATENT_TEAM = False
        assert C._EarlyLatent.team() is False and C._EarlyLatent(m.latent_ode, None, None).reserve_cus(16) == 1
        for flag in (False, True):
            C.EARLY_LATENT = flag
            outs[flag] = m.reconstruct(x.to(dev), num_points=128, y=yb.to(dev))
            torch.cuda.synchronize()
            assert m._early_latent_used == flag
        ops.LATENT_TEAM = True
        assert C._EarlyLatent.team() is True and C._EarlyLatent(m.latent_ode, None, None).reserve_cus(17) == 64
    finally:
        ops.LATENT_TEAM, C.EARLY_LATENT = prev_team, prev_early
    exact("single_kernel_early_vs_serial_x", outs[True][2], outs[False][2])
    exact("single_kernel_early_vs_serial_tnocs", outs[True][3], outs[False][3])


def test_accuracy_guard(dev, seeded_sd, stress_sd):
    """Run-time accuracy guard of the fixed-step integrators (CaSPR.check_tol; the reference's dopri5 controls its error at every call,
    flow.py:96-99, cnf.py:100-119, latent_ode_model.py:38,83).  Seeded weights at the default 8 / 2 steps: quiet, outputs and NFE
    untouched.  STRESS weights (RK4 at S = 8 is 6e-3 from the converged solution, test_stress_weights_are_a_hard_integration_problem):
    the CNF check trips with an estimate of that size -- through the deferred channel, no synchronisation in the call -- and so does
    the latent check at 2 steps per interval (1e-2 off); at the calibrated counts (64 / 16) both are quiet again."""
    from caspr_amd import ops
    from caspr_amd.models import CaSPR
    x, sp = car_sequences(2, 4, 1024, seed=5)
    ts = sp[0, :, 0, 3].to(dev)
    torch.manual_seed(9)
    yb = torch.randn(2, 4, 256, 3).to(dev)

    def model(sd, **kw):
        m = CaSPR(**kw)
        m.load_state_dict(sd)
        return m.to(dev).eval()
    # --- seeded weights: quiet; same bits and the same NFE as without the guard
    ops.reset_guard()
    m = model(seeded_sd, cnf_rk4_steps=8, latent_rk4_steps=2, check_tol=1e-5)
    got = m.reconstruct(x.to(dev), num_points=256, timestamps=ts, y=yb)
    nfe = [int(v) for v in m.get_nfe()]
    ops.check_deferred_errors()
    rep = dict(ops.GUARD_LAST)
    assert rep["cnf"]["ok"] and rep["latent"]["ok"] and rep["cnf"]["other_steps"] == 4 and rep["latent"]["other_steps"] == 1, rep
    assert rep["cnf"]["estimate"] <= 1e-6 and rep["latent"]["estimate"] <= rep["latent"]["bound"] <= 1e-2, rep    # (T = 4: intervals of 1/3, three times the headline's)
    m.check_tol = None
    ref = m.reconstruct(x.to(dev), num_points=256, timestamps=ts, y=yb)
    assert nfe == [int(v) for v in m.get_nfe()] == [4 * 2 * 3, 32]
    exact("guard_does_not_change_x", got[2], ref[2])
    exact("guard_does_not_change_tnocs", got[3], ref[3])
    REPORT["accuracy_guard_seeded"] = rep
    # --- stress weights, CNF under-resolved at S = 8 (latent at its calibrated 16 steps): raises, naming the CNF
    ops.reset_guard()
    ms = model(stress_sd, cnf_rk4_steps=8, latent_rk4_steps=16, check_tol=1e-5, check_action="raise")
    with pytest.raises(ops.CasprAccuracyError, match="point CNF"):
        ms.reconstruct(x.to(dev), num_points=256, timestamps=ts, y=yb)      # returns: the verdict is deferred (to the next guarded solve or ...)
        ops.check_deferred_errors()                                         # ... to here
    rep = dict(ops.GUARD_LAST)
    assert rep["latent"]["ok"] and not rep["cnf"]["ok"] and 2e-4 <= rep["cnf"]["estimate"] <= 2e-1, rep
    REPORT["accuracy_guard_stress_cnf_8"] = rep
    # the same as a warning, per call
    ops.reset_guard()
    ms.check_tol, ms.check_action = None, "warn"
    with pytest.warns(RuntimeWarning, match="not converged"):
        ms.reconstruct(x.to(dev), num_points=256, timestamps=ts, y=yb, check_tol=1e-5)
        ops.check_deferred_errors()
    ms.reconstruct(x.to(dev), num_points=256, timestamps=ts, y=yb)          # guard off again (the attribute is None): nothing queued
    ops.check_deferred_errors()
    # --- latent ODE under-resolved at 2 steps per interval (tolerance 100 x check_tol = 1e-3, the reference's ratio): raises, naming it
    ops.reset_guard()
    ml = model(stress_sd, cnf_rk4_steps=64, latent_rk4_steps=2, check_tol=1e-5, check_action="raise")
    with pytest.raises(ops.CasprAccuracyError, match="latent ODE"):
        ml.reconstruct(x.to(dev), num_points=256, timestamps=ts, y=yb)      # (the latent verdict may arrive while the same call queues the CNF's check)
        ops.check_deferred_errors()
    ops.check_deferred_errors()                                             # drain whatever the interrupted call had queued behind it
    REPORT["accuracy_guard_stress_latent_2"] = dict(ops.GUARD_LAST)
    # --- the density direction (forward(): y AND the log-density are integrated, cnf.py:112-126): quiet on seeded weights with the same
    # loss values as without the guard, raises on the stress weights at S = 8
    ops.reset_guard()
    sp_ = sp.to(dev)
    e_ = torch.randn(2 * 4, 1024, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    m.check_tol = 1e-5
    nll_g, tl_g = m(x.to(dev), sp_, e=e_)
    ops.check_deferred_errors()
    rep = dict(ops.GUARD_LAST)
    assert rep["cnf_fwd_y"]["ok"] and rep["cnf_fwd_logp"]["ok"] and rep["latent"]["ok"], rep
    m.check_tol = None
    nll_r, tl_r = m(x.to(dev), sp_, e=e_)
    exact("guard_does_not_change_nll", nll_g, nll_r)
    REPORT["accuracy_guard_forward_seeded"] = rep
    ops.reset_guard()
    ms.check_tol, ms.check_action = 1e-5, "raise"
    with pytest.raises(ops.CasprAccuracyError, match="density direction"):
        ms(x.to(dev), sp_, e=e_)
        ops.check_deferred_errors()
    ops.check_deferred_errors()
    REPORT["accuracy_guard_forward_stress_8"] = dict(ops.GUARD_LAST)
    # --- THE DEFAULTS (round 6): a model built with no guard arguments and loaded with weights that are under-resolved at the default
    # 8 / 2 steps WARNS -- nobody who swaps the import gets an unconverged flow silently
    ops.reset_guard()
    md = model(stress_sd)
    assert md.check_tol == 1e-5 and md.check_action == "warn" and md.point_cnf.chain[1].rk4_steps == 8 and md.latent_ode.rk4_steps == 2
    with pytest.warns(RuntimeWarning, match="not converged") as wrec:
        out_d = md.reconstruct(x.to(dev), num_points=256, timestamps=ts, y=yb)
        ops.check_deferred_errors()
    assert any("point CNF" in str(w.message) for w in wrec) and any("latent ODE" in str(w.message) for w in wrec), [str(w.message)[:80] for w in wrec]
    assert torch.isfinite(out_d[2]).all()
    # ... and after calibrate_rk4_steps on the same input the same call is quiet
    S_d, _, L_d, _ = md.calibrate_rk4_steps(x.to(dev), tol=1e-5, timestamps=ts, latent_tol=1e-4)
    assert md.check_tol == 1e-5                      # (the calibration switches the guard off while it tries under-resolved counts, and back on)
    ops.reset_guard()
    import warnings as _w
    with _w.catch_warnings():
        _w.simplefilter("error")
        md.reconstruct(x.to(dev), num_points=256, timestamps=ts, y=yb)
        ops.check_deferred_errors()
    assert ops.GUARD_LAST["cnf"]["ok"] and ops.GUARD_LAST["latent"]["ok"] and ops.GUARD_LAST["cnf"]["steps"] == S_d, ops.GUARD_LAST
    REPORT["accuracy_guard_defaults_on_stress"] = {"calibrated": [S_d, L_d], "after": dict(ops.GUARD_LAST)}
    # --- calibrated counts: quiet
    ops.reset_guard()
    mc = model(stress_sd, cnf_rk4_steps=64, latent_rk4_steps=16, check_tol=1e-5)
    mc.reconstruct(x.to(dev), num_points=256, timestamps=ts, y=yb)
    ops.check_deferred_errors()
    assert ops.GUARD_LAST["cnf"]["ok"] and ops.GUARD_LAST["latent"]["ok"], ops.GUARD_LAST
    REPORT["accuracy_guard_stress_calibrated"] = dict(ops.GUARD_LAST)


@pytest.mark.parametrize("B,P_,Cin,Cout,reserve", [(2, 1024, 512, 1600, 32), (3, 1280, 576, 1024, 1), (1, 2048, 1600, 1600, 0)])
def test_conv_layer_in_pieces_is_the_layer(ops, dev, B, P_, Cin, Cout, reserve):
    """caspr_conv1x1_x6w_part_f32 + caspr_conv_gn_finalize_f32 (ops.conv1x1_gn_early): first channel tile -> statistics of group 0 ->
    callback -> remaining tiles (+ the remainder below 512 channels) on all but `reserve` compute units -> all groups.  Output,
    scale / shift and the max over points must equal the one-call entry BIT FOR BIT, and at the time of the callback the columns of
    group 0 must already hold their final values."""
    w = rnd(1, Cout, Cin, scale=1.0 / np.sqrt(Cin))
    b = rnd(2, Cout, scale=0.3)
    x = rnd(3, B, P_, Cin)
    sc_in, sh_in = rnd(4, B, Cin).abs() + 0.5, rnd(5, B, Cin)
    gamma, beta = rnd(6, Cout) * 0.2 + 1.0, rnd(7, Cout) * 0.1
    pw = ops.PackedWeight(w.to(dev))
    assert ops.conv1x1_gn_early_ok(pw, B, P_, 16, 64)
    kw = dict(in_scale=sc_in.to(dev), in_shift=sh_in.to(dev), in_relu=True)
    whole = ops.conv1x1_gn(pw, b.to(dev), x.to(dev), gamma.to(dev), beta.to(dev), want_max=True, **kw)
    seen = {}

    def on_early(pmax):
        seen["g0"] = pmax[:, :Cout // 16].clone()
    parts = ops.conv1x1_gn_early(pw, b.to(dev), x.to(dev), gamma.to(dev), beta.to(dev), on_early, reserve_cus=reserve, **kw)
    torch.cuda.synchronize()
    for i, nm in enumerate(("y", "scale", "shift", "pmax")):
        exact("conv_pieces_%s_%dx%d" % (nm, Cin, Cout), parts[i], whole[i])
    exact("conv_pieces_early_group0_%dx%d" % (Cin, Cout), seen["g0"], whole[3][:, :Cout // 16])


def test_transposing_reduction_lane_map(dev):
    """xw_treduce16 / xw_rows_add (csrc/x6w_common.h: v_permlane16_swap + bank-masked DPP; the statistics epilogue of the persistent conv)
    against a host reduction, lane by lane: tools/micro/treduce_check (built by __graft_entry__.build())."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "micro", "treduce_check")
    if not os.path.exists(exe):
        pytest.skip("tools/micro/treduce_check is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "treduce_check: 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]
