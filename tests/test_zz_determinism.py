"""Determinism of the encoder under its own concurrency (round 6).  Kept in a file of its own that pytest collects LAST among the GPU files: the check
is bitwise and the one failure seen so far was a single run in one full-suite pass that 1,500 further runs did not reproduce --
if it ever fires again it must not hide the rest of the suite behind `-x`, and it prints where the difference sits."""
import os
import sys

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from caspr_amd.utils.synthetic import car_sequences

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    return torch.device("cuda:0")


def exact(name, got, want):
    d = (got != want)
    if bool(d.any()):
        rows = d.flatten(1).any(dim=1).nonzero().flatten().tolist()
        where = {r_: [int(d[r_].flatten()[512 * t_:512 * (t_ + 1)].sum()) for t_ in range(4)] for r_ in rows[:4]} if got.dim() == 2 else rows[:8]
        raise AssertionError("%s: %d / %d entries differ, max |diff| %.3e; rows %s (z0: per 512-channel tile %s)" % (
            name, int(d.sum()), d.numel(), float((got.double() - want.double()).abs().max()), rows[:16], where))


def test_encode_is_run_to_run_deterministic_under_its_own_concurrency(dev, seeded_sd):
    """The encoder runs on five streams (index chain, global PointNet, a set-abstraction scale, the T-NOCS regression, the caller's): its
    outputs, and every index tensor computed INSIDE that concurrency, must be the same bits run after run and equal the idle-chip index
    chain.  (Round 6: a leaner FPS kernel that was bit-exact on an idle chip chose wrong centres at the later levels beside the global
    PointNet's stats-only conv -- in a few frames, from some round on, differently every run: z0 moved by 1e-2.  No test looked at the
    indices as the pipeline computes them; this one does, at the headline shape where every compute unit is busy.)"""
    from caspr_amd.models import CaSPR
    m = CaSPR()
    m.load_state_dict(seeded_sd)
    m = m.to(dev).eval()
    x, _ = car_sequences(16, 10, 2048, seed=1234)
    xg = x.to(dev)
    le = m.encoder.local_extract

    def flat(ind):
        d = {}
        for l, s_ in enumerate(ind["sa"]):
            d["fps%d" % l], d["new_xyz%d" % l] = s_["fps_idx"], s_["new_xyz"]
            for i, b in enumerate(s_["ball_idx"]):
                d["ball%d_%d" % (l, i)] = b
        for l, t_ in enumerate(ind["nn"]):
            d["nn%d" % l] = t_[0]
        return d
    with torch.no_grad():
        idle = {k: v.clone() for k, v in flat(le.indices(xg.view(160, 2048, 4)[:, :, :3].contiguous())).items()}
    torch.cuda.synchronize()
    seen = {}
    orig = type(le).indices

    def spy(self, *a, **k):
        seen["ind"] = orig(self, *a, **k)
        return seen["ind"]
    type(le).indices = spy
    try:
        outs = []
        for r in range(4):
            with torch.no_grad():
                z0, tn = m.encode(xg)
            torch.cuda.synchronize()
            cur = flat(seen["ind"])
            bad = [k for k in cur if not torch.equal(cur[k], idle[k])]
            assert not bad, "encode %d: index tensors computed inside the pipeline differ from the idle-chip chain: %s" % (r, bad)
            outs.append((z0.clone(), tn.clone()))
        for r in range(1, 4):
            exact("encode_run%d_vs_run0_z0" % r, outs[r][0], outs[0][0])
            exact("encode_run%d_vs_run0_tnocs" % r, outs[r][1], outs[0][1])
    finally:
        type(le).indices = orig


