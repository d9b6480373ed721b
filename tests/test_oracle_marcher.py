"""CPU: hand-derived cases for the marcher restatement (lib/nerfacc/cuda/csrc/ray_marching.cu:81-192,
intersection.cu:16-91).  The vendored CUDA cannot be built in this image (needs cuda_runtime.h /
torch headers), so these cases plus structural properties are what pins M1/M3/M4."""
import numpy as np
import torch

import oracle

ROI = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])


def test_aabb_hit_miss_and_inside():
    o = torch.tensor([[0.0, 0.0, -4.0], [0.0, 0.0, -4.0], [0.0, 0.0, 0.0], [3.0, 3.0, -4.0]])
    d = torch.tensor([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    tn, tf = oracle.ray_aabb_intersect(o, d, ROI)
    assert tn.tolist() == [2.5, 1e10, 0.0, 1e10] and tf.tolist()[0] == 5.5 and tf.tolist()[2] == 1.5
    assert tf[1] == 1e10 and tf[3] == 1e10  # miss => both 1e10 (intersection.cu:36-40)


def test_axis_ray_through_occupied_block():
    """SURVEY.md A.1 case: 8^3 grid, cells [2,6)^3 occupied, AABB +-1.5, step 0.05: a ray along an
    axis through the centre crosses the occupied span of 1.5 => 30 samples; a ray that misses => 0."""
    binary = torch.zeros(8, 8, 8, dtype=torch.bool)
    binary[2:6, 2:6, 2:6] = True
    o = torch.tensor([[0.01, 0.02, -4.0], [-4.0, 0.01, 0.02], [1.4, 1.4, -4.0]])
    d = torch.tensor([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    tn, tf = oracle.ray_aabb_intersect(o, d, ROI)
    pk, ri, ts, te = oracle.ray_marching_packed(o, d, tn, tf, ROI, binary, 0.05)
    assert pk[:, 1].tolist() == [30, 30, 0]
    assert pk[:, 0].tolist() == [0, 30, 60]
    mid = (ts + te) / 2
    # every emitted midpoint lies in an occupied cell; samples are contiguous and dt-spaced
    for r in (0, 1):
        seg = slice(int(pk[r, 0]), int(pk[r, 0] + pk[r, 1]))
        pts = o[r] + d[r] * mid[seg, None]
        occ, _ = oracle.query_occ(pts, ROI, binary)
        assert bool(occ.all())
        assert torch.allclose(te[seg] - ts[seg], torch.full((30,), 0.05), atol=1e-6)
        assert torch.allclose(ts[seg][1:], te[seg][:-1])


def test_dense_marching_counts_and_ordering():
    o = torch.tensor([[0.0, 0.0, -4.0]])
    d = torch.tensor([[0.0, 0.0, 1.0]])
    ri, ts, te = oracle.ray_marching(o, d, scene_aabb=ROI, render_step_size=0.01)
    # t in [2.5, 5.5): samples while t_mid < far  => 300
    assert ri.numel() == 300 and abs(float(ts[0]) - 2.5) < 1e-6
    assert bool((ts[1:] >= ts[:-1]).all())
    # stratified jitter shifts the first sample by u * step (ray_marching.py:157-158)
    ri2, ts2, _ = oracle.ray_marching(o, d, scene_aabb=ROI, render_step_size=0.01,
                                      stratified_u=torch.tensor([0.5]))
    assert abs(float(ts2[0]) - 2.505) < 1e-6


def test_cell_index_and_inclusive_bounds():
    binary = torch.zeros(4, 4, 4, dtype=torch.bool)
    binary[3, 3, 3] = True
    pts = torch.tensor([[1.5, 1.5, 1.5], [1.5001, 1.5, 1.5], [-1.5, -1.5, -1.5], [0.0, 0.0, 0.0]])
    occ, cell = oracle.query_occ(pts, ROI, binary)
    # the upper face belongs to the box (inclusive test) and clamps into the last cell (ray_marching.cu:20-24,34-40)
    assert occ.tolist() == [True, False, False, False]
    assert cell.tolist() == [63, -1, 0, 2 * 16 + 2 * 4 + 2]


def test_visibility_pruning_policy():
    o = torch.tensor([[0.0, 0.0, -4.0]])
    d = torch.tensor([[0.0, 0.0, 1.0]])
    alpha_fn = lambda ts, te, ri: torch.full_like(ts, 0.5)
    ri, ts, te = oracle.ray_marching(o, d, scene_aabb=ROI, render_step_size=0.01, alpha_fn=alpha_fn,
                                     early_stop_eps=1e-4)
    # T_i = 0.5^i >= 1e-4  <=>  i <= 13
    assert ri.numel() == 14


def test_pack_unpack_roundtrip():
    counts = torch.tensor([3, 0, 0, 5, 1, 0])
    ri = torch.repeat_interleave(torch.arange(6), counts)
    pk = oracle.pack_info(ri, 6)
    assert pk[:, 1].tolist() == counts.tolist() and pk[:, 0].tolist() == [0, 3, 3, 3, 8, 9]
    assert torch.equal(oracle.unpack_info(pk, ri.numel()), ri)
