"""GPU parity for the steps either side of the field: occupancy-grid update (M2) and ray generation + pixel
gather (N2), vs the oracle's restatements of lib/nerfacc/grid.py:196-239 and systems/split_occ.py:58-131."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import texture as otex

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("res,n_upd", [((8, 8, 8), 200), ((16, 12, 10), 1920), ((32, 32, 32), 9000)])
def test_occupancy_update_matches_oracle(dev, res, n_upd):
    from rise_sdf_amd.nerfacc import OccGridEstimator
    n_cells = res[0] * res[1] * res[2]
    roi = torch.tensor([-1.5, -1.2, -1.0, 1.5, 1.3, 1.1])
    est = OccGridEstimator(roi, resolution=list(res)).to(dev)
    g = torch.Generator().manual_seed(n_upd)
    full = n_upd == n_cells
    idx = torch.arange(n_cells) if full else torch.randperm(n_cells, generator=g)[:n_upd]
    jit = torch.rand(n_upd, 3, generator=g)
    occ_fn = lambda x: (1.0 - x.norm(dim=-1, keepdim=True) / 3.0).clamp(0, 1) * 0.02
    occs0 = torch.rand(n_cells, generator=g) * 0.01
    est.occs.copy_(occs0)
    seen = {}

    def occ_fn_dev(x):   # evaluated on the host so that both sides see identical occupancy values
        seen["x"] = x.cpu()
        return occ_fn(seen["x"]).to(dev)

    est._update(step=1000, occ_eval_fn=occ_fn_dev, occ_thre=0.01,
                indices=est.grid_indices if full else idx.to(dev), cell_jitter=jit.to(dev))
    coords = torch.stack(torch.meshgrid(*[torch.arange(r) for r in res], indexing="ij"), -1).reshape(-1, 3)[idx]
    x = (coords + jit) / torch.tensor(res) * (roi[3:] - roi[:3]) + roi[:3]
    assert torch.equal(seen["x"], x)                      # bit-exact cell points
    occs_ref, bin_ref = oracle.occ_grid_update(occs0, idx, occ_fn(x).squeeze(-1), res, occ_thre=0.01)
    assert torch.equal(est.occs.cpu(), occs_ref)
    # the mean is accumulated in fp64 on the device, in fp32 by torch: cells within rounding of the threshold
    thre = float(torch.clamp(occs_ref.mean(), max=0.01))
    near = (occs_ref - thre).abs() < 1e-7 * max(thre, 1e-30) * 10
    got = est.binaries[0].cpu()
    assert torch.equal(got.view(-1)[~near], bin_ref.reshape(-1)[~near]) and int(near.sum()) < 4


def test_occupancy_update_duplicates_take_the_max(dev):
    from rise_sdf_amd import ops
    occs = torch.tensor([0.5, 0.2, 0.0, 0.9], device=dev)
    binary = torch.zeros(4, dtype=torch.uint8, device=dev)
    idx = torch.tensor([1, 1, 1, 3, 2], device=dev)
    occ = torch.tensor([0.1, 0.7, 0.3, 0.05, 0.0], device=dev)
    ops.occ_update(occs, binary, idx, occ, 0.5, 0.4)
    assert torch.allclose(occs.cpu(), torch.tensor([0.5, 0.7, 0.0, 0.45]))
    # mean = 0.4125 -> threshold min(mean, 0.4) = 0.4
    assert binary.cpu().tolist() == [1, 1, 0, 1]


@pytest.mark.parametrize("per_view_dirs,batch_index", [(False, True), (True, True), (False, False)])
def test_gen_rays_and_pixel_gather(dev, golden_dir, per_view_dirs, batch_index):
    from rise_sdf_amd import ops
    g = torch.Generator().manual_seed(3)
    V, H, W, n = 5, 24, 20, 3000
    dirs = oracle.get_ray_directions(W, H, 30.0, 31.0, W / 2, H / 2)
    if per_view_dirs:
        dirs = dirs[None] + 0.01 * torch.randn(V, H, W, 3, generator=g)
    rot = torch.linalg.qr(torch.randn(V, 3, 3, generator=g))[0]
    c2w = torch.cat([rot, torch.randn(V, 3, 1, generator=g)], -1)
    images = torch.rand(V, H, W, 3, generator=g)
    masks = (torch.rand(V, H, W, generator=g) > 0.4).float()
    bg = torch.tensor([1.0, 0.5, 0.0])
    index = torch.randint(0, V, (n if batch_index else 1,), generator=g)
    y, x = torch.randint(0, H, (n,), generator=g), torch.randint(0, W, (n,), generator=g)
    rays, rgb, fg = ops.gen_rays(index.to(dev), y.to(dev), x.to(dev), dirs.to(dev), c2w.to(dev), images.to(dev),
                                 masks.to(dev), bg.to(dev), apply_mask=True)
    # reference expressions (systems/split_occ.py:66-81,103,113-116)
    vi = index if batch_index else index.expand(n)
    d = dirs[vi, y, x] if per_view_dirs else dirs[y, x]
    rays_o, rays_d = oracle.get_rays(d, c2w[vi])
    rays_ref = torch.cat([rays_o, torch.nn.functional.normalize(rays_d, p=2, dim=-1)], -1)
    m = masks[vi, y, x]
    rgb_ref = images[vi, y, x] * m[:, None] + otex.rgb_to_srgb(bg * (1 - m[:, None]))
    assert torch.equal(rays.cpu()[:, :3], rays_ref[:, :3])
    assert torch.allclose(rays.cpu()[:, 3:], rays_ref[:, 3:], rtol=0, atol=2e-7)
    assert torch.equal(fg.cpu(), m)
    assert torch.allclose(rgb.cpu(), rgb_ref, rtol=1e-6, atol=1e-7)


def test_gen_rays_reference_fixture(dev, golden_dir):
    """Full-image rays of the reference's get_ray_directions + get_rays (tests/golden/rays.npz)."""
    from rise_sdf_amd import ops
    z = {k: torch.tensor(v) for k, v in np.load(os.path.join(golden_dir, "rays.npz")).items()}
    H, W = z["directions"].shape[:2]
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    rays, _, _ = ops.gen_rays(torch.zeros(1, dtype=torch.int64, device=dev), yy.reshape(-1).to(dev),
                              xx.reshape(-1).to(dev), z["directions"].to(dev), z["c2w"][None].to(dev))
    assert torch.equal(rays.cpu()[:, :3], z["rays_o"])
    ref_d = torch.nn.functional.normalize(z["rays_d"], p=2, dim=-1)
    assert torch.allclose(rays.cpu()[:, 3:], ref_d, rtol=0, atol=2e-7)


def test_orbit_view_rays_match_the_test_helper(dev):
    """The package's synthetic view generator (used by bench.py) vs tests/helpers.camera_rays (oracle get_rays)."""
    from helpers import camera_rays
    from rise_sdf_amd.ray_utils import get_ray_directions, get_rays, orbit_view_rays
    ref = camera_rays(40, 30, seed=5)
    got = orbit_view_rays(40, 30, seed=5, device=dev).cpu()
    assert torch.equal(got[:, :3], ref[:, :3])
    assert torch.allclose(got[:, 3:], ref[:, 3:], rtol=0, atol=3e-7)
    dirs = oracle.get_ray_directions(40, 30, 50.0, 51.0, 20.0, 15.0)
    assert torch.equal(get_ray_directions(40, 30, 50.0, 51.0, 20.0, 15.0), dirs)
    c2w = torch.tensor([[0.0, 1, 0, 0.5], [0, 0, 1, -0.2], [1, 0, 0, 2.0]])
    ro, rd = get_rays(dirs.to(dev), c2w.to(dev))
    ro_ref, rd_ref = oracle.get_rays(dirs, c2w)
    assert torch.equal(ro.cpu(), ro_ref) and torch.allclose(rd.cpu(), rd_ref, rtol=1e-6, atol=1e-6)
