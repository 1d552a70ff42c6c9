"""Host-side logic that needs no GPU: the get_outputs mirror's torch post-ops, camera helpers, synthetic inputs,
bench accounting.  (The mirror is exercised with the oracle plugged in for the two gsplat calls.)"""
import math
import os
import sys

import pytest
import torch

from _scenes import assert_close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_get_viewmat_inverts_the_camera(dns):
    from dn_splatter_amd import synthetic

    cam = synthetic.orbit_camera(3, width=64, height=48, focal=50.0)
    vm = dns.get_viewmat(cam.camera_to_worlds)[0]
    c2w = cam.camera_to_worlds[0]
    # OpenGL -> OpenCV: flip y, z of the camera axes; viewmat @ [cam centre, 1] = 0
    centre = torch.cat([c2w[:, 3], torch.ones(1)])
    assert torch.allclose(vm @ centre, torch.tensor([0.0, 0.0, 0.0, 1.0]), atol=1e-5)
    R = vm[:3, :3]
    assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-5)
    # the camera looks at the origin: the origin lies on the +z (forward) axis in OpenCV coordinates
    o = vm @ torch.tensor([0.0, 0.0, 0.0, 1.0])
    assert abs(float(o[0])) < 1e-4 and abs(float(o[1])) < 1e-4 and abs(float(o[2]) - 8.0) < 1e-4


def test_normal_from_depth_image_of_a_plane():
    """A fronto-parallel plane z = 2 gives the normal (0,0,+-1); a tilted plane gives its analytic normal."""
    sys.path.insert(0, ROOT)
    from dn_splatter_amd.model import normal_from_depth_image

    W, H, f = 40, 30, 35.0
    d = torch.full((H, W, 1), 2.0)
    n = normal_from_depth_image(d, f, f, W / 2, H / 2, (W, H), torch.eye(4))
    inner = n[1:-1, 1:-1]
    assert torch.allclose(inner.abs(), torch.tensor([0.0, 0.0, 1.0]).expand_as(inner), atol=1e-5)
    assert float(n[0].abs().max()) == 0.0 and float(n[:, 0].abs().max()) == 0.0   # zero-padded border
    # plane n.X = c with n = (a, 0, 1)/|.|: z = c / (a x/z + 1) ... sample depth analytically along pixel rays
    a, c = 0.3, 2.0
    xs = (torch.arange(W) + 0.5 - W / 2) / f
    z = c / (a * xs + 1.0)
    d2 = z[None, :, None].expand(H, W, 1).contiguous()
    n2 = normal_from_depth_image(d2, f, f, W / 2, H / 2, (W, H), torch.eye(4))[1:-1, 1:-1]
    expect = torch.tensor([a, 0.0, 1.0]) / math.sqrt(a * a + 1)
    assert torch.allclose(n2.abs(), expect.abs().expand_as(n2), atol=2e-3)


def test_synthetic_inputs_follow_the_reference_init(dns):
    from dn_splatter_amd import synthetic

    gp = synthetic.make_gauss_params(5000, sh_degree=3, seed=0)
    assert gp["means"].shape == (5000, 3) and float(gp["means"].detach().abs().max()) <= 5.0          # (rand-0.5)*10
    assert gp["features_rest"].shape == (5000, 15, 3) and gp["features_dc"].shape == (5000, 3)
    assert torch.allclose(torch.sigmoid(gp["opacities"]), torch.full((5000, 1), 0.1), atol=1e-6)   # logit(0.1)
    assert torch.allclose(gp["scales"][:, 0], gp["scales"][:, 1])                                  # isotropic
    assert torch.allclose(gp["quats"].norm(dim=-1), torch.ones(5000), atol=1e-5)
    # closed-form 3-NN distance tracks the kNN value the reference computes with sklearn
    knn = torch.exp(gp["scales"][:, 0]).mean().item()
    cf = synthetic.mean_3nn_distance_closed_form(5000)
    assert abs(knn - cf) / cf < 0.08
    # determinism
    gp2 = synthetic.make_gauss_params(5000, sh_degree=3, seed=0)
    assert torch.equal(gp["means"], gp2["means"]) and torch.equal(gp["quats"], gp2["quats"])


def test_bench_stage_bytes_sum_to_the_survey_formula():
    sys.path.insert(0, ROOT)
    import bench

    N, Nv, I, P, T = 1_000_000, 716_866, 21_243_626, 1920 * 1080, 8160
    sb = bench.stage_bytes(N, Nv, I, P, T)
    assert sum(sb.values()) == 84 * N + 796 * Nv + 216 * I + 76 * P + 12 * T
    assert set(bench.WORKLOADS) == {"c1", "c2", "c3", "c5"} and bench.WORKLOADS["c2"][:3] == (1_000_000, 1920, 1080)


def test_mirror_output_contract_and_sh_schedule(dns, orc):
    from dn_splatter_amd import synthetic

    gp = synthetic.make_gauss_params(400, sh_rest_std=0.2, seed=2)
    cam = synthetic.orbit_camera(0, width=48, height=32, focal=30.0)
    params = {k: v.detach().clone().requires_grad_(k != "normals") for k, v in gp.items()}
    m = dns.DNSplatterRenderer(params, fused=False, rasterization_fn=orc.rasterization,
                               rasterize_gaussians_fn=orc.rasterize_gaussians)
    m.step = 0            # dn_model.py:487-490: degree = min(step // interval, sh_degree)
    assert m._sh_degree_to_use() == 0
    out0 = m.get_outputs(cam)
    m.step = 2500
    assert m._sh_degree_to_use() == 2
    m.step = 10 ** 9
    out3 = m.get_outputs(cam)
    assert not torch.allclose(out0["rgb"], out3["rgb"])      # higher bands change the colour
    assert torch.allclose(out0["depth"], out3["depth"])      # ... but not the geometry
    for k, c in (("rgb", 3), ("depth", 1), ("normal", 3), ("surface_normal", 3), ("accumulation", 1)):
        assert out3[k].shape == (32, 48, c), k
    assert out3["background"].shape == (3,)
    assert float(out3["rgb"].detach().min()) >= 0 and float(out3["rgb"].detach().max()) <= 1
    assert m.xys.shape == (1, 400, 2) and m.radii.shape == (400,) and m.radii.dtype == torch.int32
    with pytest.raises(ValueError):
        dns.DNSplatterRenderer(params, config=dns.RendererConfig(rasterize_mode="bogus")).get_outputs(cam)


def test_torch_postops_equal_their_closed_forms(dns, orc):
    """Background blend, depth fill and normal normalisation of the mirror (dn_model.py:526-537, 577-578)."""
    from dn_splatter_amd import synthetic

    gp = synthetic.make_gauss_params(300, sh_rest_std=0.0, seed=4)
    cam = synthetic.orbit_camera(2, width=40, height=24, focal=25.0)
    params = {k: v.detach().clone() for k, v in gp.items()}
    m = dns.DNSplatterRenderer(params, fused=False, rasterization_fn=orc.rasterization,
                               rasterize_gaussians_fn=orc.rasterize_gaussians)
    out = m.get_outputs(cam)
    n = out["normal"] * 2 - 1
    assert torch.allclose(n.norm(dim=-1), torch.ones(24, 40), atol=1e-5)
    acc = out["accumulation"]
    empty = acc[..., 0] == 0
    if bool(empty.any()):
        assert torch.allclose(out["rgb"][empty], out["background"].expand(int(empty.sum()), 3), atol=1e-6)
        assert torch.allclose(out["depth"][empty], out["depth"].max().expand(int(empty.sum()), 1))


def test_torch_loss_stack_restatement():
    """dn_splatter/losses.py + regularization_strategy.py:146-199 restated in torch (they stay in PyTorch)."""
    sys.path.insert(0, ROOT)
    from dn_splatter_amd import torch_losses as tl

    H, W = 24, 32
    batch = tl.synthetic_batch(W, H, "cpu", seed=1)
    # identical images: SSIM == 1, L1 == 0
    assert abs(float(tl.ssim(batch["image"], batch["image"])) - 1.0) < 1e-6
    g = torch.Generator().manual_seed(2)
    out = {"rgb": torch.rand(H, W, 3, generator=g, requires_grad=True),
           "depth": (torch.rand(H, W, 1, generator=g) * 5 + 1).requires_grad_(True),
           "normal": torch.rand(H, W, 3, generator=g, requires_grad=True)}
    scales = torch.randn(50, 3, requires_grad=True)
    loss = tl.dn_loss(out, batch, scales)
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(v.grad).all() for v in out.values()) and torch.isfinite(scales.grad).all()
    # TV of a constant image is 0; EdgeAwareLogL1 of a perfect depth is 0
    assert float(tl.tv_loss(torch.ones(H, W, 3))) == 0.0
    assert float(tl.edge_aware_log_l1(batch["mono_depth"], batch["mono_depth"], batch["image"], batch["mono_depth"] > 0.1)) == 0.0
    # scale term: mean over Gaussians of the smallest exp(scale)  (regularization_strategy.py:195-199)
    only_scale = tl.dn_loss({k: v.detach() for k, v in out.items()}, {"image": out["rgb"].detach()}, scales)
    assert abs(float(only_scale.detach()) - float(torch.exp(scales.detach()).min(dim=1)[0].mean())) < 1e-6


@pytest.mark.parametrize("step,kw", [(3500, {}), (2500, {}), (3100, {}), (16000, {}), (3500, dict(cull_alpha_thresh=0.005)),
                                     (16000, dict(continue_cull_post_densification=False)), (400, {})])
def test_refinement_step_equals_the_reference_sequence(step, kw):
    """N3: densify.refinement_after (flag byte -> index gathers) against the statement-by-statement restatement of
    DNSplatterModel.refinement_after + the nerfstudio helpers (oracle/densify_ref.py: boolean masks, cat, cull over the
    concatenation) on identical split noise — every branch: densify with and without the screen-size / too-big rules,
    opacity reset, cull-only after stop_split_at, dn-splatter-big's thresholds, warm-up.  Includes the reference's quirk
    that a split parent whose shrunk scale falls under densify_size_thresh is duplicated as well."""
    from dn_splatter_amd import densify, synthetic
    from oracle import densify_ref as ref

    N = 4000
    gp = {k: v.detach() for k, v in synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=1).items()}
    g = torch.Generator().manual_seed(2)
    gp["scales"] = gp["scales"] + torch.randn(N, 3, generator=g) * 1.5 - 3.5
    gp["opacities"] = gp["opacities"] + torch.randn(N, 1, generator=g) * 2
    gp["normals"] = torch.randn(N, 3, generator=g)
    stats = densify.DensifyStats(N, "cpu")
    stats.xys_grad_norm = torch.rand(N, generator=g) * 0.01
    stats.vis_counts = torch.randint(1, 5, (N,), generator=g).float()
    stats.max_2Dsize = torch.rand(N, generator=g) * 0.1
    adam = {k: {"exp_avg": torch.randn(v.shape, generator=g), "exp_avg_sq": torch.rand(v.shape, generator=g), "step": torch.tensor(7.0)}
            for k, v in gp.items() if k != "normals"}
    cfg = densify.RefineConfig(**kw)
    seen = {}

    def split_fn(params, parents, noise):
        seen["noise"] = noise
        return ref.split_children_torch(params, parents, noise)

    new, new_adam, report = densify.refinement_after(gp, stats, cfg, step, 100, (480, 640), adam_state=adam, seed=3,
                                                     classify_fn=ref.classify_torch, split_fn=split_fn)
    m = ref.Model(gp, cfg, step, 100, (480, 640), stats.xys_grad_norm.clone(), stats.vis_counts.clone(), stats.max_2Dsize.clone(), adam)
    m.refinement_after(lambda n: seen["noise"] if n else torch.zeros(0, 3))
    for k in gp:
        assert new[k].shape == m.gauss_params[k].shape, (k, new[k].shape, m.gauss_params[k].shape)
        assert torch.allclose(new[k], m.gauss_params[k], atol=1e-6), k
    for k in adam:
        assert torch.equal(new_adam[k]["exp_avg"], m.adam[k]["exp_avg"]) and torch.equal(new_adam[k]["exp_avg_sq"], m.adam[k]["exp_avg_sq"]), k
    assert report["n_after"] == new["means"].shape[0]
    if step in (3500, 2500):
        assert report["n_split"] > 1000 and report["n_dup"] > 100 and report["n_culled"] > 1000
    if step == 3100:
        assert report["opacity_reset"] and float(new["opacities"].max()) <= float(torch.logit(torch.tensor(0.2))) + 1e-6
        assert float(new_adam["opacities"]["exp_avg"].abs().max()) == 0.0


def test_spatial_reorder_is_a_permutation_of_the_reference_refinement():
    """densify.spatial_order / reorder / refinement_after(spatial_reorder=True): a Morton-curve layout of the rows.  The refined set
    is the reference's refined set row for row under report["perm"] (parameters, normals, Adam moments); the order is a bijection,
    neighbouring rows are neighbours in space, and what a frustum culls are runs of rows, not scattered ones."""
    from dn_splatter_amd import densify, synthetic
    from oracle import densify_ref as ref

    N = 4000
    gp = {k: v.detach() for k, v in synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=1).items()}
    g = torch.Generator().manual_seed(2)
    gp["scales"] = gp["scales"] + torch.randn(N, 3, generator=g) * 1.5 - 3.5
    gp["opacities"] = gp["opacities"] + torch.randn(N, 1, generator=g) * 2
    gp["normals"] = torch.randn(N, 3, generator=g)
    stats = densify.DensifyStats(N, "cpu")
    stats.xys_grad_norm = torch.rand(N, generator=g) * 0.01
    stats.vis_counts = torch.randint(1, 5, (N,), generator=g).float()
    stats.max_2Dsize = torch.rand(N, generator=g) * 0.1
    adam = {k: {"exp_avg": torch.randn(v.shape, generator=g), "exp_avg_sq": torch.rand(v.shape, generator=g), "step": torch.tensor(7.0)}
            for k, v in gp.items() if k != "normals"}
    cfg = densify.RefineConfig()
    kw = dict(adam_state=adam, seed=3, classify_fn=ref.classify_torch, split_fn=ref.split_children_torch)
    plain, plain_adam, rep0 = densify.refinement_after(gp, stats, cfg, 3500, 100, (480, 640), **kw)
    new, new_adam, rep = densify.refinement_after(gp, stats, cfg, 3500, 100, (480, 640), spatial_reorder=True, **kw)
    perm = rep["perm"]
    assert "perm" not in rep0 and rep["n_after"] == rep0["n_after"] == perm.shape[0]
    assert torch.equal(torch.sort(perm)[0], torch.arange(perm.shape[0]))
    for k in plain:
        assert torch.equal(new[k], plain[k][perm]), k
    for k in adam:
        assert torch.equal(new_adam[k]["exp_avg"], plain_adam[k]["exp_avg"][perm]) and float(new_adam[k]["step"]) == 7.0
    # warm-up (nothing changes): no reorder either
    same, _, rep_w = densify.refinement_after(gp, stats, cfg, 400, 100, (480, 640), spatial_reorder=True, **kw)
    assert same is gp and "perm" not in rep_w
    # locality of the curve on the reference's initialisation
    big = synthetic.make_gauss_params(100_000, seed=0)
    order = densify.spatial_order(big["means"])
    assert torch.equal(order, densify.spatial_order(big["means"]))               # deterministic (every rank gets the same order)
    m, _ = densify.reorder(big, order)
    step_sorted = (m["means"][1:] - m["means"][:-1]).norm(dim=1).mean()
    step_random = (big["means"].detach()[1:] - big["means"].detach()[:-1]).norm(dim=1).mean()
    assert float(step_sorted) < 0.06 * float(step_random)
    cam = synthetic.orbit_camera(0)
    R, t = cam.camera_to_worlds[0][:, :3], cam.camera_to_worlds[0][:, 3]

    def mixed_blocks(means):
        pc = (means.detach() - t) @ R
        z = -pc[:, 2]
        vis = (z > 0.01) & ((pc[:, 0] / z * cam.fx).abs() < cam.cx * 1.15) & ((pc[:, 1] / z * cam.fy).abs() < cam.cy * 1.15)
        b = vis[: (means.shape[0] // 64) * 64].view(-1, 64).float().mean(1)
        return float(((b > 0) & (b < 1)).float().mean())

    assert mixed_blocks(big["means"]) > 0.99 and mixed_blocks(m["means"]) < 0.3


def test_bench_gpus_n_starts_n_ranks():
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run with two ranks (gloo here:
    no GPU) and refuses to print a line for fewer ranks than it was asked for (a single-GPU run must never be
    labelled n_gpus: N)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c1", "--rendezvous-only"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["multi_gpu"]["ranks_seen"] == 2
    # a launcher that started another number of ranks than --gpus says is an error, not a relabelled run
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                         capture_output=True, text=True, timeout=300, env=env2, cwd=root)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)


def test_kernel_source_hash_ignores_comments_but_not_code(tmp_path):
    """profiles/pmc_traffic.json is stamped with a hash of the kernel sources; rewording a comment must not un-stamp it, touching
    code (or a string literal that only looks like a comment) must."""
    import shutil
    import bench

    root = tmp_path / "tree"
    shutil.copytree(os.path.join(ROOT, "dn-splatter_amd", "csrc"), root / "dn-splatter_amd" / "csrc",
                    ignore=shutil.ignore_patterns("*.o", "*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), root / "include")
    base = bench.kernel_source_sha16(str(root))
    assert base == bench.kernel_source_sha16()
    f = root / "dn-splatter_amd" / "csrc" / "postops.hip"
    src = f.read_text()
    f.write_text("// a new remark\n" + src.replace("\n", "   \n", 3) + "\n/* and a block\n one */\n")
    assert bench.kernel_source_sha16(str(root)) == base
    f.write_text(src + "\nstatic const char *dns_probe = \"// not a comment\";\n")
    assert bench.kernel_source_sha16(str(root)) != base
    f.write_text(src.replace("0.5f", "0.25f", 1))
    assert bench.kernel_source_sha16(str(root)) != base


def test_digit_ranges_of_a_box_equal_the_pair_by_pair_count(tmp_path):
    """csrc/bin_ranges.h (the first tile pass's histogram without generating pairs) against the definition: pair k of a record has
    tile id base + (k / w) * tw + k % w and counts on digit (tile & (2^dbits - 1)).  The difference-array bookkeeping of
    radix_hist_ranges_kernel (cyclic ranges, sink slot, constant share) is replayed here as the kernel does it."""
    import subprocess

    src = tmp_path / "ranges.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "bin_ranges.h"
int main() {
    unsigned long long seed = 12345;
    auto rnd = [&](unsigned m) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)((seed >> 33) % m); };
    long cases = 0;
    for (int it = 0; it < 200000; ++it) {
        const int dbits = 1 + rnd(8);
        const unsigned ND = 1u << dbits;
        const unsigned tw = 1 + rnd(it % 3 == 0 ? 600 : 130);
        const unsigned w = 1 + rnd(tw), rows = 1 + rnd(70);
        const unsigned base = rnd(70000);
        const unsigned total = w * rows;
        unsigned lo = rnd(total), hi = lo + 1 + rnd(total - lo);
        if (it % 7 == 0) { lo = 0; hi = total; }
        std::vector<unsigned> want(ND, 0), diff(ND + 1, 0);
        for (unsigned k = lo; k < hi; ++k) want[(base + (k / w) * tw + k % w) & (ND - 1)]++;
        unsigned all = dns_record_digit_ranges(base, w, tw, lo, hi, dbits, [&](uint32_t d0, uint32_t len) {
            if (d0 >= ND || len == 0 || len >= ND) { printf("bad range %u %u\n", d0, len); exit(1); }
            diff[d0] += 1u;
            const unsigned end = d0 + len;
            if (end <= ND) diff[end] += 0xFFFFFFFFu;
            else { diff[0] += 1u; diff[end - ND] += 0xFFFFFFFFu; }
        });
        unsigned run = 0;
        for (unsigned d = 0; d < ND; ++d) {
            run += diff[d];
            if (run + all != want[d]) { printf("MISMATCH it %d dbits %d tw %u w %u rows %u base %u lo %u hi %u digit %u: %u vs %u\n", it, dbits, tw, w, rows, base, lo, hi, d, run + all, want[d]); return 1; }
        }
        ++cases;
    }
    printf("ok %ld\n", cases);
    return 0;
}
''')
    exe = tmp_path / "ranges"
    inc = os.path.join(ROOT, "dn-splatter_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", inc, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


def test_lazy_info_copies_force_their_lazy_entries(dns):
    """``info`` of the fused path is a dict subclass whose isect_ids / flatten_ids / n_isects are computed on first read;
    dict(info), {**info}, info.copy() and pickling must hand out the computed values, not the placeholder."""
    import pickle

    from dn_splatter_amd._ops import LazyInfo

    calls = []

    def make():
        calls.append(1)
        return 42

    fresh = lambda: LazyInfo({"width": 7}, {"isect_ids": make})   # noqa: E731
    assert dict(fresh())["isect_ids"] == 42
    assert {**fresh()}["isect_ids"] == 42
    assert fresh().copy()["isect_ids"] == 42
    assert pickle.loads(pickle.dumps(fresh()))["isect_ids"] == 42
    assert dict(fresh().items())["isect_ids"] == 42 and list(fresh().values()) == [7, 42]
    i = fresh()
    assert "isect_ids" in i and len(i) == 2 and list(i) == ["width", "isect_ids"] and i.get("isect_ids") == 42
    n = len(calls)
    assert i["isect_ids"] == 42 and len(calls) == n          # computed once
    assert isinstance(i, dict)


def test_capturable_torch_losses_equal_the_reference_form():
    """torch_losses.dn_loss(capturable=True) — masked means as sum / count, so that the step can be replayed as a HIP graph —
    against the reference's boolean-mask gathers (losses.py:216-222): same value, same gradients."""
    from dn_splatter_amd import torch_losses

    W, H, N = 40, 28, 50
    g = torch.Generator().manual_seed(3)
    batch = torch_losses.synthetic_batch(W, H, "cpu", seed=2)
    batch["mono_depth"][::3, ::2] = 0.05                     # below the depth tolerance: masked out
    res = []
    for cap in (False, True):
        out = {"rgb": torch.rand(H, W, 3, generator=torch.Generator().manual_seed(4)).requires_grad_(True),
               "depth": (torch.rand(H, W, 1, generator=torch.Generator().manual_seed(5)) * 8 + 0.3).requires_grad_(True),
               "normal": torch.rand(H, W, 3, generator=torch.Generator().manual_seed(6)).requires_grad_(True)}
        scales = torch.randn(N, 3, generator=torch.Generator().manual_seed(7)).requires_grad_(True)
        loss = torch_losses.dn_loss(out, batch, scales, capturable=cap)
        loss.backward()
        res.append((loss.detach(), out["rgb"].grad, out["depth"].grad, out["normal"].grad, scales.grad))
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-8), float((a - b).abs().max())


def test_ssim_as_block_toeplitz_gemms_equals_the_conv2d_form():
    """torch_losses.ssim_gemm (what the capturable stack uses: the five 11-tap Gaussian blurs of SSIM stacked and run as two GEMMs over
    unfolded blocks) == torch_losses.ssim (pytorch_msssim's separable conv2d form), value and gradient, at sizes below, at and beyond
    one block and with ragged last blocks."""
    from dn_splatter_amd import torch_losses as tl

    g = torch.Generator().manual_seed(1)
    for H, W in ((11, 11), (28, 40), (74, 75), (97, 203), (150, 139)):
        a = torch.rand(H, W, 3, generator=g).requires_grad_(True)
        b = torch.rand(H, W, 3, generator=g)
        s1, s2 = tl.ssim(a, b), tl.ssim_gemm(a, b)
        g1, = torch.autograd.grad(s1, a)
        g2, = torch.autograd.grad(s2, a)
        assert abs(float(s1.detach()) - float(s2.detach())) < 1e-6, (H, W)
        assert float((g1 - g2).abs().max()) <= 2e-6 * float(g1.abs().max()), (H, W)


def test_capacity_guesses_are_carried_over_a_refinement(dns):
    """densify.after_refinement(report=...) files the binning's capacity guesses of the old Gaussian count under the new one, scaled
    by n_new / n_old (x 1.1), instead of dropping them: the capture that follows needs no eager frames to size its buffers again.
    Without a report (or on another device) nothing is guessed."""
    from dn_splatter_amd import _ops, densify

    dev = torch.device("cpu")
    hints = _ops.BUFFERS.capacity_hint
    saved = dict(hints)
    try:
        hints.clear()
        hints[(dev, 1000, 640, 480)] = 50_000
        hints[(dev, 1000, 320, 240)] = 20_000
        hints[(dev, 777, 640, 480)] = 9_000                       # another Gaussian set: untouched
        new = {"means": torch.zeros(1500, 3)}
        densify.after_refinement(new, report={"n_before": 1000})
        assert hints[(dev, 1500, 640, 480)] == int(50_000 * 1.5 * 1.1) + 4096 and hints[(dev, 1500, 320, 240)] == int(20_000 * 1.5 * 1.1) + 4096
        assert (dev, 1000, 640, 480) not in hints and hints[(dev, 777, 640, 480)] == 9_000
        densify.after_refinement({"means": torch.zeros(1500, 3)}, report={"n_before": 1500})     # size unchanged: guesses stay
        assert hints[(dev, 1500, 640, 480)] == int(50_000 * 1.5 * 1.1) + 4096
        densify.after_refinement({"means": torch.zeros(900, 3)})                                  # no report: dropped (all of the device)
        assert not any(k[0] == dev for k in hints)
    finally:
        hints.clear()
        hints.update(saved)
