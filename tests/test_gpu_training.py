"""The pieces composed the way a trainer composes them (dn_splatter/dn_model.py:271-386, :404-612, :614-729; nerfstudio's
optimizer loop): a captured step — get_outputs, loss, backward, densification statistics — replayed with a new pose and a new
target per iteration, torch's Adam updating the parameters IN PLACE between replays, a refinement + re-capture in the middle.
Not a parity test (those are test_gpu_parity.py): what is asserted is that the loop LEARNS — the loss against images rendered
from a "ground-truth" Gaussian set falls — and that nothing overflows on the way."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")


def test_a_captured_training_loop_fits_rendered_targets(dns):
    from dn_splatter_amd import _ops, densify, dp, synthetic
    from dn_splatter_amd.graph import GraphedStep

    N, W, H, focal, n_views = 6000, 160, 120, 110.0, 4
    truth = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=21, device=DEV)
    with torch.no_grad():
        truth["opacities"] += 1.5                                           # some structure to fit: denser, larger splats
        truth["scales"] += 0.3
    poses = [synthetic.orbit_camera(i, n_views=n_views, width=W, height=H, focal=focal).camera_to_worlds.to(DEV) for i in range(n_views)]
    cam = synthetic.orbit_camera(0, n_views=n_views, width=W, height=H, focal=focal).to(DEV)
    targets = []
    with torch.no_grad():
        r_t = dns.DNSplatterRenderer(truth, fused=True)
        for p in poses:
            cam.camera_to_worlds.copy_(p)
            out = r_t.get_outputs(cam)
            targets.append({k: out[k].detach().clone() for k in ("rgb", "depth")})
    # the model: the truth with its colours, opacities and positions disturbed
    g = torch.Generator(device=DEV).manual_seed(5)
    gp = {k: v.detach().clone() for k, v in truth.items()}
    gp["features_dc"] += torch.randn(N, 3, device=DEV, generator=g) * 0.5
    gp["opacities"] += torch.randn(N, 1, device=DEV, generator=g) * 0.5
    gp["means"] += torch.randn(N, 3, device=DEV, generator=g) * 0.02
    prev_policy = _ops.BIN_POLICY["mode"]

    def build(gp):
        for k in KEYS:
            gp[k].requires_grad_(True)
        renderer = dns.DNSplatterRenderer(gp, fused=True)
        arena = dp.GradArena(gp)
        dns.set_grad_arena(arena)
        stats = densify.DensifyStats(gp["means"].shape[0], DEV)
        tgt = {k: torch.empty_like(v) for k, v in targets[0].items()}      # the captured loss reads the target from these buffers
        loss_out = torch.zeros((), device=DEV)

        def compute():
            for k in KEYS:
                gp[k].grad = None
            out = renderer.get_outputs(cam)
            loss = (out["rgb"] - tgt["rgb"]).abs().mean() + 0.1 * (out["depth"] - tgt["depth"]).abs().mean()
            loss.backward()
            stats.after_train(renderer, W, H)
            loss_out.copy_(loss.detach())

        dns.set_bin_policy("capacity")
        for i, p in enumerate(poses):                                       # sizes the buffers for every pose of the cycle
            cam.camera_to_worlds.copy_(p)
            for k in tgt:
                tgt[k].copy_(targets[i][k])
            compute()
        torch.cuda.synchronize()
        renderer.forget()
        step = GraphedStep(compute, params={k: gp[k] for k in KEYS})
        stats.xys_grad_norm.zero_(); stats.vis_counts.fill_(1.0); stats.max_2Dsize.zero_()
        # the geometry keeps small steps: positions by far the smallest, as in the reference's parameter groups (dn_config.py)
        opt = torch.optim.Adam([{"params": [gp["means"]], "lr": 2e-4}, {"params": [gp["features_dc"], gp["features_rest"]], "lr": 2e-2},
                                {"params": [gp["opacities"]], "lr": 3e-2}, {"params": [gp["scales"], gp["quats"]], "lr": 2e-3}])
        return renderer, stats, step, opt, tgt, loss_out

    def run(step, opt, tgt, loss_out, first, n):
        losses = []
        for it in range(n):
            v = (first + it) % n_views
            cam.camera_to_worlds.copy_(poses[v])
            for k in tgt:
                tgt[k].copy_(targets[v][k])
            step()
            opt.step()                                                      # in place: the next replay reads the updated parameters
            losses.append(loss_out.clone())
            if (it + 1) % 10 == 0:
                step.check()
        return torch.stack(losses).cpu()

    try:
        renderer, stats, step, opt, tgt, loss_out = build(gp)
        l1 = run(step, opt, tgt, loss_out, 0, 60)
        assert torch.isfinite(l1).all()
        start, mid = float(l1[:n_views].mean()), float(l1[-n_views:].mean())
        assert mid < 0.75 * start, f"the captured loop does not learn: loss {start:.4f} -> {mid:.4f} over 60 steps"
        # refinement on the statistics the replays accumulated, then a new capture on the refined set
        assert float(stats.vis_counts.max()) > 1.0 and float(stats.xys_grad_norm.max()) > 0.0, "densify statistics were not accumulated by the replays"
        step.close()
        params = {k: v.detach() for k, v in gp.items()}
        # the tenth of the visible Gaussians with the largest screen-space gradients is split / duplicated; the cull thresholds are
        # those of a scene this size (the defaults are meant for metric room scans: they would cull most of this toy scene as "too big")
        q90 = float(torch.quantile((stats.xys_grad_norm / stats.vis_counts * 0.5 * max(W, H))[stats.vis_counts > 1], 0.9))
        cfg = densify.RefineConfig(densify_grad_thresh=q90, cull_alpha_thresh=0.005, cull_scale_thresh=100.0, cull_screen_size=10.0,
                                   split_screen_size=10.0)
        new, _adam, report = densify.refinement_after(params, stats, cfg, 3500, n_views, (H, W), seed=3)
        assert report["n_after"] != report["n_before"] and report["n_split"] + report["n_dup"] > 0, report
        gp2 = {k: (v.requires_grad_(True) if k != "normals" else v) for k, v in new.items()}
        del renderer, stats, step, opt
        densify.after_refinement(gp2, report=report)
        renderer, stats, step, opt, tgt, loss_out = build(gp2)
        l2 = run(step, opt, tgt, loss_out, 60, 40)
        step.check()
        assert torch.isfinite(l2).all()
        end = float(l2[-n_views:].mean())
        # a split replaces a Gaussian by two smaller ones at sampled positions (nerfstudio's split_gaussians): the images jump, and the
        # loop — new parameter tensors, new gradient bucket, new capture, fresh optimiser state — learns on from there
        assert report["n_after"] > report["n_before"] and end < 0.6 * float(l2[:n_views].mean()) and end < start, \
            (start, mid, float(l2[:n_views].mean()), end)
        print(f"[training] loss {start:.4f} -> {mid:.4f} (60 steps, N = {report['n_before']}) -> refinement to N = {report['n_after']} "
              f"({report['n_split']} split, {report['n_dup']} duplicated, {report['n_culled']} culled) -> {end:.4f} (40 steps)")
        step.close()
    finally:
        dns.set_grad_arena(None)
        dns.set_bin_policy(prev_policy)
        _ops.forget_capacity_guesses(torch.device(DEV))
