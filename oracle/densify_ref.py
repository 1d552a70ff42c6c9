"""Plain-torch restatement of dn-splatter's refinement step (TEST INFRASTRUCTURE ONLY).

Follows ``DNSplatterModel.refinement_after`` (``dn_splatter/dn_model.py:271-386``) statement by statement — boolean masks,
``torch.cat`` of [params, split children, duplicates], then ``cull_gaussians`` over the concatenation — together with the
helpers it inherits from nerfstudio's ``SplatfactoModel`` (1.1.3, pinned by the reference's pyproject.toml:7 but not
vendored; restated from its published source): ``split_gaussians``, ``dup_gaussians``, ``cull_gaussians``,
``dup_in_optim`` (zeros appended to exp_avg / exp_avg_sq) and ``remove_from_optim``.  PARITY UNPINNED for the nerfstudio
helpers (no copy of that package here); the control flow and thresholds of refinement_after itself are read off the
reference file.

The only liberty: ``split_gaussians`` draws ``torch.randn`` internally; here the samples are an argument so that the
product (``dn-splatter_amd/densify.py``) and this restatement can be compared on identical noise.
Only tests may import this module.
"""
from __future__ import annotations

import torch


def quat_to_rotmat(quat):
    w, x, y, z = torch.unbind(torch.nn.functional.normalize(quat, dim=-1), dim=-1)
    return torch.stack([1 - 2 * (y**2 + z**2), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x**2 + z**2), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x**2 + y**2)], dim=-1).reshape(quat.shape[:-1] + (3, 3))


class Model:
    """Just enough of SplatfactoModel's state for refinement_after."""

    def __init__(self, gauss_params, cfg, step, num_train_data, last_size, xys_grad_norm, vis_counts, max_2Dsize, adam):
        self.gauss_params = {k: v.detach().clone() for k, v in gauss_params.items()}
        self.config, self.step, self.num_train_data, self.last_size = cfg, step, num_train_data, last_size
        self.xys_grad_norm, self.vis_counts, self.max_2Dsize = xys_grad_norm, vis_counts, max_2Dsize
        self.adam = None if adam is None else {n: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for n, st in adam.items()}

    def __getattr__(self, name):
        gp = self.__dict__.get("gauss_params", {})
        if name in gp:
            return gp[name]
        raise AttributeError(name)

    # ---- nerfstudio SplatfactoModel helpers -------------------------------------------------------------------------
    def split_gaussians(self, split_mask, samps, centered_samples):
        scaled_samples = torch.exp(self.scales[split_mask].repeat(samps, 1)) * centered_samples
        quats = self.quats[split_mask] / self.quats[split_mask].norm(dim=-1, keepdim=True)
        rots = quat_to_rotmat(quats.repeat(samps, 1))
        rotated_samples = torch.bmm(rots, scaled_samples[..., None]).squeeze(-1)
        new_means = rotated_samples + self.means[split_mask].repeat(samps, 1)
        size_fac = 1.6
        new_scales = torch.log(torch.exp(self.scales[split_mask]) / size_fac).repeat(samps, 1)
        self.scales[split_mask] = torch.log(torch.exp(self.scales[split_mask]) / size_fac)
        out = {"means": new_means, "features_dc": self.features_dc[split_mask].repeat(samps, 1),
               "features_rest": self.features_rest[split_mask].repeat(samps, 1, 1),
               "opacities": self.opacities[split_mask].repeat(samps, 1), "scales": new_scales,
               "quats": self.quats[split_mask].repeat(samps, 1)}
        for name, param in self.gauss_params.items():
            if name not in out:
                out[name] = param[split_mask].repeat(samps, 1)
        return out

    def dup_gaussians(self, dup_mask):
        return {name: param[dup_mask] for name, param in self.gauss_params.items()}

    def cull_gaussians(self, extra_cull_mask=None):
        cfg = self.config
        culls = (torch.sigmoid(self.opacities) < cfg.cull_alpha_thresh).squeeze()
        if extra_cull_mask is not None:
            culls = culls | extra_cull_mask
        if self.step > cfg.refine_every * cfg.reset_alpha_every:
            toobigs = (torch.exp(self.scales).max(dim=-1).values > cfg.cull_scale_thresh).squeeze()
            if self.step < cfg.stop_screen_size_at:
                if self.max_2Dsize is not None:
                    toobigs = toobigs | (self.max_2Dsize > cfg.cull_screen_size).squeeze()
            culls = culls | toobigs
        for name, param in self.gauss_params.items():
            self.gauss_params[name] = param[~culls]
        return culls

    def dup_in_all_optim(self, idcs, n):
        if self.adam is None:
            return
        for st in self.adam.values():
            for key in ("exp_avg", "exp_avg_sq"):
                t = st[key]
                rep = [n] + [1] * (t.dim() - 1)
                st[key] = torch.cat([t, torch.zeros_like(t[idcs.squeeze()]).repeat(*rep)], dim=0)

    def remove_from_all_optim(self, deleted_mask):
        if self.adam is None:
            return
        for st in self.adam.values():
            for key in ("exp_avg", "exp_avg_sq"):
                st[key] = st[key][~deleted_mask]

    # ---- dn_model.py:271-386 ----------------------------------------------------------------------------------------
    def refinement_after(self, centered_samples_fn):
        cfg = self.config
        if self.step <= cfg.warmup_length:
            return
        with torch.no_grad():
            reset_interval = cfg.reset_alpha_every * cfg.refine_every
            do_densification = (self.step < cfg.stop_split_at
                                and self.step % reset_interval > self.num_train_data + cfg.refine_every)
            if do_densification:
                avg_grad_norm = (self.xys_grad_norm / self.vis_counts) * 0.5 * max(self.last_size[0], self.last_size[1])
                high_grads = (avg_grad_norm > cfg.densify_grad_thresh).squeeze()
                splits = (self.scales.exp().max(dim=-1).values > cfg.densify_size_thresh).squeeze()
                if self.step < cfg.stop_screen_size_at:
                    splits |= (self.max_2Dsize > cfg.split_screen_size).squeeze()
                splits &= high_grads
                nsamps = cfg.n_split_samples
                split_params = self.split_gaussians(splits, nsamps, centered_samples_fn(nsamps * int(splits.sum())))
                dups = (self.scales.exp().max(dim=-1).values <= cfg.densify_size_thresh).squeeze()
                dups &= high_grads
                dup_params = self.dup_gaussians(dups)
                for name, param in self.gauss_params.items():
                    self.gauss_params[name] = torch.cat([param.detach(), split_params[name], dup_params[name]], dim=0)
                self.max_2Dsize = torch.cat([self.max_2Dsize, torch.zeros_like(split_params["scales"][:, 0]),
                                             torch.zeros_like(dup_params["scales"][:, 0])], dim=0)
                self.dup_in_all_optim(torch.where(splits)[0], nsamps)
                self.dup_in_all_optim(torch.where(dups)[0], 1)
                splits_mask = torch.cat((splits, torch.zeros(nsamps * int(splits.sum()) + int(dups.sum()), dtype=torch.bool,
                                                             device=splits.device)))
                deleted_mask = self.cull_gaussians(splits_mask)
            elif self.step >= cfg.stop_split_at and cfg.continue_cull_post_densification:
                deleted_mask = self.cull_gaussians()
            else:
                deleted_mask = None
            if deleted_mask is not None:
                self.remove_from_all_optim(deleted_mask)
            if self.step < cfg.stop_split_at and self.step % reset_interval == cfg.refine_every:
                reset_value = cfg.cull_alpha_thresh * 2.0
                self.gauss_params["opacities"] = torch.clamp(self.gauss_params["opacities"],
                                                             max=torch.logit(torch.tensor(reset_value)).item())
                if self.adam is not None:
                    st = self.adam["opacities"]
                    st["exp_avg"] = torch.zeros_like(st["exp_avg"])
                    st["exp_avg_sq"] = torch.zeros_like(st["exp_avg_sq"])
            self.xys_grad_norm = self.vis_counts = self.max_2Dsize = None


def classify_torch(params, stats, cfg, step, last_size, do_densify):
    """The flag byte of dnsplat_densify_classify in torch (what the gloo data-parallel test runs on the CPU)."""
    scales, opac = params["scales"], params["opacities"].reshape(-1)
    smax = scales.exp().max(dim=-1).values
    N = scales.shape[0]
    f = torch.zeros(N, dtype=torch.uint8, device=scales.device)
    z = stats.max_2Dsize if (stats is not None and stats.max_2Dsize is not None) else torch.zeros(N, device=scales.device)
    screen = step < cfg.stop_screen_size_at
    smax_child = torch.log(scales.exp() / 1.6).exp().max(dim=-1).values
    smax_dup = smax
    if do_densify:
        high = ((stats.xys_grad_norm / stats.vis_counts) * 0.5 * max(last_size[0], last_size[1])) > cfg.densify_grad_thresh
        split = smax > cfg.densify_size_thresh
        if screen:
            split = split | (z > cfg.split_screen_size)
        split = split & high
        smax_dup = torch.where(split, smax_child, smax)        # split_gaussians shrank the parents in place (dn_model.py:309-315)
        dup = (smax_dup <= cfg.densify_size_thresh) & high
        f |= split.to(torch.uint8) * 1 | dup.to(torch.uint8) * 2
    low = torch.sigmoid(opac) < cfg.cull_alpha_thresh
    big = torch.zeros_like(low)
    big_child = torch.zeros_like(low)
    big_dup = torch.zeros_like(low)
    if step > cfg.refine_every * cfg.reset_alpha_every:
        big = smax > cfg.cull_scale_thresh
        big_dup = smax_dup > cfg.cull_scale_thresh
        if screen and stats is not None and stats.max_2Dsize is not None:
            big = big | (z > cfg.cull_screen_size)
        big_child = smax_child > cfg.cull_scale_thresh
    f |= (((f & 1) != 0) | low | big).to(torch.uint8) * 4
    f |= (low | big_child).to(torch.uint8) * 8
    f |= (low | big_dup).to(torch.uint8) * 16
    return f


def split_children_torch(params, parents, noise):
    n_par = parents.shape[0]
    samps = noise.shape[0] // max(n_par, 1)
    sc = params["scales"][parents]
    q = params["quats"][parents]
    rots = quat_to_rotmat((q / q.norm(dim=-1, keepdim=True)).repeat(samps, 1))
    means = torch.bmm(rots, (torch.exp(sc.repeat(samps, 1)) * noise)[..., None]).squeeze(-1) + params["means"][parents].repeat(samps, 1)
    return means, torch.log(torch.exp(sc) / 1.6).repeat(samps, 1)
