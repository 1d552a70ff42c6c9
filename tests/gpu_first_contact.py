"""Ad-hoc first-contact script (not a pytest file): HIP path vs oracle on a small scene, with verbose
diagnostics.  Run on the GPU box: python tests/gpu_first_contact.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dn_splatter_amd as dns
from dn_splatter_amd import synthetic
from oracle import oracle as orc


def rel(a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    d = (a - b).abs()
    scale = b.abs().mean().item() + 1e-12
    return d.max().item(), (d / (b.abs() + scale)).max().item(), b.abs().max().item()


def main():
    dev = torch.device("cuda:0")
    print(torch.cuda.get_device_name(0))
    N, W, H = 10_000, 256, 256
    gp = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0)
    cam = synthetic.orbit_camera(0, width=W, height=H, focal=160.0)
    viewmat = dns.get_viewmat(cam.camera_to_worlds)
    K = cam.get_intrinsics_matrices()

    def inputs(device):
        means = gp["means"].detach().to(device).requires_grad_(True)
        quats = gp["quats"].detach()
        quats = (quats / quats.norm(dim=-1, keepdim=True)).to(device).requires_grad_(True)
        scales = torch.exp(gp["scales"].detach()).to(device).requires_grad_(True)
        opac = torch.sigmoid(gp["opacities"].detach()).squeeze(-1).to(device).requires_grad_(True)
        colors = torch.cat([gp["features_dc"].detach()[:, None], gp["features_rest"].detach()], 1).to(device).requires_grad_(True)
        return means, quats, scales, opac, colors

    torch.manual_seed(1)
    v_r = torch.rand(1, H, W, 4) * 2 - 1
    v_a = torch.rand(1, H, W, 1) * 2 - 1

    # oracle
    ci = inputs("cpu")
    t0 = time.time()
    r_o, a_o, info_o = orc.rasterization(*ci, viewmat, K, W, H, sh_degree=3, render_mode="RGB+ED", absgrad=True, packed=False)
    info_o["means2d"].retain_grad()
    ((r_o * v_r).sum() + (a_o * v_a).sum()).backward()
    print("oracle fwd+bwd %.2fs  n_isects=%d visible=%d" % (time.time() - t0, info_o["flatten_ids"].shape[0], int((info_o["radii"] > 0).sum())))

    gi = inputs(dev)
    r_g, a_g, info_g = dns.rasterization(*gi, viewmat.to(dev), K.to(dev), W, H, sh_degree=3, render_mode="RGB+ED", absgrad=True, packed=False)
    torch.cuda.synchronize()
    print("gpu fwd ok, n_isects", info_g["n_isects"])
    print("radii equal:", torch.equal(info_g["radii"].cpu(), info_o["radii"]),
          " tiles equal:", torch.equal(info_g["tiles_per_gauss"].cpu(), info_o["tiles_per_gauss"]))
    print("flatten_ids equal:", torch.equal(info_g["flatten_ids"].cpu(), info_o["flatten_ids"]),
          " offsets equal:", torch.equal(info_g["isect_offsets"].cpu(), info_o["isect_offsets"]))
    print("isect_ids equal:", torch.equal(info_g["isect_ids"].get().cpu(), info_o["isect_ids"]))
    for k in ("means2d", "depths", "conics"):
        print(k, "max abs/rel/scale", rel(info_g[k], info_o[k]))
    print("render", rel(r_g, r_o), "alpha", rel(a_g, a_o))
    info_g["means2d"].retain_grad()
    ((r_g * v_r.to(dev)).sum() + (a_g * v_a.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    for name, x, y in zip(["means", "quats", "scales", "opac", "colors"], gi, ci):
        print("grad", name, rel(x.grad, y.grad))
    print("means2d.grad", rel(info_g["means2d"].grad, info_o["means2d"].grad))
    print("means2d.absgrad", rel(info_g["means2d"].absgrad, info_o["means2d"].absgrad))

    # fused vs two-call (GPU vs GPU and vs oracle through the model mirror)
    def run_model(fused, device, raster=None, raster_legacy=None):
        params = {k: v.detach().to(device).requires_grad_(k != "normals") for k, v in gp.items()}
        m = dns.DNSplatterRenderer(params, fused=fused, rasterization_fn=raster, rasterize_gaussians_fn=raster_legacy)
        out = m.get_outputs(cam.to(device))
        torch.manual_seed(2)
        loss = 0
        for k in ("rgb", "depth", "normal", "accumulation"):
            loss = loss + (out[k] * (torch.rand(out[k].shape) * 2 - 1).to(device)).sum()
        loss.backward()
        return out, params, m

    out_f, p_f, m_f = run_model(True, dev)
    out_t, p_t, m_t = run_model(False, dev)
    out_o, p_o, m_o = run_model(False, "cpu", orc.rasterization, orc.rasterize_gaussians)
    torch.cuda.synchronize()
    for k in ("rgb", "depth", "normal", "surface_normal", "accumulation"):
        print("model", k, "fused-vs-oracle", rel(out_f[k], out_o[k]), "twocall-vs-oracle", rel(out_t[k], out_o[k]))
    for k in ("means", "scales", "quats", "features_dc", "features_rest", "opacities"):
        print("model grad", k, "fused-vs-oracle", rel(p_f[k].grad, p_o[k].grad), "twocall-vs-oracle", rel(p_t[k].grad, p_o[k].grad))
    print("xys.grad fused-vs-oracle", rel(m_f.xys.grad, m_o.xys.grad), "absgrad", rel(m_f.xys.absgrad, m_o.xys.absgrad))

    # rough timing at C2
    N, W, H = 1_000_000, 1920, 1080
    gp2 = synthetic.make_gauss_params(N, sh_rest_std=0.1, seed=0, device=dev)
    cam2 = synthetic.orbit_camera(0, width=W, height=H).to(dev)
    m = dns.DNSplatterRenderer(gp2, fused=True)
    vs = None
    for mode in ("sync", "capacity"):
        dns.set_bin_policy(mode)
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.time()
            out = m.get_outputs(cam2)
            torch.cuda.synchronize(); t1 = time.time()
            if vs is None:
                vs = {k: torch.rand_like(out[k]) * 2 - 1 for k in ("rgb", "depth", "normal", "accumulation")}
            loss = sum((out[k] * vs[k]).sum() for k in vs)
            loss.backward()
            torch.cuda.synchronize(); t2 = time.time()
            for p in gp2.values():
                p.grad = None
            print(mode, it, "fwd %.2f ms  bwd %.2f ms  n_isects %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, m.last_info["n_isects"]))


if __name__ == "__main__":
    main()
