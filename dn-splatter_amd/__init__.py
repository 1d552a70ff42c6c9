"""dn-splatter_amd — MI355X-native (gfx950) renderer that drops in behind
``DNSplatterModel.get_outputs`` of maturk/dn-splatter.

Public surface = the four gsplat symbols dn-splatter imports (dn_splatter/dn_model.py:29-35) plus
the fused one-pass renderer and the get_outputs mirror:

    from dn_splatter_amd import rasterization, rasterize_gaussians, quat_to_rotmat, num_sh_bases
    from dn_splatter_amd import render_dn, DNSplatterRenderer

All rendering arithmetic runs in hand-written HIP kernels (``csrc/*.hip`` -> ``libdnsplat.so``,
C ABI in ``include/dnsplat.h``).  There is no CPU fallback: without a GPU, or without the built
library, the ops raise.
"""
from ._lib import DnsplatError, build as build_library, lib as load_library  # noqa: F401
from .legacy import num_sh_bases, quat_to_rotmat, rasterize_gaussians  # noqa: F401
from .rendering import rasterization  # noqa: F401
from .fused import render_dn  # noqa: F401
from .model import Camera, DNSplatterRenderer, RendererConfig, get_viewmat  # noqa: F401
from ._ops import set_bin_policy, set_deterministic, set_grad_arena, set_sh_exchange  # noqa: F401
from . import dp  # noqa: F401
from .densify import DensifyStats  # noqa: F401
from .install import install, install_losses, install_ssim, uninstall  # noqa: F401

__all__ = [
    "rasterization", "rasterize_gaussians", "quat_to_rotmat", "num_sh_bases", "render_dn",
    "DNSplatterRenderer", "RendererConfig", "Camera", "get_viewmat", "set_bin_policy", "set_deterministic", "set_grad_arena", "set_sh_exchange", "dp", "DensifyStats",
    "install", "install_losses", "install_ssim", "uninstall", "build_library", "load_library", "DnsplatError",
]
