"""ctypes binding of ``libdnsplat.so`` (C ABI in ``include/dnsplat.h``).

The library is the product: there is NO fallback.  If the shared object is missing or a call
fails, this module raises — it never routes to a CPU path.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch
from ctypes import c_float, c_int32, c_int64, c_size_t, c_void_p
from pathlib import Path

_PKG_DIR = Path(__file__).resolve().parent
# DNSPLAT_LIB: load another build of the same library (kernel A/B runs, tools/ab_libs.sh); never a fallback
LIB_PATH = Path(os.environ["DNSPLAT_LIB"]).resolve() if os.environ.get("DNSPLAT_LIB") else _PKG_DIR / "libdnsplat.so"
CSRC_DIR = _PKG_DIR / "csrc"

ABI_VERSION = 15
RECORD_FLOATS = 16
MAX_CHANNELS = 8


class DnsplatError(RuntimeError):
    pass


class Scene(ctypes.Structure):
    _fields_ = [
        ("N", c_int32),
        ("means", c_void_p), ("quats", c_void_p), ("scales", c_void_p), ("opacities", c_void_p),
        ("scales_are_log", c_int32), ("opacities_are_logit", c_int32),
        ("sh_degree", c_int32), ("sh_K", c_int32),
        ("sh0", c_void_p), ("sh0_stride", c_int32),
        ("shN", c_void_p), ("shN_stride", c_int32),
        ("colors", c_void_p), ("n_colors", c_int32), ("colors_are_logit", c_int32),
    ]


class Camera(ctypes.Structure):
    _fields_ = [
        ("viewmat", c_void_p), ("K", c_void_p), ("normal_frame", c_void_p),
        ("width", c_int32), ("height", c_int32), ("tile_size", c_int32),
        ("eps2d", c_float), ("near_plane", c_float), ("far_plane", c_float), ("radius_clip", c_float),
        ("antialiased", c_int32), ("tight_tiles", c_int32),
    ]


class ProjOut(ctypes.Structure):
    _fields_ = [
        ("radii", c_void_p), ("means2d", c_void_p), ("depths", c_void_p), ("conics", c_void_p),
        ("compensations", c_void_p), ("tiles_per_gauss", c_void_p), ("splats", c_void_p),
        ("normals_world", c_void_p),
        ("with_depth_channel", c_int32), ("with_normal_channels", c_int32), ("saturation_flag", c_void_p),
        ("tiles_bin", c_void_p), ("tile_boxes", c_void_p), ("phase", c_int32), ("skip_culled_records", c_int32),
    ]


class BinArgs(ctypes.Structure):
    _fields_ = [
        ("N", c_int32), ("n_cameras", c_int32), ("width", c_int32), ("height", c_int32), ("tile_size", c_int32),
        ("means2d", c_void_p), ("radii", c_void_p), ("depths", c_void_p), ("tiles_per_gauss", c_void_p),
        ("isect_capacity", c_int64),
        ("flatten_ids", c_void_p), ("tile_offsets", c_void_p),
        ("n_isects", c_void_p), ("n_isects_host", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("splats", c_void_p), ("tight_tiles", c_int32),
        ("tile_ends", c_void_p), ("skip_offsets_fill", c_int32), ("tile_boxes", c_void_p), ("n_isects_max", c_void_p),
    ]


class DnPost(ctypes.Structure):
    _fields_ = [
        ("background_rgb", c_void_p), ("rgb", c_void_p), ("depth", c_void_p), ("normal", c_void_p),
        ("depth_max", c_void_p),
        ("v_rgb", c_void_p), ("v_depth", c_void_p), ("v_normal", c_void_p), ("v_accumulation", c_void_p),
    ]


class DensifyArgs(ctypes.Structure):
    _fields_ = [
        ("N", c_int32), ("scales", c_void_p), ("opacities", c_void_p), ("xys_grad_norm", c_void_p), ("vis_counts", c_void_p),
        ("max_2Dsize", c_void_p),
        ("do_densify", c_int32), ("screen_rules", c_int32), ("cull_big", c_int32),
        ("max_image_side", c_float),
        ("densify_grad_thresh", c_float), ("densify_size_thresh", c_float), ("split_screen_size", c_float),
        ("cull_alpha_thresh", c_float), ("cull_scale_thresh", c_float), ("cull_screen_size", c_float),
        ("flags", c_void_p),
    ]


class DnLossArgs(ctypes.Structure):
    _fields_ = [
        ("width", c_int32), ("height", c_int32),
        ("rgb", c_void_p), ("depth", c_void_p), ("normal", c_void_p),
        ("gt_rgb", c_void_p), ("gt_depth", c_void_p), ("gt_normal", c_void_p), ("depth_counts", c_void_p),
        ("ssim_lambda", c_float), ("depth_weight", c_float), ("depth_tolerance", c_float),
        ("maps", c_void_p), ("v_rgb", c_void_p), ("v_depth", c_void_p), ("v_normal", c_void_p), ("sums", c_void_p),
    ]


class RasterArgs(ctypes.Structure):
    _fields_ = [
        ("width", c_int32), ("height", c_int32), ("tile_size", c_int32), ("D", c_int32),
        ("splats", c_void_p), ("flatten_ids", c_void_p), ("tile_offsets", c_void_p), ("background", c_void_p),
        ("ed_channel", c_int32),
        ("render", c_void_p), ("alphas", c_void_p), ("last_ids", c_void_p),
        ("v_render", c_void_p), ("v_alphas", c_void_p),
        ("xy_split", c_int32),
        ("v_splats", c_void_p),
        ("dn", ctypes.POINTER(DnPost)),
        ("n_cameras", c_int32), ("keep_masks", c_void_p), ("keep_mask_stride", c_int64), ("pair_counters", c_void_p),
        ("saturation_flag", c_void_p), ("tile_ends", c_void_p), ("zero_fill", c_void_p), ("zero_fill_bytes", c_int64),
        ("det_partials", c_void_p), ("det_capacity", c_int64),
    ]


class DetArgs(ctypes.Structure):
    _fields_ = [
        ("n_records", c_int32), ("capacity", c_int64), ("n_isects", c_void_p), ("flatten_ids", c_void_p),
        ("partials", c_void_p), ("v_splats", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t),
    ]


class ProjGrads(ctypes.Structure):
    _fields_ = [
        ("radii", c_void_p), ("v_splats", c_void_p), ("v_means2d", c_void_p), ("v_depths", c_void_p),
        ("v_conics", c_void_p), ("v_compensations", c_void_p),
        ("v_means", c_void_p), ("v_quats", c_void_p), ("v_scales", c_void_p), ("v_opacities", c_void_p),
        ("v_sh0", c_void_p), ("v_sh0_stride", c_int32),
        ("v_shN", c_void_p), ("v_shN_stride", c_int32),
        ("v_colors", c_void_p),
        ("sh_grads_skip", c_int32),
        ("sh_factors", c_void_p),
        ("sh_grad_scale", c_float), ("sh_zero_state", c_void_p), ("sh_packed", c_void_p),
        ("zero_state_geometry", c_int32),
    ]


# every symbol include/dnsplat.h declares (tests/test_abi.py checks the .so exports all of them)
EXPORTS = [
    "dnsplat_strerror", "dnsplat_abi_version",
    "dnsplat_stamp",
    "dnsplat_project_fwd", "dnsplat_pack_splats",
    "dnsplat_bin_workspace_bytes", "dnsplat_bin_status_offset", "dnsplat_bin_prepare", "dnsplat_bin_emit_sort", "dnsplat_bin_isect_ids",
    "dnsplat_raster_fwd", "dnsplat_raster_bwd", "dnsplat_det_workspace_bytes", "dnsplat_det_reduce",
    "dnsplat_dn_depth_normals", "dnsplat_camera_prepare", "dnsplat_densify_stats", "dnsplat_densify_classify",
    "dnsplat_densify_split", "dnsplat_dn_loss", "dnsplat_ssim", "dnsplat_edge_aware_logl1", "dnsplat_tv_loss", "dnsplat_scale_reg", "dnsplat_sh_grads_from_factors", "dnsplat_sh_factors",
    "dnsplat_project_bwd", "dnsplat_sh_grads_add_factors", "dnsplat_packed_slab_floats", "dnsplat_visible_index",
    "dnsplat_sh_grads_from_packed",
]

_lib = None


def build(force: bool = False) -> Path:
    """Compile the HIP sources for gfx950 into ``libdnsplat.so`` next to this file (hipcc cross-compiles
    without a GPU)."""
    if os.environ.get("DNSPLAT_LIB"):
        return LIB_PATH
    srcs = list(CSRC_DIR.glob("*.hip")) + list(CSRC_DIR.glob("*.h")) + list((_PKG_DIR.parent / "include").glob("*.h"))
    stale = force or not LIB_PATH.exists() or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs)
    if stale:
        subprocess.run(["bash", str(CSRC_DIR / "build.sh")], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """The loaded library.  Raises if it is not built — no silent fallback."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise DnsplatError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `bash {CSRC_DIR / 'build.sh'}`. There is no CPU fallback.")
        L = ctypes.CDLL(str(LIB_PATH))
        L.dnsplat_strerror.restype = ctypes.c_char_p
        L.dnsplat_strerror.argtypes = [ctypes.c_int]
        L.dnsplat_abi_version.restype = ctypes.c_int
        L.dnsplat_stamp.argtypes = [c_void_p, c_void_p, ctypes.c_uint32, c_void_p]
        L.dnsplat_bin_workspace_bytes.restype = c_size_t
        L.dnsplat_bin_workspace_bytes.argtypes = [c_int32, c_int64, c_int32]
        L.dnsplat_bin_status_offset.restype = c_size_t
        L.dnsplat_bin_status_offset.argtypes = [c_int32, c_int64]
        L.dnsplat_project_fwd.argtypes = [ctypes.POINTER(Scene), ctypes.POINTER(Camera), ctypes.POINTER(ProjOut), c_void_p]
        L.dnsplat_project_bwd.argtypes = [ctypes.POINTER(Scene), ctypes.POINTER(Camera), ctypes.POINTER(ProjOut),
                                          ctypes.POINTER(ProjGrads), c_void_p]
        L.dnsplat_pack_splats.argtypes = [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]
        L.dnsplat_bin_prepare.argtypes = [ctypes.POINTER(BinArgs), c_void_p]
        L.dnsplat_bin_emit_sort.argtypes = [ctypes.POINTER(BinArgs), c_void_p]
        L.dnsplat_bin_isect_ids.argtypes = [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
        L.dnsplat_densify_classify.argtypes = [ctypes.POINTER(DensifyArgs), c_void_p]
        L.dnsplat_densify_split.argtypes = [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p]
        L.dnsplat_raster_fwd.argtypes = [ctypes.POINTER(RasterArgs), c_void_p]
        L.dnsplat_raster_bwd.argtypes = [ctypes.POINTER(RasterArgs), c_void_p]
        L.dnsplat_det_workspace_bytes.restype = c_size_t
        L.dnsplat_det_workspace_bytes.argtypes = [c_int64]
        L.dnsplat_det_reduce.argtypes = [ctypes.POINTER(DetArgs), c_void_p]
        L.dnsplat_dn_depth_normals.argtypes = [c_int32, c_int32, c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p]
        L.dnsplat_densify_stats.argtypes = [c_int32, c_void_p, c_void_p, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p]
        L.dnsplat_sh_grads_from_factors.argtypes = [c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p, c_int32,
                                                    c_void_p, c_int32, c_void_p]
        L.dnsplat_dn_loss.argtypes = [ctypes.POINTER(DnLossArgs), c_void_p]
        L.dnsplat_scale_reg.argtypes = [c_int32, c_void_p, c_float, c_void_p, c_void_p, c_void_p]
        L.dnsplat_ssim.argtypes = [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.dnsplat_edge_aware_logl1.argtypes = [c_int32, c_int32] + [c_void_p] * 9
        L.dnsplat_tv_loss.argtypes = [c_int32, c_int32, c_int32] + [c_void_p] * 5
        L.dnsplat_camera_prepare.argtypes = [c_void_p, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_int32, c_void_p]
        L.dnsplat_sh_factors.argtypes = [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.dnsplat_sh_grads_add_factors.argtypes = [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p,
                                                   c_int32, c_void_p, c_int32, c_void_p]
        L.dnsplat_packed_slab_floats.restype = c_size_t
        L.dnsplat_packed_slab_floats.argtypes = [c_int32, c_int32]
        L.dnsplat_visible_index.argtypes = [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.dnsplat_sh_grads_from_packed.argtypes = [c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_float,
                                                   c_void_p, c_int32, c_void_p, c_int32, c_void_p]
        for name in EXPORTS:
            if name not in ("dnsplat_strerror", "dnsplat_bin_workspace_bytes", "dnsplat_bin_status_offset", "dnsplat_det_workspace_bytes",
                            "dnsplat_packed_slab_floats"):
                getattr(L, name).restype = ctypes.c_int
        if L.dnsplat_abi_version() != ABI_VERSION:
            raise DnsplatError(f"libdnsplat ABI {L.dnsplat_abi_version()} != binding {ABI_VERSION}; rebuild")
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise DnsplatError(f"{what} failed: {lib().dnsplat_strerror(rc).decode()} (code {rc})")


class StageTimer:
    """Per-stage GPU time from HIP events recorded on the stream the kernels are enqueued on (torch's current
    stream — the one every dnsplat_* call receives).  Used by bench.py for the roofline figures; off by default."""

    def __init__(self, only=None):
        self.events = {}
        self.only = set(only) if only else None   # bracket these entry points only (an event pair costs ~10 us of stream time)

    def add(self, name, e0, e1):
        self.events.setdefault(name, []).append((e0, e1))

    def summary(self):
        """name -> (launches, mean ms, total ms); call after a stream/device synchronize."""
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = (len(ms), sum(ms) / len(ms), sum(ms))
        return out


class StampTimer:
    """Brackets entry points with device time stamps instead of HIP events (dnsplat_stamp): the only bracket that can be part of
    a frame captured into a HIP graph on ROCm 7.2.  Every call — and every REPLAY of a captured call — appends two stamps to a
    ring on the device; ``intervals_ticks()`` returns them pairwise after a synchronize."""

    RING = 1 << 14

    def __init__(self, only, device):
        self.only = set(only)
        self.ring = torch.zeros(self.RING, dtype=torch.int64, device=device)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=device)

    def stamp(self) -> None:
        check(lib().dnsplat_stamp(c_void_p(self.ring.data_ptr()), c_void_p(self.cursor.data_ptr()), self.RING,
                                  c_void_p(torch.cuda.current_stream().cuda_stream)), "dnsplat_stamp")

    def reset(self) -> None:
        self.cursor.zero_()

    def intervals_ticks(self):
        n = int(self.cursor.item())
        assert n <= self.RING and n % 2 == 0, n
        r = self.ring[:n].tolist()
        return [r[i + 1] - r[i] for i in range(0, n, 2)]


TIMER = None  # set to a StageTimer to record


def run(name: str, fn, *args) -> None:
    """Call one C-ABI entry point, raising on a non-zero code; brackets it with HIP events if TIMER is set."""
    t = TIMER
    if t is None or (t.only is not None and name not in t.only):
        check(fn(*args), name)
        return
    if isinstance(t, StampTimer):
        t.stamp()
        rc = fn(*args)
        t.stamp()
        check(rc, name)
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    t.add(name, e0, e1)
    check(rc, name)
