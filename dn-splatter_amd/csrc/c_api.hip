// c_api.hip — the non-kernel part of the C ABI declared in include/dnsplat.h.
#include "splat_common.h"

extern "C" const char *dnsplat_strerror(int code)
{
    switch (code) {
        case DNSPLAT_OK: return "ok";
        case DNSPLAT_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or inconsistent channel setup)";
        case DNSPLAT_ERR_WORKSPACE: return "workspace too small (see dnsplat_bin_workspace_bytes)";
        case DNSPLAT_ERR_LAUNCH: return "HIP kernel launch / async copy failed";
        case DNSPLAT_ERR_UNSUPPORTED: return "unsupported configuration (tile_size != 16, > 8 channels, or SH degree > 3)";
        default: return "unknown dnsplat error code";
    }
}

extern "C" int dnsplat_abi_version(void) { return DNSPLAT_ABI_VERSION; }
