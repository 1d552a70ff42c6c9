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

// ---- time stamps inside a captured frame -----------------------------------------------------------------------------------
// A frame replayed as a HIP graph has no host code between its kernels, and on ROCm 7.2 an event cannot be recorded into a graph
// under capture (hipEventRecordWithFlags(hipEventRecordExternal) returns hipErrorInvalidValue; PyTorch-ROCm refuses such events
// outright).  The bracket around a stage is therefore a pair of one-thread kernels that append the constant-rate wall clock
// (s_memrealtime) to a ring: every replay leaves its own pair, the host reads them all after the timed region and calibrates the
// tick against HIP events recorded around a stamped interval outside the graph (bench.py).
namespace {
__global__ void stamp_kernel(unsigned long long *ring, unsigned int *cursor, unsigned int ring_size)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned int i = atomicAdd(cursor, 1u);
        ring[i % ring_size] = wall_clock64();
    }
}
}  // namespace

extern "C" int dnsplat_stamp(uint64_t *ring, uint32_t *cursor, uint32_t ring_size, dnsplat_stream_t stream)
{
    if (!ring || !cursor || ring_size == 0) return DNSPLAT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long *>(ring), cursor,
                       ring_size);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}
