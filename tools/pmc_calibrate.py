"""Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this chip (MI355X_MICROARCH.md "HBM":
"WRITE_SIZE [is] uncalibrated: calibrate on a known byte count in your own access pattern").

    cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out -o p -- python tools/pmc_calibrate.py

Launches, each on buffers far larger than the 256 MiB Infinity Cache:
  fill      1 GiB streaming write (torch fill_)                        expected write 1 GiB, read 0
  copy      1 GiB -> 1 GiB (torch copy_)                               expected read 1 GiB, write 1 GiB
  strided12 N x 12-byte rows written by one lane per row (the access pattern of project_bwd's v_means stores)
tools/pmc_summary.py prints per-kernel averages; the ratio reported / expected is the correction factor.
"""
import torch

dev = "cuda:0"
n = 256 * 1024 * 1024            # floats = 1 GiB
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
for _ in range(3):
    a.fill_(1.0)
torch.cuda.synchronize()
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
# one lane per 12-byte row: index_put of [N,3] rows through a strided view (elementwise kernel, lane = row element triple)
rows = torch.empty(64 * 1024 * 1024, 3, dtype=torch.float32, device=dev)   # 768 MiB
src = torch.ones(64 * 1024 * 1024, 1, dtype=torch.float32, device=dev)
for _ in range(3):
    rows.copy_(src.expand(-1, 3))
torch.cuda.synchronize()
print("calibration launches done: fill 1 GiB x3, copy 1 GiB x3, rows 768 MiB x3")
