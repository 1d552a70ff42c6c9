"""pytest plugin (diagnostic): records every call site that converts a tensor with requires_grad=True to a Python float — torch warns
about it only ONCE per process, so `-W error::UserWarning` finds one site per run.   python -m pytest -p tools.float_sites_plugin ...
Writes gpurun_out/float_sites.txt."""
import os
import traceback

import torch

_sites = set()
_orig = torch.Tensor.__float__


def _patched(self):
    if self.requires_grad:
        for fr in reversed(traceback.extract_stack()[:-1]):
            if os.sep + "tests" + os.sep in fr.filename or fr.filename.endswith(("bench.py", "__graft_entry__.py")):
                _sites.add(f"{fr.filename}:{fr.lineno}: {fr.line}")
                break
        return _orig(self.detach())
    return _orig(self)


torch.Tensor.__float__ = _patched


def pytest_sessionfinish(session, exitstatus):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/float_sites.txt", "w") as f:
        f.write("\n".join(sorted(_sites)) + "\n")
