#!/usr/bin/env bash
# Round-4 call 18: the whole GPU suite and smoke() on the final sources of the round.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04r; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== pytest -m gpu"; DNSPLAT_MARGIN_LOG=$R/$O/margins.tsv timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_gpu.log | head -20
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
