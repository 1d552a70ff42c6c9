"""Static check of the hand-placed LDS loads in raster_bwd.hip.

row_issue() starts three ds_read_b128 whose destination registers the compiler believes to be written
immediately; row_wait() carries the s_waitcnt.  The ISA must not touch those registers in between (a
compiler-inserted copy there would read data that has not landed), and it should not wait for them early
(that only costs time).  Compiles the file with -save-temps and
scans every raster_bwd_kernel instantiation.  Also checked: the step loop (the innermost loop around those loads) holds no
scratch access — the benchmark instantiation is pinned to 128 VGPRs and may spill, but only outside that loop.
Exit code 1 on a hazard.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dn-splatter_amd", "csrc", "raster_bwd.hip")


MIN_COVER = 8   # instructions the row-independent arithmetic places between row_issue() and row_wait()


def regs_of(line):
    used = set()
    for a, b, c in re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", line):
        if c:
            used.add(int(c))
        else:
            used |= set(range(int(a), int(b) + 1))
    return used


def check() -> int:
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-save-temps", "-c", SRC,
                        "-o", os.path.join(tmp, "rb.o")], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, "raster_bwd-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    kernels = re.findall(r"^(_ZN\S*raster_bwd_kernel\S*):", asm, re.M)
    assert kernels, "no raster_bwd_kernel found in the ISA"
    hazards = groups = early = spills = 0
    for name in kernels:
        body = re.search(re.escape(name) + r":(.*?)\.Lfunc_end", asm, re.S).group(1)
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        label_at = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r"(\.LBB\w+):", l)] if m}
        i = 0
        while i < len(lines):
            if all(i + k < len(lines) and lines[i + k].startswith("ds_read_b128") for k in range(3)):
                dest = set()
                for l in lines[i:i + 3]:
                    a, b = re.search(r"v\[(\d+):(\d+)\]", l).groups()
                    dest |= set(range(int(a), int(b) + 1))
                groups += 1
                # the step loop: from the target of the first backward branch after the loads to that branch
                for e in range(i, len(lines)):
                    m = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)", lines[e])
                    if m and label_at.get(m.group(1), len(lines)) <= i:
                        in_loop = [l for l in lines[label_at[m.group(1)]:e] if l.startswith("scratch_")]
                        if in_loop:
                            spills += 1
                            print(f"SCRATCH IN THE STEP LOOP of {name}: {in_loop[:3]}")
                        break
                else:
                    raise AssertionError("step loop not found")
                j = i + 3
                # a wait fewer than MIN_COVER instructions after the loads was put there by the compiler (a pending LDS
                # result from before the loop, WAW on one of its registers): correct, but the LDS latency is then exposed
                k = j
                while k < len(lines) and not (lines[k].startswith("s_waitcnt") and "lgkmcnt(0)" in lines[k]):
                    k += 1
                if k - j < MIN_COVER:
                    early += 1
                    print(f"EARLY WAIT in {name}: only {k - j} instructions between the row loads and `{lines[k] if k < len(lines) else '?'}`")
                while j < len(lines) and not (lines[j].startswith("s_waitcnt") and "lgkmcnt(0)" in lines[j]):
                    if not lines[j].startswith(".") and regs_of(lines[j]) & dest:
                        hazards += 1
                        print(f"HAZARD in {name}: `{lines[j]}` touches in-flight v{sorted(regs_of(lines[j]) & dest)}")
                    if lines[j].startswith(("s_cbranch", "s_branch", "s_endpgm")):
                        hazards += 1
                        print(f"HAZARD in {name}: control flow `{lines[j]}` before the wait")
                    j += 1
                i = j
            else:
                i += 1
    print(f"{len(kernels)} kernels, {groups} load groups, {hazards} hazards, {early} early waits, {spills} step loops with scratch accesses")
    return 1 if hazards or early or spills or groups < len(kernels) else 0


if __name__ == "__main__":
    sys.exit(check())
