"""Live VGPR count at every point of a kernel, from its ISA (CFG liveness over the physical registers hipcc chose).

    python tools/vgpr_pressure.py raster_bwd 'raster_bwd_kernelILi7ELi4ELb1' [extra hipcc flags]

Prints the blocks with the highest pressure.  The register COUNT hipcc reports is the highest index it used, not the
peak number of simultaneously live values; this tells the two apart (how far a kernel is from the next occupancy step)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(s):
    out = set()
    for a, b, c in re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", s):
        out |= {int(c)} if c else set(range(int(a), int(b) + 1))
    return out


def def_use(line):
    parts = line.split(None, 1)
    op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
    ops = [o.strip() for o in rest.split(",")]
    if op.startswith(("v_cmp", "ds_write", "global_store", "global_atomic", "scratch_store", "s_", "buffer_store")):
        return set(), regs(rest)
    if op.startswith(("v_", "ds_read", "ds_bpermute", "global_load", "scratch_load", "buffer_load")):
        d, u = regs(ops[0]), regs(",".join(ops[1:]))
        if op.startswith(("v_fmac", "v_mov_b32_dpp", "v_mac")):
            u |= d
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            return set(), regs(rest)
        return d, u
    return set(), regs(rest)


def main():
    name, sym = sys.argv[1], sys.argv[2]
    flags = sys.argv[3:]
    src = name if os.path.exists(name) else os.path.join(ROOT, "dn-splatter_amd", "csrc", name + ".hip")
    extra = ["-fno-slp-vectorize"] if "raster_bwd" in name else []
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "dn-splatter_amd", "csrc"),
                        *extra, *flags, "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    m = re.search(r"^(_Z\S*" + re.escape(sym) + r"\S*):(.*?)s_endpgm", asm, re.S | re.M)
    lines = [l.split(";")[0].strip() for l in (m.group(2) + "s_endpgm").split("\n")]
    lines = [l for l in lines if l]
    # basic blocks
    blocks, cur, label_of = [], {"label": "entry", "ins": []}, {}
    for l in lines:
        lab = re.match(r"^(\.LBB\d+_\d+):$", l)
        if lab:
            if cur["ins"] or cur["label"] == "entry":
                blocks.append(cur)
            cur = {"label": lab.group(1), "ins": []}
            continue
        cur["ins"].append(l)
        if re.match(r"s_c?branch|s_endpgm|s_setpc", l):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    if cur["ins"]:
        blocks.append(cur)
    for i, b in enumerate(blocks):
        if b["label"]:
            label_of[b["label"]] = i
    succ = []
    for i, b in enumerate(blocks):
        s = []
        last = b["ins"][-1] if b["ins"] else ""
        t = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", last)
        if t and t.group(1) in label_of:
            s.append(label_of[t.group(1)])
        if not last.startswith(("s_branch", "s_endpgm")) and i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    live_in = [set() for _ in blocks]
    live_out = [set() for _ in blocks]
    changed = True
    while changed:
        changed = False
        for i in reversed(range(len(blocks))):
            lo = set().union(*[live_in[j] for j in succ[i]]) if succ[i] else set()
            li = set(lo)
            for l in reversed(blocks[i]["ins"]):
                d, u = def_use(l)
                li = (li - d) | u
            if lo != live_out[i] or li != live_in[i]:
                live_out[i], live_in[i], changed = lo, li, True
    rows = []
    for i, b in enumerate(blocks):
        live = set(live_out[i])
        peak, at = len(live), "(block end)"
        for l in reversed(b["ins"]):
            d, u = def_use(l)
            live = (live - d) | u
            if len(live | d) > peak:
                peak, at = len(live | d), l
        rows.append((peak, i, b["label"] or "(fallthrough)", len(b["ins"]), at))
    rows.sort(reverse=True)
    print(f"{m.group(1)[-44:]}: {len(blocks)} blocks, peak live VGPRs {rows[0][0]}")
    for peak, i, lab, n, at in rows[:12]:
        print(f"  block {i:3d} {lab:14s} {n:4d} instrs  peak {peak:3d}  at `{at[:70]}`")


if __name__ == "__main__":
    main()
