"""Static cost estimate of a kernel's hot loop from its ISA: vector-issue cycles per iteration with the per-opcode
costs measured by tools/ubench/valu_rate.hip (nominal 2.4 GHz cycles per wave64 instruction per SIMD).

    python tools/valu_model.py raster_bwd 'raster_bwd_kernelILi7ELi4ELb1' [extra hipcc flags]
    python tools/valu_model.py raster_fwd 'raster_fwd_kernelILi7ELb1'

The loop is found as the innermost backward branch region that contains a v_exp_f32.  Scalar, LDS and branch
instructions are listed but not priced (they issue on other ports); the model matched the measured step time of the
backward kernel within 3 %."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COST = [
    (r"v_pk_", 4.7), (r"v_exp_|v_rcp_|v_rsq_|v_sqrt_|v_log_", 8.3), (r"v_cmp", 4.9), (r"v_cndmask_b32_e64", 4.9),
    (r"v_cndmask_b32_e32", 2.6), (r"v_mov_b32_dpp", 4.8), (r"v_fma_f32|v_mad_|v_add_f32_e64|v_sub_f32_e64|v_mul_f32_e64|v_min_f32_e64|v_max_f32_e64|v_med3|v_add3|v_lshl_add|v_mul_lo_u32|v_mul_hi", 3.75),
    (r"v_cvt_", 4.4), (r"v_mov_b64", 4.7), (r"v_", 2.75),
]


def cost(op, line):
    for pat, c in COST:
        if re.match(pat, op):
            if c == 2.75 and re.search(r"0x[0-9a-f]{5,}", line):     # 32-bit literal: 8-byte encoding
                return 3.75
            return c
    return 0.0


def main():
    name, sym = sys.argv[1], sys.argv[2]
    flags = sys.argv[3:]
    src = os.path.join(ROOT, "dn-splatter_amd", "csrc", name + ".hip")
    extra = ["-fno-slp-vectorize"] if name == "raster_bwd" else []
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, *flags, "-S",
                        "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    m = re.search(r"^(_Z\S*" + re.escape(sym) + r"\S*):(.*?)s_endpgm", asm, re.S | re.M)
    assert m, "kernel not found"
    lines = [l.split(";")[0].strip() for l in m.group(2).split("\n")]
    lines = [l for l in lines if l]
    # label positions
    pos = {l[:-1]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:$", l)}
    best = None
    for i, l in enumerate(lines):
        b = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if b and b.group(1) in pos and pos[b.group(1)] < i:
            body = lines[pos[b.group(1)]:i + 1]
            if any(x.startswith("v_exp_f32") for x in body) and (best is None or len(body) < len(best)):
                best = body
    assert best, "no loop with v_exp_f32 found"
    hist, total, nv, ns, nl = {}, 0.0, 0, 0, 0
    for l in best:
        op = l.split()[0]
        if op.startswith("v_"):
            c = cost(op, l)
            total += c
            nv += 1
            hist[op] = hist.get(op, 0) + 1
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch")):
            ns += 1
        elif op.startswith("ds_"):
            nl += 1
    print(f"{m.group(1)[-48:]}: loop of {len(best)} lines: {nv} vector ({total:.0f} modelled cycles), {ns} scalar, {nl} LDS")
    print("  " + "  ".join(f"{k}:{v}" for k, v in sorted(hist.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
