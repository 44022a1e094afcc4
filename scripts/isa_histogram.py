#!/usr/bin/env python3
"""Developer tool: static opcode histogram of the blend kernels' per-splat loop bodies.
   python scripts/isa_histogram.py > profiles/<tag>_blend_opcode_histogram.txt
Compiles rasterize.hip to gfx950 assembly with the library's flags and counts, for the innermost loop (depth 2) of
K16 (rasterize_kernel<true,false>) and of K17 (rasterize_backward_kernel<false>: both the clamp and the no-clamp copy),
instructions by issue class.  Static counts = one trip with all four quadrants taken and the reduction executed; the
dynamic figure (SQ_INSTS_VALU per blended intersection, profiles/*_sq_counters.csv) is lower because quadrants are skipped."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -S --cuda-device-only".split()
QUARTER = ("v_rcp_", "v_sqrt_", "v_rsq_", "v_exp_", "v_log_", "v_sin_", "v_cos_", "v_permlane")
HALF = ("v_cmp", "v_cndmask", "v_min_", "v_max_", "v_med3", "v_max3", "v_min3")


def classify(op, operands):
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait/nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("ds_",)):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("v_"):
        if op.startswith(QUARTER):
            return "VALU quarter-rate (transcendental / permlane swap)"
        if op.startswith(HALF) or "_dpp" in op or " row_" in operands or "quad_perm" in operands:
            return "VALU half-rate (cmp / cndmask / min / max / DPP)"
        if re.search(r"(^|[ ,\[])s\d+|s\[\d+:\d+\]", operands.split(",", 1)[1] if "," in operands else ""):
            return "VALU with an SGPR source (half rate)"
        return "VALU full-rate"
    return "other"


def loops(lines):
    """{(header label, depth): [instructions]}: a block belongs to the loop named in its label comment ('in Loop: Header=X
    Depth=N'); a header block ('=>This [Inner] Loop Header: Depth=N', possibly on the comment line after the label) to itself."""
    out = {}
    cur = None
    for n, ln in enumerate(lines):
        if ln.startswith(".LBB"):
            lab = ln.split(":")[0]
            ctx = ln + " " + " ".join(l for l in lines[n + 1:n + 4] if l.lstrip().startswith(";"))
            own = re.search(r"=>\s*This (Inner )?Loop Header: Depth=(\d+)", ctx)
            hdr = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", ln)
            if own:
                cur = (lab, int(own.group(2)))
            elif hdr:
                cur = (".L" + hdr.group(1), int(hdr.group(2)))
            else:
                cur = None
            if cur:
                out.setdefault(cur, [])
            continue
        if cur and ln.startswith("\t") and not ln.strip().startswith((";", ".")):
            out[cur].append(ln.strip())
    return out


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "r.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "brush_amd/csrc"), "-I" + os.path.join(ROOT, "include"),
                                                                  os.path.join(ROOT, "brush_amd/csrc/rasterize.hip"), "-o", asm], stderr=subprocess.DEVNULL)
        text = open(asm).read().split("\n")
    kernels = {"K16 rasterize_kernel<BWD_INFO=true, SMOOTH=false, PHASE=1>": "_ZN2bh16rasterize_kernelILb1ELb0ELi1EEEv",
               "K17 rasterize_backward_kernel<SMOOTH=false>": "_ZN2bh25rasterize_backward_kernelILb0EEEv"}
    for title, prefix in kernels.items():
        start = next(i for i, l in enumerate(text) if l.startswith(prefix) and ": ; @" in l)
        end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
        body = text[start:end]
        vg = next((l.split()[-1] for l in text[end:end + 400] if "amdhsa_next_free_vgpr" in l), "?")
        print("=" * 100)
        print("%s   (%d lines of assembly, %s VGPRs)" % (title, len(body), vg))
        allloops = loops(body)
        deepest = max(k[1] for k in allloops)   # the per-splat loops (K16: depth 2; K17: depth 3 since the two-segment loop of round 3)
        inner = {k: v for k, v in allloops.items() if k[1] == deepest}
        for (lab, depth), ins in inner.items():
            hist = collections.Counter()
            ops = collections.Counter()
            for l in ins:
                parts = l.split(None, 1)
                op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
                hist[classify(op, rest)] += 1
                ops[re.sub(r"_e(32|64)$", "", op)] += 1
            valu = sum(n for c, n in hist.items() if c.startswith("VALU"))
            print("-" * 100)
            print("per-splat loop at %s: %d instructions, %d VALU" % (lab, sum(hist.values()), valu))
            for c, n in sorted(hist.items(), key=lambda x: -x[1]):
                print("    %-62s %4d" % (c, n))
            print("    opcodes: " + "  ".join("%s x%d" % (o, n) for o, n in ops.most_common(28)))


if __name__ == "__main__":
    main()
