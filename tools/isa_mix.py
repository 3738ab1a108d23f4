"""Static instruction mix of a kernel's ISA, segment by segment (a segment = the code between two s_barrier).

    hipcc ... --cuda-device-only -S mcd_inst.hip -o k.s ; python tools/isa_mix.py k.s score_kernel [--ops]

Classes: mfma | dpp (v_*_dpp) | fp (other floating-point VALU) | int (integer / move / select / lane VALU) | salu | lds | vmem |
scratch | other.  Branches make the static count an upper bound of what one wave issues; the step loop's segments are the ones
holding MFMAs."""
import collections
import re
import sys

FP = re.compile(r"v_(pk_)?(add|sub|mul|fma|fmac|mac|mad|med3|max|min|exp|log|rcp|rsq|sqrt|sin|cos|cvt|fract|floor|ceil|trunc|rndne|ldexp|cmp_\w+|cmpx_\w+)_(f32|f16|f64|legacy_f32)")


def classify(op: str, line: str) -> str:
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        if "_dpp" in op or "row_newbcast" in line or "quad_perm" in line or "row_mirror" in line or "row_half_mirror" in line:
            return "dpp"
        if FP.match(op) or op.startswith("v_cvt_") or op.startswith("v_dot"):
            return "fp"
        return "int"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def main() -> None:
    path, name = sys.argv[1], sys.argv[2]
    show_ops = "--ops" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + name + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    segs, cur = [], collections.Counter()
    ops = collections.Counter()
    seg_ops = []
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if op == "s_barrier":
            segs.append(cur); seg_ops.append(ops)
            cur, ops = collections.Counter(), collections.Counter()
            continue
        c = classify(op, s)
        cur[c] += 1
        if c == "int":
            ops[op] += 1
    segs.append(cur); seg_ops.append(ops)
    cols = ["mfma", "dpp", "fp", "int", "salu", "lds", "vmem", "scratch", "other"]
    print("seg  " + " ".join(f"{c:>7s}" for c in cols))
    tot = collections.Counter()
    tot_ops = collections.Counter()
    for i, (sg, so) in enumerate(zip(segs, seg_ops)):
        print(f"{i:3d}  " + " ".join(f"{sg[c]:7d}" for c in cols) + ("   <- step loop" if sg["mfma"] else ""))
        if sg["mfma"]:
            tot.update(sg); tot_ops.update(so)
            if show_ops:
                print("       int ops: " + ", ".join(f"{k} {v}" for k, v in so.most_common(12)))
    print("segments with MFMAs: " + " ".join(f"{c} {tot[c]}" for c in cols))
    print("integer / move VALU opcodes there: " + ", ".join(f"{k} {v}" for k, v in tot_ops.most_common(30)))


if __name__ == "__main__":
    main()
