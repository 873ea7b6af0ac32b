#!/usr/bin/env python3
"""Static instruction statistics of one kernel in a hipcc -S listing.

    hipcc --offload-arch=gfx950 ... -S --cuda-device-only -o k.s file.hip
    python tools/isa_hist.py k.s <substring of the kernel symbol> [--blocks] [--hot] [--mnem]

Prints the basic blocks (label, instruction count, VALU count, terminator), the
outermost backward branch (= the row loop of the marching kernels) and a histogram
of the mnemonics inside that loop, grouped into classes.  Blocks of lazily evaluated
branches (shock paths) are inside the loop's address range and counted; --blocks lists
them one by one so that the straight-line path can be summed by hand.
"""
import collections
import re
import sys


def kernel_lines(path, key):
    out, on = [], False
    for ln in open(path):
        s = ln.rstrip("\n")
        if not on:
            if re.match(r"^_Z\S*:\s*(;.*)?$", s) and key in s:
                on = True
            continue
        if s.startswith("\t.section") or s.startswith(".Lfunc_end"):
            break
        out.append(s)
    return out


def classify(m):
    if m.startswith("v_mov_b32_dpp") or m.endswith("_dpp"):
        return "dpp"
    if m.startswith(("v_fma_f64", "v_fmac_f64")):
        return "fma64"
    if m.startswith("v_mul_f64"):
        return "mul64"
    if m.startswith("v_add_f64"):
        return "add64"
    if m.startswith(("v_max_f64", "v_min_f64")):
        return "minmax64"
    if m.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_", "v_frexp", "v_ldexp",
                     "v_rcp_", "v_rsq_", "v_log", "v_exp", "v_trig", "v_fract")):
        return "trans/div"
    if m.startswith("v_cmp") or m.startswith("v_cmpx"):
        return "cmp"
    if m.startswith("v_cndmask"):
        return "cndmask"
    if m.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr")):
        return "mov"
    if m.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if m.startswith("v_"):
        return "valu-other"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if m.startswith("s_waitcnt"):
        return "waitcnt"
    if m.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = kernel_lines(path, key)
    # instruction list with the label each one sits under
    insts, labels, cur = [], {}, "entry"
    for s in lines:
        t = s.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\S+):", t)
            if m:
                cur = m.group(1)
                labels[cur] = len(insts)
            continue
        m = re.match(r"^(\.LBB\S+):", t)
        if m:
            cur = m.group(1)
            labels[cur] = len(insts)
            continue
        mn = t.split()[0]
        insts.append((mn, t, cur))
    print(f"{len(insts)} instructions in kernel")
    # backward branches
    back = []
    for n, (mn, t, _) in enumerate(insts):
        if mn.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[1]
            if tgt in labels and labels[tgt] <= n:
                back.append((labels[tgt], n, tgt))
    back.sort(key=lambda b: b[0] - b[1])
    if not back:
        lo, hi = 0, len(insts) - 1
        print("no loop found")
    elif "--hot" in sys.argv:
        # The row loop proper: the compiler moves the bodies of unlikely branches (the two-rarefaction
        # `pow` blocks of the Riemann solver, the supersonic fluxes) BEHIND the loop and lets them
        # jump back into it, so the widest backward branch spans the loop AND that cold code.  The
        # loop's own back edge is the one to the EARLIEST head; what lies behind its source and
        # branches back into the loop is listed apart as out-of-line code.
        # (a loop's range holds no branch to before its head -- a few instructions of slack: a
        # `continue` may re-enter through a short preamble of the head block)
        def closed(lo_, hi_):
            for n_, (mn_, t_, _) in enumerate(insts[lo_:hi_ + 1]):
                if mn_.startswith(("s_cbranch", "s_branch")):
                    tg = t_.split()[1]
                    if tg in labels and labels[tg] < lo_ - 8:
                        return False
            return True
        ok = [b for b in back if closed(b[0], b[1])]
        lo, hi, tgt = max(ok, key=lambda b: b[1] - b[0]) if ok else back[0]
        cold = [b for b in back if b[1] > hi and lo <= b[0] <= hi]
        ncold = (max(b[1] for b in cold) - hi) if cold else 0
        print(f"row loop: {tgt} .. instruction {hi}  ({hi - lo + 1} instructions in line); "
              f"{ncold} instructions of out-of-line blocks behind it ({len(cold)} jump back into the loop)")
    else:
        lo, hi, tgt = back[0]
        print(f"outermost loop: {tgt} .. instruction {hi}  ({hi - lo + 1} instructions); "
              f"{len(back)} backward branches")
    hist = collections.Counter()
    mnem = collections.Counter()
    for mn, t, _ in insts[lo:hi + 1]:
        hist[classify(mn)] += 1
        mnem[mn] += 1
    valu = sum(v for k, v in hist.items() if k in ("dpp", "fma64", "mul64", "add64", "minmax64",
                                                    "trans/div", "cmp", "cndmask", "mov", "lane",
                                                    "valu-other"))
    print(f"VALU in loop range: {valu}")
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
        print(f"  {k:12s} {v}")
    if "--mnem" in sys.argv:
        for k, v in sorted(mnem.items(), key=lambda kv: -kv[1]):
            print(f"    {k:28s} {v}")
    if show_blocks:
        blocks = collections.OrderedDict()
        for n, (mn, t, lab) in enumerate(insts[lo:hi + 1]):
            b = blocks.setdefault(lab, [0, 0, ""])
            b[0] += 1
            if mn.startswith("v_"):
                b[1] += 1
            if mn.startswith(("s_cbranch", "s_branch")):
                b[2] += " " + mn.replace("s_cbranch_", "").replace("s_branch", "br") + "->" + t.split()[1]
        for lab, (n, nv, term) in blocks.items():
            print(f"  {lab:14s} n={n:5d} valu={nv:5d} {term}")


if __name__ == "__main__":
    main()
