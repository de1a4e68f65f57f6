"""Static instruction-class histogram of the gfx950 code of selected kernels (no GPU needed).

    python tools/isa_histogram.py [--src mw_dycore.hip] [--csrc DIR] [--md] <kernel-substring> [...]     (-- extra hipcc flags)

Compiles the source to device assembly (hipcc -S --offload-device-only) and, per kernel whose demangled name contains one of the
substrings, counts instructions by class -- for the whole kernel and for its largest loop (the marching loop: the basic blocks
between the back-edge target with the most instructions and its s_cbranch).  Classes:
  fp64 arithmetic (v_fma/mul/add/... _f64, v_rcp/rsq/...), v_mov (window shifts), DPP moves, v_cndmask, readlane/writelane
  (SGPR spills live in VGPR lanes), integer / address VALU, v_cmp, scalar ALU, scalar memory, vector memory, LDS, waits / branches.
"""
import os
import re
import subprocess
import sys
from collections import Counter, OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=on", "-I/opt/rocm/include"]

CLASSES = ["fp64 arith", "v_mov", "dpp mov", "v_cndmask", "readlane/writelane", "int/address VALU", "v_cmp", "other VALU",
           "SALU", "SMEM", "VMEM load", "VMEM store", "LDS", "wait/branch/misc"]


def classify(op, rest):
    if op.startswith("v_"):
        if "dpp" in rest or op.endswith("_dpp"):
            return "dpp mov"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return "readlane/writelane"
        if op.startswith("v_cndmask"):
            return "v_cndmask"
        if op.startswith("v_cmp"):
            return "v_cmp"
        if op.startswith(("v_mov_b", "v_accvgpr")):
            return "v_mov"
        if re.search(r"_f64|_f32", op) and not op.startswith("v_cvt"):
            return "fp64 arith"
        if re.match(r"v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|bfe|bfi|min|max|not|mbcnt|add3|lshl_add|lshl_or|and_or|or3|addc|subb|perm|alignbit|cvt)", op):
            return "int/address VALU"
        return "other VALU"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store"):
        return "SMEM"
    if op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load")):
        return "VMEM load"
    if op.startswith(("global_store", "flat_store", "buffer_store", "scratch_store", "global_atomic")):
        return "VMEM store"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_nop", "s_barrier", "s_endpgm", "s_setpc", "s_swappc", "s_getpc", "s_sleep", "s_setprio")):
        return "wait/branch/misc"
    if op.startswith("s_"):
        return "SALU"
    return "wait/branch/misc"


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void mw::", "")
    except Exception:
        return n


def parse_kernels(asm):
    """-> {mangled: [(label or None, op, rest), ...]}"""
    kernels, cur, name = OrderedDict(), None, None
    for ln in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            name = m.group(1); cur = []; kernels[name] = cur; continue
        if cur is None:
            continue
        s = ln.strip()
        if s.startswith(".Lfunc_end"):
            cur = None; continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur.append((m.group(1), None, None)); continue
        if not s or s.startswith((";", ".", "//")):
            continue
        parts = s.split(None, 1)
        cur.append((None, parts[0], parts[1] if len(parts) > 1 else ""))
    return kernels


def largest_loop(items):
    """the instruction range [target label, backward branch] that contains the most instructions"""
    pos = {}
    for i, (lab, op, rest) in enumerate(items):
        if lab:
            pos[lab] = i
    best = (0, 0, 0)
    for i, (lab, op, rest) in enumerate(items):
        if op and op.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"(\.LBB\d+_\d+)", rest)
            if m and m.group(1) in pos and pos[m.group(1)] < i:
                n = sum(1 for x in items[pos[m.group(1)]:i + 1] if x[1])
                if n > best[0]:
                    best = (n, pos[m.group(1)], i + 1)
    return items[best[1]:best[2]]


def hist(items):
    c = Counter()
    for lab, op, rest in items:
        if op:
            c[classify(op, rest)] += 1
    return c


def detail_fp(items):
    c = Counter()
    for lab, op, rest in items:
        if op and classify(op, rest) == "fp64 arith":
            c[re.sub(r"_e(32|64)$", "", op)] += 1
    return c


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--"); extra = args[i + 1:]; args = args[:i]
    src = "mw_dycore.hip"
    csrc = os.path.join(ROOT, "miniweatherml_amd", "csrc")
    md = False
    filt = []
    it = iter(args)
    for a in it:
        if a == "--src":
            src = next(it)
        elif a == "--csrc":                      # another checkout's csrc directory (e.g. `git archive <rev>` for a before / after table)
            csrc = next(it)
        elif a == "--md":
            md = True
        else:
            filt.append(a)
    os.makedirs("/tmp/rr", exist_ok=True)
    out = "/tmp/rr/isa_%s.s" % os.path.splitext(src)[0]
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-x", "hip", "--offload-device-only", "-S", os.path.join(csrc, src), "-o", out] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr); sys.exit(1)
    kernels = parse_kernels(open(out).read())
    for mangled, items in kernels.items():
        name = demangle(mangled)
        if filt and not any(f in name for f in filt):
            continue
        whole, loop = hist(items), hist(largest_loop(items))
        tw, tl = sum(whole.values()), sum(loop.values())
        valu_cls = CLASSES[:8]
        vw, vl = sum(whole[c] for c in valu_cls), sum(loop[c] for c in valu_cls)
        if md:
            print("\n### `%s`\n\n| class | whole kernel | marching loop |\n|---|---|---|" % name)
            for c in CLASSES:
                print("| %s | %d | %d |" % (c, whole[c], loop[c]))
            print("| **all instructions** | %d | %d |\n| **VALU** | %d | %d |\n| VALU that is not fp arithmetic | %d (%.0f %%) | %d (%.0f %%) |" % (
                tw, tl, vw, vl, vw - whole["fp64 arith"], 100.0 * (vw - whole["fp64 arith"]) / max(vw, 1), vl - loop["fp64 arith"],
                100.0 * (vl - loop["fp64 arith"]) / max(vl, 1)))
            fp = detail_fp(largest_loop(items))
            print("\nfp instructions of the loop: " + ", ".join("%s %d" % kv for kv in fp.most_common(8)))
        else:
            print("%-40s all %5d / loop %5d | VALU %5d / %5d | " % (name[:40], tw, tl, vw, vl) +
                  " ".join("%s %d/%d" % (c.split()[0], whole[c], loop[c]) for c in CLASSES[:8]))


if __name__ == "__main__":
    main()
