"""Development check (needs /root/reference, never run on the GPU box): every constant that tools/gen_weno_tables.py derived equals,
as a double, the corresponding `_fp` literal of the reference (parsed as long double, then converted -- main_header.h:61-63), and the
terms appear in the same order.    python tools/check_weno_tables.py"""
import os
import re
import sys

import numpy as np

REF = "/root/reference/model/modules/helpers"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ref_double(lit):
    return float(np.longdouble(lit))


def split_terms(expr):
    out, cur, depth = [], "", 0
    for ch in expr:
        if ch == "(": depth += 1
        if ch == ")": depth -= 1
        if ch in "+-" and depth == 0 and cur and not cur.endswith("e"):
            out.append(cur); cur = ch
        else:
            cur += ch
    out.append(cur)
    return out


def parse_ref():
    src = open(os.path.join(REF, "WenoLimiter_recon.h")).read().splitlines()
    res = {}
    for i, ln in enumerate(src):
        m = re.search(r"static real TV\(SArray<real,1,(\d)>", ln)
        if m and m.group(1) in "79":
            expr = src[i + 2].strip().split("=", 1)[1].rstrip(";")
            res["tv%s" % m.group(1)] = [(ref_double(re.match(r"[+-]?([0-9.e]+)_fp", t).group(1)), tuple(int(v) for v in re.findall(r"a\((\d)\)", t)))
                                        for t in split_terms(expr)]
        m = re.search(r"static void coefs([79])\(", ln)
        if m:
            N = int(m.group(1)); rows = []
            for r in range(N):
                expr = src[i + 1 + r].strip().split("=", 1)[1].rstrip(";")
                row = []
                for t in split_terms(expr):
                    mm = re.match(r"([+-]?)([0-9.e-]+)_fp\*v(\d)", t)
                    row.append(((-1 if mm.group(1) == "-" else 1) * ref_double(mm.group(2)), int(mm.group(3))))
                rows.append(row)
            res["coefs%d" % N] = rows
    tm = open(os.path.join(REF, "TransformMatrices.h")).read()
    return res, tm


def parse_mine():
    txt = open(os.path.join(ROOT, "oracle", "weno79.inc")).read()
    res = {}
    for N in (7, 9):
        body = re.search(r"void mw_coefs%d\(.*?\{\n(.*?)\n\}" % N, txt, re.S).group(1)
        rows = []
        for ln in body.splitlines():
            if "=" not in ln: continue
            expr = ln.split("=", 1)[1].strip().rstrip(";")
            row = []
            for t in re.findall(r"([+-]?)\s*(0x[0-9a-f.]+p[+-]\d+)\*s(\d)", expr):
                row.append(((-1 if t[0] == "-" else 1) * float.fromhex(t[1]), int(t[2])))
            rows.append(row)
        res["coefs%d" % N] = rows
        body = re.search(r"REAL mw_tv%d\(.*?\{\n.*?return (.*?);\n\}" % N, txt, re.S).group(1)
        res["tv%d" % N] = [(float.fromhex(re.match(r"(0x[0-9a-f.]+p[+-]\d+)", t.strip()).group(1)), tuple(int(v) for v in re.findall(r"a\[(\d)\]", t)))
                           for t in body.split(" + ")]
        for what in ("PTS", "WTS"):
            vals = re.search(r"#define MW_GLL%d_%s \{ (.*?) \}" % (N, what), txt).group(1)
            res["gll%d_%s" % (N, what)] = [float.fromhex(v) if v.startswith(("0x", "-0x")) else float(v) for v in vals.split(", ")]
    return res


def ref_gll(tm, N):
    """get_gll_points / get_gll_weights(SArray<FP,1,N>) of TransformMatrices.h"""
    out = {}
    for what, key in (("get_gll_points", "PTS"), ("get_gll_weights", "WTS")):
        m = re.search(r"void %s\(SArray<FP,1,%d> &rslt\) \{(.*?)\n\s*\}" % (what, N), tm, re.S)
        vals = re.findall(r"rslt\(\d+\)\s*=\s*([-0-9.e]+)", m.group(1))
        out[key] = [float(np.longdouble(v)) for v in vals]
    return out


def main():
    if not os.path.isdir(REF):
        print("reference not present: nothing to check"); return 0
    ref, tm = parse_ref()
    mine = parse_mine()
    bad = 0
    for k in ("coefs7", "coefs9", "tv7", "tv9"):
        same = ref[k] == mine[k]
        print(k, "terms", sum(len(r) for r in ref[k]) if k.startswith("coefs") else len(ref[k]), "identical" if same else "DIFFERENT")
        bad += not same
    for N in (7, 9):
        g = ref_gll(tm, N)
        for key in ("PTS", "WTS"):
            same = g[key] == mine["gll%d_%s" % (N, key)]
            print("gll%d %s" % (N, key), "identical" if same else "DIFFERENT", "" if same else (g[key], mine["gll%d_%s" % (N, key)]))
            bad += not same
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
