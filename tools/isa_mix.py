#!/usr/bin/env python3
"""Static instruction mix of one kernel in a device assembly file (hipcc --cuda-device-only -S):
   tools/isa_mix.py file.s <substring of the mangled kernel name> [--loop]   (--loop: only the largest loop body)"""
import collections
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l][0]
end = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end") and i > start][0]
body = lines[start:end]
if "--loop" in sys.argv:
    # largest backward branch span
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
    best = (0, 0, 0)
    for i, l in enumerate(body):
        t = l.split()
        if len(t) >= 2 and t[0].startswith("s_cbranch") or (len(t) >= 2 and t[0] == "s_branch"):
            tgt = labels.get(t[1])
            if tgt is not None and tgt < i and i - tgt > best[0]:
                best = (i - tgt, tgt, i)
    body = body[best[1]:best[2] + 1]
cnt = collections.Counter()
for l in body:
    l = l.strip()
    l = l.split(";")[0].strip()
    if not l or l[0] in "." or l.endswith(":"):
        continue
    cnt[l.split()[0]] += 1
groups = collections.Counter()
for op, n in cnt.items():
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64")):
        groups["fp64 arith"] += n
    elif op.startswith("v_cndmask"):
        groups["v_cndmask"] += n
    elif op.startswith(("v_mov", "v_accvgpr")):
        groups["v_mov/accvgpr"] += n
    elif op.startswith("v_cmp"):
        groups["v_cmp"] += n
    elif op.startswith(("v_rcp", "v_rsq", "v_sqrt")):
        groups[op] += n
    elif op.startswith("ds_"):
        groups["lds"] += n
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        groups[op.split("_")[0] + "_" + op.split("_")[1]] += n
    elif op.startswith("s_waitcnt"):
        groups["s_waitcnt"] += n
    elif op.startswith("s_barrier"):
        groups["s_barrier"] += n
    elif op.startswith("s_"):
        groups["salu"] += n
    elif op.startswith("v_"):
        groups["other valu"] += n
    else:
        groups[op] += n
print("instructions", sum(cnt.values()), " valu", sum(n for o, n in cnt.items() if o.startswith("v_")))
for g, n in groups.most_common():
    print(f"  {g:18s} {n}")
print("  other valu:", [(o, n) for o, n in cnt.most_common() if o.startswith("v_") and not o.startswith(
    ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_cndmask", "v_mov", "v_accvgpr", "v_cmp", "v_rcp", "v_rsq", "v_sqrt"))][:14])
