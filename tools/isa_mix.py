"""Instruction mix of one kernel in a hipcc -S listing, per loop nest level (the innermost loop with MFMAs = one tile).
    python tools/isa_mix.py file.s <kernel name substring>"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
sub = sys.argv[2]
start = next(i for i, l in enumerate(txt) if re.match(r"^_Z\S*" + re.escape(sub) + r"\S*:", l))
end = next(i for i in range(start + 1, len(txt)) if txt[i].startswith(".Lfunc_end"))
body = txt[start:end]
# basic blocks and loops: find back edges (branch to an earlier label)
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch\S*\s+(\.LBB\S+)|\s+s_branch\s+(\.LBB\S+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.append((labels[t], i))


def mix(lo, hi):
    c = collections.Counter()
    for l in body[lo:hi + 1]:
        l = l.strip()
        if not l or l[0] in ".;/" or l.endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith(("ds_read", "ds_load", "ds_write", "ds_store")):
            c[op] += 1
        elif op.startswith("scratch_"):
            c[op] += 1
        elif op.startswith("s_barrier"):
            c["s_barrier"] += 1
        elif op.startswith("s_waitcnt"):
            c["s_waitcnt"] += 1
        elif op.startswith("v_accvgpr"):
            c[op] += 1
        elif op.startswith(("v_exp", "v_log", "v_rcp", "v_sqrt", "v_rsq")):
            c["transcendental"] += 1
        elif op.startswith("v_cvt_pk_bf16"):
            c["v_cvt_pk_bf16"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c[op] += 1
        elif op.startswith("v_"):
            c["valu_other"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


print("kernel lines", len(body), "loops", [(a, b, b - a) for a, b in loops])
for lo, hi in sorted(loops, key=lambda ab: ab[0] - ab[1])[:3]:
    c = mix(lo, hi)
    print(f"loop [{lo},{hi}] total {sum(c.values())}")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
        print("   ", k, v)
