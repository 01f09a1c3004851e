#!/usr/bin/env python3
"""Static screen of a gfx950 assembly listing (hipcc -S --cuda-device-only) for uses of LDS-read results before the wait
that releases them.  The streaming kernels issue ds_read_* as inline asm and wait by hand (counted s_waitcnt lgkmcnt);
the compiler believes the result is there right after the asm statement, so a copy / hoisted consumer placed between the
read and the wait captures garbage (seen twice while writing chain3f.hip: a split hoisted out of both arms of a branch,
and register copies behind conditionally executed reads).  Linear scan per function: pending = destination registers of
the LDS reads issued since the last wait; `s_waitcnt lgkmcnt(N)` keeps the N youngest; any instruction that names a
pending register is reported.  Scalar-memory loads share the counter and only make a wait stricter, so they are ignored.
The scan is LINEAR in listing order and knows nothing of control flow: code laid out behind a loop whose exit passes a wait
(an epilogue after a `continue`, the far arm of a branch) can be reported although no path reaches it with the read in
flight -- read the block structure around a report before believing it (chain2f.hip and gemm3s.hip list such reports and
pass their parity tests); zero reports on straight-line kernels (chain3f, the quad kernel, gemm4) is the useful signal.
usage: asm_inflight_check.py file.s [function-substring]"""
import re, sys
src = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
reg_re = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
def regs(text):
    out = set()
    for m in reg_re.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
func, pending, bad = None, [], 0
for ln, line in enumerate(src, 1):
    s = line.strip()
    m = re.match(r"^(_Z\w+):", s)
    if m:
        func, pending = m.group(1), []
        continue
    if func is None or want not in func or not s or s.startswith((";", ".", "//")):
        continue
    op = s.split()[0]
    if op == "s_endpgm":
        func = None
        continue
    if op == "s_waitcnt":
        m = re.search(r"lgkmcnt\((\d+)\)", s)
        if m:
            n = int(m.group(1))
            pending = pending[len(pending) - n:] if n else []
        continue
    if op.startswith("ds_read") or op.startswith("ds_load"):
        ops = s[len(op):].split(",")
        dst = regs(ops[0])
        used = regs(",".join(ops[1:]))
        hit = used & set().union(*[p for p, _ in pending]) if pending else set()
        if hit:
            print(f"{sys.argv[1]}:{ln}: {func[:60]}: address uses in-flight v{sorted(hit)}: {s}"); bad += 1
        pending.append((dst, ln))
        continue
    if op.startswith("ds_") or op.startswith("s_") :
        continue
    if pending:
        allp = set().union(*[p for p, _ in pending])
        hit = regs(s) & allp
        if hit:
            print(f"{sys.argv[1]}:{ln}: {func[:60]}: uses in-flight v{sorted(hit)[:8]}: {s}"); bad += 1
print(f"{bad} suspicious use(s)")
sys.exit(1 if bad else 0)
