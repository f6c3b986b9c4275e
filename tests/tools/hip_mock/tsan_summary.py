#!/usr/bin/env python3
"""Condense ThreadSanitizer output: one block per report -- its kind and the first frames of each stack that lie in this repository."""
import re
import sys

reports, cur = [], None
for line in sys.stdin:
    if line.startswith("WARNING: ThreadSanitizer"):
        cur = [line.strip()]
        reports.append(cur)
    elif cur is not None:
        if line.startswith("=================="):
            cur = None
            continue
        m = re.match(r"\s+(Write|Read|Previous|Atomic|Location|Mutex|Thread|As if|Cycle)", line)
        if m:
            cur.append("  " + line.strip()[:160])
        m = re.match(r"\s+#(\d+) (.*?) (/root/repo/\S+|\S+:\d+)", line)
        if m and "/root/repo/" in line and int(m.group(1)) < 12:
            fn = re.sub(r"\(anonymous namespace\)::", "", m.group(2))[:110]
            cur.append(f"      #{m.group(1)} {fn} {m.group(3).replace('/root/repo/', '')}")
    elif not line.startswith("=================="):
        sys.stdout.write(line)
seen = {}
for r in reports:
    key = "\n".join(x for x in r if x.startswith("      #"))[:600]
    seen.setdefault(key, []).append(r)
print(f"--- {len(reports)} ThreadSanitizer reports, {len(seen)} distinct ---")
for key, rs in seen.items():
    print(f"[x{len(rs)}]")
    print("\n".join(rs[0][:40]))
