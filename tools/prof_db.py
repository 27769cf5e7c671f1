"""Summarise a rocprofv3 results .db (kernel trace): per (kernel, grid) time per step, registers, LDS.
usage: prof_db.py <results.db> <steps-in-run> [top-n] [--by-grid]"""
import re
import sqlite3
import sys
from collections import defaultdict

db, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 40
by_grid = "--by-grid" in sys.argv
c = sqlite3.connect(db)
rows = c.execute("select name, duration, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, lds_size, start, end "
                 "from kernels order by start").fetchall()
agg = defaultdict(lambda: [0, 0.0, None])
for name, dur, gx, wx, vg, ag, lds, s, e in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    short = re.sub(r"^void ", "", short)[:46]
    key = (short, gx // max(wx, 1)) if by_grid else (short,)
    a = agg[key]
    a[0] += 1
    a[1] += dur
    a[2] = (wx, vg, ag, lds)
tot = sum(a[1] for a in agg.values())
span = rows[-1][8] - rows[0][7]
print("kernel ms/step %.3f   launches/step %.0f   (GPU span/step %.3f ms incl. warm-up gaps)" % (
    tot / steps / 1e6, len(rows) / steps, span / steps / 1e6))
for key, (n, d, meta) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    label = key[0] + (" g%d" % key[1] if by_grid else "")
    print("%-54s calls/step %6.1f avg %8.1f us  ms/step %.3f  wg %d vgpr %d+%d lds %d" % (
        label, n / steps, d / n / 1e3, d / steps / 1e6, *meta))

# concurrency over the last 40% of the dispatches (the timed replays): wall span, union of busy intervals, kernel sum
tail = rows[int(len(rows) * 0.6):]
iv = sorted((r[7], r[8]) for r in tail)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s_, e_ in iv[1:]:
    if s_ > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
ksum = sum(r[1] for r in tail)
print("last 40%% of dispatches: span %.2f ms, GPU busy (union) %.2f ms (%.0f%%), kernel-time sum %.2f ms (overlap x%.2f)" % (
    span / 1e6, busy / 1e6, 100.0 * busy / span, ksum / 1e6, ksum / max(busy, 1)))
