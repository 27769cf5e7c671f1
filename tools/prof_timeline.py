"""Timeline of ONE step out of a rocprofv3 kernel-trace .db: start offset, duration, overlap, name.
usage: prof_timeline.py <results.db> <step-index> [anchor-kernel-substring]"""
import re
import sqlite3
import sys

db, which = sys.argv[1], int(sys.argv[2])
anchor = sys.argv[3] if len(sys.argv) > 3 else "softmax_nll_f"  # (softmax_nll_fwd_kernel | softmax_nll_from_stats_kernel)
c = sqlite3.connect(db)
cols = [d[1] for d in c.execute("pragma table_info(kernels)").fetchall()] or \
       [d[0] for d in c.execute("select * from kernels limit 0").description]
qcol = next((q for q in ("queue_id", "stream_id", "queue") if q in cols), None)
rows = c.execute("select name, start, end, grid_x, workgroup_x, %s from kernels order by start" % (qcol or "0")).fetchall()
qids = {q: i for i, q in enumerate(sorted({r[5] for r in rows}, key=str))}
marks = [i for i, r in enumerate(rows) if anchor in r[0]]
# a step (one graph replay) = n consecutive kernels in start order, n = the anchor spacing. Where a replay begins
# relative to its anchor is found over ALL steady-state steps at once: the offset whose idle gaps add up to the most (a
# graph launch leaves ~8-10 us every time, dependencies inside the graph leave <= 6 us and not at a fixed place).
n = marks[which + 1] - marks[which]
steady = [m for i, m in enumerate(marks[:-1]) if marks[i + 1] - m == n and m >= n]
def gap_at(j):
    return rows[j][1] - max(r[2] for r in rows[max(0, j - 8):j])
best, off = -1, 0
for a in range(n):
    tot = sum(gap_at(m - a) for m in steady)
    if tot > best:
        best, off = tot, a
lo = marks[which] - off
hi = lo + n
seg = rows[lo:hi]
t0 = seg[0][1]
end_max = 0
busy = 0
print("step %d: %d kernels, span %.1f us" % (which, len(seg), (max(r[2] for r in seg) - t0) / 1e3))
for i, (name, s, e, gx, wx, q) in enumerate(seg):
    short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")[:64]
    ov = sum(1 for r in seg if r[1] < e and r[2] > s) - 1
    gap = (s - end_max) / 1e3 if end_max else 0.0
    print("%8.1f %7.1f us  q%d ov %d  gap %5.1f  g%-5d %s" % ((s - t0) / 1e3, (e - s) / 1e3, qids[q], ov, max(gap, 0.0),
                                                               gx // max(wx, 1), short))
    end_max = max(end_max, e)
