#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / avg / min / max duration (like --stats CSV)."""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
scols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
namecol = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
q = f"""select s.{namecol}, count(*), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), sum(d.end - d.start),
        max(d.workgroup_size_x), max(d.grid_size_x)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.{namecol} order by 6 desc"""
rows = cur.execute(q).fetchall()
tot = sum(r[5] for r in rows) or 1
print(f"{'kernel':100s} {'calls':>7s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'pct':>6s}")
for name, n, avg, mn, mx, sm, wg, grid in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    name = re.sub(r"\(anonymous namespace\)::|void |\(GemmParams\)", "", name)[:100]
    print(f"{name:100s} {n:7d} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {sm / 1e6:9.3f} {100 * sm / tot:6.1f}")
