#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) result into the markdown kept under profiles/.

usage: tools/rocpd_summary.py <results.db> [--title T] [--cmd CMD] > profiles/rNN_name.md
Equivalent of `rocprofv3 --kernel-trace --stats` kernel_stats.csv (name, calls, total, avg, min, max, %).
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--title", default="rocprofv3 --kernel-trace --stats")
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(grid_y), max(grid_z), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# {a.title}\n")
    if a.cmd:
        print(f"command: `{a.cmd}`\n")
    print("| kernel | calls | total us | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch | max grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0].split("(")[0]
        print(f"| `{name}` | {r[1]} | {r[2] / 1e3:.1f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | "
              f"{100 * r[2] / tot:.2f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]}x{r[12]}x{r[13]} | {r[14]} |")
    try:
        pm = c.execute("select * from pmc_events limit 1").fetchall()
        if pm:
            cur = c.execute("select name, counter_name, avg(value), count(*) from (select k.name as name, p.* from pmc_events p "
                            "join kernels k on k.dispatch_id = p.dispatch_id) group by name, counter_name")
            print("\n## PMC (average per dispatch)\n\n| kernel | counter | avg | n |\n|---|---|---|---|")
            for r in cur:
                print(f"| `{r[0].split('(')[0]}` | {r[1]} | {r[2]:.1f} | {r[3]} |")
    except sqlite3.Error as e:
        print(f"\n(no PMC table: {e})")


if __name__ == "__main__":
    main()
