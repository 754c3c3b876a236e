"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel table:
    python tools/rocpd_stats.py gpurun_out/prof/run_results.db [--top 40] [--md out.md]
"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--md", default="")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    rows = db.execute(
        "select s.display_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.display_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    span = db.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, t, mn, mx in rows[: a.top]:
        lines.append(f"| {short(n)} | {c} | {t / 1e6:.3f} | {t / c / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * t / total:.1f} |")
    lines.append(f"\nkernel time total {total / 1e6:.2f} ms over {len(rows)} kernels, {sum(r[1] for r in rows)} dispatches; "
                 f"first-to-last dispatch span {(span[1] - span[0]) / 1e6:.2f} ms")
    out = "\n".join(lines)
    print(out)
    if a.md:
        with open(a.md, "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
