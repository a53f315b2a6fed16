"""Turn a rocprofv3 (ROCm 7.2) rocpd SQLite result into the per-kernel stats table committed under
profiles/:  python profiles/summarize_rocpd.py gpurun_out/prof_x/x_results.db > profiles/x.stats.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':<70} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for name, calls, tot, avg, mn, mx in rows:
        short = name if len(name) <= 70 else name[:67] + "..."
        print(f"{short:<70} {calls:>7} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100.0 * tot / total:>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
