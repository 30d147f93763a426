"""Per-kernel summary of a rocprofv3 results database (the default output of `rocprofv3 --kernel-trace`).

    python tools/prof_db.py gpurun_out/prof/x_results.db [divide_by_steps]
"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 "
         f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc")
    rows = list(con.execute(q))
    total = sum(r[3] for r in rows)
    print("kernel,calls,avg_us,total_us,percent")
    for name, n, avg, tot in rows:
        print(f"\"{name}\",{n},{avg:.2f},{tot:.1f},{100 * tot / total:.2f}")
    print(f"# GPU busy per step: {total / steps / 1000.0:.3f} ms")


if __name__ == "__main__":
    main()
