"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / avg / min / max duration and,
if present, PMC counter sums per kernel.  Usage: python tools/rocpd_summary.py results.db"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    suf = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
    q = f"""select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3,
            max(d.end-d.start)/1e3, sum(d.end-d.start)/1e3, max(s.arch_vgpr_count), max(s.accum_vgpr_count),
            max(s.sgpr_count), max(d.group_segment_size)
            from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id=s.id
            group by s.kernel_name order by 6 desc limit 15"""
    print(f"{'kernel':64s} {'calls':>5s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_us':>11s} vgpr* agpr sgpr lds")
    print("# vgpr* = rocpd's arch_vgpr_count; on gfx950 it reads HALF the wave64 allocation in the code-object metadata "
          "(.vgpr_count: e.g. 84 here = 168 there)")
    for r in c.execute(q):
        print(f"{r[0][:64]:64s} {r[1]:5d} {r[2]:10.1f} {r[3]:10.1f} {r[4]:10.1f} {r[5]:11.1f} {r[6]} {r[7]} {r[8]} {r[9]}")
    try:
        q = f"""select s.kernel_name, p.name, count(*), avg(e.value)
                from rocpd_pmc_event{suf} e join rocpd_info_pmc{suf} p on e.pmc_id=p.id
                join rocpd_kernel_dispatch{suf} d on e.event_id=d.event_id
                join rocpd_info_kernel_symbol{suf} s on d.kernel_id=s.id
                group by s.kernel_name, p.name order by s.kernel_name, p.name"""
        rows = list(c.execute(q))
        if rows:
            print("\nPMC counters (average per dispatch, summed over the chip):")
            for r in rows:
                if "at::native" in r[0] or "rocclr" in r[0]:
                    continue
                print(f"  {r[0][:56]:56s} {r[1]:28s} n={r[2]:4d} avg={r[3]:.4g}")
    except sqlite3.Error as exc:
        print("no PMC data:", exc)


if __name__ == "__main__":
    main(sys.argv[1])
