"""Summarises a tools/profile_bench.sh output directory (rocprofv3 writes rocpd SQLite databases: <dir>/{trace,fetch,write}/
bench_results.db) as markdown: per-kernel time from the kernel trace, HBM-side bytes per dispatch from the two PMC passes.
FETCH_SIZE is doubled (MI355X_MICROARCH.md §HBM: on gfx950 it reports half of a wide coalesced read); both counters are in KiB.
usage: python tools/summarize_prof.py gpurun_out/prof_<tag> [--json profiles/traffic.json]"""
import glob
import json
import os
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name


def db(path):
    hits = glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True)
    return sqlite3.connect(hits[0]).cursor() if hits else None


def main():
    root = sys.argv[1]
    print(f"# rocprofv3 summary of `{os.path.basename(root.rstrip('/'))}`\n")
    c = db(os.path.join(root, "trace"))
    if c is None:
        print("no kernel trace database")
    else:
        rows = list(c.execute("select name, grid_x, workgroup_x, count(*), avg(duration), sum(duration) from kernels group by name, grid_x order by 6 desc"))
        total = sum(r[5] for r in rows)
        print("## Kernel time\n\n| kernel | grid (threads) x workgroup | launches | avg us | total ms | % |\n|---|---|---|---|---|---|")
        for name, gx, wx, n, avg, tot in rows:
            if tot / total < 0.001:
                continue
            print(f"| `{short(name)}` | {gx} x {wx} | {n} | {avg / 1e3:.1f} | {tot / 1e6:.2f} | {100 * tot / total:.1f} |")
        print(f"\nTotal kernel time {total / 1e6:.1f} ms.\n")
    traffic = {}
    for key, counter, scale in (("read", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
        c = db(os.path.join(root, "fetch" if key == "read" else "write"))
        if c is None:
            print(f"no {counter} database")
            continue
        q = ("select kernel_name, grid_size_x, count(*), avg(value) from counters_collection where counter_name = ? "
             "group by kernel_name, grid_size_x")
        for name, gx, n, val in c.execute(q, (counter,)):
            traffic.setdefault((short(name), gx), {})[key] = val * 1024.0 * scale
            traffic[(short(name), gx)]["n_" + key] = n
    if traffic:
        print("## HBM-side traffic per dispatch (PMC; FETCH_SIZE x2, WRITE_SIZE raw; KiB -> GB)\n\n| kernel | grid | read GB | write GB |\n|---|---|---|---|")
        for (name, gx), t in sorted(traffic.items(), key=lambda kv: -(kv[1].get("read", 0) + kv[1].get("write", 0))):
            if t.get("read", 0) + t.get("write", 0) < 5e7:
                continue
            print(f"| `{name}` | {gx} | {t.get('read', 0) / 1e9:.2f} | {t.get('write', 0) / 1e9:.2f} |")
    if "--json" in sys.argv:
        out = sys.argv[sys.argv.index("--json") + 1]
        fused = [(k, t) for k, t in traffic.items() if "fused" in k[0]]
        if fused:
            (name, gx), t = max(fused, key=lambda kt: kt[1].get("read", 0) + kt[1].get("write", 0))
            rec = {"fused_samples": {"bytes_per_launch": t.get("read", 0) + t.get("write", 0), "read_bytes": t.get("read", 0),
                                     "write_bytes": t.get("write", 0), "kernel": name, "grid_threads": gx,
                                     "source": f"profiles/{os.path.basename(root.rstrip('/')).replace('prof_', '')}_*_rocprof.md "
                                               "(rocprofv3 --pmc FETCH_SIZE x2, --pmc WRITE_SIZE, per dispatch)"}}
            # one frame = every kernel that runs once (or more) per bench step; the passes ran `frames` steps (warm-up + timed)
            frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 2
            rd = sum(t.get("read", 0) * (t.get("n_read", 0) // frames) for t in traffic.values())
            wr = sum(t.get("write", 0) * (t.get("n_write", 0) // frames) for t in traffic.values())
            rec["frame"] = {"read_bytes": rd, "write_bytes": wr, "bytes_per_frame": rd + wr,
                            "source": rec["fused_samples"]["source"].replace("per dispatch", "every dispatch of one 65536-ray frame")}
            try:
                with open(out) as f:
                    old = json.load(f)
                for k in ("bound", "ta_busy", "mfma_busy", "l1_bytes_per_launch", "l1_note"):
                    if k in old.get("fused_samples", {}):
                        rec["fused_samples"][k] = old["fused_samples"][k]
            except (OSError, ValueError):
                pass
            with open(out, "w") as f:
                json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
