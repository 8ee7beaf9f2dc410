"""Summarise a tools/profile_bench.sh output directory: per-kernel time table (from rocprofv3 --stats) and per-kernel
mean FETCH_SIZE / WRITE_SIZE per dispatch (from the two --pmc passes).  Prints markdown."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def short(name, n=70):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def kernel_stats(root):
    files = find(os.path.join(root, "trace"), "*kernel_stats.csv")
    if not files:
        print("no kernel_stats.csv found under", root)
        return
    rows = list(csv.DictReader(open(files[0])))
    print("## Kernel time (rocprofv3 --kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in rows[:25]:
        name = r.get("Name") or r.get("KernelName") or "?"
        calls = r.get("Calls") or r.get("Count") or "?"
        tot = float(r.get("TotalDurationNs") or r.get("TotalDuration(ns)") or 0) / 1e6
        avg = float(r.get("AverageNs") or r.get("Average(ns)") or 0) / 1e3
        pct = r.get("Percentage") or r.get("Percentage(%)") or "?"
        print(f"| `{short(name)}` | {calls} | {tot:.3f} | {avg:.1f} | {pct} |")
    print()


def counters(root, sub, counter):
    files = find(os.path.join(root, sub), "*counter_collection.csv")
    if not files:
        print(f"no counter_collection.csv for {counter}")
        return
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if (r.get("Counter_Name") or r.get("CounterName")) != counter:
            continue
        k = r.get("Kernel_Name") or r.get("KernelName") or "?"
        v = float(r.get("Counter_Value") or r.get("CounterValue") or 0)
        agg[k][0] += 1
        agg[k][1] += v
    print(f"## {counter} per dispatch (rocprofv3 --pmc {counter}; raw counter units as reported, KiB on gfx950)\n")
    print("| kernel | dispatches | mean per dispatch | total |")
    print("|---|---|---|---|")
    for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
        print(f"| `{short(k)}` | {n} | {s / n:.4g} | {s:.4g} |")
    print()


if __name__ == "__main__":
    root = sys.argv[1]
    print(f"# rocprofv3 summary of `{os.path.basename(root)}`\n")
    kernel_stats(root)
    counters(root, "fetch", "FETCH_SIZE")
    counters(root, "write", "WRITE_SIZE")
