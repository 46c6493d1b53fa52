"""Print per-kernel medians of the counters in rocprofv3 counter_collection CSVs."""
import csv, collections, sys, statistics
for d in sys.argv[1:]:
    rows = list(csv.DictReader(open(f"{d}/p_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        k = r["Kernel_Name"]
        k = k[k.find("::") + 2:][:40] if "::" in k else k[:40]
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        if any(t in k for t in ("gemm", "attn", "remask", "ce_")): print(f"{d[-6:]} {k:40s} {c:36s} {statistics.median(v):.4e}  n={len(v)}")
