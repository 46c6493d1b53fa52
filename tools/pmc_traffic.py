"""Summarise two rocprofv3 --pmc passes (TCC_EA0_RDREQ*, TCC_EA0_WRREQ*) over tools/gemm_one.py into the per-launch
memory-side traffic record bench.py quotes: pmc_traffic.py <read-pass dir> <write-pass dir> <kernel substring> <out.json>
<algorithmic bytes> <description> <command>.  128-byte read requests are counted at 128 B (on gfx950 TCC_EA0_RDREQ
counts requests of 32 / 64 / 128 B; the _32B / _64B / _128B counters split them), write requests at 64 B."""
import collections, csv, datetime, glob, json, statistics, sys

rd_dir, wr_dir, needle, out, alg_bytes, desc, cmd = sys.argv[1:8]


def medians(d):
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


rd, n_rd = medians(rd_dir)
wr, n_wr = medians(wr_dir)
r128, r64, r32 = rd.get("TCC_EA0_RDREQ_128B_sum", 0.0), rd.get("TCC_EA0_RDREQ_64B_sum", 0.0), rd.get("TCC_EA0_RDREQ_32B_sum", 0.0)
total = rd.get("TCC_EA0_RDREQ_sum", 0.0)
other = max(total - r128 - r64 - r32, 0.0)          # requests not in any size bucket: counted at 64 B
read_bytes = 128 * r128 + 64 * (r64 + other) + 32 * r32
w64 = wr.get("TCC_EA0_WRREQ_64B_sum", 0.0)
write_bytes = 64 * w64 + 32 * max(wr.get("TCC_EA0_WRREQ_sum", 0.0) - w64, 0.0)
rec = {"kernel": desc, "command": cmd, "date": datetime.date.today().isoformat(),
       "per_launch": {k.replace("_sum", ""): v for k, v in {**rd, **wr}.items()}, "launches_sampled": {**n_rd, **n_wr},
       "read_bytes": int(read_bytes), "write_bytes": int(write_bytes), "traffic_bytes": int(read_bytes + write_bytes),
       "algorithmic_bytes": int(float(alg_bytes)), "traffic_over_algorithmic": round((read_bytes + write_bytes) / float(alg_bytes), 3),
       "note": "memory-side (fabric) requests of the eight L2s, Infinity-Cache hits included; medians over the sampled launches"}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
