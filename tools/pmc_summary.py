#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_probe.py into
profiles/pmc_summary.json: per kernel and grid size, HBM bytes per launch.

Counter units and the gfx950 correction follow MI355X_MICROARCH.md section HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read (16 B per lane, global_load and LDS-DMA alike), so reads are doubled; WRITE_SIZE is
exact for 16-B-per-lane stores.  The first launch of each shape is dropped (cold instruction/L2).

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> out.json
"""
import collections
import csv
import json
import re
import sys


REPS = 5   # tools/pmc_probe.py launches every shape REPS times in a row


def load(path, counter):
    """(kernel, grid, run index) -> [(value, duration ns)]: consecutive launches of one kernel and
    grid are cut into runs of REPS, one run per probed shape (shapes can share a grid)."""
    rows = collections.defaultdict(list)
    seen = collections.Counter()
    data = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    data.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in data:
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = name.replace("mixdq::(anonymous namespace)::", "")
        name = re.sub(r"\(.*$", "", name)
        base = (name, int(r["Grid_Size"]))
        run = seen[base] // REPS
        seen[base] += 1
        rows[base + (run,)].append(
            (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return rows


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for key in sorted(fetch):
        name, grid, run = key
        if "rocclr" in name:
            continue
        f = fetch[key][1:] or fetch[key]
        w = write.get(key, [(0.0, 0)])
        w = w[1:] or w
        fkb = sum(v for v, _ in f) / len(f)
        wkb = sum(v for v, _ in w) / len(w)
        us = sum(d for _, d in f) / len(f) / 1e3
        out[f"{name} grid={grid} shape#{run}"] = dict(
            launches=len(f), avg_us_under_pmc=round(us, 2),
            fetch_size_kib_raw=round(fkb, 1), write_size_kib=round(wkb, 1),
            hbm_read_bytes_corrected=int(2 * fkb * 1024), hbm_write_bytes=int(wkb * 1024),
            hbm_bytes_per_launch=int((2 * fkb + wkb) * 1024))
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        print(f"{v['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch (rd {v['hbm_read_bytes_corrected'] / 1e6:7.2f} "
              f"wr {v['hbm_write_bytes'] / 1e6:7.2f})  {v['avg_us_under_pmc']:7.1f} us  {k}")


if __name__ == "__main__":
    main()
