"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM-side bytes per kernel family.

usage: python tools/pmc_traffic.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <out.json> [note]

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: the counters are in
KB; on gfx950 FETCH_SIZE reports half of the bytes of 16-byte-per-lane streaming reads (global_load and LDS-DMA alike), so
it is doubled; WRITE_SIZE is exact for 16-byte stores.  Infinity-Cache hits are counted, not excluded.
"""
import csv
import json
import sys
from collections import defaultdict

FAMILIES = (("conv_igemm", "conv_igemm_kernel"), ("conv3x3_patch", "conv3x3_patch"), ("attn_flash", "attn_flash"),
            ("cross_attn", "cross_attn_kernel"), ("absorb_q", "absorb_q_kernel"), ("absorb_v", "absorb_v_kernel"),
            ("decode_attn", "decode_attn"), ("layernorm", "layernorm"), ("beam_candidates", "beam_candidates"))


def family(name: str) -> str:
    for fam, pat in FAMILIES:
        if pat in name:
            return fam
    return "other"


def fold(path):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            fam = family(row["Kernel_Name"])
            tot[fam] += float(row["Counter_Value"])
            cnt[fam] += 1
    return tot, cnt


def main():
    fetch, n1 = fold(sys.argv[1])
    write, n2 = fold(sys.argv[2])
    config = json.loads(sys.argv[5]) if len(sys.argv) > 5 else None
    out = {"config": config, "units": "bytes", "fetch_correction": "FETCH_SIZE x 2 (gfx950, 16-byte-per-lane reads)", "note": sys.argv[4] if len(sys.argv) > 4 else "",
           "kernels": {}}
    for fam in sorted(set(fetch) | set(write)):
        launches = max(n1.get(fam, 0), n2.get(fam, 0))
        rd, wr = 2.0 * fetch.get(fam, 0.0) * 1024.0, write.get(fam, 0.0) * 1024.0
        out["kernels"][fam] = {"launches": launches, "read_bytes": rd, "write_bytes": wr,
                               "bytes_per_launch": (rd + wr) / max(launches, 1)}
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    for fam, v in out["kernels"].items():
        print(f"{fam:16s} {v['launches']:6d} launches  read {v['read_bytes'] / 1e9:8.2f} GB  write {v['write_bytes'] / 1e9:8.2f} GB  "
              f"{v['bytes_per_launch'] / 1e6:8.2f} MB/launch")


if __name__ == "__main__":
    main()
