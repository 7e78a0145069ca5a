"""Per-launch breakdown of one detector page from a rocprofv3 --kernel-trace CSV (pages workload).
usage: python profiles/per_page.py <kernel_trace.csv>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rocclr" not in r["Kernel_Name"]]
idx = [i for i, r in enumerate(rows) if "resize_linear" in r["Kernel_Name"]]
page = rows[idx[-2]:idx[-1]]
tot = 0.0
for r in page:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    nm = r["Kernel_Name"].replace("_ZN12_GLOBAL__N_1", "").replace("(anonymous namespace)::", "")[:58]
    print("%-60s %8.1f us  grid=%9s" % (nm, d, r["Grid_Size_X"]))
wall = (int(page[-1]["End_Timestamp"]) - int(page[0]["Start_Timestamp"])) / 1e3
print("sum of kernels %.1f us; wall first->last %.1f us" % (tot, wall))
