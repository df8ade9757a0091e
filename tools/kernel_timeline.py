"""Prints the last frames' kernel dispatches (start, duration) from a rocprofv3 --kernel-trace CSV (profiling helper)."""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mi355rt" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
t0 = int(rows[-n]["Start_Timestamp"])
for r in rows[-n:]:
    print("%-44s start %9.3f dur %8.3f ms" % (r["Kernel_Name"].replace("mi355rt::", "").replace("void ", "")[:44],
                                             (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
