"""Per-kernel LDS bank-conflict ratio from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE pass (dev).
usage: python tools/lds_conflicts.py <dir with *_counter_collection.csv>"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0][-70:]
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
            n[name] += 1
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0.0)):
    act, conf = c.get("SQ_LDS_IDX_ACTIVE", 0.0), c.get("SQ_LDS_BANK_CONFLICT", 0.0)
    if act > 0:
        print(f"{name:70s} launches {n[name]:4d}  LDS active {act:14.0f}  conflict cycles {conf:14.0f}  = {100 * conf / act:5.1f} %")
