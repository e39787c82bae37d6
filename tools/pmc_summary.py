#!/usr/bin/env python
"""Average rocprofv3 --pmc counter values per launch and kernel from one or more `*_counter_collection.csv` files
(separate passes, one counter group each) -> JSON {"kernels": {name: {counter: mean value per launch}}}.
usage: pmc_summary.py out.json pass1_counter_collection.csv [pass2_counter_collection.csv ...]"""
import csv
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = {}
    for path in files:
        per = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if "skp_" not in k:
                    continue
                if "_conv_" in k:                      # one kernel, several launch shapes: key by grid size too
                    k += "@grid" + row["Grid_Size"]
                key = (k, row["Counter_Name"])
                d = per.setdefault(key, {})
                disp = row["Dispatch_Id"]
                d[disp] = d.get(disp, 0.0) + float(row["Counter_Value"])        # sum over XCDs / instances
        for (k, c), d in per.items():
            acc.setdefault(k, {})[c] = sum(d.values()) / len(d)
            acc[k].setdefault("_launches", len(d))
    json.dump({"note": "mean counter value per launch (summed over XCD instances); FETCH_SIZE/WRITE_SIZE in KiB; "
                       "HBM read bytes = 2 x FETCH_SIZE x 1024 on gfx950 (MI355X_MICROARCH.md, HBM section)",
               "kernels": acc}, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(acc.items()):
        print(k, {c: round(x, 1) for c, x in v.items()})


if __name__ == "__main__":
    main()
