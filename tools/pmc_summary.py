"""Summarise rocprofv3 --pmc passes: mean counter value per kernel and counter over the dispatches of every
*counter_collection.csv under the given directories.  Usage: python tools/pmc_summary.py <dir> [<dir> ...] [--json out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(dirs):
    acc = defaultdict(list)
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    return acc


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if out_json in args:
        args.remove(out_json)
    acc = collect(args)
    table = {}
    for (kernel, counter), values in sorted(acc.items()):
        if "trt::" not in kernel:
            continue
        short = kernel.replace("void ", "").split("(")[0]
        table.setdefault(short, {})[counter] = {"n": len(values), "mean": sum(values) / len(values)}
        print(f"{short:45s} {counter:24s} n={len(values):<3d} mean={sum(values) / len(values):.4g}")
    if out_json:
        with open(out_json, "w") as fh:
            json.dump(table, fh, indent=1)


if __name__ == "__main__":
    main()
