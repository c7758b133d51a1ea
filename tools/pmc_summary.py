#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, mean counter value per dispatch."""
import csv, sys, collections, re
def main(path, filt=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if filt and filt not in k: continue
            m = re.search(r"(k_\w+(<[^>]*>)?)", k); k = m.group(1) if m else k[:60]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(k, "dispatches=%d" % max(len(v) for v in cs.values()))
        for c, v in sorted(cs.items()):
            print("    %-28s %16.1f" % (c, sum(v) / len(v)))
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
