"""Per-kernel averages of rocprofv3 --pmc CSV output:
   python profiles/summarize_pmc.py gpurun_out/pmc_x/pmc_counter_collection.csv [more.csv ...]"""
import collections
import csv
import sys


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name") or row.get("Kernel Name")
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, counters in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        short = k if len(k) < 90 else k[:87] + "..."
        print(short)
        for name, vals in sorted(counters.items()):
            print(f"    {name:<34} launches {len(vals):>4}   mean {sum(vals) / len(vals):>18.1f}")


if __name__ == "__main__":
    main(sys.argv[1:])
