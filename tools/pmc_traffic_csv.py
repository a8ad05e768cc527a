"""Aggregate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in SEPARATE runs, as the MI355X guide prescribes)
into the rows of profiles/r01/pmc_hbm_traffic.csv: kernel,config,counter,avg_per_launch_KB,launches.

    python3 tools/pmc_traffic_csv.py c3 gpurun_out/pmc_fetch gpurun_out/pmc_write >> profiles/r01/pmc_hbm_traffic.csv
"""
import collections
import csv
import glob
import re
import sys

config = sys.argv[1]
for d in sys.argv[2:]:
    total, count = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
            name = re.sub(r"\([^()]*\)$", "", name).strip()  # drop the argument list only
            key = (name, row["Counter_Name"])
            total[key] += float(row["Counter_Value"])
            count[key] += 1
    for (name, counter), value in sorted(total.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        if name.startswith(("jd::", "void jd::")):
            print(f'"{name}",{config},{counter},{value / count[(name, counter)]:.1f},{count[(name, counter)]}')
