"""Print (launch index, kernel, IDX_ACTIVE - baseline, BANK_CONFLICT) rows from a rocprofv3 --pmc csv directory."""
import csv, glob, sys, collections
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:24])
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
base = float(sys.argv[2]) if len(sys.argv) > 2 else 512.0
for (d, k), c in sorted(rows.items()):
    print(d, k, "active-base=%d" % (c.get("SQ_LDS_IDX_ACTIVE", 0) - base), "conflict=%d" % c.get("SQ_LDS_BANK_CONFLICT", 0))
