import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    if "gemm_f32" not in r["Kernel_Name"]: continue
    key = r["Dispatch_Id"]
    by.setdefault(key, {"wgs": int(r["Grid_Size"]) // int(r["Workgroup_Size"])})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in by.items():
    print(k, v)
