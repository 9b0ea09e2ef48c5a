import csv, glob, sys, collections
d = sys.argv[1]
plan = [l for l in open(d + "/plan.txt").read().strip().split("\n") if not l.startswith("total")]
f = glob.glob(d + "/*/*_counter_collection.csv")[0]
rows = list(csv.DictReader(open(f)))
# group by dispatch id
disp = collections.OrderedDict()
for r in rows:
    k = int(r["Dispatch_Id"])
    disp.setdefault(k, {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
ds = [v for k, v in sorted(disp.items()) if any(t in v["name"] for t in ("obb::k_conv", "k_dwconv3", "k_maxpool5", "k_upsample2", "k_attention"))]
ds = ds[-len(plan):]
names = [c for c in ds[0] if c != "name"]
print("counters:", names)
for o, v in zip(plan, ds):
    if any(k in o for k in sys.argv[2:]) or len(sys.argv) == 2:
        print(o[:70].ljust(70), " ".join("%s=%.3g" % (c, v.get(c, 0)) for c in names))
