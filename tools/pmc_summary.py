"""Mean counter value per dispatch and kernel from the counter_collection CSVs of tools/pmc.sh."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = defaultdict(float)
    names = {}
    for row in csv.DictReader(open(f)):
        k = (row.get("Dispatch_Id"), row["Counter_Name"])
        per_dispatch[k] += float(row["Counter_Value"])
        names[row.get("Dispatch_Id")] = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("hsdev::", "").replace("void ", "")
    for (d, c), v in per_dispatch.items():
        a = acc[(names[d], c)]
        a[0] += v; a[1] += 1
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "mean_per_dispatch", "dispatches"])
for (k, c), (s, n) in sorted(acc.items()):
    w.writerow([k, c, f"{s / n:.1f}", n])
