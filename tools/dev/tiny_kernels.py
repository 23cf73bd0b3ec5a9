"""Dev tool: list the small kernels of one training step from a rocprofv3 --kernel-trace csv, in launch order with gaps."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_weff" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = starts[-k - 1], starts[-k]
t0 = rows[a][0]
import re
for s, e, n in rows[a:b]:
    short = re.sub(r"\(.*", "", n)
    short = re.sub(r"void |at::native::|<.*", "", short)[:70]
    detail = ""
    m = re.search(r"(elementwise_kernel|vectorized_elementwise_kernel|reduce_kernel|CatArrayBatchedCopy|index|fill|copy|Functor\w*|\w+_kernel_cuda)", n)
    f2 = re.findall(r"(\w+Functor\w*|\w+_cuda\w*|FillFunctor|\w+Op\b)", n)
    print("%8.3f %7.1f us  %-50s %s" % ((s - t0) / 1e6, (e - s) / 1e3, short, ",".join(dict.fromkeys(f2))[:90]))
