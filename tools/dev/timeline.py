"""Dev tool: condense a rocprofv3 --kernel-trace csv into a per-step timeline (large kernels, runs of tiny ones, idle gaps).
usage: python tools/dev/timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Stream_Id", "")))
rows.sort()
# step boundaries: k_weff launches
starts = [i for i, r in enumerate(rows) if "k_weff" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = starts[-k - 1], starts[-k]
step = rows[a:b]
t0 = step[0][0]
print("step span %.3f ms, %d kernels" % ((rows[b][0] - t0) / 1e6, len(step)))
busy_end = t0
tiny_n, tiny_t, tiny_start = 0, 0, None
idle = 0
def flush():
    global tiny_n, tiny_t, tiny_start
    if tiny_n:
        print("  %8.3f  [%d tiny kernels, %.0f us busy]" % ((tiny_start - t0) / 1e6, tiny_n, tiny_t / 1e3))
    tiny_n, tiny_t, tiny_start = 0, 0, None
for s, e, n, st in step:
    gap = s - busy_end
    if gap > 0:
        idle += gap
    if gap > 15000:
        flush()
        print("  %8.3f  ---- idle %.0f us" % ((busy_end - t0) / 1e6, gap / 1e3))
    if e - s >= 30000:
        flush()
        print("  %8.3f  %-60s %7.0f us  s%s" % ((s - t0) / 1e6, n, (e - s) / 1e3, st))
    else:
        if tiny_n == 0:
            tiny_start = s
        tiny_n += 1; tiny_t += e - s
    busy_end = max(busy_end, e)
flush()
print("total idle in step: %.3f ms" % (idle / 1e6))
