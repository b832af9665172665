"""Kernel timeline of the last World::Update in a rocprofv3 --kernel-trace CSV: busy time, idle gaps, per-kernel list.
usage: timeline.py <kernel_trace.csv> [marker-kernel-substring=k_integrate_velocity]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_integrate_velocity"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
busy = 0; prev_end = t0; gaps = []
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    if s - prev_end > 3000: gaps.append(((s - prev_end) / 1e3, (prev_end - t0) / 1e3, r["Kernel_Name"][:50]))
    prev_end = max(prev_end, e)
span = (prev_end - t0) / 1e3
print("step span %.1f us, kernels %d, busy %.1f us, idle %.1f us" % (span, len(step), busy / 1e3, span - busy / 1e3))
# the step proper ends with k_integrate_position (the settle's mailbox post sits in front of it); what follows in the cut is the
# measuring script reading statistics (round trips of its own)
names = [r["Kernel_Name"] for r in step]
ip = [i for i, n in enumerate(names) if "k_integrate_position" in n]
if ip:
    last = ip[-1]
    print("step proper (up to IntegratePosition): %.1f us, %d kernels" % ((int(step[last]["End_Timestamp"]) - t0) / 1e3, last + 1))
print("gaps > 3 us (gap us, at us, next kernel):")
for g in sorted(gaps, reverse=True)[:25]: print("  %.1f  @%.1f  %s" % g)
if "-v" in sys.argv:
    for r in step: print("%9.1f %8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]))
