"""A steady-state step of a rocprofv3 kernel trace as a timeline: per kernel name the launches per step, their mean duration and the mean
idle time in front of them (start minus the previous kernel's end).  The last `steps` repetitions of a `period`-launch pattern are used.
    python tools/trace_gaps.py <stats_kernel_trace.csv> [steps = 50] [substring of the kernel that ends a step = mt_clip_adamw]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
names = [r["Kernel_Name"] for r in rows]
# period: distance between the last two launches of the optimizer kernel (one per step)
marker = sys.argv[3] if len(sys.argv) > 3 else "mt_clip_adamw"
marks = [i for i, n in enumerate(names) if marker in n]
period = marks[-1] - marks[-2]
seg = rows[marks[-1 - steps] + 1: marks[-1] + 1]
dur = collections.defaultdict(float); gap = collections.defaultdict(float); cnt = collections.Counter()
prev_end = None
for r in seg:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n] += (e - s) / 1e3; cnt[n] += 1
    if prev_end is not None:
        gap[n] += max(0, s - prev_end) / 1e3
    prev_end = e
wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3 / steps
print(f"launches per step {period}, wall per step {wall:.1f} us, kernel time {sum(dur.values()) / steps:.1f} us, idle {sum(gap.values()) / steps:.1f} us")
for n in sorted(dur, key=lambda k: -(dur[k] + gap[k])):
    print(f"{cnt[n] / steps:6.1f} x {dur[n] / cnt[n]:7.1f} us  + idle in front {gap[n] / cnt[n]:5.1f} us  = {(dur[n] + gap[n]) / steps:7.1f} us/step  {n}")
