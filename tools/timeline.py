"""The last bench step of a rocprofv3 kernel trace: kernels in start order, with durations and the idle gaps between
them (a gap = start - latest end so far, when positive).  usage: timeline.py kernel_trace.csv"""
import csv, sys, re

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# the last step = everything after the last synth/first long idle stretch: take the kernels after the last gap > 20 ms
cut, end = 0, ks[0][1]
for i, (s, e, n) in enumerate(ks):
    if s - end > 20e6:
        cut = i
    end = max(end, e)
ks = ks[cut:]
t0, end = ks[0][0], ks[0][0]
busy = gaps = 0
short = lambda n: re.sub(r"\(.*", "", n.replace("goss::", "").replace("void ", ""))[:70]
for s, e, n in ks:
    g = s - end
    if g > 0:
        gaps += g
    if g > 100e3:
        print("%9.3f   -- idle %.3f ms --" % ((end - t0) / 1e6, g / 1e6))
    if e - s > 200e3 or g > 100e3:
        print("%9.3f %8.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, short(n)))
    busy += max(0, e - max(s, end))
    end = max(end, e)
print("step: %.3f ms from first kernel to last; kernels %.3f ms, idle %.3f ms; %d kernels" % ((end - t0) / 1e6, busy / 1e6, gaps / 1e6, len(ks)))
