#!/usr/bin/env python3
"""Probe: do two contexts (two streams), each counting half of C2, overlap on the GPU?
The extraction is VALU bound, the partition and counting kernels memory / LDS bound; if the
hardware runs them side by side, two half-size jobs in flight finish sooner than one after the
other.  usage (through gpurun): PYTHONPATH=. python tools/overlap_probe.py"""
import threading
import time

import torch

import gossamer_amd as g

n, L, G = 100_000_000, 150, 100_000_000
half = n // 2
buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
free_b, _ = torch.cuda.mem_get_info(0)
ctxs = [g.Context(25, g.MODE_KMER_SET, hbm_budget=int(free_b * 0.45)) for _ in range(2)]
ctxs[0].synth_reads(buf.data_ptr(), n, L, G, seed=1)
torch.cuda.synchronize()
parts = [(buf.data_ptr(), half * (L + 1)), (buf.data_ptr() + half * (L + 1), half * (L + 1))]


def job(i):
    ctxs[i].reset()
    ctxs[i].push_device(*parts[i])
    ctxs[i].finish()


for i in (0, 1):
    job(i)                                    # warm-up: arena mapping
import sys
stagger = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for mode in ("sequential", "concurrent", "sequential", "concurrent"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "sequential":
        job(0); job(1)
    else:
        th = [threading.Thread(target=job, args=(i,)) for i in (0, 1)]
        th[0].start()
        time.sleep(stagger)
        th[1].start()
        [t.join() for t in th]
    torch.cuda.synchronize()
    print(mode, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), [c.counts.distinct for c in ctxs])
