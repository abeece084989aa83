#!/usr/bin/env python3
"""What sequencing errors do to the counting pipeline: C2's reads (100 M x 150 bp, k = 25) with a share of the
bases substituted at random -- every error makes up to k new k-mers, so the distinct set grows from 1e8 to several
1e9 and the duplication that the segment tables live on shrinks.  Prints the step time and the path taken.
usage (GPU box): python tools/error_probe.py [error_rate ...]"""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import gossamer_amd as g  # noqa: E402


def main():
    rates = [float(x) for x in sys.argv[1:]] or [0.0, 0.001, 0.01]
    n, L, G, k = 100_000_000, 150, 100_000_000, 25
    dev = torch.device("cuda", 0)
    nbytes = n * (L + 1)
    bases = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    free_b, _ = torch.cuda.mem_get_info(dev)
    ctx = g.Context(k, g.MODE_KMER_SET, device=0, hbm_budget=int((free_b - (8 << 30)) * 0.97))
    lut = torch.tensor([ord(c) for c in "ACGT"], dtype=torch.uint8, device=dev)
    for rate in rates:
        ctx.synth_reads(bases.data_ptr(), n, L, G, seed=1)
        torch.cuda.synchronize()
        if rate > 0:
            gen = torch.Generator(device=dev)
            gen.manual_seed(7)
            step = 1 << 28
            for at in range(0, nbytes, step):
                v = bases[at:at + step]
                hit = torch.rand(v.numel(), device=dev, generator=gen) < rate
                sub = lut[torch.randint(0, 4, (v.numel(),), device=dev, generator=gen)]
                nl = v == 10
                v[hit & ~nl] = sub[hit & ~nl]
                del hit, sub, nl
            torch.cuda.synchronize()
        times = []
        for it in range(2):
            ctx.reset()
            ctx.timing(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.push_device(bases.data_ptr(), nbytes)
            c = ctx.finish()
            ctx.emit_device()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        tim = {kk: round(v["ms"], 1) for kk, v in ctx.timing().as_dict().items()}
        stats = {s: ctx.stat(s) for s in ("fused_chunks", "fused_msd_chunks", "rem32_chunks", "rem32_bits", "rem32_split", "big_table_chunks", "segment_retries", "runs", "seg_merges")}
        print(json.dumps({"error_rate": rate, "ms": round(min(times) * 1e3, 1), "windows": c.windows, "distinct": c.distinct,
                          "G_kmers_per_s": round(c.windows / min(times) / 1e9, 1), "device_ms": tim, "stats": stats}))
        sys.stdout.flush()
    ctx.close()


if __name__ == "__main__":
    main()
