import os, random, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["GOSS_GPU_FUSED_MIN"] = "0"; os.environ["GOSS_GPU_DEBUG"] = "1"
import gossamer_amd as g
rng = random.Random(8)
prefix = "ACGTTGCAAGCTGAGGCATC"
tails = ["".join(rng.choice("ACGT") for _ in range(5)) for _ in range(1_100_000)]
txt = ("\n".join(prefix + t for t in tails) + "\n").encode()
with g.Context(25, 0, hbm_budget=2 << 30) as ctx:
    ctx.push_host(txt)
    c = ctx.finish()
    print(c.windows, c.distinct, {n: ctx.stat(n) for n in ("fused_chunks", "rem32_chunks", "fused_overflows", "segment_retries")})
