import random, os
os.environ["GOSS_GPU_DEBUG"]="1"
import gossamer_amd as g
rng = random.Random(5)
seq = "".join(rng.choice("ACGT") for _ in range(1_000_000))
reads = ("\n".join([seq] * 40) + "\n").encode()
with g.Context(27, g.MODE_KMER_SET, hbm_budget=8 << 30) as ctx:
    ctx.push_host(reads)
    c = ctx.finish()
    print(c.windows, c.distinct, ctx.stat("fused_chunks"), ctx.stat("runs"))
