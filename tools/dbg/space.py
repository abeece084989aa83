import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["GOSS_GPU_FUSED_MIN"] = "0"
os.environ["GOSS_GPU_DEBUG"] = "1"
import gossamer_amd as g
reads = g.synth_reads_host(300000, 150, 1500000, seed=23)
with g.Context(25, g.MODE_KMER_SET, hbm_budget=400 << 20) as ctx:
    ctx.push_host(reads)
    c = ctx.finish()
    print({n: ctx.stat(n) for n in ("rep_chunks", "canon_chunks", "fused_chunks")})
