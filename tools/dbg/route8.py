"""debug: one rank routing for P parts (GOSS_DIST_ROUTE_PARTS) against the oracle, two-word keys"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
import gossamer_amd as g
from gossamer_amd import dist as gd
import oracle_lib as o
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)
reads = g.synth_reads_host(20000, 150, 100000, seed=13)
buf = torch.frombuffer(bytearray(reads), dtype=torch.uint8).cuda()
for k, graph in ((55, True), (45, False), (40, True), (31, True), (63, False)):
    exp, nwin = (o.build_graph if graph else o.build_kmer_set)([(o.LINE, "reads", reads)], k, out="ob")
    exp = {n[2:]: d for n, d in exp.items()}
    for parts in (1, 2, 3, 8):
        os.environ["GOSS_DIST_ROUTE_PARTS"] = str(parts)
        with g.Context(k, g.MODE_GRAPH if graph else g.MODE_KMER_SET, hbm_budget=1 << 30) as ctx:
            r = gd.count_distributed(ctx, buf.data_ptr(), buf.numel(), 2 * (k + 1 if graph else k), torch.device("cuda", 0), exchange="records")
            got = gd.assemble_files([ctx.files()])
        bad = [n for n in exp if got.get(n) != exp[n]]
        print(k, graph, parts, "windows", r["windows"], nwin, "M", r["M"], "sizes", gd._ROUTE_SIZES.get((buf.data_ptr(), buf.numel(), parts, k, 1 if graph else 0)), "differ:", bad, flush=True)
