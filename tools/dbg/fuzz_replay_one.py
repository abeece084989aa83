"""One context, the dumped reads pushed in `pieces` host pushes; variants of the environment given as ENV=V,ENV=V
arguments (one run each); keys and counts against the oracle."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gossamer_amd as g
import oracle_lib as o
d, case, budget = sys.argv[1], sys.argv[2], int(sys.argv[3])
meta = json.load(open(os.path.join(d, "case%s.json" % case)))
reads = open(os.path.join(d, "case%s.reads" % case), "rb").read()
graph, k = meta["graph"], meta["k"]
L = k + 1 if graph else k
ek, ec, _, nwin = o.count([(o.LINE, "r", reads)], L, 1 if graph else 0)
want = dict(zip(ek, ec))
for spec in sys.argv[4:] or [""]:
    env = dict(meta["env"])
    path = 0
    for item in [x for x in spec.split(",") if x]:
        a, b = item.split("=", 1)
        if a == "PATH_": path = int(b)
        elif b == "-": env.pop(a, None)
        else: env[a] = b
    old = {n: os.environ.get(n) for n in env}
    for n in meta["env"]:
        if n not in env: old[n] = os.environ.get(n); os.environ.pop(n, None)
    os.environ.update(env)
    try:
        with g.Context(k, g.MODE_GRAPH if graph else g.MODE_KMER_SET, hbm_budget=budget) as ctx:
            ctx.set_path(path)
            ctx.push_host(reads)
            c = ctx.finish()
            gk, gc = ctx.result()
            st = {s: ctx.stat(s) for s in ("fused_chunks", "runs", "seg_merges", "hash_merges", "segment_retries", "table96_chunks", "wide_table_chunks", "big_table_chunks")}
    except g.GossGpuError as e:
        print(spec, "refused", e); continue
    finally:
        for n, v in old.items():
            if v is None: os.environ.pop(n, None)
            else: os.environ[n] = v
    gc = [int(x) for x in gc]
    diffs = [(x, a, want.get(x)) for x, a in zip(gk, gc) if want.get(x) != a]
    print("[%s] windows %d/%d distinct %d/%d sum %d/%d diffs %d %s %s" % (spec, c.windows, nwin, len(gk), len(ek), sum(gc), sum(ec), len(diffs),
          [(o.kmer_string(x, L)[:10] + ".." + o.kmer_string(x, L)[-6:], a, b) for x, a, b in diffs[:3]], st), flush=True)
