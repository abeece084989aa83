"""Replay a case tests/fuzz_parity.py dumped (FUZZ_DUMP_DIR): the same reads and settings through one context, files
against the oracle, with variations of the settings to find what the difference depends on.
usage: python tools/dbg/fuzz_replay.py <dir> <case> [ENV=VALUE ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gossamer_amd as g
import oracle_lib as o
d, case = sys.argv[1], sys.argv[2]
meta = json.load(open(os.path.join(d, "case%s.json" % case)))
reads = open(os.path.join(d, "case%s.reads" % case), "rb").read()
graph, k = meta["graph"], meta["k"]
exp, nwin = (o.build_graph if graph else o.build_kmer_set)([(o.LINE, "r", reads)], k, out="o")
exp = {n[1:]: b for n, b in exp.items()}
ek, ec, _, _ = o.count([(o.LINE, "r", reads)], k + 1 if graph else k, 1 if graph else 0)
variants = [dict(meta["env"])]
for extra in sys.argv[3:]:
    a, b = extra.split("=", 1)
    v = dict(meta["env"]);
    if b == "-":
        v.pop(a, None)
    else:
        v[a] = b
    variants.append(v)
for env in variants:
    old = {n: os.environ.get(n) for n in env}
    os.environ.update(env)
    try:
        with g.Context(k, g.MODE_GRAPH if graph else g.MODE_KMER_SET, hbm_budget=meta["budget"]) as ctx:
            ctx.push_host(reads)
            c = ctx.finish()
            gk, gc = ctx.result()
            got = ctx.emit()
            st = {s: ctx.stat(s) for s in ("fused_chunks", "rep_chunks", "canon_chunks", "runs", "rem32_chunks", "valid_sized_chunks", "valid_resizes", "segment_retries")}
    finally:
        for n, v in old.items():
            if v is None: os.environ.pop(n, None)
            else: os.environ[n] = v
    gc = [int(x) for x in gc]
    bad = [n for n in exp if got.get(n) != exp[n]]
    print("env", env, "windows", c.windows, nwin, "distinct", len(gk), len(ek), "keys equal", gk == ek, "counts equal", gc == ec, "files differ", bad, st, flush=True)
    if gk != ek:
        sg, se = set(gk), set(ek)
        missing, extra_k = sorted(se - sg)[:5], sorted(sg - se)[:5]
        print("   missing %d e.g. %s ; extra %d e.g. %s" % (len(se - sg), [o.kmer_string(x, k + 1 if graph else k) for x in missing], len(sg - se), [o.kmer_string(x, k + 1 if graph else k) for x in extra_k]))
    elif gc != ec:
        diff = [(o.kmer_string(a, k + 1 if graph else k), x, y) for a, x, y in zip(gk, gc, ec) if x != y][:8]
        print("   counts differ at", len([1 for x, y in zip(gc, ec) if x != y]), "keys, e.g.", diff)
