"""Replay a dumped fuzz case through several contexts with the exchange before counting (the fuzz script's group
branch): all combinations of rounds and of a last exchange, files against the oracle.
usage: python tools/dbg/fuzz_replay_group.py <dir> <case> <parts>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gossamer_amd as g
from gossamer_amd import dist as gd
import oracle_lib as o
d, case, parts = sys.argv[1], sys.argv[2], int(sys.argv[3])
meta = json.load(open(os.path.join(d, "case%s.json" % case)))
reads = open(os.path.join(d, "case%s.reads" % case), "rb").read()
graph, k = meta["graph"], meta["k"]
exp, nwin = (o.build_graph if graph else o.build_kmer_set)([(o.LINE, "r", reads)], k, out="o")
exp = {n[1:]: b for n, b in exp.items()}
ek, ec, _, _ = o.count([(o.LINE, "r", reads)], k + 1 if graph else k, 1 if graph else 0)
os.environ.update(meta["env"])
for a in sys.argv[4:]:
    n, v = a.split("=", 1)
    if v == "-": os.environ.pop(n, None)
    else: os.environ[n] = v
lines = reads.split(b"\n")[:-1]
per = (len(lines) + parts - 1) // parts
shards = [b"".join(x + b"\n" for x in lines[i * per:(i + 1) * per]) for i in range(parts)]
budget = int(os.environ.get("REPLAY_BUDGET", max(meta["budget"], 256 << 20)))
for before in (False, True):
    for rounds in ((1, 2, 3) if before else (1,)):
        for last in ((True, False) if before else (False,)):
            ctxs = [g.Context(k, g.MODE_GRAPH if graph else g.MODE_KMER_SET, hbm_budget=budget) for _ in shards]
            try:
                windows = 0
                if before:
                    for cx in ctxs: cx.set_deferred(True)
                    for r in range(rounds):
                        for cx, sh in zip(ctxs, shards):
                            piece = b"".join(x + b"\n" for x in sh.split(b"\n")[:-1][r::rounds])
                            if piece: cx.push_host(piece)
                        if r + 1 < rounds or last: g.group_route_exchange(ctxs)
                per_ctx = []
                for cx, sh in zip(ctxs, shards):
                    if sh and not before: cx.push_host(sh)
                    windows += cx.finish().windows
                    ks, cs = cx.result()
                    per_ctx.append(dict(zip(ks, (int(x) for x in cs))))
                merged = {}
                for dct in per_ctx:
                    for a, b in dct.items(): merged[a] = merged.get(a, 0) + b
                mk = sorted(merged)
                pre_ok = mk == ek and [merged[x] for x in mk] == ec
                g.group_exchange(ctxs, sample_per_context=64)
                g.group_emit(ctxs)
                got = gd.assemble_files([cx.files() for cx in ctxs])
            except g.GossGpuError as e:
                print("before", before, "rounds", rounds, "last", last, "refused", e); continue
            finally:
                for cx in ctxs: cx.close()
            bad = [n for n in exp if got.get(n) != exp[n]]
            print("before", before, "rounds", rounds, "last exchange", last, "windows", windows, nwin, "counts before the range exchange right:", pre_ok, "files differ", bad, flush=True)
            if not pre_ok:
                se, sg = set(ek), set(mk)
                want = dict(zip(ek, ec))
                diffs = [(x, merged[x], want[x]) for x in mk if x in want and merged[x] != want[x]]
                L = k + 1 if graph else k
                print("   missing", len(se - sg), "extra", len(sg - se), "count diffs", len(diffs), [(o.kmer_string(x, L), a, b) for x, a, b in diffs[:6]])
                print("   per context:", [(len(dct), sum(dct.values())) for dct in per_ctx], "oracle", len(ek), sum(ec))
