import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gossamer_amd as g
import oracle_lib as o
rng = random.Random(55)
comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
genome = list("".join(rng.choice("ACGT") for _ in range(60000)))
for _ in range(int(sys.argv[3])):
    n = rng.randint(60, 140)
    piece = rng.choice("AT") * n if rng.random() < 0.7 else "A" * (n // 2) + "T" * (n - n // 2)
    at = rng.randint(0, len(genome) - n)
    genome[at:at + n] = piece
genome = "".join(genome)
reads = []
for _ in range(int(sys.argv[1])):
    p = rng.randint(0, len(genome) - 101)
    r = list(genome[p:p + rng.randint(100, 101)])
    for i in range(len(r)):
        if rng.random() < float(sys.argv[2]):
            r[i] = rng.choice("ACGT")
    r = "".join(r)
    reads.append(r if rng.random() < 0.5 else "".join(comp[c] for c in reversed(r)))
txt = ("\n".join(reads) + "\n").encode()
ek, ec, _, nwin = o.count([(o.LINE, "r", txt)], 56, 1)
want = dict(zip(ek, ec))
print("windows", nwin, "distinct", len(ek), flush=True)
for env, budget in (({"GOSS_GPU_FUSED_MIN": "0", "GOSS_GPU_NO_FUSED": "1"}, 1 << 30), ({"GOSS_GPU_FUSED_MIN": "0", "GOSS_GPU_DEBUG": "1"}, 8 << 30)):
    os.environ.update(env)
    try:
        with g.Context(55, g.MODE_GRAPH, hbm_budget=budget) as ctx:
            ctx.push_host(txt)
            c = ctx.finish()
            gk, gc = ctx.result()
            st = {n: ctx.stat(n) for n in ("hash_merges", "seg_merges", "table96_chunks", "fused_chunks", "runs")}
        gc = [int(x) for x in gc]
        diffs = [(o.kmer_string(x, 56)[:8] + ".." + o.kmer_string(x, 56)[-4:], a, want.get(x)) for x, a in zip(gk, gc) if want.get(x) != a]
        print(env, budget >> 20, "MB: diffs", len(diffs), diffs[:3], st, flush=True)
    except g.GossGpuError as e:
        print(env, budget >> 20, "refused", e)
    for n in env: os.environ.pop(n, None)
