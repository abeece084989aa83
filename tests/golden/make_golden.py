#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.

The reference (C++ with Boost, which this image lacks) cannot be built or imported here, so the
golden vectors have two sources:

* `reference_kat.json` -- known answers that come from the reference itself: the values recorded
  in SURVEY.md Appendix C (captured from the reference's own code by the survey) and the
  constants of the reference's unit tests for this path (testGossCmdBuildGraph.cc,
  testReverseComplementAdapter.cc, testUtils.cc, testVByteCodec.cc).  Written by hand below;
  this script only re-serialises them.
* `small_objects.json` -- inputs and every output file (zlib + base64) of build-kmer-set / build-graph /
  merge / set algebra / dump on small read sets, produced by the CPU oracle (oracle/), whose key
  stream, counts, headers and statistics are pinned by the vectors above.  These pin the
  PRODUCT's bytes on the GPU box without the oracle in the loop, and pin the oracle against
  accidental change here.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

REFERENCE_KAT = {
    "source": "SURVEY.md Appendix C (captured from the reference) and the reference's unit tests",
    "kmers_k25": [
        {"seq": "ACGTACGTACGTACGTACGTACGTA", "value": 119212931312748, "hash": 3023600895719869485,
         "rc_value": 874228162960155, "rc_hash": 2200065297453272164, "canonical": "rc"},
        {"seq": "TTTTTTTTTTTTTTTTTTTTTTTTT", "value": 1125899906842623, "hash": 15861409801123372236,
         "rc_value": 0, "rc_hash": 9808874869469701221, "canonical": "rc"},
        {"seq": "GATTACAGATTACAGATTACAGATT", "value": 629233934386319, "hash": 12020582313314063284,
         "rc_value": 61232791124749, "rc_hash": 14067249205108543165, "canonical": "fwd"},
        {"seq": "CCCCCCCCCCCCCAAAAAAAAAAAA", "value": 375299963355136, "hash": 6897489284872682045,
         "rc_value": 1125899884473002, "rc_hash": 17867811721608618404, "canonical": "fwd"},
    ],
    "kmer_k4": {"seq": "ACGT", "value": 27, "hash": 4535195482310230718},
    "kmer_k56": {"seq": "ACGTTGCA" * 7, "lo": 2009762000248511460, "hi": 30666534427620, "hash": 7144136964027999792,
                 "rc_lo": 16436982073461040155, "rc_hi": 250808442283035},
    "sparse_d": [[25, 126000000, 23], [25, 1000000000, 20], [25, 3000000000, 18], [56, 200000000, 84], [56, 4000000000, 80]],
    "five_key_kmer_set": {"k": 25, "keys": [3, 1000, 123456789, 2 ** 40, 2 ** 49 + 17], "D": 47, "qD": 48,
                          "sizes": {".header": 24, ".kmers.header": 64, ".kmers.high-bits": 8, ".kmers-d0": 4160,
                                    ".kmers-d1": 4144, ".kmers.low-bits.lwr": 20, ".kmers.low-bits.upr": 10}},
    "testGossCmdBuildGraph": {"polyA_k27_edges": 2, "four_N_read_k27_edges": 42},
    "testReverseComplementAdapter_items": 116,
    "testVByteCodec": {"0": "00", "1": "01", "128": "8080"},
}


def make_reads(rng, n, genome_len, lo, hi):
    genome = "".join(rng.choice("ACGT") for _ in range(genome_len))
    reads = []
    for _ in range(n):
        length = rng.randrange(lo, hi + 1)
        s = rng.randrange(0, genome_len - length)
        r = genome[s:s + length]
        if rng.random() < 0.5:
            r = r[::-1].translate(str.maketrans("ACGT", "TGCA"))
        if rng.random() < 0.1:
            p = rng.randrange(len(r))
            r = r[:p] + "N" + r[p + 1:]
        if rng.random() < 0.2:
            r = r.lower()
        reads.append(r)
    return reads


def hexfiles(files):
    """file name -> base64(zlib(bytes)): the DenseSelect header pages are mostly zeros."""
    import base64
    import zlib
    return {name: base64.b64encode(zlib.compress(data, 9)).decode() for name, data in sorted(files.items())}


def main():
    import oracle_lib as oracle
    with open(os.path.join(HERE, "reference_kat.json"), "w") as f:
        json.dump(REFERENCE_KAT, f, indent=1)

    rng = random.Random(20260101)
    a = make_reads(rng, 50, 800, 40, 120)
    b = make_reads(rng, 50, 800, 40, 120) + a[:15]
    txt_a = "\n".join(a) + "\n"
    txt_b = "\n".join(b) + "\n"
    out = {"inputs": {"a.txt": txt_a, "b.txt": txt_b}, "cases": []}

    def case(name, cmd, k, files, extra=None):
        c = {"name": name, "cmd": cmd, "k": k, "files": hexfiles(files)}
        if extra:
            c.update(extra)
        out["cases"].append(c)

    objs = {}
    for k in (25, 45):
        for tag, txt in (("a", txt_a), ("b", txt_b)):
            f, nwin = oracle.build_kmer_set([(oracle.LINE, tag, txt)], k, out="ks%d%s" % (k, tag))
            objs.update(f)
            case("ks%d%s" % (k, tag), "build-kmer-set", k, f, {"input": tag + ".txt", "windows": nwin})
    for k in (27, 55):
        for tag, txt in (("a", txt_a), ("b", txt_b)):
            f, nwin = oracle.build_graph([(oracle.LINE, tag, txt)], k, out="gr%d%s" % (k, tag))
            objs.update(f)
            case("gr%d%s" % (k, tag), "build-graph", k, f, {"input": tag + ".txt", "windows": nwin})
    case("mks25", "merge-kmer-sets", 25, oracle.merge(objs, ["ks25a", "ks25b"], 0, "mks25"), {"ins": ["ks25a", "ks25b"]})
    case("mgr55", "merge-graphs", 55, oracle.merge(objs, ["gr55a", "gr55b"], 1, "mgr55"), {"ins": ["gr55a", "gr55b"]})
    case("iks45", "intersect-kmer-sets", 45, oracle.intersect_kmer_sets(objs, ["ks45a", "ks45b"], "iks45"), {"ins": ["ks45a", "ks45b"]})
    case("sks25", "subtract-kmer-set", 25, oracle.subtract_kmer_set(objs, "ks25a", "ks25b", "sks25"), {"ins": ["ks25a", "ks25b"]})
    ann, stats = oracle.merge_and_annotate(objs, "ks25a", "ks25b", "aks25")
    case("aks25", "merge-and-annotate-kmer-sets", 25, ann, {"ins": ["ks25a", "ks25b"], "stdout": "%d\t%d\t%d\n" % stats})
    out["dumps"] = {"ks25a": oracle.dump(objs, "ks25a", 0).decode(), "gr27a": oracle.dump(objs, "gr27a", 1).decode()}
    with open(os.path.join(HERE, "small_objects.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
