"""Replay one case of tests/test_gpu_fuzz.py::fuzz_multi (same rng stream, earlier cases skipped without GPU work)
and say which of the single index / the multi-shard index disagrees with the oracle, and how.
usage: python scratch/replay_multi_case.py SEED CASE [REPEATS]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def replay(seed0, case):
    rng = np.random.default_rng(seed0)
    for c in range(case + 1):
        dim = int(rng.choice([64, 384, 768, 1024]))
        dtype = str(rng.choice(["f32", "f16"]))
        n = int(rng.choice([1, 5, 77, 1000, 12000]))
        shards = int(rng.choice([1, 2, 3, 5]))
        row_seed = int(rng.integers(1, 1 << 30))
        n_clusters = int(rng.choice([0, 6]))
        nq = int(rng.choice([1, 3, 20]))
        k = int(rng.choice([1, 10, 100]))
        q_seeds = [int(rng.integers(1, 1 << 30)) for _ in range(nq)]
        if c == case:
            return dict(dim=dim, dtype=dtype, n=n, shards=shards, row_seed=row_seed, n_clusters=n_clusters, nq=nq, k=k,
                        q_seeds=q_seeds)
        rng.choice(n, size=min(n, 9), replace=False)
        P = min(min(k, n), 40)
        if P >= 2:
            rng.integers(1, P + 1)
            rng.choice([0.0, 0.4, 1.0])


def main():
    seed0, case = int(sys.argv[1]), int(sys.argv[2])
    repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    p = replay(seed0, case)
    print("case", p, flush=True)
    dim, dtype, n, k = p["dim"], p["dtype"], p["n"], p["k"]
    rows = O.synth_rows(n, dim, seed=p["row_seed"], n_clusters=p["n_clusters"], f16=(dtype == "f16"))
    if n > 10:
        rows[n - 1] = rows[0]
    qs = np.stack([O.normalize(O.synth_query(dim, seed=s)) for s in p["q_seeds"]])
    if n > 10:
        qs[0] = O.normalize(rows[0].copy())
    want = []
    for i in range(len(qs)):
        sc = O.scan(rows, qs[i])
        order = np.argsort(-sc, kind="stable")[:k]
        want.append((order.astype(np.uint64), sc[order]))
    if len(sys.argv) > 4:        # CPU only: just show the oracle's answer
        for i, (r, c) in enumerate(want):
            print(i, list(r[:12]), [float(x) for x in c[:4]])
        return
    rlr = importlib.import_module("rust-local-rag_amd")
    for rep in range(repeats):
        one = rlr.GpuIndex(dim, dtype)
        one.upload(rows)
        mi = rlr.MultiGpuIndex(dim, [0] * p["shards"], dtype)
        mi.upload(rows)
        for name, ix in (("one", one), ("multi", mi)):
            for attempt in range(3):
                r, c = ix.search_topk(qs, k)
                bad = []
                for i, (wr, wc) in enumerate(want):
                    if not (np.array_equal(r[i], wr) and np.array_equal(bits(c[i]), bits(wc))):
                        bad.append(i)
                print("rep", rep, name, "attempt", attempt, "mismatching queries:", bad, flush=True)
                for i in bad[:3]:
                    wr, wc = want[i]
                    print("   q", i, "want rows", list(wr), "\n        got rows", list(r[i]))
                    print("        want cos", [hex(x) for x in bits(wc)], "\n        got cos ", [hex(x) for x in bits(c[i])])
        one.close()
        mi.close()


if __name__ == "__main__":
    main()
