"""The committed search-level fixtures against the oracle (CPU): pins the oracle's search / diversity / MMR / blend
outputs -- everything the reference's own tests leave unpinned -- to data in the repository, so that
an edit of the oracle cannot move together with the kernels unnoticed."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _maker():
    spec = importlib.util.spec_from_file_location("make_fixtures", os.path.join(GOLDEN, "make_fixtures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_oracle_reproduces_the_committed_fixtures():
    with open(os.path.join(GOLDEN, "search_fixtures.json")) as f:
        want = json.load(f)
    got = json.loads(json.dumps(_maker().build()))  # through JSON: tuples -> lists, exactly what the file holds
    assert [c["name"] for c in got["corpora"]] == [c["name"] for c in want["corpora"]]
    for g, w in zip(got["corpora"], want["corpora"]):
        assert g["rows_sha256"] == w["rows_sha256"], f"{w['name']}: the corpus generator changed"
        assert g["query_sha256"] == w["query_sha256"], f"{w['name']}: the query generator changed"
        assert len(g["calls"]) == len(w["calls"])
        for cg, cw in zip(g["calls"], w["calls"]):
            assert cg == cw, f"{w['name']}: {cw['kind']} {cw['args']} differs from the committed fixture"


def test_fixture_shapes_follow_the_reference_rules():
    """sizes implied by rag_engine.rs: top_k 0 acts as 1 (:490), stage-1 hands back min(N, 3k) (:544), the first MMR
    pick is the best-scored candidate (:782-785)"""
    with open(os.path.join(GOLDEN, "search_fixtures.json")) as f:
        doc = json.load(f)
    for c in doc["corpora"]:
        assert c["n"] <= 4096 and c["dim"] in (768, 1024)
        for call in c["calls"]:
            a, e = call["args"], call["expect"]
            if call["kind"] == "search":
                k = max(a["top_k"], 1)
                assert len(e["rows"]) == (min(c["n"], 3 * k) if a.get("stage") else min(c["n"], k))
            if call["kind"] == "search_with_diversity":
                assert len(e["rows"]) == min(c["n"], max(a["top_k"], 1))
            if call["kind"] == "mmr":
                assert e["order"][0] == 0 and len(set(e["order"])) == len(e["order"])
