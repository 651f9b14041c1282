"""-m gpu: short randomised differential runs (the long form is tools/fuzz_parity.py --seconds N): brute force across every dispatch path and
BM25 multi-term merges with random operators / limits / field boosts, all bit-exact against the CPU oracle."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from oracle.pyoracle import FtOracle
from .test_bm25_oracle import _multi_case

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_bruteforce_fuzz_short(rxgpu):
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_parity.py"), "--seconds", "12", "--seed", "7"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_hnsw_fuzz_short(rxgpu):
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_hnsw.py"), "--seconds", "10", "--seed", "5"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "hnsw fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_bm25_multi_term_fuzz_short(rxgpu, oracle):
    from reindexer_amd import hostapi
    ft = FtOracle(oracle)
    rng = np.random.default_rng(2024)
    for it in range(40):
        nf = int(rng.integers(1, 5))
        nterms = int(rng.integers(2, 5))
        ops = tuple(int(x) for x in rng.choice([1, 1, 2, 3], nterms))
        if all(o == 3 for o in ops):
            ops = (1,) + ops[1:]
        limit = int(rng.choice([20000, 20000, 500, 120, 37]))
        total = int(rng.choice([800, 3000, 12000]))
        fbs = [[float(rng.choice([0.0, 0.5, 1.0, 2.0])) for _ in range(nf)] for _ in range(nterms)] if rng.random() < 0.5 else None
        if fbs:
            for fb in fbs:
                if all(v == 0.0 for v in fb):
                    fb[0] = 1.0
        _, words, avg, removed, excluded, terms, store = _multi_case(5000 + it, nf, total, limit, ops, bool(rng.random() < 0.4), fbs,
                                                                     sizes=(max(2, total // 40), max(3, total // 3)))
        m = hostapi.GpuFtMerger(nf)
        m.set_docs(words, avg, removed)
        for s in store:
            m.set_word_fpos(s["word"], s)
        cfg = ft.default_config(nf, merge_limit=limit, min_rank=int(rng.choice([0, 5, 40])))
        cfg["distance_boost"], cfg["distance_weight"] = float(rng.choice([1.0, 1.7, 0.0])), float(rng.choice([0.5, 0.8, 1.0]))
        gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
        exc = excluded if rng.random() < 0.5 else None
        wd, wp, wf, wn, wpre = ft.merge_query(cfg, terms, total, words, avg, removed, exc, sort_by_rank=False, distance_boost=cfg["distance_boost"],
                                              distance_weight=cfg["distance_weight"])
        gd, gp, gf, gn, gpre = m.merge_query(cfg, gterms, exc, sort_by_rank=False)
        assert gpre == wpre and np.array_equal(gd, wd.astype(np.int32)), (it, ops, nf, limit, total)
        assert np.array_equal(gn, wn) and np.array_equal(gf, wf) and np.array_equal(gp.view(np.uint32), wp.view(np.uint32)), (it, ops)
        m.close()
