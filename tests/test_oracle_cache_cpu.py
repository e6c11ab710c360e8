"""tests/oracle_cache.py (test infrastructure for the `-m gpu` suite): the cache is content-addressed — a row is keyed by the bytes of its
own input tensor, the crop resolution and max_new_tokens, inside a file named after the oracle model — so it can miss, never be stale."""
import json

import numpy as np
import torch


def test_row_key_depends_on_every_input_and_nothing_else():
    import oracle_cache as OC
    rng = np.random.default_rng(0)
    a = rng.standard_normal((64, 64, 3)).astype(np.float32)
    b = a.copy(); b[10, 20, 1] = np.nextafter(b[10, 20, 1], np.float32(1e9))       # one ulp in one value
    k = OC.row_key(a, 64, 20)
    assert k == OC.row_key(a.copy(), 64, 20) and len(k) == 32
    assert len({k, OC.row_key(b, 64, 20), OC.row_key(a, 64, 21), OC.row_key(a, 768, 20)}) == 4


def test_model_digest_follows_the_weights():
    import oracle_cache as OC
    m1, m2 = torch.nn.Linear(8, 4), torch.nn.Linear(8, 4)
    m2.load_state_dict(m1.state_dict())
    assert OC.model_digest(m1) == OC.model_digest(m2)
    with torch.no_grad():
        m2.weight[0, 0] += 1e-3
    assert OC.model_digest(m1) != OC.model_digest(m2)


def test_put_get_flush_roundtrip_and_committed_file_is_well_formed(tmp_path, monkeypatch):
    import oracle_cache as OC
    monkeypatch.setattr(OC, "CACHE_DIR", tmp_path / "gold")
    monkeypatch.setattr(OC, "MISS_DIR", tmp_path / "out" / "misses")
    (tmp_path / "out").mkdir()
    m = torch.nn.Linear(4, 4)
    c = OC.CaptionCache(m)
    assert c.get("k1") is None
    c.put("k1", [2, 0, 17, 2], 0.25)
    c.put("k2", [2, 0, 2], float("inf"))
    c.flush()                                                     # no OMNI_ORACLE_CACHE_WRITE: rows go next to gpurun_out/, not into the goldens
    assert not (tmp_path / "gold").exists() and (tmp_path / "out" / "misses" / c.name).exists()
    c2 = OC.CaptionCache(m)
    assert c2.get("k1") == {"ids": [2, 0, 17, 2], "margin": 0.25} and c2.get("k2")["margin"] is None
    monkeypatch.setenv("OMNI_ORACLE_CACHE_WRITE", "1")
    c2.put("k3", [2, 2], 1.0); c2.flush()
    assert json.loads((tmp_path / "gold" / c.name).read_text())["rows"].keys() == {"k3"}
    # the committed cache: every row is a token list that starts with the decoder start token and ends at its EOS, margins are floats or null
    from pathlib import Path
    gold = Path(__file__).resolve().parent / "golden" / "oracle_cache"
    files = sorted(gold.glob("caption_*.json"))
    assert files and (gold / "MANIFEST.json").exists()
    rows = json.loads(files[0].read_text())["rows"]
    assert len(rows) >= 800
    for k, r in list(rows.items())[:200]:
        assert len(k) == 32 and r["ids"][0] == 2 and 3 <= len(r["ids"]) <= 21 and (r["margin"] is None or r["margin"] >= 0.0)
