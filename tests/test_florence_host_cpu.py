"""Host-side captioner pieces that need no GPU: checkpoint key mapping, processor, generation config."""
import json

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def ckpt_dir():
    from tools.make_weights import ensure_caption_checkpoint
    return ensure_caption_checkpoint(0)


def _native_to_legacy(sd):
    """inverse of omniparser_amd.florence._legacy_to_native (remote-code Florence-2 naming)."""
    out = {}
    for k, v in sd.items():
        if k.startswith("model.vision_tower."):
            nk = k[len("model."):]
            if ".convs." in nk:
                nk = nk.replace(".conv.", ".proj.")
            for a, b in ((".spatial_block.conv1.", ".spatial_block.conv1.fn.dw."), (".spatial_block.conv2.", ".spatial_block.conv2.fn.dw."),
                         (".channel_block.conv1.", ".channel_block.conv1.fn.dw."), (".channel_block.conv2.", ".channel_block.conv2.fn.dw."),
                         (".spatial_block.norm1.", ".spatial_block.window_attn.norm."), (".channel_block.norm1.", ".channel_block.channel_attn.norm."),
                         (".window_attn.qkv.", ".window_attn.fn.qkv."), (".window_attn.proj.", ".window_attn.fn.proj."),
                         (".channel_attn.qkv.", ".channel_attn.fn.qkv."), (".channel_attn.proj.", ".channel_attn.fn.proj."),
                         (".norm2.", ".ffn.norm."), (".ffn.fc1.", ".ffn.fn.net.fc1."), (".ffn.fc2.", ".ffn.fn.net.fc2.")):
                nk = nk.replace(a, b)
        elif k == "model.multi_modal_projector.image_projection.weight":
            nk, v = "image_projection", v.t().contiguous()
        elif k.startswith("model.multi_modal_projector.image_proj_norm."):
            nk = k[len("model.multi_modal_projector."):]
        elif k.startswith("model.multi_modal_projector.image_position_embed."):
            nk = "image_pos_embed." + k[len("model.multi_modal_projector.image_position_embed."):]
        elif k.startswith("model.multi_modal_projector.visual_temporal_embed."):
            nk = k[len("model.multi_modal_projector."):]
        elif k.startswith("model.language_model."):
            nk = "language_model.model." + k[len("model.language_model."):]
        else:
            nk = k
        out[nk] = v
    return out


def test_legacy_checkpoint_names_map_onto_native(ckpt_dir):
    from safetensors.torch import load_file
    from omniparser_amd.florence import _legacy_to_native
    native = load_file(str(ckpt_dir / "model.safetensors"))
    legacy = _native_to_legacy(native)
    assert any(k.startswith("vision_tower.convs.0.proj") for k in legacy) and "image_projection" in legacy
    back = _legacy_to_native(legacy)
    assert set(back) == set(native)
    for k in native:
        assert torch.equal(back[k], native[k]), k


def test_weights_and_generation_config(ckpt_dir):
    from omniparser_amd.florence import FlorenceWeights
    w = FlorenceWeights(ckpt_dir)
    assert (w.embed_dim, w.depths, w.heads) == ([128, 256, 512, 1024], [1, 1, 9, 1], [4, 8, 16, 32])
    assert (w.d_model, w.n_heads, w.enc_layers, w.dec_layers, w.vocab) == (768, 12, 6, 6, 51290)
    assert (w.ngram, w.forced_bos, w.forced_eos, w.start, w.pad, w.eos) == (3, 0, 2, 2, 1, 2)
    assert "lm_head.weight" in w.sd and w.embed_scale == 1.0


def test_processor_matches_hf_image_transforms():
    """FlorenceProcessor pixel_values == PIL bicubic + transformers' numpy rescale/normalize."""
    from PIL import Image
    from transformers.image_transforms import normalize, rescale
    from omniparser_amd.util.utils import FlorenceProcessor
    rng = np.random.default_rng(0)
    imgs = [Image.fromarray(rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)) for _ in range(2)]
    proc = FlorenceProcessor(None)
    for do_resize, R in ((True, 768), (False, 64)):
        out = proc(images=imgs, text=["<CAPTION>"] * 2, return_tensors="pt", do_resize=do_resize)
        assert out["pixel_values"].shape == (2, 3, R, R)
        assert out["input_ids"].shape == (2, (R // 32) ** 2 + 1 + 8)
        im = imgs[0].resize((768, 768), Image.Resampling.BICUBIC) if do_resize else imgs[0]
        ref = normalize(rescale(np.asarray(im), 1 / 255), mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
        assert np.array_equal(out["pixel_values"][0].numpy(), ref.transpose(2, 0, 1).astype(np.float32))
    moved = out.to(device="cpu", dtype=torch.float16)
    assert moved["pixel_values"].dtype == torch.float16 and moved["input_ids"].dtype == torch.int64
    assert proc.batch_decode(torch.tensor([[2, 0, 100, 200, 2, 1]])) == ["tok100 tok200"]


def test_product_constructors_fail_loudly_without_gpu(ckpt_dir):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    with pytest.raises(RuntimeError):
        Florence2Captioner(ckpt_dir, "cpu")
    with pytest.raises(RuntimeError):
        YOLOv9Detector(model_path="does-not-matter.pt", device="cuda")
    with pytest.raises(RuntimeError):
        YOLOv9Detector(model_path="does-not-matter.pt", device="cpu")


def test_batch_decode_through_tokenizer_json(tmp_path):
    """ref:util/utils.py:128: `processor.batch_decode(ids, skip_special_tokens=True)` -> text.  With a tokenizer.json next to the
    checkpoint the product decodes through `tokenizers` (byte-level BPE, BART layout); pinned with the committed synthetic
    tokenizer (tests/golden/gen_tokenizer.py) because the real Florence-2 tokenizer files are not on this box."""
    import shutil
    from pathlib import Path
    from tokenizers import Tokenizer
    from omniparser_amd.util.utils import FlorenceProcessor
    src = Path(__file__).resolve().parent / "golden" / "tokenizer_synth" / "tokenizer.json"
    shutil.copy(src, tmp_path / "tokenizer.json")
    proc = FlorenceProcessor(tmp_path)
    tok = Tokenizer.from_file(str(src))
    texts = ["a blue settings icon with a gear", "close window", "volume slider"]
    rows = []
    for t in texts:
        ids = tok.encode(t).ids                                   # <s> ... </s>
        rows.append([2] + ids + [1] * (21 - 1 - len(ids)))       # decoder start token 2, then pad to the plan width
    out = proc.batch_decode(torch.tensor(rows), skip_special_tokens=True)
    assert [o.strip() for o in out] == texts
    raw = proc.batch_decode(torch.tensor(rows[:1]), skip_special_tokens=False)[0]
    assert "<s>" in raw and "</s>" in raw and "<pad>" in raw
    assert FlorenceProcessor(None).batch_decode(torch.tensor(rows[1:2]))[0].startswith("tok")      # no tokenizer file: ids as text


def test_plan_cache_is_bounded_by_bytes_and_never_evicts_the_pending_batch(monkeypatch):
    """Florence2Captioner._cached_plan: one LRU for encode and decode plan sets, bounded by bytes; the plan sets the batch
    being issued has taken are pinned; decode plans follow the same policy; a twin's size is the estimate for the next slot."""
    import types
    import torch
    from omniparser_amd.florence import Florence2Captioner as F
    cap = F.__new__(F)
    cap.device = torch.device("cpu")
    cap._plans = {}
    synced = []
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: synced.append(1))
    monkeypatch.setenv("OMNI_CAPTION_PLAN_GB", "100")
    GB = 2 ** 30
    built = []

    def take(key, gb):
        def build():
            built.append(key)
            return types.SimpleNamespace(key=key)
        obj = cap._cached_plan(key, build)
        cap._plan_meta[key][0] = gb * GB                      # what memory_allocated would have measured on the device
        cap._plan_sizes[key[:4] if key[0] == "dec" else key[:3]] = gb * GB
        return obj

    cap.begin_batch()
    a = take((128, 768, 20), 40)
    b = take((128, 768, 20, 1), 40)                           # twin: 80 GB resident
    d = take(("dec", 384, 768, 20), 8)
    assert cap.plan_cache_bytes() == 88 * GB and not synced
    # same batch: a third 128-row lane does not fit (88 + 40 > 100), but everything resident is pinned -> built anyway, nothing evicted
    c = take((128, 768, 20, 2), 40)
    assert len(cap._plans) == 4 and not synced
    # next batch: touching a and d pins them; the cache is over its bound (128 GB resident).  A 96-row plan has never been built, so its
    # size is ESTIMATED from the known 128-row set of the same resolution (40 GB * 96 / 128 = 30 GB: activations are linear in the
    # rows) — the byte bound guards first builds too: the unpinned sets b and c go (128 + 30 > 100, 88 + 30 > 100, 48 + 30 fits)
    cap.begin_batch()
    assert take((128, 768, 20), 40) is a and take(("dec", 384, 768, 20), 8) is d and len(built) == 4
    e = take((96, 768, 20), 30)
    assert list(cap._plans) == [(128, 768, 20), ("dec", 384, 768, 20), (96, 768, 20)] and len(synced) == 2
    # ... its twin (30 GB known now, 78 resident) fits without evicting anything
    take((96, 768, 20, 1), 30)
    assert list(cap._plans) == [(128, 768, 20), ("dec", 384, 768, 20), (96, 768, 20), (96, 768, 20, 1)]
    assert len(synced) == 2 and cap.plan_evictions == 2 and cap.plan_cache_bytes() == (40 + 8 + 30 + 30) * GB
    # decode plans are evicted like any other set once they are the oldest unpinned entries
    cap.begin_batch()
    take((64, 768, 20), 90)                                   # estimate 40 * 64 / 128 = 20 GB: 108 + 20 > 100 -> a goes, 68 + 20 fits
    assert list(cap._plans)[0] == ("dec", 384, 768, 20)
    take((64, 768, 20, 1), 90)                                # estimate 90 (known now) -> evicts d, e, e' (all unpinned), keeps the pinned 64-row set
    assert list(cap._plans) == [(64, 768, 20), (64, 768, 20, 1)]
    # an entry placed in _plans from outside (tests do) has no size record: it is evictable and never a KeyError
    cap.begin_batch()
    cap._plans[("foreign",)] = object()
    take((32, 768, 20), 10)
    take((32, 768, 20, 1), 10)


def test_checkpoint_audit_is_strict(ckpt_dir, tmp_path):
    """the loader audits a checkpoint against config.json — the expected key list equals the tensors of the native
    transformers module tree (the stand-in checkpoint is a saved transformers model), a legacy-named (`trust_remote_code`) checkpoint
    passes after the rename, and a missing / unexpected / mis-shaped tensor is an error that names it."""
    import json
    import shutil
    from safetensors.torch import load_file, save_file
    from omniparser_amd.florence import FlorenceWeights, _OPTIONAL_KEYS, audit_checkpoint, expected_native_keys
    cfg = json.loads((ckpt_dir / "config.json").read_text())
    native = load_file(str(ckpt_dir / "model.safetensors"))
    want = expected_native_keys(cfg)
    assert set(want) == set(native) - set(_OPTIONAL_KEYS)
    assert all(tuple(native[k].shape) == tuple(v) for k, v in want.items())
    audit_checkpoint(native, cfg)

    def write(sd, name):
        d = tmp_path / name
        d.mkdir()
        for f in ("config.json", "generation_config.json"):
            if (ckpt_dir / f).exists():
                shutil.copy(ckpt_dir / f, d / f)
        save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
        return d
    # a legacy-named checkpoint loads (renamed, audited) and yields the same tensors
    w = FlorenceWeights(write(_native_to_legacy(native), "legacy"))
    assert torch.equal(w.sd["model.vision_tower.convs.0.conv.weight"], native["model.vision_tower.convs.0.conv.weight"].float())
    assert torch.equal(w.sd["model.multi_modal_projector.image_projection.weight"], native["model.multi_modal_projector.image_projection.weight"].float())
    # missing / unexpected / mis-shaped
    bad = dict(native); bad.pop("model.vision_tower.blocks.2.7.channel_block.ffn.fc2.bias")
    with pytest.raises(ValueError, match=r"1 missing \[model\.vision_tower\.blocks\.2\.7\.channel_block\.ffn\.fc2\.bias\]"):
        FlorenceWeights(write(bad, "missing"))
    bad = dict(native); bad["model.vision_tower.blocks.0.0.spatial_block.window_attn.fn.qkv.weight"] = torch.zeros(3)
    with pytest.raises(ValueError, match="1 unexpected"):
        FlorenceWeights(write(bad, "unexpected"))
    bad = dict(native); bad["model.language_model.decoder.layers.3.fc1.weight"] = torch.zeros(3072, 767)
    with pytest.raises(ValueError, match=r"1 mis-shaped \[model\.language_model\.decoder\.layers\.3\.fc1\.weight: \(3072, 767\) != \(3072, 768\)\]"):
        FlorenceWeights(write(bad, "shape"))


# ------------------------------------------------------------------------------------------ real-artefact readiness
def _tiny_native_config():
    """a Florence-2-shaped config small enough to write and audit in milliseconds (window 12 and 64-wide BART heads are structural)"""
    return {"model_type": "florence2", "image_token_id": 99,
            "vision_config": {"embed_dim": [16, 32, 64, 128], "depths": [1, 1, 2, 1], "num_heads": [1, 2, 4, 8], "num_groups": [1, 2, 4, 8],
                              "patch_size": [7, 3, 3, 3], "patch_stride": [4, 2, 2, 2], "patch_padding": [3, 1, 1, 1],
                              "patch_prenorm": [False, True, True, True], "window_size": 12, "projection_dim": 64,
                              "max_position_embeddings": 50, "max_temporal_embeddings": 100},
            "text_config": {"model_type": "bart", "d_model": 64, "vocab_size": 100, "encoder_layers": 1, "decoder_layers": 1,
                            "encoder_attention_heads": 1, "decoder_attention_heads": 1, "encoder_ffn_dim": 96, "decoder_ffn_dim": 96,
                            "max_position_embeddings": 32, "scale_embedding": False}}


def _write_legacy_checkpoint(d, gen=None, text_extra=None, top_extra=None):
    """A checkpoint in the REMOTE-CODE layout (what `AutoModelForCausalLM.from_pretrained(..., trust_remote_code=True)` of the reference
    reads, ref:util/utils.py:63-68): legacy tensor names, `image_projection` as a Parameter [C, P] used as `x @ W` (the native Linear
    stores its transpose), the tied `language_model.lm_head.weight` + zero `final_logits_bias`, and a config.json with `dim_embed` and the
    nested `image_pos_embed` / `visual_temporal_embedding` objects.  Returns the native-layout tensors it was derived from."""
    from safetensors.torch import save_file
    from omniparser_amd.florence import expected_native_keys
    cfg = _tiny_native_config()
    g = torch.Generator().manual_seed(5)
    native = {k: torch.randn(shape, generator=g) for k, shape in expected_native_keys(cfg).items()}
    legacy = _native_to_legacy(native)
    legacy["language_model.lm_head.weight"] = native["model.language_model.shared.weight"].clone()
    legacy["language_model.final_logits_bias"] = torch.zeros(1, cfg["text_config"]["vocab_size"])
    vc = dict(cfg["vision_config"])
    vc["dim_embed"] = vc.pop("embed_dim")
    vc["image_pos_embed"] = {"type": "learned_abs_2d", "max_pos_embeddings": vc.pop("max_position_embeddings")}
    vc["visual_temporal_embedding"] = {"type": "COSINE", "max_temporal_embeddings": vc.pop("max_temporal_embeddings")}
    vc["image_feature_source"] = ["spatial_avg_pool", "temporal_avg_pool"]
    lcfg = {"model_type": "florence2", "architectures": ["Florence2ForConditionalGeneration"], "vision_config": vc,
            "text_config": dict(cfg["text_config"], **(text_extra or {})), "bos_token_id": 0, "eos_token_id": 2, "pad_token_id": 1,
            **(top_extra or {})}
    d.mkdir(parents=True, exist_ok=True)
    (d / "config.json").write_text(json.dumps(lcfg))
    if gen is not None:
        (d / "generation_config.json").write_text(json.dumps(gen))
    save_file({k: v.contiguous() for k, v in legacy.items()}, str(d / "model.safetensors"))
    return native


def test_legacy_remote_code_checkpoint_loads_through_the_strict_audit(tmp_path):
    """(i) of the readiness list: legacy key names AND the legacy config spelling through FlorenceWeights — same tensors as the native
    layout, the Parameter-vs-Linear transpose of `image_projection` undone, the tied lm_head accepted."""
    from omniparser_amd.florence import FlorenceWeights
    native = _write_legacy_checkpoint(tmp_path / "legacy", gen={"no_repeat_ngram_size": 3, "forced_bos_token_id": 0, "forced_eos_token_id": 2,
                                                                 "num_beams": 3, "bos_token_id": 0, "eos_token_id": 2, "pad_token_id": 1,
                                                                 "decoder_start_token_id": 2})
    w = FlorenceWeights(tmp_path / "legacy")
    assert (w.embed_dim, w.depths, w.heads, w.groups) == ([16, 32, 64, 128], [1, 1, 2, 1], [1, 2, 4, 8], [1, 2, 4, 8])
    assert (w.d_model, w.n_heads, w.enc_layers, w.dec_layers, w.vocab) == (64, 1, 1, 1, 100)
    for k, v in native.items():
        assert torch.equal(w.sd[k], v), k
    assert w.sd["model.multi_modal_projector.image_projection.weight"].shape == (64, 128)        # Linear [P, C], the file held [C, P]
    assert torch.equal(w.sd["lm_head.weight"], native["model.language_model.shared.weight"])
    assert (w.ngram, w.forced_bos, w.forced_eos, w.start, w.pad, w.bos, w.eos) == (3, 0, 2, 2, 1, 0, 2)


def test_generation_defaults_with_and_without_generation_config(tmp_path):
    """(ii): `no_repeat_ngram_size / forced_bos / forced_eos` come from generation_config.json when it is there; without the file from
    text_config (where the remote-code config keeps them) or the top level of config.json; with none of them the transformers default
    applies (no n-gram ban, nothing forced) — the device's greedy_step takes exactly these three numbers."""
    from omniparser_amd.florence import FlorenceWeights
    _write_legacy_checkpoint(tmp_path / "a", gen={"no_repeat_ngram_size": 2, "forced_bos_token_id": 5, "forced_eos_token_id": 7,
                                                   "decoder_start_token_id": 4, "pad_token_id": 9, "eos_token_id": 7})
    w = FlorenceWeights(tmp_path / "a")
    assert (w.ngram, w.forced_bos, w.forced_eos, w.start, w.pad, w.eos) == (2, 5, 7, 4, 9, 7)
    _write_legacy_checkpoint(tmp_path / "b", text_extra={"no_repeat_ngram_size": 3, "forced_bos_token_id": 0, "forced_eos_token_id": 2,
                                                          "decoder_start_token_id": 2})
    w = FlorenceWeights(tmp_path / "b")
    assert (w.ngram, w.forced_bos, w.forced_eos, w.start, w.pad, w.bos, w.eos) == (3, 0, 2, 2, 1, 0, 2)      # pad / bos / eos: top level
    _write_legacy_checkpoint(tmp_path / "c")
    w = FlorenceWeights(tmp_path / "c")
    assert (w.ngram, w.forced_bos, w.forced_eos, w.start) == (0, -1, -1, 2)
    # a generation_config.json that exists but says nothing about them (`{"_from_model_config": true}`) behaves like its absence
    _write_legacy_checkpoint(tmp_path / "d", gen={"_from_model_config": True, "transformers_version": "4.41.0"},
                             top_extra={"no_repeat_ngram_size": 3, "forced_eos_token_id": 2})
    w = FlorenceWeights(tmp_path / "d")
    assert (w.ngram, w.forced_bos, w.forced_eos) == (3, -1, 2)


def test_config_that_names_neither_spelling_is_a_named_error(tmp_path):
    from omniparser_amd.florence import FlorenceWeights, normalise_config
    _write_legacy_checkpoint(tmp_path / "x")
    cfg = json.loads((tmp_path / "x" / "config.json").read_text())
    del cfg["vision_config"]["dim_embed"]
    (tmp_path / "x" / "config.json").write_text(json.dumps(cfg))
    with pytest.raises(ValueError, match="vision_config.embed_dim"):
        FlorenceWeights(tmp_path / "x")
    native = _tiny_native_config()
    assert normalise_config(native) == native                       # a native config passes through unchanged


def test_special_token_ids_come_from_tokenizer_json(tmp_path):
    """(iii): which ids `batch_decode(skip_special_tokens=True)` drops is what tokenizer.json flags `special` — not a hard-coded
    {0, 1, 2, 3}.  A tokenizer whose specials sit elsewhere (here: moved to 10..13 with an ordinary token at id 2) must decode right;
    without a tokenizer file the ids the checkpoint's generation config names are used."""
    from pathlib import Path
    from tokenizers import Tokenizer
    from omniparser_amd.util.utils import FlorenceProcessor
    src = Path(__file__).resolve().parent / "golden" / "tokenizer_synth" / "tokenizer.json"
    d = json.loads(src.read_text())
    assert {t["id"] for t in d["added_tokens"] if t["special"]} == {0, 1, 2, 3}
    proc = FlorenceProcessor(src.parent)
    assert proc.special == {0, 1, 2, 3, 51289}
    # add a special task token far from the first four, flag-only change for <unk>: the processor follows the file
    extra = max(d["model"]["vocab"].values()) + 1
    d["added_tokens"].append({"id": extra, "content": "<loc_0>", "single_word": False, "lstrip": False, "rstrip": False, "normalized": False, "special": True})
    d["model"]["vocab"]["<loc_0>"] = extra
    for t in d["added_tokens"]:
        if t["content"] == "<unk>":
            t["special"] = False
    (tmp_path / "tokenizer.json").write_text(json.dumps(d))
    proc2 = FlorenceProcessor(tmp_path, image_token_id=extra + 5)
    assert proc2.special == {0, 1, 2, extra, extra + 5}
    tok = Tokenizer.from_file(str(tmp_path / "tokenizer.json"))
    ids = tok.encode("close window").ids
    row = [2] + ids[:-1] + [extra, extra + 5] + ids[-1:] + [1, 1]
    assert proc2.batch_decode(torch.tensor([row]), skip_special_tokens=True)[0].strip() == "close window"
    assert "<loc_0>" in proc2.batch_decode(torch.tensor([row]), skip_special_tokens=False)[0]
    # no tokenizer file: the checkpoint's own bos / pad / eos / unk
    p3 = FlorenceProcessor(None, image_token_id=77, special_ids=(5, 6, 7, 8))
    assert p3.batch_decode(torch.tensor([[7, 5, 100, 77, 200, 7, 6]])) == ["tok100 tok200"]
