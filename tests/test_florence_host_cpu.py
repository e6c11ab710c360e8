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
    # next batch: touching a and d pins them; the cache is over its bound (128 GB resident), so taking a new 96-row plan (size unknown,
    # estimate 0) evicts the least recently used UNPINNED set (b) and stops as soon as the resident sets fit again
    cap.begin_batch()
    assert take((128, 768, 20), 40) is a and take(("dec", 384, 768, 20), 8) is d and len(built) == 4
    e = take((96, 768, 20), 30)
    assert list(cap._plans) == [(128, 768, 20, 2), (128, 768, 20), ("dec", 384, 768, 20), (96, 768, 20)] and len(synced) == 1
    # ... its twin (estimate 30 GB, 118 resident) evicts c; a, d, e belong to this batch and stay although the bound is exceeded
    take((96, 768, 20, 1), 30)
    assert list(cap._plans) == [(128, 768, 20), ("dec", 384, 768, 20), (96, 768, 20), (96, 768, 20, 1)]
    assert len(synced) == 2 and cap.plan_evictions == 2 and cap.plan_cache_bytes() == (40 + 8 + 30 + 30) * GB
    # decode plans are evicted like any other set once they are the oldest unpinned entries
    cap.begin_batch()
    take((64, 768, 20), 90)                                   # over the bound already: a goes (108 -> 68 GB fits with the estimate 0)
    assert list(cap._plans)[0] == ("dec", 384, 768, 20)
    take((64, 768, 20, 1), 90)                                # estimate 90 -> evicts d, e, e' (all unpinned), keeps the pinned 64-row set
    assert list(cap._plans) == [(64, 768, 20), (64, 768, 20, 1)]


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
