import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
    if p not in sys.path:
        sys.path.insert(0, p)

# OMNI_EMU=1 python -m pytest tests/test_gpu_*.py -m gpu : REHEARSAL of the `-m gpu` tests without a GPU — the product's device path
# runs on the host emulation of its own kernels (tests/emu).  Sizes are the GPU's, so this takes minutes to hours; it is a
# developer tool, never what the driver runs (OMNI_EMU is unset there and on the GPU box).
REHEARSE = os.environ.get("OMNI_EMU", "0") == "1"
# parity mode: a GEMM operand beyond the f16 range of the split formats is FATAL in every test that runs the pipeline (the product
# only counts it into stats["split_overflow"]; omniparser_amd/pipeline.py::_check_range)
os.environ.setdefault("OMNI_STRICT_RANGE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu or REHEARSE:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _rehearse_on_emulation(request):
    if not (REHEARSE and "gpu" in request.keywords):
        yield
        return
    import torch
    import gpu_checks  # noqa: F401  (emulated_device redirects its DEV / _sync)
    from emu_runtime import emulated_device
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with emulated_device():
            yield
    finally:
        torch.Tensor.cuda = cuda


@pytest.fixture
def emu():
    """host emulation of the device library bound behind omniparser_amd._lib for the duration of one (CPU) test."""
    import gpu_checks  # noqa: F401
    from emu_runtime import emulated_device
    with emulated_device() as L:
        yield L


def small_vocab_caption_checkpoint(seed=0, vocab=8192):
    """The stand-in caption checkpoint with its token table cut to the first `vocab` rows (every prompt id is < 6191): same DaViT / BART
    weights, an lm_head of 8192 instead of 51290 rows.  For the EMULATED tests that compare the device path with itself (merged vs
    per-micro-batch decode, C entry points vs Python objects, pipelined vs synchronous stream): a 128-row lm_head tile over the full table
    costs the host emulation 15 G multiply-adds per decode step, 6x what those tests need.  Tests against transformers use the full one."""
    import json
    from safetensors.torch import load_file, save_file
    from tools.make_weights import caption_dir, ensure_caption_checkpoint
    src = ensure_caption_checkpoint(seed)
    dst = src.with_name(src.name + f"_vocab{vocab}")
    if not (dst / "model.safetensors").exists():
        dst.mkdir(parents=True, exist_ok=True)
        sd = load_file(str(src / "model.safetensors"))
        full = sd["model.language_model.shared.weight"].shape[0]
        out = {}
        for k, v in sd.items():
            if v.dim() == 2 and v.shape[0] == full:                      # the (tied) token table
                v = v[:vocab].contiguous()
            elif "final_logits_bias" in k and v.shape[-1] == full:
                v = v[..., :vocab].contiguous()
            out[k] = v
        save_file(out, str(dst / "model.safetensors"))
        cfg = json.loads((src / "config.json").read_text())
        cfg.setdefault("text_config", {})["vocab_size"] = vocab
        (dst / "config.json").write_text(json.dumps(cfg))
        (dst / "generation_config.json").write_text((src / "generation_config.json").read_text())
    return dst
