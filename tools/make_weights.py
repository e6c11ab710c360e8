"""Create a stand-in `icon_detect_v3/model.pt` TorchScript blob with seeded random weights.

The reference downloads its detector blob from the Hugging Face hub (ref:util/yolov9.py:43-50); there
is no network and no checkpoint on this box, so benchmarks and tests feed the product a blob with the
same architecture (oracle/yolov9e_ref.py = public YOLOv9-E, 58.05 M params unfused) and the same
loading path (`torch.jit.load`).  This is input-data generation, not part of the product path.
"""
import argparse
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


# Stand-in detector version.  v5 (round 5): BatchNorm shifts centred at +1 and gains 0.15 (oracle/yolov9e_ref.py::build_random_detector) —
# the stand-in's own f32-vs-f64 head difference is <= 2.6e-5 at 1088x1920 (v4: 1e-2 ... 8e-2) and <= 2e-5 at 640x640; and the score
# threshold is placed on the CALIBRATION batch alone (seeds 0..7 at 640x640): rounds 2-4 centred it in a gap of the parity frames'
# own logits (`PARITY_FRAMES`), i.e. tuned the stand-in to the frames it was tested on.  Every frame list below is now chosen by an
# a-priori scan of the CPU oracle (tools/scan_parity_frames.py) over frames the calibration never saw (seeds >= 8, other sizes).
DETECTOR_STANDIN = "v5"

# Frames (synthetic_screenshot seed at 1920x1080 — 640x480 for the 320 entry — letterboxed to the given network size; "native" = the
# scale_img=True path, 1088x1920 network input) on which the CPU oracle is tie-free WITH MARGIN: no NMS IoU within 3e-5 of the
# threshold, no suppression by a box whose score is within 9e-6 of its victim's, no anchor logit within 1e-4 of logit(conf) — several
# times the GPU-vs-oracle differences (scores ~1e-6, boxes ~2e-4 px, logits ~2e-5).  On these frames the parity tests demand the
# oracle's boxes one for one and fail if the oracle on the test box disagrees that they are tie-free.
# Scan: tools/scan_parity_frames.py -> profiles/r5_parity_frame_scan.md.  Re-scan whenever build_random_detector or synth.py changes.
# Every listed seed is >= 8 (the calibration batch holds seeds 0..7 at 640x640) except where the frame SIZE differs from the calibration's.
EXACT_FRAMES = {(1.0, 640): (14, 18), (0.5, 640): (14, 18), (0.25, 640): (9, 10), (0.25, 320): (8, 9), (1.0, "native"): (8, 12),
                (0.25, "native"): (8, 9)}
# configs[4]: the 3840x2160 frame on which the tiled policy (2x2 tiles -> global NMS) is tie-free in all four tiles and in the merge (half width)
TILED_EXACT_SEED = 4
# configs[3]: (w, h, seed) per resolution of stream.RESOLUTION_MIX, two frames at the most frequent size first (one device batch of two)
STREAM_FRAMES = ((2560, 1440, 12), (2560, 1440, 14), (3840, 2160, 8), (1920, 1080, 14), (2880, 1800, 10), (3456, 2234, 13), (5120, 2880, 8),
                 (2560, 1600, 20))
# frames the scan found well conditioned (ties or not): candidate sets must be identical there, final boxes up to the tie rules
WELL_FRAMES = {(1.0, 640): tuple(range(8)), (0.5, 640): tuple(range(8)), (0.25, 640): (0, 1, 2, 3), (0.25, 320): (0, 1, 2, 3)}


CALIB_DIR = ROOT / "tools" / "standin_calibration"


def calibration_path(seed=0, nc=1, width=1.0):
    return CALIB_DIR / f"{DETECTOR_STANDIN}_s{seed}_nc{nc}_w{width:g}.pt"


def make_blob(path, seed=0, nc=1, width=1.0):
    """Seeded initialisation + calibration.  The calibration results (BatchNorm statistics, class-head scale / threshold placement:
    a few MB) are committed under tools/standin_calibration/, so a fresh box builds the SAME blob bit for bit in seconds instead of
    re-running ~45 full-width CPU forward passes whose statistics differ in the last bits from host to host; without the file the
    calibration runs and writes it."""
    from oracle.yolov9e_ref import build_random_detector, calibrated_keys
    cpath = calibration_path(seed, nc, width)
    if cpath.exists() and os.environ.get("OMNI_RECALIBRATE", "0") != "1":
        model = build_random_detector(seed=seed, nc=nc, width=width, calibration=torch.load(str(cpath), map_location="cpu"))
    else:
        model = build_random_detector(seed=seed, nc=nc, width=width)      # threshold placed on the calibration batch alone
        sd = model.state_dict()
        calib = {k: sd[k].clone() for k in calibrated_keys(sd)}
        calib["margin"] = torch.tensor(model.margin, dtype=torch.float64)
        calib["pass_rate"] = torch.tensor(model.pass_rate, dtype=torch.float64)
        cpath.parent.mkdir(parents=True, exist_ok=True)
        torch.save(calib, str(cpath))
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with torch.no_grad():
        ts = torch.jit.trace(model, torch.rand(1, 3, 64, 64), check_trace=False)
    ts.save(str(path))
    return model


def default_path(seed=0, nc=1, width=1.0):
    tag = f"{DETECTOR_STANDIN}_s{seed}_nc{nc}_w{width:g}"
    return ROOT / "weights" / f"icon_detect_v3_{tag}" / "icon_detect_v3" / "model.pt"


def ensure_blob(seed=0, nc=1, width=1.0):
    p = default_path(seed, nc, width)
    if not p.exists():
        make_blob(p, seed, nc, width)
    return p


# ------------------------------------------------------------------------------------------------
# Florence-2-base-shaped caption checkpoint (random weights): config.json + model.safetensors +
# generation_config.json in the layout `get_caption_model_processor` loads (ref:util/utils.py:63-68).
def florence_base_config():
    from transformers import Florence2Config
    cfg = Florence2Config(
        vision_config=dict(projection_dim=768),
        text_config=dict(model_type="bart", vocab_size=51290, d_model=768, encoder_layers=6, decoder_layers=6,
                         encoder_attention_heads=12, decoder_attention_heads=12, encoder_ffn_dim=3072,
                         decoder_ffn_dim=3072, max_position_embeddings=1024, activation_function="gelu",
                         scale_embedding=False, dropout=0.1, num_beams=3, no_repeat_ngram_size=3,
                         forced_bos_token_id=0, forced_eos_token_id=2),
    )
    cfg._attn_implementation = "eager"
    return cfg


# Stand-in captioner version.  v2 (round 3): the q and k rows of every DaViT channel-attention qkv projection are scaled by
# CHAN_QK_SCALE.  hf's ChannelAttention scales q by N^-0.5 (N = tokens: 36 864 in stage 0 at 768x768) and sums q_i k_j over all
# tokens; on the reference's REAL caption inputs (a 64x64 crop up-sampled to 768x768: smooth, spatially coherent) that sum grows
# like N, not sqrt(N), so unscaled random weights give softmax logits of ~+-100 and every channel block amplifies rounding noise
# 10-50x: transformers' OWN f32 evaluation then differs from its f64 evaluation by 0.8 % in the image features and 0.02-0.11 in
# the logits, and flips greedy tokens (1 of 4 crops, measured on the CPU: profiles/r3_caption_oracle_conditioning.md) — "token
# exact vs the f32 CPU path" is not a defined target on such a model.  With the scale the f32 / f64 difference on those crops is
# 2.7e-6 in the features and <= 3e-5 in the logits (margins >= 5e-2), as on trained weights.  randn inputs never showed this
# (f32 vs f64 1.8e-5 before and after), which is why the round-2 tests on randn pixel_values were green while the real-crop
# test was red.
# v3 (round 6): v2 + a HOST-INDEPENDENT temporal table.  transformers fills `visual_temporal_embed.pos_idx_to_embed` (100 x 768 sin / cos
# values, a persistent buffer) with torch.exp / sin / cos at construction; their vectorised kernels split the 38 400 elements over
# the host's threads and treat chunk tails differently, so the table differs in the last ulp between an 8-thread and a 128-thread
# host (found when tests/oracle_cache.py began to hash EVERY tensor: the 667 seeded tensors were identical on both boxes, this one
# was not).  A real checkpoint carries the table; the stand-in now computes it with Python's scalar libm (`temporal_table`), so the
# oracle model of the CPU container IS the oracle model of the GPU box, bit for bit, and the committed oracle rows are rows of the
# very model the device path loads.
CAPTION_STANDIN = "v3"
CHAN_QK_SCALE = 0.2


def temporal_table(max_positions: int, embed_dim: int) -> torch.Tensor:
    """hf:models/florence2/modeling_florence2.py Florence2VisionPositionalEmbeddingCosine1D.get_sinusoid_embeddings with every
    transcendental evaluated by scalar libm in float64 and rounded to f32 where the original holds an f32 (host-independent)."""
    import math
    import numpy as np
    half = embed_dim // 2
    emb = math.log(10000) / half
    freq = [float(np.float32(math.exp(float(np.float32(k) * np.float32(-emb))))) for k in range(half)]
    t = torch.empty(max_positions, embed_dim)
    for p in range(max_positions):
        ang = [float(np.float32(p) * np.float32(f)) for f in freq]
        t[p, 0::2] = torch.tensor([math.sin(a) for a in ang], dtype=torch.float64).float()
        t[p, 1::2] = torch.tensor([math.cos(a) for a in ang], dtype=torch.float64).float()
    return t


def standin_scale(standin=None):
    return CHAN_QK_SCALE if (standin or CAPTION_STANDIN) == CAPTION_STANDIN else 1.0


def caption_dir(seed=0, standin=CAPTION_STANDIN):
    return ROOT / "weights" / f"icon_caption_florence_{standin}_s{seed}"


def build_random_captioner(seed=0, init_std=0.06, chan_qk_scale=CHAN_QK_SCALE):
    """transformers-native Florence2ForConditionalGeneration, fp32, eager attention, seeded weights.
    init_std is larger than the library default (0.02) so that logits depend visibly on the image;
    chan_qk_scale: see CAPTION_STANDIN above."""
    from transformers import Florence2ForConditionalGeneration
    cfg = florence_base_config()
    torch.manual_seed(seed)
    model = Florence2ForConditionalGeneration(cfg)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * init_std)
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            else:   # LayerNorm weights
                p.copy_(1.0 + 0.05 * torch.randn(p.shape, generator=g))
            if "channel_attn.qkv" in name:
                p[: 2 * (p.shape[0] // 3)] *= chan_qk_scale
        model.tie_weights()
        tab = model.model.multi_modal_projector.visual_temporal_embed.pos_idx_to_embed
        det = temporal_table(*tab.shape)
        assert (det - tab).abs().max().item() < 5e-5, "temporal table restatement drifted from transformers' own"   # 1 ulp of an f32 frequency x 99 positions = 6e-6 rad
        tab.copy_(det)
    model.generation_config.no_repeat_ngram_size = 3
    model.generation_config.forced_bos_token_id = 0
    model.generation_config.forced_eos_token_id = 2
    return model.eval()


_CAPTIONERS = {}


def shared_random_captioner(seed=0, init_std=0.06, chan_qk_scale=CHAN_QK_SCALE):
    """One instance per (seed, init_std, scale) and process: building the 230 M-parameter stand-in costs 10-20 s of CPU, and a test
    process asks for the same read-only oracle model a dozen times (tests/gpu_checks.py).  Callers must not modify it."""
    key = (seed, init_std, chan_qk_scale)
    if key not in _CAPTIONERS:
        _CAPTIONERS[key] = build_random_captioner(seed, init_std, chan_qk_scale)
    return _CAPTIONERS[key]


def ensure_caption_checkpoint(seed=0, standin=CAPTION_STANDIN):
    """standin="v1": the round-1/2 stand-in (no channel-attention scale) — kept only so that tools/archive/r3_bisect.py can show what
    it does to the f32 arithmetic on real crops."""
    import json
    from safetensors.torch import save_file
    d = caption_dir(seed, standin)
    if not (d / "model.safetensors").exists():
        d.mkdir(parents=True, exist_ok=True)
        model = build_random_captioner(seed, chan_qk_scale=standin_scale(standin))
        skip = ("lm_head.weight", "model.language_model.encoder.embed_tokens.weight",
                "model.language_model.decoder.embed_tokens.weight")            # tied to shared.weight
        sd = {k: v.contiguous().clone() for k, v in model.state_dict().items() if k not in skip}
        save_file(sd, str(d / "model.safetensors"))
        (d / "config.json").write_text(model.config.to_json_string())
        (d / "generation_config.json").write_text(json.dumps(
            {"no_repeat_ngram_size": 3, "forced_bos_token_id": 0, "forced_eos_token_id": 2, "num_beams": 3,
             "bos_token_id": 0, "eos_token_id": 2, "pad_token_id": 1, "decoder_start_token_id": 2}))
    return d


def ensure_via_subprocess(kind: str, seed=0, nc=1, width=1.0):
    """Generate the stand-in checkpoint in a SEPARATE process (so the caller's process never imports oracle/)
    and return its path.  kind: 'detector' | 'caption'."""
    import subprocess
    path = default_path(seed, nc, width) if kind == "detector" else caption_dir(seed)
    marker = path if kind == "detector" else path / "model.safetensors"
    if not marker.exists():
        cmd = [sys.executable, str(Path(__file__).resolve()), "--ensure", kind, "--seed", str(seed), "--nc", str(nc), "--width", str(width)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 or not marker.exists():
            raise RuntimeError(f"weight generation failed ({' '.join(cmd)}):\n{r.stderr[-2000:]}")
    return path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--nc", type=int, default=1)
    ap.add_argument("--width", type=float, default=1.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--ensure", choices=["detector", "caption"], default=None)
    a = ap.parse_args()
    if a.ensure == "caption":
        print(ensure_caption_checkpoint(a.seed))
    elif a.ensure == "detector":
        print(ensure_blob(a.seed, a.nc, a.width))
    else:
        out = a.out or default_path(a.seed, a.nc, a.width)
        make_blob(out, a.seed, a.nc, a.width)
        print(out)
